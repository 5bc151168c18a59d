// Loop-structure laboratory for the LDS-direct fp32-MFMA GEMMs of gemm.hip (round 5).
// One dense problem with the operand geometry of the step's dominant launch - the weight-gradient GEMM of
// upsample2 -> conv3x3 512->256 at batch 128 (igemm_tng_kernel<128,128,2,2>): per phase p (4) and pixel split s (8)
//     part[s][p][m][n] = sum_{k in split s} x[(k + shift(m / Cin)) * Cin + m % Cin] * dy_p[k][n]
// with m = tap * Cin + ci (4 taps x 512), n < 256, 8192 pixels per phase - and the SAME inner loop (k-major LDS tiles,
// ds_read_b32 fragments, v_mfma_f32_32x32x2_f32) under different pipeline structures:
//   NBUF / MODE : 2 buffers + full drain + __syncthreads (what gemm.hip ships)  vs  a ring of NBUF buffers, loads
//                 NBUF-1 tiles ahead, counted s_waitcnt vmcnt(n) + raw s_barrier
//   UNITS       : work units (tile, split, phase) per workgroup; > 1 = persistent workgroup, the load ring runs on across the
//                 unit boundary so that the next unit's first tiles are in LDS while the epilogue stores drain
//   waves       : 4 (128x128) or 8 (256x128) per workgroup;  OCC = launch-bounds occupancy
//   EPI         : 0 no stores (accumulators kept alive: the K loop alone), 1 lean buffer stores
// Prints TFLOP/s of executed MFMA work; every variant is checked against a host fp64 sum on sampled outputs.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gemm_lab.hip -o tools/gemm_lab
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr unsigned OOB = 0x80000000u;
constexpr int BK = 16;

struct Prob {
    const float* x; const float* dy; float* part;
    int Cin, Cout, ntap, Kpix, nsplit, pchunk;   // GEMM M = ntap * Cin, N = Cout, K = Kpix per phase
    int sh0, sh1, sh2, sh3;                      // pixel shift per tap
    int ntm, ntn;                                // tiles along M, N
};

__device__ __forceinline__ void glds16(__amdgpu_buffer_rsrc_t r, float* lds, unsigned voff, int soff) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 16, voff, soff, 0, 0);
#endif
}

template <int N> __device__ __forceinline__ void wait_vm() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
    else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else static_assert(N < 0, "add the count");
}

// device-pass only: a body the host pass cannot compile makes it drop the kernel's launch stub without a word
__device__ __forceinline__ void keep_alive(const f32x16& v) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" :: "v"(v));
#endif
}
__device__ __forceinline__ void raw_barrier() {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_s_barrier();
#endif
}

struct Unit { int m0, n0, split, phase, soa, sob; };

template <int BM, int BN, int WM, int WN, int NBUF, int MODE, int EPI, int OCC, int PF = 0, int KG = 1>
__global__ __launch_bounds__(WM * WN * 64 * KG, OCC) void tn_lab(Prob p, int units_per_wg, int nunits) {
    constexpr int NW = WM * WN, NT = NW * 64;   // per wave group
    constexpr int MI = BM / WM / 32, NI = BN / WN / 32;
    constexpr int A_TILE = BK * BM, B_TILE = BK * BN, STAGE = A_TILE + B_TILE;
    constexpr int AVEC = BM / 4, ARPP = NT / AVEC, APASS = BK / ARPP, ARPW = 64 / AVEC;
    constexpr int BVEC = BN / 4, BRPP = NT / BVEC, BPASS = BK / BRPP, BRPW = 64 / BVEC;
    static_assert(APASS >= 1 && BPASS >= 1 && APASS * ARPP == BK && BPASS * BRPP == BK, "whole passes");
    constexpr int L = APASS + BPASS;           // LDS-DMA instructions per wave per K tile
    constexpr int D = MODE == 0 ? 1 : NBUF - 1;   // load distance in K tiles
    static_assert(MODE == 1 || NBUF == 2, "drain mode is the 2-buffer scheme");
    __shared__ __attribute__((aligned(16))) float smem_all[KG * NBUF * STAGE];

    // KG > 1: the workgroup's KG wave groups take consecutive K sub-ranges of the SAME output tile (own LDS ring each) and add their
    // accumulators through LDS at the end - split-K without partial sums in HBM
    const int grp = KG == 1 ? 0 : __builtin_amdgcn_readfirstlane((int)threadIdx.x / NT);
    float* smem = smem_all + grp * NBUF * STAGE;
    const int tid = KG == 1 ? (int)threadIdx.x : (int)threadIdx.x - grp * NT;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int wm0 = (wave / WN) * (BM / WM), wn0 = (wave % WN) * (BN / WN);
    const int T = p.pchunk / KG / BK;
    const int a_mv = tid % AVEC, a_kr = tid / AVEC, b_nv = tid % BVEC, b_kr = tid / BVEC;
    unsigned avoff[APASS], bvoff[BPASS];
#pragma unroll
    for (int q = 0; q < APASS; ++q) avoff[q] = (unsigned)((a_kr + q * ARPP) * p.Cin + 4 * a_mv) * 4u;
#pragma unroll
    for (int q = 0; q < BPASS; ++q) bvoff[q] = (unsigned)((b_kr + q * BRPP) * 2 * p.Cout + 4 * b_nv) * 4u;
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsd = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsp = __builtin_amdgcn_make_buffer_rsrc((void*)p.part, 0, 0x7fffffff, 0x00020000);

    // unit u -> (tile, split, phase), all wave-uniform
    auto unit = [&](int j) {
        int u = blockIdx.x + j * gridDim.x;
        if (u >= nunits) u = nunits - 1;
        const int tiles = p.ntm * p.ntn;
        const int t = u % tiles, r = u / tiles;
        Unit U;
        U.split = r % p.nsplit; U.phase = r / p.nsplit;
        U.m0 = (t / p.ntn) * BM; U.n0 = (t % p.ntn) * BN;
        const int tap = U.m0 / p.Cin, ci0 = U.m0 - tap * p.Cin;
        const int sh = tap == 0 ? p.sh0 : (tap == 1 ? p.sh1 : (tap == 2 ? p.sh2 : p.sh3));
        const int ps = U.split * p.pchunk + grp * (p.pchunk / KG);
        U.soa = ((ps + sh) * p.Cin + ci0) * 4;
        U.sob = (((U.phase >> 1) * 2 * p.Kpix + 2 * ps + (U.phase & 1)) * p.Cout + U.n0) * 4;
        return U;
    };

    const int total = units_per_wg * T;
    int dj = 0, dt = 0;            // load stream: unit index, K tile within the unit
    Unit DU = unit(0);
    auto dma = [&](auto bufc) __attribute__((always_inline)) {
        constexpr int buf = decltype(bufc)::value;
        float* A = smem + buf * STAGE;
        float* B = A + A_TILE;
        const int soa = DU.soa + dt * BK * p.Cin * 4;
        const int sob = DU.sob + dt * BK * 2 * p.Cout * 4;
#pragma unroll
        for (int q = 0; q < APASS; ++q) glds16(rsx, A + (q * ARPP + wave * ARPW) * BM, avoff[q], soa);
#pragma unroll
        for (int q = 0; q < BPASS; ++q) glds16(rsd, B + (q * BRPP + wave * BRPW) * BN, bvoff[q], sob);
        if (++dt == T) { dt = 0; ++dj; DU = unit(dj); }
    };

    f32x16 acc[MI][NI];
    auto zero = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    };
    zero();
    int cj = 0, ct = 0;
    Unit CU = DU;
    float keep = 0.f;

    auto epilogue = [&]() __attribute__((always_inline)) {
        if (EPI == 0) {
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) keep_alive(acc[i][j]);
        } else {
            if constexpr (KG > 1) {
                static_assert(KG == 2 || KG == 4, "tree of wave groups");
                float4* red = reinterpret_cast<float4*>(smem_all);
                constexpr int RT = MI * NI * 4 * NT;     // float4 per region
                __syncthreads();
                for (int stride = KG / 2; stride >= 1; stride /= 2) {
                    if (grp >= stride && grp < 2 * stride) {
#pragma unroll
                        for (int i = 0; i < MI; ++i)
#pragma unroll
                            for (int j = 0; j < NI; ++j)
#pragma unroll
                                for (int q = 0; q < 4; ++q)
                                    red[(grp - stride) * RT + ((i * NI + j) * 4 + q) * NT + tid] =
                                        make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
                    }
                    __syncthreads();
                    if (grp < stride) {
#pragma unroll
                        for (int i = 0; i < MI; ++i)
#pragma unroll
                            for (int j = 0; j < NI; ++j)
#pragma unroll
                                for (int q = 0; q < 4; ++q) {
                                    const float4 v = red[grp * RT + ((i * NI + j) * 4 + q) * NT + tid];
                                    acc[i][j][4 * q] += v.x; acc[i][j][4 * q + 1] += v.y; acc[i][j][4 * q + 2] += v.z; acc[i][j][4 * q + 3] += v.w;
                                }
                    }
                    __syncthreads();
                }
            }
            if (KG == 1 || grp == 0) {
            const int M = p.ntap * p.Cin;
            const long rowbase = (long)(CU.split * 4 + CU.phase) * M;
            const unsigned vo = ((unsigned)(rowbase + CU.m0 + wm0 + 4 * h) * (unsigned)p.Cout + (unsigned)(CU.n0 + wn0 + l31)) * 4u;
            const int c4 = p.Cout * 4;
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int soff = (i * 32 + (r & 3) + 8 * (r >> 2)) * c4;
#pragma unroll
                    for (int j = 0; j < NI; ++j) {
                        if (EPI == 3 && (r & 3)) continue;
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[i][j][r]), rsp, (int)(vo + j * 128), soff, EPI == 2 ? 2 : 0);
                    }
                }
            }
        }
    };

    // prologue: D tiles in flight, the first one landed
    if constexpr (D >= 1) dma(std::integral_constant<int, 0>{});
    if constexpr (D >= 2) { if (1 < total) dma(std::integral_constant<int, 1>{}); }
    if constexpr (D >= 3) { if (2 < total) dma(std::integral_constant<int, 2>{}); }
    if constexpr (MODE == 0) { wait_vm<0>(); __syncthreads(); }
    else {
        // the oldest tile's L loads done; up to (D-1)*L younger ones may stay in flight (fewer were issued if total < D: harmless)
        wait_vm<(D - 1) * L>();
        raw_barrier();
    }

    auto step = [&](auto bufc, int f) __attribute__((always_inline)) {
        constexpr int buf = decltype(bufc)::value;
        constexpr int nb = (buf + D) % NBUF;
        if (f + D < total) dma(std::integral_constant<int, nb>{});
        const float* A = smem + buf * STAGE + wm0 + l31;
        const float* B = smem + buf * STAGE + A_TILE + wn0 + l31;
        if constexpr (PF == 0) {
#pragma unroll
            for (int kk = 0; kk < BK; kk += 2) {
                float av[MI], bv[NI];
#pragma unroll
                for (int i = 0; i < MI; ++i) av[i] = A[(kk + h) * BM + i * 32];
#pragma unroll
                for (int j = 0; j < NI; ++j) bv[j] = B[(kk + h) * BN + j * 32];
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
            }
        } else {
            // fragments of k pair g+1 requested BEFORE the MFMAs of pair g (two register sets): the LDS latency hides under 4 MFMAs
            float av[2][MI], bv[2][NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) av[0][i] = A[h * BM + i * 32];
#pragma unroll
            for (int j = 0; j < NI; ++j) bv[0][j] = B[h * BN + j * 32];
#pragma unroll
            for (int kk = 0; kk < BK; kk += 2) {
                constexpr int dummy = 0; (void)dummy;
                const int s = (kk >> 1) & 1;
                if (kk + 2 < BK) {
#pragma unroll
                    for (int i = 0; i < MI; ++i) av[s ^ 1][i] = A[(kk + 2 + h) * BM + i * 32];
#pragma unroll
                    for (int j = 0; j < NI; ++j) bv[s ^ 1][j] = B[(kk + 2 + h) * BN + j * 32];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s][i], bv[s][j], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (++ct == T) {           // unit complete (wave-uniform)
            epilogue();
            zero();
            ct = 0; ++cj; CU = unit(cj);
        }
        if constexpr (MODE == 0) { wait_vm<0>(); __syncthreads(); }
        else {
            // loads of tile f+1 done (they are the oldest outstanding); the (D-1)*L younger ones stay in flight across the barrier
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            wait_vm<(D - 1) * L>();
            raw_barrier();
        }
    };
    for (int f = 0; f < total; f += NBUF) {
        step(std::integral_constant<int, 0>{}, f);
        if constexpr (NBUF > 1) { if (f + 1 < total) step(std::integral_constant<int, 1>{}, f + 1); }
        if constexpr (NBUF > 2) { if (f + 2 < total) step(std::integral_constant<int, 2>{}, f + 2); }
        if constexpr (NBUF > 3) { if (f + 3 < total) step(std::integral_constant<int, 3>{}, f + 3); }
    }
    if (keep == 12345.678f) p.part[0] = keep;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Host {
    std::vector<float> x, dy;
    float *dx, *ddy, *dpart;
    Prob p;
    size_t part_floats;
};

static double check(Host& H, int nsample) {
    std::vector<float> part(H.part_floats);
    CK(hipMemcpy(part.data(), H.dpart, H.part_floats * 4, hipMemcpyDeviceToHost));
    const Prob& p = H.p;
    const int M = p.ntap * p.Cin;
    const int sh[4] = {p.sh0, p.sh1, p.sh2, p.sh3};
    double worst = 0;
    unsigned s = 12345;
    for (int i = 0; i < nsample; ++i) {
        s = s * 1664525u + 1013904223u; const int split = (s >> 8) % p.nsplit;
        s = s * 1664525u + 1013904223u; const int phase = (s >> 8) % 4;
        s = s * 1664525u + 1013904223u; const int m = (s >> 8) % M;
        s = s * 1664525u + 1013904223u; const int n = (s >> 8) % p.Cout;
        const int tap = m / p.Cin, ci = m % p.Cin;
        double ref = 0;
        for (int k = split * p.pchunk; k < (split + 1) * p.pchunk; ++k) {
            const long xr = (long)(k + sh[tap]) * p.Cin + ci;
            const long dr = ((long)(phase >> 1) * 2 * p.Kpix + 2 * k + (phase & 1)) * p.Cout + n;
            ref += (double)H.x[xr] * (double)H.dy[dr];
        }
        const double got = part[((size_t)(split * 4 + phase) * M + m) * p.Cout + n];
        worst = fmax(worst, fabs(got - ref));
    }
    return worst;
}

template <typename K>
static void run(Host& H, const char* label, K kern, int threads, int grid, int units_per_wg, int nunits, bool epi) {
    CK(hipMemset(H.dpart, 0xff, H.part_floats * 4));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), 0, 0, H.p, units_per_wg, nunits);
    CK(hipDeviceSynchronize());
    const double err = epi ? check(H, 400) : -1.0;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f, sum = 0.f;
    const int R = 5, IT = 10;
    for (int r = 0; r < R; ++r) {
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < IT; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), 0, 0, H.p, units_per_wg, nunits);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms = 0.f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= IT;
        best = fminf(best, ms); sum += ms;
    }
    const Prob& p = H.p;
    const double flop = 2.0 * 4 * (double)p.Kpix * (p.ntap * p.Cin) * p.Cout;
    printf("%-78s %7.1f us (best %6.1f)  %6.1f TF  %.3f   err %.2e\n", label, 1e3 * sum / R, 1e3 * best, flop / (sum / R * 1e-3) * 1e-12,
           flop / (sum / R * 1e-3) / 157.3e12, err);
    fflush(stdout);
}

int main(int argc, char** argv) {
    Host H;
    Prob& p = H.p;
    p.Cin = 512; p.Cout = 256; p.ntap = 4; p.Kpix = 8192; p.nsplit = argc > 1 ? atoi(argv[1]) : 8;
    p.pchunk = p.Kpix / p.nsplit;
    p.sh0 = 0; p.sh1 = 1; p.sh2 = 8; p.sh3 = 9;
    const size_t xf = (size_t)(p.Kpix + 16) * p.Cin, dyf = (size_t)4 * p.Kpix * p.Cout;
    H.x.resize(xf); H.dy.resize(dyf);
    unsigned s = 1;
    for (auto& v : H.x) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 65536.f - 0.5f; }
    for (auto& v : H.dy) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 65536.f - 0.5f; }
    const int M = p.ntap * p.Cin;
    H.part_floats = (size_t)p.nsplit * 4 * M * p.Cout;
    CK(hipMalloc(&H.dx, xf * 4)); CK(hipMalloc(&H.ddy, dyf * 4)); CK(hipMalloc(&H.dpart, H.part_floats * 4));
    CK(hipMemcpy(H.dx, H.x.data(), xf * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(H.ddy, H.dy.data(), dyf * 4, hipMemcpyHostToDevice));
    p.x = H.dx; p.dy = H.ddy; p.part = H.dpart;
    printf("TN lab: 4 phases x [%d x %d]^T.[%d x %d], %d pixel splits (T = %d K tiles of %d per unit)\n", M, p.Kpix, p.Kpix, p.Cout, p.nsplit,
           p.pchunk / BK, BK);

#define RUNK(BM_, BN_, WM_, WN_, NBUF_, MODE_, EPI_, OCC_, UNITS_, PF_, KG_, label)                                                     \
    {                                                                                                                          \
        p.ntm = M / BM_; p.ntn = p.Cout / BN_;                                                                                 \
        const int nunits = p.ntm * p.ntn * p.nsplit * 4;                                                                       \
        run(H, label, tn_lab<BM_, BN_, WM_, WN_, NBUF_, MODE_, EPI_, OCC_, PF_, KG_>, WM_ * WN_ * 64 * KG_, nunits / UNITS_, UNITS_, nunits, EPI_ == 1 || EPI_ == 2); \
    }
#define RUNP(BM_, BN_, WM_, WN_, NBUF_, MODE_, EPI_, OCC_, UNITS_, PF_, label) RUNK(BM_, BN_, WM_, WN_, NBUF_, MODE_, EPI_, OCC_, UNITS_, PF_, 1, label)
#define RUN(BM_, BN_, WM_, WN_, NBUF_, MODE_, EPI_, OCC_, UNITS_, label) RUNP(BM_, BN_, WM_, WN_, NBUF_, MODE_, EPI_, OCC_, UNITS_, 0, label)
    // ---- what ships: 128x128, 4 waves, 2 buffers, drain
    RUN(128, 128, 2, 2, 2, 0, 1, 2, 1, "128x128 4w  2buf drain  (shipping structure)");
    RUN(128, 128, 2, 2, 2, 0, 0, 2, 1, "128x128 4w  2buf drain  no epilogue stores");
    RUN(128, 128, 2, 2, 2, 0, 2, 2, 1, "128x128 4w  2buf drain  stores with the nt hint");
    RUN(128, 128, 2, 2, 2, 0, 3, 2, 1, "128x128 4w  2buf drain  a quarter of the stores (timing only)");
    // ---- smaller tiles (use with fewer splits: same workgroup count, fewer partial bytes)
    RUN(128, 64, 2, 2, 2, 0, 1, 2, 1, "128x64  4w (64x32 wave tiles) 2buf drain");
    RUN(128, 64, 4, 1, 2, 0, 1, 2, 1, "128x64  4w (32x64 wave tiles) 2buf drain");
    RUN(64, 128, 2, 2, 2, 0, 1, 2, 1, "64x128  4w (32x64 wave tiles) 2buf drain");
    RUN(64, 128, 1, 4, 2, 0, 1, 2, 1, "64x128  4w (64x32 wave tiles) 2buf drain");
    RUN(128, 64, 2, 2, 2, 0, 0, 2, 1, "128x64  4w (64x32 wave tiles) 2buf drain  no epilogue stores");
    RUN(64, 64, 2, 2, 2, 0, 1, 2, 1, "64x64   4w (32x32 wave tiles) 2buf drain");
    // ---- split-K inside the workgroup: KG wave groups x (128x128, 4 waves), accumulators added through LDS
    RUNK(128, 128, 2, 2, 2, 0, 1, 1, 1, 0, 2, "128x128 2 groups x 4w (8 waves), LDS reduction");
    RUNK(128, 128, 2, 2, 2, 0, 1, 1, 1, 0, 4, "128x128 4 groups x 4w (16 waves), LDS reduction");
    RUNK(128, 128, 2, 2, 2, 0, 0, 1, 1, 0, 4, "128x128 4 groups x 4w (16 waves), no reduction / stores");
    RUNK(128, 64, 2, 2, 2, 0, 1, 1, 1, 0, 2, "128x64  2 groups x 4w (8 waves), LDS reduction");
    RUNK(128, 64, 2, 2, 2, 0, 1, 1, 1, 0, 4, "128x64  4 groups x 4w (16 waves), LDS reduction");
    return 0;
}
