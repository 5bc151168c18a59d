// 3x3 convolutions with <= 3 output planes on the exact-fp32 MFMA, input read ONCE (round 6).
//
// Stands in for cudnn.SpatialConvolution(128, C, 3, 3, 1, 1, 1, 1) (G's last layer, models.lua:222 / :154), for the data gradient of
// nn.SpatialConvolution(C, 64, 3, 3, 1, 1, 1) (D's first layer seen from its output, models.lua:646) and for the weight gradient of the
// former.  These layers move 512 (256) bytes per pixel for 2.3 kFLOP: HBM-bound, 67 MB per launch at batch 128.  Round 1's kernels
// (skinny_conv3x3_kernel / skinny_wgrad3x3_kernel, gemm.hip: still the fallback for widths that are not a multiple of 32) fetched the
// nine taps of every pixel from L2 - 9x read amplification, 0.03-0.17 of the HBM roofline.
//
// Scatter form: y[p][co] = sum_t Z[p + d_t][t, co]  with  Z[q][(t, co)] = sum_c x[q][c] w[t][c][co].
// Z is ONE GEMM over the pixels with 9 CO <= 27 columns and K = Cin, so x needs no halo at all: a row tile of 32 pixels is 32 Cin
// consecutive floats, staged through LDS once and multiplied on v_mfma_f32_32x32x2_f32 (27 of the 32 columns useful: 1.07 GFLOP per
// launch at batch 128, 7 us of MFMA time under >= 11 us of HBM time).  The halo moves to Z, which is 27 floats per pixel instead of 128:
// a workgroup computes Z for R + 2 rows of one image into LDS and adds the nine shifted planes for its R output rows.  The weight
// gradient is the same trick transposed: gradW[t][c][co] = sum_q x[q][c] dy[q - d_t][co] = X^T . DYS with DYS[q][(t, co)] read out of a
// zero-haloed LDS copy of the dy strip (12 bytes per pixel), x streamed through LDS by buffer_load ... lds exactly once.
#include "common.h"
#include <stdlib.h>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float4 ld4g(const float* p) { return *reinterpret_cast<const float4*>(p); }

__device__ __forceinline__ void glds16(__amdgpu_buffer_rsrc_t r, float* lds, unsigned voff, int soff) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 16, voff, soff, 0, 0);
#endif
}

// block -> (image, strip): the strips of one image share halo rows, so they go to ONE XCD (block b runs on XCD b % 8) when the batch allows
__device__ __forceinline__ void strip_of_block(int b, int N, int SP, int& n, int& s) {
    if ((N & 7) == 0) { n = (b & 7) + 8 * (b / (8 * SP)); s = (b >> 3) % SP; }
    else { n = b / SP; s = b - n * SP; }
}

constexpr int kXld = 36;   // floats per pixel of a 32-channel chunk buffer: 16 pixels x 36 dwords hit 16 distinct 4-bank slots (ds_read_b128)

// ---------------------------------------------------------------------------------------------------------------------------------
// forward.  w = the packed forward operand wf[(t * CIN + c) * CO + co]; y[pixel][CO] (+ bias).  One workgroup = R output rows of one
// image; wave v computes the Z row tiles v, v + 4, ... of the R + 2 rows [r0 - 1, r0 + R], everything it touches before the barrier is
// its own (a private 32 x 32-channel chunk buffer), so the four waves run unsynchronised and their loads overlap the others' MFMAs.
// ---------------------------------------------------------------------------------------------------------------------------------
template <int CO, int CIN>
__global__ __launch_bounds__(256, 2) void skinny_mfma_fwd_k(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                            float* __restrict__ y, int N, int H, int W, int R) {
    constexpr int ZS = 9 * CO;        // 27 / 9: odd, so consecutive pixels of one Z column sit in distinct banks
    constexpr int NCH = CIN / 32;
    constexpr int NQ = CIN / 2;       // MFMAs per row tile
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* Bs = sm;                   // [NQ][64]: the B fragment of MFMA q for every lane
    float* xs = sm + NQ * 64;         // [4][32][kXld]
    float* Zs = xs + 4 * 32 * kXld;   // [(R + 2) * W][ZS]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int SP = (H + R - 1) / R, NT = W >> 5;
    int n, s;
    strip_of_block((int)blockIdx.x, N, SP, n, s);
    const int r0 = s * R;
    const int ntiles = (R + 2) * NT;
    const int lp = lane >> 3, c4 = lane & 7;           // staging: pixels lp + 8 i, float4 c4 of the chunk

    auto next_valid = [&](int t) {                     // this wave's next row tile inside the image (wave-uniform)
        while (t < ntiles) {
            const int row = r0 - 1 + t / NT;
            if (row >= 0 && row < H) break;
            t += 4;
        }
        return t;
    };
    auto load_tile = [&](int tile, float4 (&st)[NCH][4]) {
        const int lrow = tile / NT, xb = (tile - lrow * NT) << 5;
        const float* src = x + (((long)n * H + (r0 - 1 + lrow)) * W + xb) * CIN + lp * CIN + c4 * 4;
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int i = 0; i < 4; ++i) st[c][i] = ld4g(src + (long)i * 8 * CIN + c * 32);
    };

    float4 sa[NCH][4], sb[NCH][4];
    int t0 = next_valid(wave);
    if (t0 < ntiles) load_tile(t0, sa);                // in flight under the B set-up below

    // B fragments: MFMA q = (chunk, sq, jj) contracts channels chunk * 32 + sq * 8 + {jj, 4 + jj} (lane half h takes the second), so that a
    // lane's four A values of one sq are ONE 16-byte LDS read.  w (9 CIN CO floats) is copied coalesced into the chunk buffers (still unused)
    // and re-ordered from there.
    for (int i = tid; i < 9 * CIN * CO; i += 256) xs[i] = w[i];     // 9 CIN CO <= 4 x 32 x kXld floats
    __syncthreads();
    for (int i = tid; i < NQ * 64; i += 256) {
        const int q = i >> 6, l = i & 63, jj = l & 31, hh = l >> 5;
        const int ch = (q >> 4) * 32 + ((q >> 2) & 3) * 8 + hh * 4 + (q & 3);
        Bs[i] = jj < ZS ? xs[((jj / CO) * CIN + ch) * CO + (jj % CO)] : 0.f;
    }
    __syncthreads();

    float* xw = xs + wave * 32 * kXld;
    const float* bl = Bs + lane;
    auto compute = [&](int tile, float4 (&st)[NCH][4]) {
        const int lrow = tile / NT, xb = (tile - lrow * NT) << 5;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
#pragma unroll
            for (int i = 0; i < 4; ++i) *reinterpret_cast<float4*>(xw + (lp + 8 * i) * kXld + c4 * 4) = st[c][i];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            float4 av[4];
#pragma unroll
            for (int sq = 0; sq < 4; ++sq) av[sq] = *reinterpret_cast<const float4*>(xw + j * kXld + sq * 8 + h * 4);
#pragma unroll
            for (int sq = 0; sq < 4; ++sq) {
                const int q = c * 16 + sq * 4;
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[sq].x, bl[(q + 0) * 64], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[sq].y, bl[(q + 1) * 64], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[sq].z, bl[(q + 2) * 64], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[sq].w, bl[(q + 3) * 64], acc, 0, 0, 0);
            }
            __builtin_amdgcn_wave_barrier();           // the next chunk's stores stay behind these reads
        }
        if (j < ZS) {
            float* zt = Zs + ((long)lrow * W + xb + 4 * h) * ZS + j;
#pragma unroll
            for (int r = 0; r < 16; ++r) zt[((r & 3) + 8 * (r >> 2)) * ZS] = acc[r];
        }
    };
    // two register sets: the next tile's 32 CIN floats are requested before the current tile's MFMAs
    while (t0 < ntiles) {
        const int t1 = next_valid(t0 + 4);
        if (t1 < ntiles) load_tile(t1, sb);
        compute(t0, sa);
        if (t1 >= ntiles) break;
        const int t2 = next_valid(t1 + 4);
        if (t2 < ntiles) load_tile(t2, sa);
        compute(t1, sb);
        t0 = t2;
    }
    __syncthreads();

    const int rows = min(R, H - r0);
    float bv[CO];
#pragma unroll
    for (int co = 0; co < CO; ++co) bv[co] = bias ? bias[co] : 0.f;
    for (int idx = tid; idx < rows * W; idx += 256) {
        const int orow = idx / W, ox = idx - orow * W;
        const int oy = r0 + orow;
        float o[CO];
#pragma unroll
        for (int co = 0; co < CO; ++co) o[co] = 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int iy = oy + t / 3 - 1, ix = ox + t % 3 - 1;
            if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
                const float* z = Zs + ((orow + t / 3) * W + ix) * ZS + t * CO;
#pragma unroll
                for (int co = 0; co < CO; ++co) o[co] += z[co];
            }
        }
        float* yo = y + (((long)n * H + oy) * W + ox) * CO;
#pragma unroll
        for (int co = 0; co < CO; ++co) yo[co] = o[co] + bv[co];
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// weight + bias gradient partials, the layout skinny_wgrad3x3_kernel wrote: part[block][(t * CIN + c) * CO + co], bias_part[block][co].
// One workgroup walks strips of R rows of one image: dy strip (with halo, zero outside the image) -> LDS, then the strip's x rows as
// 32-pixel stages (32 CIN consecutive floats each) by LDS-direct loads, two buffers.  Wave = (32-channel slice, K part of the stage).
// ---------------------------------------------------------------------------------------------------------------------------------
constexpr int kDylFloats = 4096;

template <int CO, int CIN>
__global__ __launch_bounds__(256, 2) void skinny_mfma_wgrad_k(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ part,
                                                              float* __restrict__ bias_part, int N, int H, int W, int R, int nstrips) {
    constexpr int ZS = 9 * CO;
    constexpr int NS = CIN / 32, KP = 4 / NS, PPK = 32 / KP;       // slices, K parts, pixels of a stage per K part
    constexpr int CALLS = 32 * CIN / 4 / 256;                      // 16-byte LDS-direct loads per lane and stage
    __shared__ __attribute__((aligned(16))) float xs0[32 * CIN];
    __shared__ __attribute__((aligned(16))) float xs1[32 * CIN];
    __shared__ __attribute__((aligned(16))) float dyl[kDylFloats];  // [(R + 2)][(W + 2)][CO]
    __shared__ float bsh[4][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int slice = wave % NS, kp = wave / NS;
    const int SP = (H + R - 1) / R, NT = W >> 5, W2 = W + 2;
    // lane constant of the B address: column j = (t, co) reads dy[q - d_t][co], d_t = (t / 3 - 1, t % 3 - 1); halo origin (-1, -1)
    const int jt = j < ZS ? j / CO : 4, jco = j < ZS ? j % CO : 0;
    const int jc = ((2 - jt / 3) * W2 + (2 - jt % 3)) * CO + jco;

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float bacc[CO];
#pragma unroll
    for (int co = 0; co < CO; ++co) bacc[co] = 0.f;

    for (int st = blockIdx.x; st < nstrips; st += gridDim.x) {
        const int n = st / SP, r0 = (st - n * SP) * R;
        const int rows = min(R, H - r0);
        const float* xg = x + (((long)n * H + r0) * W) * CIN;      // rows r0 .. r0 + rows - 1: rows * W * CIN consecutive floats
        const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void*)xg, 0, 0x7fffffff, 0x00020000);
        const int nstage = rows * NT;
        auto dma = [&](int k, float* buf) {
#pragma unroll
            for (int q = 0; q < CALLS; ++q) {
                const int blk = wave * CALLS + q;                   // 256 floats each
                glds16(rsx, buf + blk * 256, (unsigned)lane * 16u, (k * 32 * CIN + blk * 256) * 4);
            }
        };
        __syncthreads();                                            // the previous strip's readers of dyl / xs are done
        dma(0, xs0);
        for (int idx = tid; idx < (R + 2) * W2 * CO; idx += 256) {
            const int lr = idx / (W2 * CO), rem = idx - lr * (W2 * CO);
            const int lc = rem / CO, co = rem - lc * CO;
            const int iy = r0 - 1 + lr, ix = lc - 1;
            dyl[idx] = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? dy[(((long)n * H + iy) * W + ix) * CO + co] : 0.f;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int idx = tid; idx < rows * W; idx += 256) {           // gradBias: this strip's own pixels
            const int a = idx / W, qx = idx - a * W;
            const float* d = dyl + ((a + 1) * W2 + qx + 1) * CO;
#pragma unroll
            for (int co = 0; co < CO; ++co) bacc[co] += d[co];
        }
        int a = 0, xt = 0;
        for (int k = 0; k < nstage; ++k) {
            float* cur = (k & 1) ? xs1 : xs0;
            if (k + 1 < nstage) dma(k + 1, (k & 1) ? xs0 : xs1);
            const float* A = cur + (kp * PPK + h) * CIN + slice * 32 + j;
            const float* B = dyl + (a * W2 + (xt << 5) + kp * PPK + h) * CO + jc;
#pragma unroll
            for (int s2 = 0; s2 < PPK / 2; ++s2)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[2 * s2 * CIN], B[2 * s2 * CO], acc, 0, 0, 0);
            if (++xt == NT) { xt = 0; ++a; }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }

    // the K parts of a slice (CIN = 64: two) meet in LDS, added in a fixed order
    if (KP > 1) {
        __syncthreads();
        if (kp > 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) xs0[(((kp - 1) * NS + slice) * 16 + r) * 64 + lane] = acc[r];
        }
        __syncthreads();
        if (kp == 0) {
#pragma unroll
            for (int p = 1; p < KP; ++p)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] += xs0[(((p - 1) * NS + slice) * 16 + r) * 64 + lane];
        }
    }
    if (kp == 0 && j < ZS) {
        float* po = part + (long)blockIdx.x * 9 * CIN * CO + ((long)(j / CO) * CIN + slice * 32 + 4 * h) * CO + (j % CO);
#pragma unroll
        for (int r = 0; r < 16; ++r) po[((r & 3) + 8 * (r >> 2)) * CO] = acc[r];
    }
    if (bias_part) {
#pragma unroll
        for (int co = 0; co < CO; ++co) {
            const float v = cg::wave_sum(bacc[co]);
            if (lane == 0) bsh[wave][co] = v;
        }
        __syncthreads();
        if (tid < CO) bias_part[(long)blockIdx.x * CO + tid] = ((bsh[0][tid] + bsh[1][tid]) + bsh[2][tid]) + bsh[3][tid];
    }
}

constexpr size_t kFwdLdsCap = 96 * 1024;   // dynamic LDS of the forward kernel (hipFuncAttributeMaxDynamicSharedMemorySize)
size_t fwd_fixed_floats(int Cin) { return (size_t)Cin / 2 * 64 + 4 * 32 * kXld; }
int fwd_row_cap(int W, int Cin, int CO) { return (int)((kFwdLdsCap / 4 - fwd_fixed_floats(Cin)) / ((size_t)W * 9 * CO)) - 2; }
// Rows per workgroup: a strip computes Z for R + 2 rows (the halo costs MFMA time and L2 reads, not HBM reads: the strips of an image
// share an XCD) and the launch wants two workgroups per CU, all resident at once.  Measured at 32 x 32 x 128 planes (profiles/r06_skinny.txt):
// batch 128: R = 8 and 10 (512 workgroups) 25.5 us, R = 6 (768) 37 us, R = 14 (384) 35 us; batch 64: R = 4, 6, 10 all 17-18 us.
int fwd_rows(int N, int H, int W, int Cin, int CO) {
    const int cap = std::min(fwd_row_cap(W, Cin, CO), H);
    const int want = (int)std::min<long>(cap, (long)N * H / (2 * cg::kNumCU));
    return std::max(want, std::min(cap, 2));
}

}  // namespace

namespace cg {

bool skinny_mfma_ok(int Cin, int Cout, int H, int W) {
    if (!((Cin == 64 || Cin == 128) && (Cout == 1 || Cout == 3) && W % 32 == 0 && W <= 128 && H >= 1)) return false;
    return fwd_row_cap(W, Cin, Cout) >= 1 && 3 * (W + 2) * Cout <= kDylFloats;
}

int skinny_mfma_forward(hipStream_t st, const float* x, const float* w, const float* bias, float* y, int N, int H, int W, int Cin, int Cout) {
    const int R = fwd_rows(N, H, W, Cin, Cout);
    const int SP = (H + R - 1) / R;
    const size_t shb = (fwd_fixed_floats(Cin) + (size_t)(R + 2) * W * 9 * Cout) * sizeof(float);
    const dim3 grid((unsigned)(N * SP));
#define CG_SK_FWD(CO, CI)                                                                                                              \
    do {                                                                                                                               \
        static bool attr = false;                                                                                                      \
        if (!attr) {                                                                                                                   \
            CG_HIP(hipFuncSetAttribute((const void*)skinny_mfma_fwd_k<CO, CI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFwdLdsCap)); \
            attr = true;                                                                                                               \
        }                                                                                                                              \
        hipLaunchKernelGGL((skinny_mfma_fwd_k<CO, CI>), grid, dim3(256), shb, st, x, w, bias, y, N, H, W, R);                          \
    } while (0)
    if (Cout == 3 && Cin == 128) CG_SK_FWD(3, 128);
    else if (Cout == 3 && Cin == 64) CG_SK_FWD(3, 64);
    else if (Cout == 1 && Cin == 128) CG_SK_FWD(1, 128);
    else if (Cout == 1 && Cin == 64) CG_SK_FWD(1, 64);
    else return cg::fail("skinny_mfma_forward: no kernel for %d -> %d planes", Cin, Cout);
#undef CG_SK_FWD
    return 0;
}

// blocks <= max_blocks partial planes; returns the number written (the reduce adds exactly those) or -1
int skinny_mfma_wgrad(hipStream_t st, const float* x, const float* dy, float* part, float* bias_part, int N, int H, int W, int Cin, int Cout,
                      int max_blocks) {
    int R = 8;
    while (R > 1 && (R + 2) * (W + 2) * Cout > kDylFloats) R >>= 1;
    const int SP = (H + R - 1) / R;
    const long nstrips = (long)N * SP;
    const int blocks = (int)std::min<long>(nstrips, max_blocks);
#define CG_SK_WG(CO, CI) hipLaunchKernelGGL((skinny_mfma_wgrad_k<CO, CI>), dim3(blocks), dim3(256), 0, st, x, dy, part, bias_part, N, H, W, R, (int)nstrips)
    if (Cout == 3 && Cin == 128) CG_SK_WG(3, 128);
    else if (Cout == 3 && Cin == 64) CG_SK_WG(3, 64);
    else if (Cout == 1 && Cin == 128) CG_SK_WG(1, 128);
    else if (Cout == 1 && Cin == 64) CG_SK_WG(1, 64);
    else { cg::fail("skinny_mfma_wgrad: no kernel for %d -> %d planes", Cin, Cout); return -1; }
#undef CG_SK_WG
    return blocks;
}

}  // namespace cg
