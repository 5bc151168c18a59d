// Fused-transform Winograd F(2x2,3x3) for the 64 -> 64 plane 3x3 convolutions of D32_st3 (models.lua:648,655,664,673: the layers at
// 32 x 32 and 16 x 16; updateOutput and, with flipped filters, updateGradInput) - round 6, VERDICT r05 #1, developed in tools/wino_lab.hip.
//
// With K = 64 an unfused Winograd pipeline writes and re-reads a V four times the size of the input: memory-bound at the time the direct
// kernel (igemm_nn_kernel<128,64>, 0.63 of the fp32 MFMA peak) takes.  Here only x, y and the transformed filters U (256 KB, L2) move:
//   workgroup = 32 output tiles (8 x 16 pixels) of one image; its 10 x 18 pixel input patch (zero outside the image) goes to LDS once;
//   wave      = ONE position row xi of the 4 x 4 Winograd positions x ONE 32-column tile of the output planes (8 waves; 113 VGPRs: four waves
//               per SIMD).  Per position the A operand B^T d B is formed in registers from 16-byte LDS reads (the row transform is shared
//               by the row's four positions), the B operand U[pos] is a 16-byte buffer load from L2 with an SGPR offset, 32
//               v_mfma_f32_32x32x2_f32 contract the 64 input planes; the row's two output-transform sums T_b = sum_nu A[nu][b] M[xi][nu]
//               stay in registers, the four rows meet in LDS (the dead patch) and every wave finishes one of the tile's four outputs;
//   lanes -> tiles are permuted so that every 16-lane group of ds_read_b128 covers two whole tile rows whose plane quads are XORed by the
//               row-pair parity: 16 distinct 16-byte slots per group, no bank conflicts (PMC: SQ_LDS_BANK_CONFLICT 0).
// Executed MFMA work = 16 / 36 of the direct count; results differ from the direct kernel by fp32 re-association only (the transforms'
// coefficients are 0, +-1, +-1/2).  Measured alone at batch 128 (profiles/r06_wino_lab.txt): 32 x 32: 61 us against 96 us direct.
//
// The transformed filters live in a library-owned cache keyed by the PACKED operand's address: cg_pack_conv_weight* (gemm.hip) call
// wino3_note_pack for every 64 -> 64 3x3 layer they pack, which (re)computes U for the forward operand wf and the flipped, transposed U
// for the data-gradient operand wb from the canonical weights in the same stream; cg_conv2d_forward_grouped looks its operands up and
// takes this path when every group's operand is known.  No new entry point, nothing for a host to manage.
#include "common.h"
#include <mutex>
#include <unordered_map>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int C = 64;          // input planes = output planes
constexpr int PLD = 68;        // floats per patch pixel
constexpr int PW = 18, PH = 10;
constexpr int kUFloats = 16 * C * C;

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 bufld4(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    const f32x4 f = __builtin_bit_cast(f32x4, v);        // cast the WHOLE vector (see gemm.hip)
    return make_float4(f.x, f.y, f.z, f.w);
}

// U[pos][ci / 4][co][4] = (G g G^T)[xi][nu], pos = xi * 4 + nu, for the filter g of (input plane ci, output plane co) of the convolution
// the operand serves: forward g = w[co][ci][:][:]; data gradient (planes swapped, taps flipped) g[a][b] = w[ci][co][2 - a][2 - b].
struct PackJobs { const float* w[8]; float* uf[8]; float* ub[8]; int n; };
__global__ void wino3_pack_k(PackJobs jobs) {
    const float* __restrict__ w = jobs.w[blockIdx.y];
    const int idx = blockIdx.x * 256 + threadIdx.x;      // < C * C
    const int co = idx / C, ci = idx % C;
    const float G[4][3] = {{1.f, 0.f, 0.f}, {.5f, .5f, .5f}, {.5f, -.5f, .5f}, {0.f, 0.f, 1.f}};
    for (int flip = 0; flip < 2; ++flip) {
        float* U = flip ? jobs.ub[blockIdx.y] : jobs.uf[blockIdx.y];
        if (!U) continue;
        float g[3][3];
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) g[a][b] = flip ? w[((long)ci * C + co) * 9 + (2 - a) * 3 + (2 - b)] : w[((long)co * C + ci) * 9 + a * 3 + b];
        float t[4][3];
        for (int i = 0; i < 4; ++i)
            for (int b = 0; b < 3; ++b) t[i][b] = G[i][0] * g[0][b] + G[i][1] * g[1][b] + G[i][2] * g[2][b];
        for (int i = 0; i < 4; ++i)
            for (int k = 0; k < 4; ++k)
                U[(((long)(i * 4 + k) * (C / 4) + ci / 4) * C + co) * 4 + (ci & 3)] = t[i][0] * G[k][0] + t[i][1] * G[k][1] + t[i][2] * G[k][2];
    }
}

// row i of an MFMA tile (= lane & 31 of the A operand) -> output tile (ty, tx) of the 4 x 8 block: ty = result >> 3, tx = result & 7.
// The 16-lane groups of ds_read_b128 are lanes {0-3, 12-15, 20-27} / {4-11, 16-19, 28-31} (+32): the first takes tile rows 0, 1, the second 2, 3
constexpr int tile_of_row_c(int l) {
    return ((l < 4 || (l >= 12 && l < 16) || (l >= 20 && l < 28)) ? 0 : 16) +
           ((l < 4 || (l >= 12 && l < 16) || (l >= 20 && l < 28)) ? (l < 4 ? l : (l < 16 ? l - 8 : l - 12)) : (l < 12 ? l - 4 : (l < 20 ? l - 8 : l - 16)));
}
__device__ __forceinline__ int tile_of_row(int l) {
    const bool ga = l < 4 || (l >= 12 && l < 16) || (l >= 20 && l < 28);
    const int rank = ga ? (l < 4 ? l : (l < 16 ? l - 8 : l - 12)) : (l < 12 ? l - 4 : (l < 20 ? l - 8 : l - 16));
    return (ga ? 0 : 16) + rank;
}

template <int XI>
__device__ __forceinline__ void wino_row(const float* P, int tyl, int txl, int h, __amdgpu_buffer_rsrc_t rsU, unsigned uvoff, f32x16 (&T)[2]) {
    // B^T rows: xi 0: d0 - d2, 1: d1 + d2, 2: d2 - d1, 3: d1 - d3
    constexpr int I1 = XI == 0 ? 0 : (XI == 2 ? 2 : 1), I2 = XI == 0 ? 2 : (XI == 1 ? 2 : (XI == 2 ? 1 : 3));
    constexpr bool PLUS = XI == 1;
    const float* pt = P + ((2 * tyl) * PW + 2 * txl) * PLD;
    const float* r1 = pt + I1 * PW * PLD + ((h ^ ((tyl + (I1 >> 1)) & 1)) << 2);
    const float* r2 = pt + I2 * PW * PLD + ((h ^ ((tyl + (I2 >> 1)) & 1)) << 2);
    f32x16 acc[4];                                     // the row's four positions nu
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
#pragma unroll 2
    for (int s = 0; s < CG_PROBE_HALF(8, 8); ++s) {    // MFMA (s, comp) contracts input planes 8 s + {comp, 4 + comp} (lane half h: the second)
        float4 e[4];                                   // the row transform, shared by the four positions
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float4 d1 = ld4(r1 + c * PLD + 8 * s), d2 = ld4(r2 + c * PLD + 8 * s);
            e[c].x = PLUS ? d1.x + d2.x : d1.x - d2.x; e[c].y = PLUS ? d1.y + d2.y : d1.y - d2.y;
            e[c].z = PLUS ? d1.z + d2.z : d1.z - d2.z; e[c].w = PLUS ? d1.w + d2.w : d1.w - d2.w;
        }
        float4 v[4], ua[4];
#pragma unroll
        for (int nu = 0; nu < 4; ++nu) {
            constexpr int J1[4] = {0, 1, 2, 1}, J2[4] = {2, 2, 1, 3};      // B columns: nu 0: e0 - e2, 1: e1 + e2, 2: e2 - e1, 3: e1 - e3
            const float4 a = e[J1[nu]], b = e[J2[nu]];
            if (nu == 1) { v[nu].x = a.x + b.x; v[nu].y = a.y + b.y; v[nu].z = a.z + b.z; v[nu].w = a.w + b.w; }
            else { v[nu].x = a.x - b.x; v[nu].y = a.y - b.y; v[nu].z = a.z - b.z; v[nu].w = a.w - b.w; }
            ua[nu] = bufld4(rsU, uvoff, (((XI * 4 + nu) * 16 + 2 * s) * C) * 16);
        }
        // the MFMAs rotate over the four accumulators (an instruction between two MFMAs on the SAME accumulator costs ~43 cycles)
#define CG_W3_MF(comp) \
    _Pragma("unroll") for (int nu = 0; nu < 4; ++nu) acc[nu] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[nu].comp, ua[nu].comp, acc[nu], 0, 0, 0);
        CG_W3_MF(x) CG_W3_MF(y) CG_W3_MF(z) CG_W3_MF(w)
#undef CG_W3_MF
    }
    // columns of A: T_0 = M_0 + M_1 + M_2, T_1 = M_1 - M_2 - M_3
#pragma unroll
    for (int r = 0; r < 16; ++r) { T[0][r] = (acc[0][r] + acc[1][r]) + acc[2][r]; T[1][r] = (acc[1][r] - acc[2][r]) - acc[3][r]; }
}

struct W3Args {
    const float* x[4]; const float* u[4]; const float* bias[4]; float* y[4];
    int N, H, W;
};

__global__ __launch_bounds__(512, 2) void wino3_fused_k(W3Args a) {
    __shared__ __attribute__((aligned(16))) float P[PH * PW * PLD];     // 48 960 bytes
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int xi = wave & 3, nt = wave >> 2;
    const int grp = blockIdx.y;
    const float* __restrict__ x = a.x[grp];
    const float* __restrict__ U = a.u[grp];
    float* __restrict__ y = a.y[grp];
    const int H = a.H, W = a.W;
    const int bxn = W >> 4, byn = H >> 3;
    int b = blockIdx.x;
    const int bx = b % bxn; b /= bxn;
    const int by = b % byn;
    const int n = b / byn;
    const int y0 = by * 8, x0 = bx * 16;

    // ---- patch -> LDS, plane quads XORed with the parity of the patch row pair.  Lean form (PMC r06: the first version spent 9.6 VALU per
    // MFMA of the whole launch, most of them index arithmetic here and in the stores): a thread keeps its plane quad, its pixel advances by
    // 32 per pass = (1 row, 14 columns) of the 18-wide patch, out-of-image pixels are buffer loads at an out-of-range offset (-> 0)
    {
        const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void*)(x + (long)n * H * W * C), 0, (unsigned)(H * W * C * 4), 0x00020000);
        const int q = tid & 15;
        int pix = tid >> 4, pr = pix / PW, pc = pix - pr * PW;          // pix < 32
#pragma unroll
        for (int it = 0; it < (PH * PW + 31) / 32; ++it) {
            const int iy = y0 - 1 + pr, ix = x0 - 1 + pc;
            const bool in = iy >= 0 && iy < H && ix >= 0 && ix < W;
            const unsigned voff = in ? (unsigned)(((iy * W + ix) * C + q * 4) * 4) : 0x80000000u;
            const float4 v = bufld4(rsx, voff, 0);
            if (pix < PH * PW) *reinterpret_cast<float4*>(P + pix * PLD + ((q ^ ((pr >> 1) & 1)) << 2)) = v;
            pix += 32; pr += 1; pc += 14;
            if (pc >= PW) { pc -= PW; pr += 1; }
        }
    }
    __syncthreads();

    const int tl = tile_of_row(j), tyl = tl >> 3, txl = tl & 7;
    const __amdgpu_buffer_rsrc_t rsU = __builtin_amdgcn_make_buffer_rsrc((void*)U, 0, kUFloats * 4, 0x00020000);
    const unsigned uvoff = (unsigned)(h * C + nt * 32 + j) * 16u;
    f32x16 T[2];
    if (xi == 0) wino_row<0>(P, tyl, txl, h, rsU, uvoff, T);
    else if (xi == 1) wino_row<1>(P, tyl, txl, h, rsU, uvoff, T);
    else if (xi == 2) wino_row<2>(P, tyl, txl, h, rsU, uvoff, T);
    else wino_row<3>(P, tyl, txl, h, rsU, uvoff, T);

    // rows of A^T: Y[0][b] = T0[b] + T1[b] + T2[b], Y[1][b] = T1[b] - T2[b] - T3[b] (subscript = position row).  Wave (xi, nt) finishes output
    // (a = xi >> 1, b = xi & 1) of its column tile from its own sums and the others' in LDS (the dead patch), in a fixed order.  Published per
    // column tile, 1024 floats each: slot 0 T1[0], 1 T2[0], 2 T0[1], 3 T2[1], 4 T3[0], 5 T1[1]
    __syncthreads();
    float* X = P + nt * 6 * 1024;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int k = r * 64 + lane;
        if (xi == 1) { X[0 * 1024 + k] = T[0][r]; X[5 * 1024 + k] = T[1][r]; }
        if (xi == 2) { X[1 * 1024 + k] = T[0][r]; X[3 * 1024 + k] = T[1][r]; }
        if (xi == 0) X[2 * 1024 + k] = T[1][r];
        if (xi == 3) X[4 * 1024 + k] = T[0][r];
    }
    __syncthreads();
    const int oa = xi >> 1, ob = xi & 1;
    const float* bias = a.bias[grp];
    const float bco = bias ? bias[nt * 32 + j] : 0.f;
    // accumulator row r of lane half h is MFMA row (r & 3) + 8 (r >> 2) + 4 h -> tile_of_row() -> pixel (2 ty + oa, 2 tx + ob): the tile of
    // every (r, h) is a compile-time constant, so a store is one select + one buffer store at (lane base) + (select)
    const __amdgpu_buffer_rsrc_t rsy = __builtin_amdgcn_make_buffer_rsrc((void*)(y + (long)n * H * W * C), 0, (unsigned)(H * W * C * 4), 0x00020000);
    const int lane_base = (((y0 + oa) * W + x0 + ob) * C + nt * 32 + j) * 4;
    const int rowb = 2 * W * C * 4, colb = 2 * C * 4;           // bytes per tile row / tile column
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int k = r * 64 + lane;
        float v;
        if (xi == 0) v = (T[0][r] + X[0 * 1024 + k]) + X[1 * 1024 + k];
        else if (xi == 1) v = (X[2 * 1024 + k] + T[1][r]) + X[3 * 1024 + k];
        else if (xi == 2) v = (X[0 * 1024 + k] - T[0][r]) - X[4 * 1024 + k];
        else v = (X[5 * 1024 + k] - X[3 * 1024 + k]) - T[1][r];
        const int row0 = (r & 3) + 8 * (r >> 2);
        const int t0 = tile_of_row_c(row0), t1 = tile_of_row_c(row0 + 4);               // constants after unrolling
        const int off0 = (t0 >> 3) * rowb + (t0 & 7) * colb, off1 = (t1 >> 3) * rowb + (t1 & 7) * colb;   // wave-uniform
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v + bco), rsy, lane_base + (h ? off1 : off0), 0, 0);
    }
}

struct Entry { float* u; };
std::mutex g_mu;
std::unordered_map<const float*, Entry> g_cache;     // packed operand (wf or wb) -> its transformed filters

float* slot_for(const float* packed, bool create) {
    auto it = g_cache.find(packed);
    if (it != g_cache.end()) return it->second.u;
    if (!create) return nullptr;
    float* u = nullptr;
    // the first pack of a layer: outside any graph capture by construction (plans are compiled and packed in the eager warm-up passes)
    if (hipMalloc((void**)&u, (size_t)kUFloats * sizeof(float)) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    g_cache[packed] = Entry{u};
    return u;
}

}  // namespace

namespace cg {

// the fused kernel's geometry: a plain 3x3 / pad 1 convolution 64 -> 64 planes whose image divides into 8 x 16 pixel blocks, and enough of
// them: below 2 workgroups per CU the direct kernel's finer tiles win (D32_st3's 8 x 8 layers: 192 blocks at batch 128)
bool wino3_geom_ok(int ngroups, int N, int H, int W, int Cin, int Cout, int kH, int kW, int padH, int padW, int ups) {
    if (!(ups == 0 && kH == 3 && kW == 3 && padH == 1 && padW == 1 && Cin == C && Cout == C && H % 8 == 0 && W % 16 == 0)) return false;
    if (ngroups < 1 || ngroups > 4) return false;
    const long v = opt(OPT_WINO3);
    const bool full = (long)ngroups * N * (H / 8) * (W / 16) >= 2L * kNumCU;
    return v == 2 || (v == 1 && full);
}

// called by cg_pack_conv_weight / _batch for every layer they pack: 64 -> 64 3x3 layers get their transformed filters (re)computed,
// anything else packed into a known address drops that address from the cache
int wino3_note_pack(hipStream_t st, int n, const float* const* w, float* const* wf, float* const* wb, const int* Cout, const int* Cin,
                    const int* kH, const int* kW, const int* wb_map) {
    if (opt(OPT_WINO3) == 0) return 0;
    std::lock_guard<std::mutex> lk(g_mu);
    PackJobs jobs;
    jobs.n = 0;
    auto flush = [&]() {
        if (jobs.n) hipLaunchKernelGGL(wino3_pack_k, dim3(C * C / 256, jobs.n), dim3(256), 0, st, jobs);
        jobs.n = 0;
    };
    for (int i = 0; i < n; ++i) {
        const bool fits = Cout[i] == C && Cin[i] == C && kH[i] == 3 && kW[i] == 3 && !(wb_map && wb_map[i]);
        if (!fits) {
            if (wf[i]) g_cache.erase(wf[i]);     // (the buffers stay allocated: a handful of 256 KB blocks per process)
            if (wb[i]) g_cache.erase(wb[i]);
            continue;
        }
        float* uf = wf[i] ? slot_for(wf[i], true) : nullptr;
        float* ub = wb[i] ? slot_for(wb[i], true) : nullptr;
        if ((wf[i] && !uf) || (wb[i] && !ub)) return cg::fail("wino3: cannot allocate the transformed filters");
        jobs.w[jobs.n] = w[i]; jobs.uf[jobs.n] = uf; jobs.ub[jobs.n] = ub;
        if (++jobs.n == 8) flush();
    }
    flush();
    CG_LAUNCH_CHECK();
    return 0;
}

// 1 = launched, 0 = not this path (an operand without transformed filters), -1 = error
int wino3_forward(hipStream_t st, int ngroups, const float* const* x, const float* const* wpk, const float* const* bias, float* const* y,
                  int N, int H, int W) {
    W3Args a;
    memset(&a, 0, sizeof(a));
    {
        std::lock_guard<std::mutex> lk(g_mu);
        for (int g = 0; g < ngroups; ++g) {
            const float* u = slot_for(wpk[g], false);
            if (!u || !x[g] || !y[g] || (uintptr_t)x[g] % 16) return 0;
            a.x[g] = x[g]; a.u[g] = u; a.bias[g] = bias ? bias[g] : nullptr; a.y[g] = y[g];
        }
    }
    a.N = N; a.H = H; a.W = W;
    hipLaunchKernelGGL(wino3_fused_k, dim3((unsigned)(N * (H / 8) * (W / 16)), ngroups), dim3(512), 0, st, a);
    if (hipGetLastError() != hipSuccess) { cg::fail("wino3_forward: launch failed"); return -1; }
    return 1;
}

}  // namespace cg
