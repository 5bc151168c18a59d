// Implicit-GEMM convolution / linear kernels on the exact-fp32 MFMA
// (v_mfma_f32_32x32x2_f32) for gfx950.
//
// Replaces the THNN/THCUNN/cudnn kernels the reference reaches through
// cudnn.SpatialConvolution (models.lua:206,212,218,222), nn.SpatialConvolution
// (models.lua:646-685,844-846) and nn.Linear (models.lua:199,697,700,850,853):
//   igemm_nn : updateOutput and updateGradInput  (Y[m][n] = sum_k A(m,k) W[k][n])
//   igemm_tn : accGradParameters                 (dW[k][n] = sum_m A(m,k) dY[m][n])
// A(m,k) is never materialised: m -> grid pixel (n,oy,ox), k -> (tap,ci), gathered from the NHWC tensor.
//
// nn.SpatialUpSamplingNearest(2) -> conv (models.lua:205-206,211-212,217-218) is executed as FOUR PHASE
// CONVOLUTIONS on the low-resolution input: output pixel (2i+a, 2j+b) only ever sees ceil((k+1)/2)^2 distinct
// low-res taps, so the k x k weights are pre-summed per phase (a,b) into k' x k' kernels (k'=2 for k=3, 3 for
// k=5): 2.25x / 2.78x fewer MACs than convolving the materialised upsampled map, identical up to fp32
// re-association of the weight sums.  The data gradient w.r.t. the low-res input (upsampling's 2x2 block sum
// folded in) is ONE GEMM over the 4 phases' taps; the weight gradient is taken per phase and mapped back onto
// the canonical taps by the reduce kernel.
//
// fp32 MFMA issues at exactly the fp32 VALU rate, so every VALU instruction in the main loop costs MFMA
// throughput (measured: 4.9 VALU per MFMA = 73 % pipe utilisation).  The FAST path therefore keeps the gather
// lean: tap bookkeeping lives in SGPRs, per-row validity is a precomputed 64-bit tap mask, and loads are
// buffer_load_dwordx4 whose out-of-range offset returns 0 (no branches, no zero-fills).
//
// Tiling (wave64, 4 waves / workgroup): block tile BM x BN, K step 16 or 32, each wave owns MI x NI MFMA 32x32
// accumulators; LDS tiles are double buffered, one barrier per K tile.  How a K tile gets into LDS:
//   * igemm_nn_kernel / igemm_tn_kernel (register staging): both operands K-major in LDS ([k][m] XOR-swizzled, [k][n]),
//     a fragment read is one conflict-free ds_read_b32 per operand per MFMA; the next tile's global loads are issued
//     before the current tile's MFMAs and stored to LDS behind them.  (The k-quad LDS layouts, the loads two tiles ahead and the
//     quad weight-gradient kernel of rounds 2-4 lost every A/B they were in and were removed in round 5: profiles/NOTES_r01_r02.md.)
//   * igemm_nng_kernel / igemm_tng_kernel (LDS-direct loads, buffer_load_dwordx4 ... lds): the default where the
//     geometry allows (round 2: the ablation builds below located the loop's loss in the register -> LDS staging).
#include "common.h"
#include <stdlib.h>
#include <algorithm>
#include <mutex>
#include <type_traits>
#include <unordered_map>
#include <vector>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int BK = 16;
constexpr unsigned OOB = 0x80000000u;
// -DCG_TRACE (CG_BUILD_DEFINES=-DCG_TRACE, build.py): every igemm_nn_kernel workgroup records 100 MHz timestamps at its start, behind
// the prologue (first tile in LDS), behind the K loop and at its end, plus the hardware id of the CU it ran on, into the
// buffer handed to cg_debug_set_trace - scripts/wg_trace.py turns that into the dispatch balance and the phase times.
#ifdef CG_TRACE
__device__ unsigned long long* g_trace;
#define CG_STAMP(slot)                                                                                                  \
    do {                                                                                                                \
        if (threadIdx.x == 0 && g_trace) {                                                                              \
            const long wg = blockIdx.x + (long)gridDim.x * (blockIdx.y + (long)gridDim.y * blockIdx.z);                  \
            g_trace[wg * 8 + (slot)] = wall_clock64();                                                                  \
            if ((slot) == 0) {                                                                                          \
                g_trace[wg * 8 + 4] = __builtin_amdgcn_s_getreg((31 << 11) | 4);  /* HW_REG_HW_ID */                     \
                g_trace[wg * 8 + 5] = __builtin_amdgcn_s_getreg((31 << 11) | 20); /* HW_REG_XCC_ID */                    \
            }                                                                                                           \
        }                                                                                                               \
    } while (0)
#else
#define CG_STAMP(slot) do { } while (0)
#endif  // >= num_records of every descriptor below: the load returns 0

// tap t of a launch -> (ty, tx, element offset):  group = t / kk, (ry,rx) = (t % kk) / kw, % kw
//   phase bits (a,b) = launch phase (nphase == 4)  or  group bits (ngroups == 4)  or  0
//   ty = sgn*(r0y(a)+ry), tx = sgn*(r0x(b)+rx);  source pixel = ((oy+ty)*ss + a*(ss==2), (ox+tx)*ss + b*(ss==2))
struct TapDesc {
    int kw, kk, ngroups, sgn, ss;
    int r0y0, r0y1, r0x0, r0x1;   // tap origin per phase bit (no arrays: runtime indexing would go to scratch)
};

struct Geom {
    int M;                 // grid pixels per phase: N*Hg*Wg
    int Hg, Wg;            // grid per image
    int lgW, lgHW;         // log2(Wg), log2(Hg*Wg) when both are powers of two, else -1
    int Hv, Wv;            // (oy+ty, ox+tx) must lie in [0,Hv) x [0,Wv)
    int Hs, Ws;            // source tensor spatial dims
    int Cin, Cout;         // GEMM K-channels / N
    int so, Hout, Wout;    // grid pixel -> (n, oy*so+a, ox*so+b) of an [N,Hout,Wout] tensor (NN: y; TN: dy)
    int nphase;            // 1 or 4 (blockIdx.z)
    int ntaps, Ktot;       // per phase; Ktot = ntaps*Cin
    TapDesc td;
    int minoff0, minoff1, minoff2, minoff3;   // per phase: min(0, smallest tap element offset) - host-computed (finish_geom), the kernels' descriptor base
    // POSITION-MAJOR rows (round 5; igemm_nng_kernel<..., PM>): pmn = N (a power of two, a multiple of the row tile) puts grid pixel
    // (n, oy, ox) at row m = (oy * Wg + ox) * N + n, so that a row tile holds ONE pixel position of 64 images and all its rows share
    // their zero-padding taps, which the K loop then skips (D32_st3's 7x7 layer at 8x8: 38 % of the MACs multiply padding).  0 = the
    // image-major order everywhere else.  Split partials are in row order either way; out_row() maps a row to its output pixel.
    int pmn, pm_lg;
};

constexpr int MAXG = 4;  // groups per launch: same geometry, separate tensors (D32_st3's identical branches)
constexpr int kMaxStridedGroups = 36;   // ... or up to 36 equally spaced ones (cg_conv2d_wgrad_strided: 16 planes of F(2x2,3x3), 4 x 9 of F(2x2,2x2))

struct NNArgs {
    const float* x0; const float* x1; const float* x2; const float* x3;   // per group (no arrays: see TapDesc)
    const float* w0; const float* w1; const float* w2; const float* w3;   // [nphase][Ktot][Cout]
    const float* b0; const float* b1; const float* b2; const float* b3;   // [Cout] or null
    float* y0; float* y1; float* y2; float* y3;                           // output tensors
    float* part;        // split partials [S][ngroups*nphase][M][Cout] (nsplit > 1)
    int ngroups;
    Geom g;
    int kchunk;         // K range per split (multiple of BK)
    int nsplit;
    // fused epilogue (cg_conv2d_forward_ex): act 0 none, 1 PReLU (slope read from al<group>), 2 LeakyReLU (slope);
    // z<group> receives act(y) while y keeps the pre-activation (what the activation's backward needs).
    int act;
    float slope;
    const float* al0; const float* al1; const float* al2; const float* al3;
    float* z0; float* z1; float* z2; float* z3;
    // stats != null: per (phase, row tile, wave row) column sums of y and y^2 -> stats[row][2][Cout] (batch-norm statistics
    // of the layer behind this convolution, models.lua:206-207; single group, unsplit launches only)
    float* stats;
    int xcd_swizzle;
};

__device__ __forceinline__ float apply_act(int act, float v, float a) {
    return act == 1 ? (v > 0.f ? v : a * v) : (v >= 0.f ? v : a * v);
}

struct TNArgs {
    const float* x0; const float* x1; const float* x2; const float* x3;
    const float* d0; const float* d1; const float* d2; const float* d3;   // dy per group
    int ngroups;
    float* part;        // [S][ngroups*nphase][Ktot][Cout]
    float* bias_part;   // [S][ngroups*nphase][Cout] column sums of dy (gradBias partials) or null
    Geom g;
    int pchunk;         // pixels per split (multiple of BK)
    int xcd_swizzle;
    int pm_n;           // igemm_tng_kernel mode 2 (position-major K tiles): images per launch, a multiple of BK
    long xgs, dgs;      // != 0: group g reads x0 + g * xgs / d0 + g * dgs (up to kMaxStridedGroups equally spaced groups: the
                        // Winograd-domain weight-gradient GEMMs of winograd.hip in one launch), x1.. / d1.. unused
};

template <typename T>
__device__ __forceinline__ T sel4(int i, T a, T b, T c, T d) { return i == 0 ? a : (i == 1 ? b : (i == 2 ? c : d)); }

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float f4c(const float4& v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : (i == 2 ? v.z : v.w)); }

typedef float f32x4 __attribute__((ext_vector_type(4)));
// NOTE: cast the WHOLE vector.  Extracting the four dwords one by one (bit_cast(float, v.x) ...) makes LLVM's
// InstCombine (ROCm 7.2, -O3) narrow the v4i32 buffer load to a single i32 and splat it - a silent miscompile.
__device__ __forceinline__ float4 bufld4(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    const f32x4 f = __builtin_bit_cast(f32x4, v);
    return make_float4(f.x, f.y, f.z, f.w);
}

// buffer_load_dwordx4 ... lds: lane l's 16 bytes go to lds + 16 l (lds wave-uniform), zeros for an out-of-range voff.
// The builtin exists in the device pass only (the host pass would silently drop the kernel's launch stub).
__device__ __forceinline__ void glds16(__amdgpu_buffer_rsrc_t r, float* lds, unsigned voff, int soff) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 16, voff, soff, 0, 0);
#endif
}

__host__ __device__ __forceinline__ void tap_decode(const Geom& g, int t, int pa, int pb, int& ty, int& tx, int& off) {
    const TapDesc& d = g.td;
    const int grp = d.ngroups > 1 ? t / d.kk : 0;
    const int tt = t - grp * d.kk;
    const int ry = tt / d.kw, rx = tt - (tt / d.kw) * d.kw;
    int a = pa, b = pb;
    if (d.ngroups > 1) { a = grp >> 1; b = grp & 1; }
    ty = d.sgn * ((a ? d.r0y1 : d.r0y0) + ry);
    tx = d.sgn * ((b ? d.r0x1 : d.r0x0) + rx);
    const int ay = d.ss == 2 ? a : 0, ax = d.ss == 2 ? b : 0;
    off = ((ty * d.ss + ay) * g.Ws + (tx * d.ss + ax)) * g.Cin;
}

__device__ __forceinline__ void pix_decode(const Geom& g, int m, int& n, int& oy, int& ox) {
    if (g.lgW >= 0) {
        n = m >> g.lgHW;
        oy = (m >> g.lgW) & (g.Hg - 1);
        ox = m & (g.Wg - 1);
    } else {
        const int hw = g.Hg * g.Wg;
        n = m / hw;
        const int r = m - n * hw;
        oy = r / g.Wg;
        ox = r - oy * g.Wg;
    }
}

// element offset of grid pixel m (phase bits pa,pb) in the [N,Hout,Wout,Cout] tensor
__device__ __forceinline__ long out_row(const Geom& g, int m, int pa, int pb) {
    if (g.pmn) return ((long)(m & (g.pmn - 1)) * (g.Hg * g.Wg) + (m >> g.pm_lg)) * g.Cout;   // so == 1, nphase == 1 (host)
    if (g.so == 1 && g.nphase == 1) return (long)m * g.Cout;
    int n, oy, ox;
    pix_decode(g, m, n, oy, ox);
    return ((long)(n * g.Hout + oy * g.so + pa) * g.Wout + ox * g.so + pb) * g.Cout;
}

// position-major launches (Geom::pmn): the number of taps of a plain convolution that fall inside the image at pixel position pos.
// A tile at that position walks them in cdiv(taps * Cin, kchunk) equal K units (igemm_nng_kernel<..., PM>); the reduce adds as many.
__device__ __forceinline__ int pm_valid_taps(const Geom& g, int pos) {
    const TapDesc& d = g.td;
    const int kh = d.kk / d.kw;
    const int oy = g.lgW >= 0 ? pos >> g.lgW : pos / g.Wg, ox = pos - oy * g.Wg;
    const int ylo = d.sgn > 0 ? d.r0y0 : -(d.r0y0 + kh - 1), xlo = d.sgn > 0 ? d.r0x0 : -(d.r0x0 + d.kw - 1);
    const int vy = min(g.Hv - 1 - oy, ylo + kh - 1) - max(-oy, ylo) + 1;
    const int vx = min(g.Wv - 1 - ox, xlo + d.kw - 1) - max(-ox, xlo) + 1;
    return max(vy, 0) * max(vx, 0);
}

// Column XOR of the K-major A tile, as a function of the k quad kv = k / 4: a half-wave of the transposed store holds
// (K step / 4) quads x (128 / K step) consecutive rows and must land in 32 distinct banks; a fragment read (32 consecutive
// rows at fixed k) must stay a permutation of them.  K step 16: 4 quads x 8 rows -> flip row bits 3,4; K step 32: 8 quads x 4
// rows -> flip row bits 2,3,4 (PMC: the bits-3,4 form gave the K-step-32 kernel 10 % LDS bank-conflict cycles).
template <int BKT>
__device__ __forceinline__ int a_swizzle(int kv) { return BKT == 32 ? ((kv & 7) << 2) : ((kv & 3) << 3); }

// Lean epilogue of the NN kernels for a FULL tile whose output rows are consecutive ([row][Cout], i.e. stride-1 layers and split
// partials): one buffer store per value at (per-lane byte offset) + (row offset in an SGPR).  The generic loop below it - bounds checks,
// 64-bit addresses and an output-row decode per row - is ~2000 instructions per wave; with all workgroups of a launch reaching their
// epilogue together that was 10 us in which no MFMA ran (profiles/r03_wg_timeline_nn_kernels.txt).
template <int MI, int NI, bool ACT>
__device__ __forceinline__ void nn_store_lean(const f32x16 (&acc)[MI][NI], const float (&bj)[NI], float* ybase, float* zbase, unsigned vo,
                                              int c4, int act, float aslope) {
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)ybase, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc((void*)(ACT ? zbase : ybase), 0, 0x7fffffff, 0x00020000);
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int soff = (i * 32 + (r & 3) + 8 * (r >> 2)) * c4;   // wave-uniform
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const float v = acc[i][j][r] + bj[j];
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), ry, (int)(vo + j * 128), soff, 0);
                if (ACT) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(apply_act(act, v, aslope)), rz, (int)(vo + j * 128), soff, 0);
            }
        }
    }
}

// ---------------------------------------------------------------------------
// NN: Y[m][n] = sum_k A(m,k) * W[k][n]
// ---------------------------------------------------------------------------
//
template <int BM, int BN, int WM, int WN, bool FAST, bool VECB, int BK>
__global__ __launch_bounds__(256, BK == 16 ? 4 : 2) void igemm_nn_kernel(NNArgs a) {
    static_assert(WM * WN == 4, "4 waves per workgroup");
    constexpr int MI = BM / WM / 32;
    constexpr int NI = BN / WN / 32;
    static_assert(MI >= 1 && NI >= 1, "wave tile >= 32x32");
    // A tile [k][m], column XOR-swizzled by ((k>>2)&3)<<3: the transposed ds_write_b32 of a float4 (4
    // consecutive k of one pixel) hits 32 distinct banks per half-wave, and a fragment read (32 consecutive m
    // at fixed k) stays a permutation of the 32 banks.  No padding.
    constexpr int LDA = BM, LDB = BN;
    constexpr int A_TILE = BK * LDA, B_TILE = BK * LDB;
    constexpr int KV = BK / 4;        // float4 per A row
    constexpr int ARPP = 256 / KV;    // A rows per pass
    constexpr int AROWS = BM / ARPP;  // float4 per thread per tile (A)
    static_assert(BM % ARPP == 0 && AROWS >= 1, "BM multiple of rows-per-pass");
    constexpr int NVEC = BN / 4;
    constexpr int BRPP = 256 / NVEC;  // B rows per pass
    constexpr int BPASS = (BK + BRPP - 1) / BRPP;

    __shared__ __attribute__((aligned(16))) float smem[2 * A_TILE + 2 * B_TILE];
    float* As = smem;
    float* Bs = smem + 2 * A_TILE;

    CG_STAMP(0);
    const Geom& g = a.g;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wm0 = (wave / WN) * (BM / WM);
    const int wn0 = (wave % WN) * (BN / WN);

    const int ntn = (g.Cout + BN - 1) / BN;
    // XCD-aware tile order: workgroup b is dispatched to XCD b % 8 (each XCD has its own L2), so hand every XCD a CONTIGUOUS
    // range of tiles - neighbours then share their A rows / B columns through one L2 instead of eight
    int bid = blockIdx.x;
    if ((a.xcd_swizzle & 1) && (gridDim.x & 7) == 0) bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);
    const int tn = ntn == 1 ? 0 : bid % ntn, tm = ntn == 1 ? bid : bid / ntn;
    const int m0 = tm * BM, n0 = tn * BN;
    const int split = blockIdx.y;
    const int zz = blockIdx.z;
    const int group = g.nphase == 1 ? zz : (g.nphase == 4 ? zz >> 2 : zz / g.nphase);
    const int phase = zz - group * g.nphase, pa = phase >> 1, pb = phase & 1;
    const float* gx = sel4(group, a.x0, a.x1, a.x2, a.x3);
    const float* gbias = sel4(group, a.b0, a.b1, a.b2, a.b3);
    float* gy = sel4(group, a.y0, a.y1, a.y2, a.y3);
    const int ks = split * a.kchunk;
    const int kend = min(g.Ktot, ks + a.kchunk);
    const int T = CG_PROBE_HALF(1, (kend - ks + BK - 1) / BK);
    const float* wph = sel4(group, a.w0, a.w1, a.w2, a.w3) + (long)phase * g.Ktot * g.Cout;

    // ---- A staging: thread owns k-vector a_kv (4 consecutive k) of rows a_r + 64p
    const int a_kv = tid % KV, a_r = tid / KV;
    int rowb[AROWS], r_oy[AROWS], r_ox[AROWS];
    bool r_ok[AROWS];
#pragma unroll
    for (int p = 0; p < AROWS; ++p) {
        const int m = m0 + a_r + ARPP * p;
        r_ok[p] = m < g.M;
        int n, oy, ox;
        pix_decode(g, r_ok[p] ? m : 0, n, oy, ox);
        r_oy[p] = oy; r_ox[p] = ox;
        rowb[p] = ((n * g.Hs + oy * g.td.ss) * g.Ws + ox * g.td.ss) * g.Cin;
    }
    const int b_nv = tid % NVEC, b_kr = tid / NVEC;

    float4 areg[AROWS];
    float4 breg[BPASS];

    // ---- FAST-path state -------------------------------------------------------------------------------
    unsigned long long cur[AROWS];     // tap-validity mask of each row, shifted so bit 0 = next tile's tap
    unsigned rowbytes[AROWS];
    unsigned bvoff[BPASS];
    int tapi = 0, ci0 = 0, toff = 0, minoff = 0;
    __amdgpu_buffer_rsrc_t rsx, rsw;
    if (FAST) {
        const TapDesc& d = g.td;
        const int kh = d.kk / d.kw;
#pragma unroll
        for (int p = 0; p < AROWS; ++p) cur[p] = 0ull;
        for (int grp = 0; grp < d.ngroups; ++grp) {
            int ga = pa, gb = pb;
            if (d.ngroups > 1) { ga = grp >> 1; gb = grp & 1; }
            unsigned xb[AROWS];
#pragma unroll
            for (int p = 0; p < AROWS; ++p) xb[p] = 0u;
            for (int rx = 0; rx < d.kw; ++rx) {
                const int tx = d.sgn * ((gb ? d.r0x1 : d.r0x0) + rx);
#pragma unroll
                for (int p = 0; p < AROWS; ++p)
                    xb[p] |= ((unsigned)(r_ox[p] + tx) < (unsigned)g.Wv ? 1u : 0u) << rx;
            }
            for (int ry = 0; ry < kh; ++ry) {
                const int ty = d.sgn * ((ga ? d.r0y1 : d.r0y0) + ry);
                const int sh = grp * d.kk + ry * d.kw;
#pragma unroll
                for (int p = 0; p < AROWS; ++p)
                    if (r_ok[p] && (unsigned)(r_oy[p] + ty) < (unsigned)g.Hv) cur[p] |= (unsigned long long)xb[p] << sh;
            }
        }
        minoff = sel4(phase, g.minoff0, g.minoff1, g.minoff2, g.minoff3);   // host-computed (finish_geom)
        tapi = ks == 0 ? 0 : ks / g.Cin;
        ci0 = ks - tapi * g.Cin;
        { int ty, tx; tap_decode(g, tapi, pa, pb, ty, tx, toff); }
#pragma unroll
        for (int p = 0; p < AROWS; ++p) {
            cur[p] >>= tapi;
            rowbytes[p] = (unsigned)(rowb[p] + 4 * a_kv) * 4u;
        }
        // base shifted down by the most negative tap offset so that the SGPR offset stays non-negative
        rsx = __builtin_amdgcn_make_buffer_rsrc((void*)(gx + minoff), 0, 0x7fffffff, 0x00020000);
        rsw = __builtin_amdgcn_make_buffer_rsrc((void*)wph, 0, 0x7fffffff, 0x00020000);
#pragma unroll
        for (int q = 0; q < BPASS; ++q) {
            const int kr = b_kr + q * BRPP;
            const int n = n0 + 4 * b_nv;
            bvoff[q] = (kr < BK && n < g.Cout) ? (unsigned)(kr * g.Cout + n) * 4u : OOB;
        }
    }

    auto load_tile = [&](int k0) {
        if (FAST) {
            // (tapi, ci0, toff) describe this tile; all SGPR arithmetic
            const int soff = (toff - minoff + ci0) * 4;
#pragma unroll
            for (int p = 0; p < AROWS; ++p) {
                const unsigned voff = ((unsigned)cur[p] & 1u) ? rowbytes[p] : OOB;
                areg[p] = bufld4(rsx, voff, soff);
            }
            if (VECB) {
                const int sb = k0 * g.Cout * 4;
#pragma unroll
                for (int q = 0; q < BPASS; ++q) breg[q] = bufld4(rsw, bvoff[q], sb);
            }
            ci0 += BK;
            if (ci0 >= g.Cin) {
                ci0 = 0;
                ++tapi;
#pragma unroll
                for (int p = 0; p < AROWS; ++p) cur[p] >>= 1;
                if (tapi < g.ntaps) { int ty, tx; tap_decode(g, tapi, pa, pb, ty, tx, toff); }
            }
        } else {
            float tmp[AROWS][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int kj = k0 + 4 * a_kv + j;
                const bool kok = kj < kend;
                const int tap = kok ? kj / g.Cin : 0;
                const int ci = kj - tap * g.Cin;
                int ty, tx, off;
                tap_decode(g, tap, pa, pb, ty, tx, off);
#pragma unroll
                for (int p = 0; p < AROWS; ++p) {
                    const bool ok = r_ok[p] && kok && (unsigned)(r_oy[p] + ty) < (unsigned)g.Hv &&
                                    (unsigned)(r_ox[p] + tx) < (unsigned)g.Wv;
                    float v = 0.f;
                    if (ok) v = gx[(long)rowb[p] + off + ci];
                    tmp[p][j] = v;
                }
            }
#pragma unroll
            for (int p = 0; p < AROWS; ++p) areg[p] = make_float4(tmp[p][0], tmp[p][1], tmp[p][2], tmp[p][3]);
        }
        if (!(FAST && VECB)) {
#pragma unroll
            for (int q = 0; q < BPASS; ++q) {
                const int kr = b_kr + q * BRPP;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (kr < BK) {
                    const int kk = k0 + kr;
                    const int n = n0 + 4 * b_nv;
                    if (kk < kend) {
                        const float* wp = wph + (long)kk * g.Cout + n;
                        if (VECB) {
                            if (n < g.Cout) v = ld4(wp);
                        } else {
                            if (n + 0 < g.Cout) v.x = wp[0];
                            if (n + 1 < g.Cout) v.y = wp[1];
                            if (n + 2 < g.Cout) v.z = wp[2];
                            if (n + 3 < g.Cout) v.w = wp[3];
                        }
                    }
                }
                breg[q] = v;
            }
        }
    };

    // the LDS double-buffer index is a compile-time constant everywhere (the K loop is unrolled by two), so that every
    // ds_read / ds_write address is a loop-invariant register plus an immediate offset - no address VALU in the loop
    auto store_tile = [&](auto bufc) {
        constexpr int buf = decltype(bufc)::value;
        float* A = As + buf * A_TILE;
        float* B = Bs + buf * B_TILE;
#pragma unroll
        for (int p = 0; p < AROWS; ++p) {
            const int rs = (a_r + ARPP * p) ^ a_swizzle<BK>(a_kv);
            A[(4 * a_kv + 0) * LDA + rs] = areg[p].x;
            A[(4 * a_kv + 1) * LDA + rs] = areg[p].y;
            A[(4 * a_kv + 2) * LDA + rs] = areg[p].z;
            A[(4 * a_kv + 3) * LDA + rs] = areg[p].w;
        }
#pragma unroll
        for (int q = 0; q < BPASS; ++q) {
            const int kr = b_kr + q * BRPP;
            if (kr < BK) *reinterpret_cast<float4*>(B + kr * LDB + 4 * b_nv) = breg[q];
        }
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    CG_STAMP(6);   // trace builds: end of the integer set-up (tile origin, tap-validity masks, descriptors)
    if (T > 0) {
        load_tile(ks);
        store_tile(std::integral_constant<int, 0>{});
    }
    __syncthreads();
    CG_STAMP(1);

    auto k_tile = [&](auto bufc, int t) {
        constexpr int buf = decltype(bufc)::value;
        if (t + 1 < T) load_tile(ks + (t + 1) * BK);   // tile t + 1 into the staging registers, stored to LDS behind this tile's MFMAs
        const float* A = As + buf * A_TILE + wm0;
        const float* B = Bs + buf * B_TILE + wn0 + l31;
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float av[MI], bv[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) av[i] = A[(kk + h) * LDA + i * 32 + (l31 ^ a_swizzle<BK>(kk >> 2))];
#pragma unroll
            for (int j = 0; j < NI; ++j) bv[j] = B[(kk + h) * LDB + j * 32];
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
        if (t + 1 < T) store_tile(std::integral_constant<int, buf ^ 1>{});
        __syncthreads();
    };
    for (int t = 0; t < T; t += 2) {
        k_tile(std::integral_constant<int, 0>{}, t);
        if (t + 1 < T) k_tile(std::integral_constant<int, 1>{}, t + 1);
    }

    CG_STAMP(2);
    // ---- epilogue: D[i][j], i = (r&3) + 8*(r>>2) + 4*h (pixel), j = l31 (channel)
    const bool partial = a.nsplit > 1;
    const bool add_bias = (gbias != nullptr) && !partial;
    float* yout = partial ? a.part : gy;
    const int act = partial ? 0 : a.act;
    float* zout = act ? sel4(group, a.z0, a.z1, a.z2, a.z3) : nullptr;
    float aslope = a.slope;
    if (act == 1) aslope = *sel4(group, a.al0, a.al1, a.al2, a.al3);
    const bool stats = a.stats != nullptr && !partial;
    float bj[NI], s1[NI], s2[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int n = n0 + wn0 + j * 32 + l31;
        bj[j] = (add_bias && n < g.Cout) ? gbias[n] : 0.f;
        s1[j] = 0.f; s2[j] = 0.f;
    }
    {
        // lean path: full tile, consecutive output rows, byte offsets inside the 2 GB a buffer descriptor spans
        const bool lin = partial || (g.so == 1 && g.nphase == 1);
        const long rows_total = partial ? (long)a.nsplit * a.ngroups * g.nphase * g.M : (long)g.M;
        if (lin && !stats && m0 + BM <= g.M && n0 + BN <= g.Cout && rows_total * g.Cout < 0x1fffffffL) {
            const long rowbase = partial ? (long)(split * (a.ngroups * g.nphase) + zz) * g.M : 0L;
            const unsigned vo = ((unsigned)(rowbase + m0 + wm0 + 4 * h) * (unsigned)g.Cout + (unsigned)(n0 + wn0 + l31)) * 4u;
            if (act) nn_store_lean<MI, NI, true>(acc, bj, yout, zout, vo, g.Cout * 4, act, aslope);
            else nn_store_lean<MI, NI, false>(acc, bj, yout, nullptr, vo, g.Cout * 4, 0, 0.f);
            CG_STAMP(3);
            return;
        }
    }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (m >= g.M) continue;
            const long ro = partial ? ((long)(split * (a.ngroups * g.nphase) + zz) * g.M + m) * g.Cout : out_row(g, m, pa, pb);
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int n = n0 + wn0 + j * 32 + l31;
                if (n < g.Cout) {
                    const float v = acc[i][j][r] + bj[j];
                    yout[ro + n] = v;
                    if (act) zout[ro + n] = apply_act(act, v, aslope);
                    if (stats) { s1[j] += v; s2[j] += v * v; }
                }
            }
        }
    }
    if (stats) {   // the two half-waves hold the two row halves of the same columns
        const int srow = (zz * (int)((g.M + BM - 1) / BM) + tm) * WM + wave / WN;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const float t1 = s1[j] + __shfl_xor(s1[j], 32, 64), t2 = s2[j] + __shfl_xor(s2[j], 32, 64);
            const int n = n0 + wn0 + j * 32 + l31;
            if (h == 0 && n < g.Cout) {
                a.stats[((long)srow * 2 + 0) * g.Cout + n] = t1;
                a.stats[((long)srow * 2 + 1) * g.Cout + n] = t2;
            }
        }
    }
    CG_STAMP(3);
}

// ---------------------------------------------------------------------------
// NN with LDS-direct loads (buffer_load_dwordx4 ... lds, gfx950): the tiles go from global memory straight into LDS - no
// staging registers, no ds_write, no wait for a load in front of a store.  (Ablation of igemm_nn_kernel, CG_EXP builds: the
// K loop of the 512 -> 256 data gradient runs 0.318 ms; without its LDS stores and the loads feeding them 0.271 ms.)
// A lane's 16 bytes land at M0 + 16 * lane, so a wave instruction fills 1 KB of consecutive LDS and the tile layouts follow
// from which global quad every lane asks for:
//   A: row-major [m][BK] (a pixel's BK consecutive channels = 8 or 4 quads); the lane at position (m, j) loads quad
//      j ^ swz(m), swz(m) = (m >> 1) & 7 (BK 32) / (m >> 2) & 3 (BK 16), so that a ds_read_b128 fragment read (16 lanes = 16
//      rows, one quad each) covers all 16 bank slots while 8 (4) neighbouring lanes still fetch one pixel's 128 (64)
//      consecutive bytes.  A lane whose tap falls into the padding asks for the out-of-range offset and its quad is
//      written as zeros (checked on the hardware: tools/glds_test.hip).
//   B: [k][n] as in igemm_nn_kernel (a wave instruction = 64 / (BN/4) consecutive weight rows), ds_read_b32 fragments.
// MFMA step (g, s) multiplies A[m][8g + 4h + s] with B[8g + 4h + s][n] (h = lane / 32).
// FAST && VECB geometries only (Cin % BK == 0, 16-byte aligned operands, Cout % 4 == 0); prologue and epilogue as above.
// ---------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN, int BK, bool PM = false>
__global__ __launch_bounds__(256, 2) void igemm_nng_kernel(NNArgs a) {
    static_assert(WM * WN == 4, "4 waves per workgroup");
    constexpr int MI = BM / WM / 32;
    constexpr int NI = BN / WN / 32;
    constexpr int A_TILE = BK * BM, B_TILE = BK * BN;
    constexpr int KV = BK / 4;        // quads per A row
    constexpr int RPW = 64 / KV;      // A rows per wave instruction
    constexpr int ARPP = 256 / KV;    // A rows per pass of the workgroup
    constexpr int AROWS = BM / ARPP;
    static_assert(BM % ARPP == 0 && AROWS >= 1, "BM multiple of rows-per-pass");
    constexpr int NVEC = BN / 4;
    constexpr int BRPW = 64 / NVEC;   // B rows per wave instruction
    constexpr int BRPP = 256 / NVEC;  // B rows per pass
    constexpr int BPASS = (BK + BRPP - 1) / BRPP;
    constexpr int G2 = KV / 2;

    // separate arrays per buffer: the compiler can then tell the buffer the DMA writes from the one the fragments are read from
    __shared__ __attribute__((aligned(16))) float As0[A_TILE];
    __shared__ __attribute__((aligned(16))) float As1[A_TILE];
    __shared__ __attribute__((aligned(16))) float Bs0[B_TILE];
    __shared__ __attribute__((aligned(16))) float Bs1[B_TILE];

    CG_STAMP(0);
    const Geom& g = a.g;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int wm0 = (wave / WN) * (BM / WM);
    const int wn0 = (wave % WN) * (BN / WN);

    const int ntn = (g.Cout + BN - 1) / BN;
    int bid = blockIdx.x;
    // (PM: the natural order - consecutive tiles on consecutive XCDs - spreads the short border positions and the long centre ones
    // evenly over the XCDs; contiguous ranges would give one XCD the top row of the image and another the middle)
    int tn, tm;
    const int tiles_m = (int)gridDim.x / ntn, xg = ntn <= 8 ? 8 / ntn : 0;       // xg: XCDs per column tile
    if (!PM && (a.xcd_swizzle & 4) && (ntn == 2 || ntn == 4 || ntn == 8) && (gridDim.x & 7) == 0 && tiles_m % xg == 0) {
        // WEIGHTS-stationary XCDs (round 5): XCD k = bid % 8 owns column tile k % ntn and one of 8 / ntn row ranges, so its
        // L2 holds ONE [Ktot x BN] slice of the (phase) kernel - 1 MB for conv1's forward, where the row-range order below made all eight
        // XCDs fetch all four column slices: 8 x 16.8 MB of kernels per launch against 4 x 4.2 MB of input now (r x c XCD grid over
        // rows x columns: traffic = c * |x| + r * |w|)
        const int k = bid & 7, j = bid >> 3;
        tn = k % ntn;
        tm = (k / ntn) * (tiles_m / xg) + j;
    } else {
        if (!PM && (a.xcd_swizzle & 1) && (gridDim.x & 7) == 0) bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);
        tn = ntn == 1 ? 0 : bid % ntn; tm = ntn == 1 ? bid : bid / ntn;
    }
    const int m0 = tm * BM, n0 = tn * BN;
    const int split = blockIdx.y;
    const int zz = blockIdx.z;
    const int group = g.nphase == 1 ? zz : (g.nphase == 4 ? zz >> 2 : zz / g.nphase);
    const int phase = zz - group * g.nphase, pa = phase >> 1, pb = phase & 1;
    const float* gx = sel4(group, a.x0, a.x1, a.x2, a.x3);
    const float* gbias = sel4(group, a.b0, a.b1, a.b2, a.b3);
    float* gy = sel4(group, a.y0, a.y1, a.y2, a.y3);
    int ks = split * a.kchunk;
    const int kend = min(g.Ktot, ks + a.kchunk);
    int T = (kend - ks + BK - 1) / BK;
    const int pos_u = PM ? (m0 >> g.pm_lg) : 0;      // PM: the tile's pixel position (wave-uniform; N is a multiple of BM)
    const float* wph = sel4(group, a.w0, a.w1, a.w2, a.w3) + (long)phase * g.Ktot * g.Cout;

    // ---- A: lane -> (row a_r + ARPP p, LDS quad position a_kv); the quad it loads is a_kv ^ swz(row)
    const int a_kv = tid % KV, a_r = tid / KV;
    const int a_sw = KV == 8 ? ((a_r >> 1) & 7) : ((a_r >> 2) & 3);   // ARPP is a multiple of 16: the same for every pass
    int rowb[AROWS], r_oy[AROWS], r_ox[AROWS];
    bool r_ok[AROWS];
#pragma unroll
    for (int p = 0; p < AROWS; ++p) {
        const int m = m0 + a_r + ARPP * p;
        r_ok[p] = m < g.M;
        int n, oy, ox;
        if (PM) {
            n = m & (g.pmn - 1);
            oy = g.lgW >= 0 ? pos_u >> g.lgW : pos_u / g.Wg;
            ox = pos_u - oy * g.Wg;
        } else pix_decode(g, r_ok[p] ? m : 0, n, oy, ox);
        r_oy[p] = oy; r_ox[p] = ox;
        rowb[p] = ((n * g.Hs + oy * g.td.ss) * g.Ws + ox * g.td.ss) * g.Cin;
    }
    const int b_nv = tid % NVEC, b_kr = tid / NVEC;

    unsigned long long cur[AROWS];     // tap-validity mask of each row, shifted so bit 0 = next tile's tap
    unsigned long long um = 0ull;      // PM: the tile's tap-validity mask (unshifted, wave-uniform)
    unsigned rowbytes[AROWS];
    unsigned bvoff[BPASS];
    int tapi = 0, ci0 = 0, toff = 0, minoff = 0;
    {
        const TapDesc& d = g.td;
        const int kh = d.kk / d.kw;
#pragma unroll
        for (int p = 0; p < AROWS; ++p) cur[p] = 0ull;
        for (int grp = 0; grp < d.ngroups; ++grp) {
            int ga = pa, gb = pb;
            if (d.ngroups > 1) { ga = grp >> 1; gb = grp & 1; }
            unsigned xb[AROWS];
#pragma unroll
            for (int p = 0; p < AROWS; ++p) xb[p] = 0u;
            for (int rx = 0; rx < d.kw; ++rx) {
                const int tx = d.sgn * ((gb ? d.r0x1 : d.r0x0) + rx);
#pragma unroll
                for (int p = 0; p < AROWS; ++p)
                    xb[p] |= ((unsigned)(r_ox[p] + tx) < (unsigned)g.Wv ? 1u : 0u) << rx;
            }
            for (int ry = 0; ry < kh; ++ry) {
                const int ty = d.sgn * ((ga ? d.r0y1 : d.r0y0) + ry);
                const int sh = grp * d.kk + ry * d.kw;
#pragma unroll
                for (int p = 0; p < AROWS; ++p)
                    if (r_ok[p] && (unsigned)(r_oy[p] + ty) < (unsigned)g.Hv) cur[p] |= (unsigned long long)xb[p] << sh;
            }
        }
        minoff = sel4(phase, g.minoff0, g.minoff1, g.minoff2, g.minoff3);   // host-computed (finish_geom)
        if (PM) {
            // every row of the tile has the mask of row 0; the tile's VALID taps (16 .. 49 of the 7x7 kernel at 8x8) are walked in tap
            // order, in as many equal K units as their length asks for (a.kchunk = the unit the host aims at), one workgroup each
            um = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(cur[0] >> 32)) << 32) |
                 (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)cur[0]);   // (the builtin returns int: no sign extension)
            const int kv = __builtin_popcountll(um) * g.Cin;
            const int nunits = (kv + a.kchunk - 1) / a.kchunk;        // <= gridDim.y; a workgroup past them has nothing to do:
            if (split >= nunits) return;                              // the reduce adds this tile's nunits partials only
            const int kch = ((kv / BK + nunits - 1) / nunits) * BK;
            const int ksv = split * kch, kev = min(kv, ksv + kch);
            T = kev > ksv ? (kev - ksv) / BK : 0;
            const int vi = ksv / g.Cin;
            ci0 = ksv - vi * g.Cin;
            unsigned long long r = um;
            for (int c = 0; c < vi; ++c) r &= r - 1;
            tapi = r ? __builtin_ctzll(r) : g.ntaps;
        } else {
        tapi = ks == 0 ? 0 : ks / g.Cin;
        ci0 = ks - tapi * g.Cin;
        }
        { int ty, tx; tap_decode(g, tapi, pa, pb, ty, tx, toff); }
#pragma unroll
        for (int p = 0; p < AROWS; ++p) {
            cur[p] >>= tapi;
            rowbytes[p] = (unsigned)(rowb[p] + 4 * (a_kv ^ a_sw)) * 4u;
        }
#pragma unroll
        for (int q = 0; q < BPASS; ++q) {
            const int kr = b_kr + q * BRPP;
            const int n = n0 + 4 * b_nv;
            bvoff[q] = (kr < BK && n < g.Cout) ? (unsigned)(kr * g.Cout + n) * 4u : OOB;
        }
    }
    // base shifted down by the most negative tap offset so that the SGPR offset stays non-negative
    __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void*)(gx + minoff), 0, 0x7fffffff, 0x00020000);
    __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)wph, 0, 0x7fffffff, 0x00020000);

    // request tile k0 (the FAST-path state describes it) into buffer `buf`; the wave's own 1 KB pieces
    auto dma_tile = [&](int k0, auto bufc) {
        constexpr int buf = decltype(bufc)::value;
        float* A = buf ? As1 : As0;
        float* B = buf ? Bs1 : Bs0;
        const int soff = (toff - minoff + ci0) * 4;
        if (PM) k0 = tapi * g.Cin + ci0;       // the weight rows of the tap the walk is at
#pragma unroll
        for (int p = 0; p < AROWS; ++p) {
            const unsigned voff = (PM || ((unsigned)cur[p] & 1u)) ? rowbytes[p] : OOB;   // PM: only valid taps are visited
            glds16(rsx, A + (ARPP * p + wave * RPW) * BK, voff, soff);
        }
        const int sb = k0 * g.Cout * 4;
#pragma unroll
        for (int q = 0; q < BPASS; ++q) {
            const int row0 = q * BRPP + wave * BRPW;   // wave-uniform
            if (BPASS * BRPP == BK || row0 < BK)
                glds16(rsw, B + row0 * BN, bvoff[q], sb);
        }
        ci0 += BK;
        if (ci0 >= g.Cin) {
            ci0 = 0;
            if (PM) {
                const unsigned long long rest = um >> (tapi + 1);      // tapi + 1 <= 64 taps - 1 (finish_geom caps a launch at 64 taps ...)
                tapi = rest ? tapi + 1 + __builtin_ctzll(rest) : g.ntaps;
            } else {
            ++tapi;
#pragma unroll
            for (int p = 0; p < AROWS; ++p) cur[p] >>= 1;
            }
            if (tapi < g.ntaps) { int ty, tx; tap_decode(g, tapi, pa, pb, ty, tx, toff); }
        }
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    T = CG_PROBE_HALF(1, T);
    if (T > 0) dma_tile(ks, std::integral_constant<int, 0>{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    CG_STAMP(1);

    // fragment addresses: A quad (row, kq) at float4 index row * KV + (kq ^ swz(row)); swz depends on l31 only
    const int f_sw = KV == 8 ? ((l31 >> 1) & 7) : ((l31 >> 2) & 3);
    int qa_rd[G2];
#pragma unroll
    for (int gq = 0; gq < G2; ++gq) qa_rd[gq] = (wm0 + l31) * KV + ((2 * gq + h) ^ f_sw);
    const int qb_rd = 4 * h * BN + wn0 + l31;

    auto k_tile = [&](auto bufc, int t) {
        constexpr int buf = decltype(bufc)::value;
        if (t + 1 < T) dma_tile(ks + (t + 1) * BK, std::integral_constant<int, buf ^ 1>{});
        const float4* A4 = reinterpret_cast<const float4*>(buf ? As1 : As0);
        const float* B = (buf ? Bs1 : Bs0) + qb_rd;
#pragma unroll
        for (int gq = 0; gq < G2; ++gq) {
            float4 af[MI];
#pragma unroll
            for (int i = 0; i < MI; ++i) af[i] = A4[qa_rd[gq] + i * 32 * KV];
#pragma unroll
            for (int sidx = 0; sidx < 4; ++sidx) {
                float bv[NI];
#pragma unroll
                for (int j = 0; j < NI; ++j) bv[j] = B[(8 * gq + sidx) * BN + j * 32];
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(f4c(af[i], sidx), bv[j], acc[i][j], 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of tile t + 1 are in LDS
        __syncthreads();
    };
    for (int t = 0; t < T; t += 2) {
        k_tile(std::integral_constant<int, 0>{}, t);
        if (t + 1 < T) k_tile(std::integral_constant<int, 1>{}, t + 1);
    }

    CG_STAMP(2);
    // ---- epilogue (as igemm_nn_kernel): D[i][j], i = (r&3) + 8*(r>>2) + 4*h (pixel), j = l31 (channel)
    const bool partial = a.nsplit > 1;
    const bool add_bias = (gbias != nullptr) && !partial;
    float* yout = partial ? a.part : gy;
    const int act = partial ? 0 : a.act;
    float* zout = act ? sel4(group, a.z0, a.z1, a.z2, a.z3) : nullptr;
    float aslope = a.slope;
    if (act == 1) aslope = *sel4(group, a.al0, a.al1, a.al2, a.al3);
    const bool stats = a.stats != nullptr && !partial;
    float bj[NI], s1[NI], s2[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int n = n0 + wn0 + j * 32 + l31;
        bj[j] = (add_bias && n < g.Cout) ? gbias[n] : 0.f;
        s1[j] = 0.f; s2[j] = 0.f;
    }
    {
        // lean path: full tile, consecutive output rows, byte offsets inside the 2 GB a buffer descriptor spans
        if (PM && !partial) {
            // position-major tile: 64 images at one pixel - the accumulator rows are an image apart (host: full tiles, < 2 GB)
            const int hwc = g.Hg * g.Wg * g.Cout;
            const unsigned vo = ((unsigned)((m0 & (g.pmn - 1)) + wm0 + 4 * h) * (unsigned)hwc + (unsigned)(pos_u * g.Cout + n0 + wn0 + l31)) * 4u;
            if (act) nn_store_lean<MI, NI, true>(acc, bj, yout, zout, vo, hwc * 4, act, aslope);
            else nn_store_lean<MI, NI, false>(acc, bj, yout, nullptr, vo, hwc * 4, 0, 0.f);
            CG_STAMP(3);
            return;
        }
        const bool lin = partial || (g.so == 1 && g.nphase == 1);
        const long rows_total = partial ? (long)a.nsplit * a.ngroups * g.nphase * g.M : (long)g.M;
        if (lin && !stats && m0 + BM <= g.M && n0 + BN <= g.Cout && rows_total * g.Cout < 0x1fffffffL) {
            const long rowbase = partial ? (long)(split * (a.ngroups * g.nphase) + zz) * g.M : 0L;
            const unsigned vo = ((unsigned)(rowbase + m0 + wm0 + 4 * h) * (unsigned)g.Cout + (unsigned)(n0 + wn0 + l31)) * 4u;
            if (act) nn_store_lean<MI, NI, true>(acc, bj, yout, zout, vo, g.Cout * 4, act, aslope);
            else nn_store_lean<MI, NI, false>(acc, bj, yout, nullptr, vo, g.Cout * 4, 0, 0.f);
            CG_STAMP(3);
            return;
        }
        // lean STRIDED path (round 5): the phase-folded forward of a layer behind an upsampling (output pixel (2 oy + a, 2 ox + b), so == 2)
        // with or without the batch-norm statistics of the layer behind it - power-of-two grids, full tiles, no fused activation, an
        // output below 2 GB: the pixel decode is shifts and masks, the byte offset of an accumulator row one 32-bit value, a row past
        // the end stores to the out-of-range offset.  (The generic loop below decodes, bounds-checks and builds a 64-bit address per
        // VALUE.)  Same values, same order of the statistics' sums.
        if (!partial && !act && g.lgW >= 0 && m0 + BM <= g.M && n0 + BN <= g.Cout && !lin &&
            (long)(g.M >> g.lgHW) * g.Hout * g.Wout * g.Cout * 4L < 0x7fffffffL) {
            const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)yout, 0, 0x7fffffff, 0x00020000);
            const int colb = (n0 + wn0 + l31) * 4;
#pragma unroll
            for (int i = 0; i < MI; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    const int n = m >> g.lgHW, oy = (m >> g.lgW) & (g.Hg - 1), ox = m & (g.Wg - 1);
                    const int vo = ((n * g.Hout + oy * g.so + pa) * g.Wout + ox * g.so + pb) * g.Cout * 4 + colb;
#pragma unroll
                    for (int j = 0; j < NI; ++j) {
                        const float v = acc[i][j][r] + bj[j];
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), ry, vo, j * 128, 0);
                        if (stats) { s1[j] += v; s2[j] += v * v; }
                    }
                }
            }
            if (stats) {
                const int srow = (zz * (int)((g.M + BM - 1) / BM) + tm) * WM + wave / WN;
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    const float t1 = s1[j] + __shfl_xor(s1[j], 32, 64), t2 = s2[j] + __shfl_xor(s2[j], 32, 64);
                    if (h == 0) {
                        a.stats[((long)srow * 2 + 0) * g.Cout + n0 + wn0 + j * 32 + l31] = t1;
                        a.stats[((long)srow * 2 + 1) * g.Cout + n0 + wn0 + j * 32 + l31] = t2;
                    }
                }
            }
            CG_STAMP(3);
            return;
        }
    }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (m >= g.M) continue;
            const long ro = partial ? ((long)(split * (a.ngroups * g.nphase) + zz) * g.M + m) * g.Cout : out_row(g, m, pa, pb);
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int n = n0 + wn0 + j * 32 + l31;
                if (n < g.Cout) {
                    const float v = acc[i][j][r] + bj[j];
                    yout[ro + n] = v;
                    if (act) zout[ro + n] = apply_act(act, v, aslope);
                    if (stats) { s1[j] += v; s2[j] += v * v; }
                }
            }
        }
    }
    if (stats) {   // the two half-waves hold the two row halves of the same columns
        const int srow = (zz * (int)((g.M + BM - 1) / BM) + tm) * WM + wave / WN;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const float t1 = s1[j] + __shfl_xor(s1[j], 32, 64), t2 = s2[j] + __shfl_xor(s2[j], 32, 64);
            const int n = n0 + wn0 + j * 32 + l31;
            if (h == 0 && n < g.Cout) {
                a.stats[((long)srow * 2 + 0) * g.Cout + n] = t1;
                a.stats[((long)srow * 2 + 1) * g.Cout + n] = t2;
            }
        }
    }
    CG_STAMP(3);
}

// split-K reduce for NN: y_group[out_row(m)][n] = bias_group[n] + sum_s part[s][group*nphase+phase][m][n]
// LANES > 1: 256/LANES outputs per workgroup, the S partials of one output shared by LANES threads (small outputs
// with many splits would otherwise be one long serial load chain per thread).
template <int LANES, bool V4 = false>
__global__ __launch_bounds__(256) void nn_splitk_reduce_kernel(NNArgs a, int S) {
    constexpr int OUTS = 256 / LANES;
    __shared__ float sh[LANES > 1 ? LANES : 1][OUTS + 1];
    const Geom& g = a.g;
    const long PMN = (long)a.ngroups * g.nphase * g.M * g.Cout;
    if (LANES == 1 && V4) {
        // four consecutive channels of one output row per thread (host: Cout % 4 == 0, 16-byte aligned tensors): 16-byte loads, four
        // partial planes in flight, each element's splits still added in order 0 .. S-1 - the same bits as the scalar form below,
        // which moved 12 MB in 41 us beside the other queue's GEMMs (round 5)
        const long PMN4 = PMN >> 2;
        const int C4 = g.Cout >> 2;
        const float4* part4 = reinterpret_cast<const float4*>(a.part);
        for (long i4 = blockIdx.x * 256L + threadIdx.x; i4 < PMN4; i4 += (long)gridDim.x * 256) {
            const int n = (int)(i4 % C4) * 4;
            const long pm = i4 / C4;
            const int m = (int)(pm % g.M);
            const int zz = (int)(pm / g.M);
            int Se = S;
            if (g.pmn) Se = (pm_valid_taps(g, m >> g.pm_lg) * g.Cin + a.kchunk - 1) / a.kchunk;
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
            int k = 0;
            for (; k + 4 <= Se; k += 4) {
                const float4 p0 = part4[(long)k * PMN4 + i4], p1 = part4[(long)(k + 1) * PMN4 + i4];
                const float4 p2 = part4[(long)(k + 2) * PMN4 + i4], p3 = part4[(long)(k + 3) * PMN4 + i4];
                s.x += p0.x; s.y += p0.y; s.z += p0.z; s.w += p0.w;
                s.x += p1.x; s.y += p1.y; s.z += p1.z; s.w += p1.w;
                s.x += p2.x; s.y += p2.y; s.z += p2.z; s.w += p2.w;
                s.x += p3.x; s.y += p3.y; s.z += p3.z; s.w += p3.w;
            }
            for (; k < Se; ++k) {
                const float4 p0 = part4[(long)k * PMN4 + i4];
                s.x += p0.x; s.y += p0.y; s.z += p0.z; s.w += p0.w;
            }
            const int group = zz / g.nphase, phase = zz - group * g.nphase;
            const float* gbias = sel4(group, a.b0, a.b1, a.b2, a.b3);
            float* gy = sel4(group, a.y0, a.y1, a.y2, a.y3);
            if (gbias) { const float4 b = ld4(gbias + n); s.x += b.x; s.y += b.y; s.z += b.z; s.w += b.w; }
            const long o = out_row(g, m, phase >> 1, phase & 1) + n;
            *reinterpret_cast<float4*>(gy + o) = s;
            if (a.act) {
                const float al = a.act == 1 ? *sel4(group, a.al0, a.al1, a.al2, a.al3) : a.slope;
                *reinterpret_cast<float4*>(sel4(group, a.z0, a.z1, a.z2, a.z3) + o) =
                    make_float4(apply_act(a.act, s.x, al), apply_act(a.act, s.y, al), apply_act(a.act, s.z, al), apply_act(a.act, s.w, al));
            }
        }
        return;
    }
    const int ol = threadIdx.x % OUTS, ln = threadIdx.x / OUTS;
    for (long i0 = blockIdx.x * (long)OUTS; i0 < PMN; i0 += (long)gridDim.x * OUTS) {
        const long i = i0 + ol;
        float s = 0.f;
        if (i < PMN) {
            int Se = S;
            if (g.pmn) Se = (pm_valid_taps(g, (int)((i / g.Cout) % g.M) >> g.pm_lg) * g.Cin + a.kchunk - 1) / a.kchunk;   // this pixel's K units
            for (int k = ln; k < Se; k += LANES) s += a.part[(long)k * PMN + i];
        }
        if (LANES > 1) {
            __syncthreads();
            sh[ln][ol] = s;
            __syncthreads();
            if (ln != 0) continue;
            s = 0.f;
#pragma unroll
            for (int r = 0; r < LANES; ++r) s += sh[r][ol];
        }
        if (i >= PMN) continue;
        const int n = (int)(i % g.Cout);
        const long pm = i / g.Cout;
        const int m = (int)(pm % g.M);
        const int zz = (int)(pm / g.M);
        const int group = zz / g.nphase, phase = zz - group * g.nphase;
        const float* gbias = sel4(group, a.b0, a.b1, a.b2, a.b3);
        float* gy = sel4(group, a.y0, a.y1, a.y2, a.y3);
        if (gbias) s += gbias[n];
        const long o = out_row(g, m, phase >> 1, phase & 1) + n;
        gy[o] = s;
        if (a.act) {
            const float al = a.act == 1 ? *sel4(group, a.al0, a.al1, a.al2, a.al3) : a.slope;
            sel4(group, a.z0, a.z1, a.z2, a.z3)[o] = apply_act(a.act, s, al);
        }
    }
}

// ---------------------------------------------------------------------------
// TN: dW[k][n] = sum_m A(m,k) * dY[m][n]   (k = (tap,ci) rows, m = grid pixels reduced)
// ---------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN, bool VECA, bool VECB>
__global__ __launch_bounds__(256, 4) void igemm_tn_kernel(TNArgs a) {
    static_assert(WM * WN == 4, "4 waves per workgroup");
    constexpr int MI = BM / WM / 32;
    constexpr int NI = BN / WN / 32;
    constexpr int LDA = BM, LDB = BN;
    constexpr int A_TILE = BK * LDA, B_TILE = BK * LDB;
    constexpr int AVEC = BM / 4, ARPP = 256 / AVEC, APASS = (BK + ARPP - 1) / ARPP;
    constexpr int BVEC = BN / 4, BRPP = 256 / BVEC, BPASS = (BK + BRPP - 1) / BRPP;
    constexpr int NJ = VECA ? 1 : 4;

    __shared__ __attribute__((aligned(16))) float smem[2 * A_TILE + 2 * B_TILE];
    float* As = smem;
    float* Bs = smem + 2 * A_TILE;

    const Geom& g = a.g;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wm0 = (wave / WN) * (BM / WM);
    const int wn0 = (wave % WN) * (BN / WN);

    const int ntn = (g.Cout + BN - 1) / BN;
    int bid = blockIdx.x, split = blockIdx.y;
    if ((a.xcd_swizzle & 2) && (gridDim.y & 7) == 0) {
        // XCD = pixel chunk: workgroup L of the dispatch order (x fastest, then y) runs on XCD L % 8; give every XCD the splits
        // congruent to its number, for all tiles - it then reads one eighth of x and dy (each pixel chunk is wanted by every
        // tile), instead of every XCD reading all of dy for its few tiles
        const int L = (int)blockIdx.y * (int)gridDim.x + (int)blockIdx.x, s8 = (int)gridDim.y >> 3;
        split = (L & 7) + 8 * ((L >> 3) % s8);
        bid = (L >> 3) / s8;
    } else if ((a.xcd_swizzle & 1) && (gridDim.x & 7) == 0) {
        bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);   // see igemm_nn_kernel
    }
    const int tn = ntn == 1 ? 0 : bid % ntn, tm = ntn == 1 ? bid : bid / ntn;
    const int m0 = tm * BM, n0 = tn * BN;  // m0: row of dW (tap,ci)
    const int zz = blockIdx.z;
    const int group = g.nphase == 1 ? zz : (g.nphase == 4 ? zz >> 2 : zz / g.nphase);
    const int phase = zz - group * g.nphase, pa = phase >> 1, pb = phase & 1;
    const float* gx = a.xgs ? a.x0 + (long)group * a.xgs : sel4(group, a.x0, a.x1, a.x2, a.x3);
    const float* gdy = a.dgs ? a.d0 + (long)group * a.dgs : sel4(group, a.d0, a.d1, a.d2, a.d3);
    const int ps = split * a.pchunk;
    const int pend = min(g.M, ps + a.pchunk);
    const int T = CG_PROBE_HALF(2, (pend - ps + BK - 1) / BK);

    // A staging: fixed (tap,ci) columns per thread, pixel rows vary per tile
    const int a_mv = tid % AVEC, a_kr = tid / AVEC;
    int c_off[NJ], c_ty[NJ], c_tx[NJ];
    bool c_ok[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int mm = m0 + 4 * a_mv + j;
        c_ok[j] = mm < g.Ktot;
        const int mc = c_ok[j] ? mm : 0;
        const int tap = mc / g.Cin;
        int off;
        tap_decode(g, tap, pa, pb, c_ty[j], c_tx[j], off);
        c_off[j] = off + (mc - tap * g.Cin);
    }
    const int b_nv = tid % BVEC, b_kr = tid / BVEC;
    const bool b_nok = n0 + 4 * b_nv < g.Cout;
    __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void*)gx, 0, 0x7fffffff, 0x00020000);
    __amdgpu_buffer_rsrc_t rsd = __builtin_amdgcn_make_buffer_rsrc((void*)gdy, 0, 0x7fffffff, 0x00020000);

    float4 areg[APASS];
    float4 breg[BPASS];
    // gradBias rides along: the dW row-tile 0 of every column tile also sums the dy rows it streams
    const bool do_bias = a.bias_part != nullptr && tm == 0;
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);

    // Lean per-tile address math (every VALU op here costs MFMA issue slots): 32-bit offsets, the pixel decoded
    // once per row and shared by both operands when their row mappings coincide, source base = pix*Cin when the
    // grid and the source tensor have the same geometry (always true for "same"-padded layers).
    const bool lin_src = g.td.ss == 1 && g.Hs == g.Hg && g.Ws == g.Wg;
    const bool lin_out = g.so == 1 && g.nphase == 1;
    constexpr bool SAME_ROWS = (AVEC == BVEC);

    // ---- lean tile addressing for power-of-two grids (all layers of the models): a K tile is 16 consecutive grid pixels
    // starting at a multiple of 16, so the image and the tile's first row / column are wave-uniform (SALU), while a
    // thread's own row / column offset inside the tile, its tap and its byte offsets are loop invariants.  Per tile and
    // operand row that leaves two adds, two compares and a select instead of ~25 VALU of pixel decoding.
    const int HWg = g.Hg * g.Wg;
    const bool lean = VECA && VECB && g.lgW >= 0 && g.lgHW >= 0 && HWg >= BK && lin_src && (lin_out || g.so == 2) &&
                      (ps % BK) == 0 && (pend % BK) == 0 && APASS * ARPP == BK && BPASS * BRPP == BK;
    int l_ay[APASS], l_ax[APASS];
    unsigned l_avoff[APASS], l_bvoff[BPASS];
    int l_minoff = 0;
    __amdgpu_buffer_rsrc_t rsx_l = rsx;
    if (lean) {
        l_minoff = sel4(phase, g.minoff0, g.minoff1, g.minoff2, g.minoff3);   // host-computed (finish_geom)
        rsx_l = __builtin_amdgcn_make_buffer_rsrc((void*)(gx + l_minoff), 0, 0x7fffffff, 0x00020000);
#pragma unroll
        for (int q = 0; q < APASS; ++q) {
            const int kr = a_kr + q * ARPP;
            const int kyo = g.Wg >= BK ? 0 : (kr >> g.lgW), kxo = g.Wg >= BK ? kr : (kr & (g.Wg - 1));
            l_ay[q] = kyo + c_ty[0];
            l_ax[q] = kxo + c_tx[0];
            l_avoff[q] = c_ok[0] ? (unsigned)(kr * g.Cin + c_off[0] - l_minoff) * 4u : OOB;
        }
#pragma unroll
        for (int q = 0; q < BPASS; ++q) {
            const int kr = b_kr + q * BRPP;
            const int kyo = g.Wg >= BK ? 0 : (kr >> g.lgW), kxo = g.Wg >= BK ? kr : (kr & (g.Wg - 1));
            const int rel = lin_out ? kr * g.Cout : (kyo * g.so * g.Wout + kxo * g.so) * g.Cout;
            l_bvoff[q] = b_nok ? (unsigned)(rel + n0 + 4 * b_nv) * 4u : OOB;
        }
    }
    auto load_tile = [&](int p0) {
        if (lean) {
            const int r0 = p0 & (HWg - 1);
            const int oyb = r0 >> g.lgW, oxb = r0 & (g.Wg - 1);
            const int soa = p0 * g.Cin * 4;
#pragma unroll
            for (int q = 0; q < APASS; ++q) {
                const bool ok = (unsigned)(oyb + l_ay[q]) < (unsigned)g.Hv && (unsigned)(oxb + l_ax[q]) < (unsigned)g.Wv;
                areg[q] = bufld4(rsx_l, ok ? l_avoff[q] : OOB, soa);
            }
            int sob;
            if (lin_out) sob = p0 * g.Cout * 4;
            else {
                const int nimg = p0 >> g.lgHW;
                sob = (((nimg * g.Hout + oyb * g.so + pa) * g.Wout + oxb * g.so + pb) * g.Cout) * 4;
            }
#pragma unroll
            for (int q = 0; q < BPASS; ++q) breg[q] = bufld4(rsd, l_bvoff[q], sob);
            return;
        }
        int rn[APASS], roy[APASS], rox[APASS];
#pragma unroll
        for (int q = 0; q < APASS; ++q) {
            const int kr = a_kr + q * ARPP;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (APASS * ARPP == BK || kr < BK) {
                const int pix = p0 + kr;
                const bool pok = pix < pend;
                int n, oy, ox;
                pix_decode(g, pok ? pix : 0, n, oy, ox);
                rn[q] = n; roy[q] = oy; rox[q] = ox;
                const int base = lin_src ? pix * g.Cin : ((n * g.Hs + oy * g.td.ss) * g.Ws + ox * g.td.ss) * g.Cin;
                if (VECA) {
                    const bool ok = pok && c_ok[0] && (unsigned)(oy + c_ty[0]) < (unsigned)g.Hv &&
                                    (unsigned)(ox + c_tx[0]) < (unsigned)g.Wv;
                    v = bufld4(rsx, ok ? (unsigned)(base + c_off[0]) * 4u : OOB, 0);
                } else {
                    float t4[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int jj = VECA ? 0 : j;
                        const bool ok = pok && c_ok[jj] && (unsigned)(oy + c_ty[jj]) < (unsigned)g.Hv &&
                                        (unsigned)(ox + c_tx[jj]) < (unsigned)g.Wv;
                        float e = 0.f;
                        if (ok) e = gx[(long)base + c_off[jj]];
                        t4[j] = e;
                    }
                    v = make_float4(t4[0], t4[1], t4[2], t4[3]);
                }
            }
            areg[q] = v;
        }
#pragma unroll
        for (int q = 0; q < BPASS; ++q) {
            const int kr = b_kr + q * BRPP;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (BPASS * BRPP == BK || kr < BK) {
                const int pix = p0 + kr;
                const bool pok = pix < pend;
                const int n = n0 + 4 * b_nv;
                int ro;
                if (lin_out) {
                    ro = pix * g.Cout;
                } else {
                    int pn, poy, pox;
                    if (SAME_ROWS) { pn = rn[q < APASS ? q : 0]; poy = roy[q < APASS ? q : 0]; pox = rox[q < APASS ? q : 0]; }
                    else pix_decode(g, pok ? pix : 0, pn, poy, pox);
                    ro = ((pn * g.Hout + poy * g.so + pa) * g.Wout + pox * g.so + pb) * g.Cout;
                }
                if (VECB) {
                    v = bufld4(rsd, (pok && b_nok) ? (unsigned)(ro + n) * 4u : OOB, 0);
                } else if (pok) {
                    const float* dp = gdy + (long)ro + n;
                    if (n + 0 < g.Cout) v.x = dp[0];
                    if (n + 1 < g.Cout) v.y = dp[1];
                    if (n + 2 < g.Cout) v.z = dp[2];
                    if (n + 3 < g.Cout) v.w = dp[3];
                }
            }
            breg[q] = v;
        }
    };

    auto store_tile = [&](auto bufc) {   // compile-time buffer index, as in igemm_nn_kernel
        constexpr int buf = decltype(bufc)::value;
        float* A = As + buf * A_TILE;
        float* B = Bs + buf * B_TILE;
#pragma unroll
        for (int q = 0; q < APASS; ++q) {
            const int kr = a_kr + q * ARPP;
            if (APASS * ARPP == BK || kr < BK) *reinterpret_cast<float4*>(A + kr * LDA + 4 * a_mv) = areg[q];
        }
#pragma unroll
        for (int q = 0; q < BPASS; ++q) {
            const int kr = b_kr + q * BRPP;
            if (BPASS * BRPP == BK || kr < BK) {
                *reinterpret_cast<float4*>(B + kr * LDB + 4 * b_nv) = breg[q];
                if (do_bias) { bsum.x += breg[q].x; bsum.y += breg[q].y; bsum.z += breg[q].z; bsum.w += breg[q].w; }
            }
        }
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (T > 0) {
        load_tile(ps);
        store_tile(std::integral_constant<int, 0>{});
    }
    __syncthreads();

    auto k_tile = [&](auto bufc, int t) {
        constexpr int buf = decltype(bufc)::value;
        if (t + 1 < T) load_tile(ps + (t + 1) * BK);
        const float* A = As + buf * A_TILE + wm0 + l31;
        const float* B = Bs + buf * B_TILE + wn0 + l31;
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float av[MI], bv[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) av[i] = A[(kk + h) * LDA + i * 32];
#pragma unroll
            for (int j = 0; j < NI; ++j) bv[j] = B[(kk + h) * LDB + j * 32];
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
        if (t + 1 < T) store_tile(std::integral_constant<int, buf ^ 1>{});
        __syncthreads();
    };
    for (int t = 0; t < T; t += 2) {
        k_tile(std::integral_constant<int, 0>{}, t);
        if (t + 1 < T) k_tile(std::integral_constant<int, 1>{}, t + 1);
    }

    if (do_bias) {  // all waves are past the loop's final barrier: reuse the A tile as scratch
        float* red = smem;
        *reinterpret_cast<float4*>(red + b_kr * BN + 4 * b_nv) = bsum;
        __syncthreads();
        if (tid < BN && n0 + tid < g.Cout) {
            float t = 0.f;
#pragma unroll
            for (int r = 0; r < 256 / BVEC; ++r) t += red[r * BN + tid];
            a.bias_part[(long)(split * (a.ngroups * g.nphase) + zz) * g.Cout + n0 + tid] = t;
        }
    }
    float* pout = a.part + (long)(split * (a.ngroups * g.nphase) + zz) * g.Ktot * g.Cout;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int n = n0 + wn0 + j * 32 + l31;
        if (n >= g.Cout) continue;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (m < g.Ktot) pout[(long)m * g.Cout + n] = acc[i][j][r];
            }
        }
    }
}

// ---------------------------------------------------------------------------
// TN with LDS-direct loads (see igemm_nng_kernel): both operands are pixel rows of consecutive channels, i.e. already the
// K-major [pixel][column] tiles the MFMA fragments are read from, so a wave instruction simply drops 64 / (BM / 4) pixel
// rows of 4 BM bytes into place (no swizzle, ds_read_b32 fragments exactly as in igemm_tn_kernel).  Lean addressing only
// (the host checks, tng_ok): 16-byte aligned channel quads, a source tensor with the grid's geometry, a reduction range of whole K
// tiles, and either a power-of-two grid with HWg >= 16 (a K tile never straddles two images) or a 1 x 1 kernel without padding
// (`flat`, mode 1: every row is valid and rows are simply consecutive - the linear layers and the Winograd-domain GEMMs).
//
// Mode 2, POSITION-MAJOR K tiles (round 6; models.lua:681,685 - D32_st3's 5x5 and 7x7 layers - and :696-697, the View -> Linear head as
// an H x W convolution): a K tile is ONE grid position (oy, ox) of 16 consecutive images, so all its rows share their zero-padding taps
// and a workgroup - whose dW rows belong to one or two taps - runs its K loop only over the rectangle of positions at which one of its
// taps reads inside the image: the 38 % / 14 % of the MACs that multiply padding are not issued.  The splits divide each workgroup's
// OWN tile list evenly, rows of a tile are a whole image apart (a per-lane offset, as before), and the grid needs no power-of-two
// geometry, nor 16 pixels per image.  gradBias rides on the row tile that holds the centre tap (its rectangle is the whole grid).
// ---------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256, 2) void igemm_tng_kernel(TNArgs a, int mode) {
    const int flat = mode == 1;
    const bool pm = mode == 2;
    static_assert(WM * WN == 4, "4 waves per workgroup");
    constexpr int MI = BM / WM / 32;
    constexpr int NI = BN / WN / 32;
    constexpr int LDA = BM, LDB = BN;
    constexpr int A_TILE = BK * LDA, B_TILE = BK * LDB;
    constexpr int AVEC = BM / 4, ARPP = 256 / AVEC, APASS = (BK + ARPP - 1) / ARPP, ARPW = 64 / AVEC;
    constexpr int BVEC = BN / 4, BRPP = 256 / BVEC, BPASS = (BK + BRPP - 1) / BRPP, BRPW = 64 / BVEC;

    __shared__ __attribute__((aligned(16))) float As0[A_TILE];
    __shared__ __attribute__((aligned(16))) float As1[A_TILE];
    __shared__ __attribute__((aligned(16))) float Bs0[B_TILE];
    __shared__ __attribute__((aligned(16))) float Bs1[B_TILE];

    const Geom& g = a.g;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    // (Round 5 tried INTERLEAVING a wave's 32-row blocks with the other waves' - 64 floats apart, so that a fragment pair is one
    // ds_read2st64_b32 with immediate offsets and the K loop has no address VALU at all (the ds_read2_b32 pairs below need one v_add_u32
    // per pair: their 8-bit offsets reach 1 KB).  Alone the weight gradients measured the same; in the step it cost 0.12 ms
    // (6.00 against 5.88 ms, profiles/r05_sweeps.txt) - the two reads of a pair then hit the same LDS banks.  Not kept.)
    const int wm0 = (wave / WN) * (BM / WM);
    const int wn0 = (wave % WN) * (BN / WN);

    const int ntn = (g.Cout + BN - 1) / BN;
    int bid = blockIdx.x, split = blockIdx.y;
    if ((a.xcd_swizzle & 2) && (gridDim.y & 7) == 0) {   // XCD = pixel chunk, see igemm_tn_kernel
        const int L = (int)blockIdx.y * (int)gridDim.x + (int)blockIdx.x, s8 = (int)gridDim.y >> 3;
        split = (L & 7) + 8 * ((L >> 3) % s8);
        bid = (L >> 3) / s8;
    } else if ((a.xcd_swizzle & 1) && (gridDim.x & 7) == 0) {
        bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);
    }
    const int tn = ntn == 1 ? 0 : bid % ntn, tm = ntn == 1 ? bid : bid / ntn;
    const int m0 = tm * BM, n0 = tn * BN;
    const int zz = blockIdx.z;
    const int group = g.nphase == 1 ? zz : (g.nphase == 4 ? zz >> 2 : zz / g.nphase);
    const int phase = zz - group * g.nphase, pa = phase >> 1, pb = phase & 1;
    const float* gx = a.xgs ? a.x0 + (long)group * a.xgs : sel4(group, a.x0, a.x1, a.x2, a.x3);
    const float* gdy = a.dgs ? a.d0 + (long)group * a.dgs : sel4(group, a.d0, a.d1, a.d2, a.d3);
    const int ps = split * a.pchunk;
    const int pend = min(g.M, ps + a.pchunk);
    int T = CG_PROBE_HALF(2, (pend - ps) / BK);
    int q_c = 0, q_x = 0, q_y = 0, q_x0 = 0, q_x1 = 0, q_nc = 1;   // mode 2: image chunk / position of the next tile, the rectangle's columns
    if (pm) {
        const int t_lo = m0 / g.Cin, t_hi = min(g.Ktot - 1, m0 + BM - 1) / g.Cin;
        int y0 = g.Hg, y1 = 0, x0 = g.Wg, x1 = 0;
        for (int t = t_lo; t <= t_hi; ++t) {
            int ty, tx, off;
            tap_decode(g, t, pa, pb, ty, tx, off);
            y0 = min(y0, max(0, -ty)); y1 = max(y1, min(g.Hg, g.Hv - ty));
            x0 = min(x0, max(0, -tx)); x1 = max(x1, min(g.Wg, g.Wv - tx));
        }
        const int rw = max(0, x1 - x0), rh = max(0, y1 - y0);
        q_nc = a.pm_n / BK;
        const long L = (long)rw * rh * q_nc;
        const int u0 = (int)(L * split / (int)gridDim.y), u1 = (int)(L * (split + 1) / (int)gridDim.y);
        T = CG_PROBE_HALF(2, u1 - u0);
        if (T > 0) {
            const int pos = u0 / q_nc;
            q_c = u0 - pos * q_nc;
            q_y = y0 + pos / rw; q_x = x0 + pos % rw;
        }
        q_x0 = x0; q_x1 = x1;
    }

    const int a_mv = tid % AVEC, a_kr = tid / AVEC;
    const int b_nv = tid % BVEC, b_kr = tid / BVEC;
    const bool lin_out = g.so == 1 && g.nphase == 1;
    const int HWg = g.Hg * g.Wg;
    int l_minoff = 0, c_ty = 0, c_tx = 0, c_off = 0;
    bool c_ok;
    {
        const int mm = m0 + 4 * a_mv;
        c_ok = mm < g.Ktot;
        const int mc = c_ok ? mm : 0;
        const int tap = mc / g.Cin;
        int off;
        tap_decode(g, tap, pa, pb, c_ty, c_tx, off);
        c_off = off + (mc - tap * g.Cin);
        l_minoff = sel4(phase, g.minoff0, g.minoff1, g.minoff2, g.minoff3);   // host-computed (finish_geom)
    }
    __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void*)(gx + l_minoff), 0, 0x7fffffff, 0x00020000);
    __amdgpu_buffer_rsrc_t rsd = __builtin_amdgcn_make_buffer_rsrc((void*)gdy, 0, 0x7fffffff, 0x00020000);
    const bool b_nok = n0 + 4 * b_nv < g.Cout;
    int l_ay[APASS], l_ax[APASS];
    unsigned l_avoff[APASS], l_bvoff[BPASS];
    const bool wide = flat || g.Wg >= BK;
#pragma unroll
    for (int q = 0; q < APASS; ++q) {
        const int kr = a_kr + q * ARPP;
        const int kyo = (wide || pm) ? 0 : (kr >> g.lgW), kxo = (flat || pm) ? 0 : (wide ? kr : (kr & (g.Wg - 1)));
        l_ay[q] = kyo + c_ty;
        l_ax[q] = kxo + c_tx;
        l_avoff[q] = (c_ok && kr < BK) ? (unsigned)(kr * (pm ? g.Hs * g.Ws : 1) * g.Cin + c_off - l_minoff) * 4u : OOB;
    }
#pragma unroll
    for (int q = 0; q < BPASS; ++q) {
        const int kr = b_kr + q * BRPP;
        const int byo = wide ? 0 : (kr >> g.lgW), bxo = wide ? kr : (kr & (g.Wg - 1));
        const int rel = pm ? kr * g.Hout * g.Wout * g.Cout : (lin_out ? kr * g.Cout : (byo * g.so * g.Wout + bxo * g.so) * g.Cout);
        l_bvoff[q] = (b_nok && kr < BK) ? (unsigned)(rel + n0 + 4 * b_nv) * 4u : OOB;
    }
    // gradBias rides along on the row-tile 0 workgroups (mode 2: the tile of the tap at offset (0, 0), which visits every position):
    // column sums of the dy rows, read back from the LDS tile
    const int tm_bias = pm ? ((-g.td.r0y0) * g.td.kw + (-g.td.r0x0)) * g.Cin / BM : 0;
    const bool do_bias = a.bias_part != nullptr && tm == tm_bias;
    float bsum = 0.f;

    auto dma_tile = [&](int p0, auto bufc) {
        constexpr int buf = decltype(bufc)::value;
        float* A = buf ? As1 : As0;
        float* B = buf ? Bs1 : Bs0;
        int oyb, oxb, soa, sob;
        if (pm) {   // wave-uniform: the next tile of this workgroup's list
            oyb = q_y; oxb = q_x;
            soa = ((q_c * BK * g.Hs + q_y) * g.Ws + q_x) * g.Cin * 4;
            sob = ((q_c * BK * g.Hout + q_y) * g.Wout + q_x) * g.Cout * 4;
            if (++q_c == q_nc) {
                q_c = 0;
                if (++q_x == q_x1) { q_x = q_x0; ++q_y; }
            }
        } else {
            const int r0 = flat ? 0 : (p0 & (HWg - 1));
            oyb = flat ? 0 : (r0 >> g.lgW); oxb = flat ? 0 : (r0 & (g.Wg - 1));
            soa = p0 * g.Cin * 4;
            if (lin_out) sob = p0 * g.Cout * 4;
            else {
                const int nimg = p0 >> g.lgHW;
                sob = (((nimg * g.Hout + oyb * g.so + pa) * g.Wout + oxb * g.so + pb) * g.Cout) * 4;
            }
        }
#pragma unroll
        for (int q = 0; q < APASS; ++q) {
            const int row0 = q * ARPP + wave * ARPW;   // wave-uniform
            if (APASS * ARPP == BK || row0 < BK) {
                const bool ok = (unsigned)(oyb + l_ay[q]) < (unsigned)g.Hv && (unsigned)(oxb + l_ax[q]) < (unsigned)g.Wv;
                glds16(rsx, A + row0 * LDA, ok ? l_avoff[q] : OOB, soa);
            }
        }
#pragma unroll
        for (int q = 0; q < BPASS; ++q) {
            const int row0 = q * BRPP + wave * BRPW;
            if (BPASS * BRPP == BK || row0 < BK) glds16(rsd, B + row0 * LDB, l_bvoff[q], sob);
        }
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (T > 0) dma_tile(ps, std::integral_constant<int, 0>{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    auto k_tile = [&](auto bufc, int t) {
        constexpr int buf = decltype(bufc)::value;
        if (t + 1 < T) dma_tile(ps + (t + 1) * BK, std::integral_constant<int, buf ^ 1>{});
        const float* A = (buf ? As1 : As0) + wm0 + l31;
        const float* B = (buf ? Bs1 : Bs0) + wn0 + l31;
        if (do_bias && tid < BN) {
#pragma unroll
            for (int kk = 0; kk < BK; ++kk) bsum += (buf ? Bs1 : Bs0)[kk * LDB + tid];
        }
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float av[MI], bv[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) av[i] = A[(kk + h) * LDA + i * 32];
#pragma unroll
            for (int j = 0; j < NI; ++j) bv[j] = B[(kk + h) * LDB + j * 32];
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };
    for (int t = 0; t < T; t += 2) {
        k_tile(std::integral_constant<int, 0>{}, t);
        if (t + 1 < T) k_tile(std::integral_constant<int, 1>{}, t + 1);
    }

    if (do_bias && tid < BN && n0 + tid < g.Cout)
        a.bias_part[(long)(split * (a.ngroups * g.nphase) + zz) * g.Cout + n0 + tid] = bsum;
    float* pout = a.part + (long)(split * (a.ngroups * g.nphase) + zz) * g.Ktot * g.Cout;
    // lean path (round 5, as nn_store_lean): a FULL tile stores through a buffer descriptor at (per-lane byte offset) + (row offset in an
    // SGPR) - no bounds test, no 64-bit address per value (the generic loop below is ~2000 instructions per wave, 1 VALU per MFMA of the
    // whole launch; the partial sums of one (split, phase) plane always fit the 2 GB a descriptor spans)
    if (m0 + BM <= g.Ktot && n0 + BN <= g.Cout && (long)g.Ktot * g.Cout * 4L < 0x7fffffffL) {
        const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc((void*)pout, 0, 0x7fffffff, 0x00020000);
        const unsigned vo = ((unsigned)(m0 + wm0 + 4 * h) * (unsigned)g.Cout + (unsigned)(n0 + wn0 + l31)) * 4u;
        const int c4 = g.Cout * 4;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int soff = (i * 32 + (r & 3) + 8 * (r >> 2)) * c4;   // wave-uniform
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[i][j][r]), rp, (int)(vo + j * 128), soff, 0);
            }
        return;
    }
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int n = n0 + wn0 + j * 32 + l31;
        if (n >= g.Cout) continue;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (m < g.Ktot) pout[(long)m * g.Cout + n] = acc[i][j][r];
            }
        }
    }
}

// canonical tap d of a k-tap kernel (pad p) seen from output phase a -> index of the low-res tap it folds into
__device__ __host__ __forceinline__ int phase_map(int a, int d, int pad) { return ((a + d - pad) >> 1) - ((a - pad) >> 1); }

// Split-K reduce + transpose into Torch7's canonical layout, accumulate semantics:
//   plain: gw[co][ci][tap] += scale * sum_s part[s][tap*Cin+ci][co]
//   UPS  : gw[co][ci][dy][dx] += scale * sum_s sum_{a,b} part[s][2a+b][(map(a,dy)*kp + map(b,dx))*Cin + ci][co]
// One workgroup owns CI_T input channels x 32 output channels x all taps: reads are coalesced along co, the sums
// are transposed through LDS, writes are contiguous runs of CI_T*KK floats per co.
template <bool UPS>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* part, float* gw, int Cin, int Cout, int k, int KK,
                                                           int pad, int kp, int S, float scale, int CI_T, long sstride) {
    extern __shared__ float sh[];  // [CI_T][KK][33]
    const int ci0 = blockIdx.x * CI_T, co0 = blockIdx.y * 32;
    const long plane = (long)(UPS ? kp * kp : KK) * Cin * Cout;
    const int n1 = KK * CI_T * 32;
    if (UPS && k == 3 && kp == 2 && pad == 1 && co0 + 32 <= Cout && ci0 + CI_T <= Cin && (CI_T & 31) == 0 && (Cout & 3) == 0 && ((uintptr_t)part & 15) == 0 &&
        (sstride & 3) == 0) {
        // 3x3 behind an upsampling, full block, 16-byte loads (round 5): a thread owns four output channels of one input channel and
        // reads each of the 16 (phase, low-res tap) partial elements of a split ONCE (the scalar form below re-reads them per canonical
        // tap: 36 loads for 16 values), adding them to the nine canonical taps in the same order - split by split, phase by phase.
        const int c4 = threadIdx.x & 7;
        for (int ci_l = threadIdx.x >> 3; ci_l < CI_T; ci_l += 32) {
            float4 acc[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
            const float* base = part + (long)(ci0 + ci_l) * Cout + co0 + 4 * c4;
            for (int sp = 0; sp < S; ++sp) {
                float4 v[4][4];
#pragma unroll
                for (int p = 0; p < 4; ++p)
#pragma unroll
                    for (int tp = 0; tp < 4; ++tp) v[p][tp] = ld4(base + (long)sp * sstride + (long)p * plane + (long)tp * Cin * Cout);
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int dy = t / 3, dx = t - 3 * (t / 3);
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        const float4 q = v[p][phase_map(p >> 1, dy, 1) * 2 + phase_map(p & 1, dx, 1)];
                        acc[t].x += q.x; acc[t].y += q.y; acc[t].z += q.z; acc[t].w += q.w;
                    }
                }
            }
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                float* d = sh + (ci_l * KK + t) * 33 + 4 * c4;
                d[0] = acc[t].x; d[1] = acc[t].y; d[2] = acc[t].z; d[3] = acc[t].w;
            }
        }
    } else
    for (int idx = threadIdx.x; idx < n1; idx += 256) {
        const int co_l = idx & 31;
        const int r = idx >> 5;
        const int ci_l = r % CI_T, tap = r / CI_T;
        const int ci = ci0 + ci_l, co = co0 + co_l;
        float s = 0.f;
        if (ci < Cin && co < Cout) {
            if (UPS) {
                const int dy = tap / k, dx = tap - dy * k;
                for (int sp = 0; sp < S; ++sp)
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        const int tp = phase_map(p >> 1, dy, pad) * kp + phase_map(p & 1, dx, pad);
                        s += part[(long)sp * sstride + (long)p * plane + ((long)tp * Cin + ci) * Cout + co];
                    }
            } else {
                const long o = ((long)tap * Cin + ci) * Cout + co;
                for (int sp = 0; sp < S; ++sp) s += part[(long)sp * sstride + o];
            }
        }
        sh[(ci_l * KK + tap) * 33 + co_l] = s;   // row = position in the output run: the transposed read below walks rows with stride 33 (odd: conflict-free)
    }
    __syncthreads();
    const int run = CI_T * KK;
    for (int idx = threadIdx.x; idx < n1; idx += 256) {
        const int j = idx % run, co_l = idx / run;
        const int ci_l = j / KK, tap = j - ci_l * KK;
        const int ci = ci0 + ci_l, co = co0 + co_l;
        if (ci < Cin && co < Cout) {
            float* dst = gw + ((long)co * Cin + ci) * KK + tap;
            *dst += scale * sh[j * 33 + co_l];
        }
    }
}

// Same reduction for small weight tensors (where the tiled form above would not fill the chip), all groups of a
// grouped launch and their bias gradients in ONE launch: grid (weight blocks + bias blocks, ngroups); a workgroup
// owns 32 consecutive partial elements (or 32 bias channels) and splits the S partials over 8 lanes.
struct RedPtrs { float* gw0; float* gw1; float* gw2; float* gw3; float* gb0; float* gb1; float* gb2; float* gb3;
                 long gws; };   // gws != 0: group g accumulates into gw0 + g * gws (strided groups, no bias gradients)

template <bool UPS>
__device__ __forceinline__ void wgrad_reduce_small_body(float (*sh)[33], int bx, int group, const float* part, const float* bias_part,
                                                        const RedPtrs& rp, int Cin, int Cout, int k, int KK, int pad, int kp, int S,
                                                        int P, float scale, long sstride, long gstride, long bsstride, int wblocks) {
    const int ol = threadIdx.x & 31, ln = threadIdx.x >> 5;
    float s = 0.f;
    if (bx < wblocks) {
        const long total = (long)KK * Cin * Cout;
        const long plane = (long)(UPS ? kp * kp : KK) * Cin * Cout;
        const long i = bx * 32L + ol;
        const float* pg = part + (long)group * gstride;
        int co = 0, ci = 0, tap = 0;
        if (i < total) {
            co = (int)(i % Cout);
            const long r = i / Cout;
            ci = (int)(r % Cin);
            tap = (int)(r / Cin);
            if (UPS) {
                const int dy = tap / k, dx = tap - dy * k;
                for (int sp = ln; sp < S; sp += 8)
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        const int tp = phase_map(p >> 1, dy, pad) * kp + phase_map(p & 1, dx, pad);
                        s += pg[(long)sp * sstride + (long)p * plane + ((long)tp * Cin + ci) * Cout + co];
                    }
            } else {
#pragma unroll 4
                for (int sp = ln; sp < S; sp += 8) s += pg[(long)sp * sstride + i];
            }
        }
        sh[ln][ol] = s;
        __syncthreads();
        if (ln == 0 && i < total) {
            float t = 0.f;
#pragma unroll
            for (int r = 0; r < 8; ++r) t += sh[r][ol];
            float* gw = rp.gws ? rp.gw0 + (long)group * rp.gws : sel4(group, rp.gw0, rp.gw1, rp.gw2, rp.gw3);
            gw[((long)co * Cin + ci) * KK + tap] += scale * t;
        }
    } else {
        float* gb = sel4(group, rp.gb0, rp.gb1, rp.gb2, rp.gb3);
        if (!gb) return;
        const int c = (bx - wblocks) * 32 + ol;
        const float* bp = bias_part + (long)group * P * Cout;
        if (c < Cout)
            for (int i = ln; i < S * P; i += 8) s += bp[(long)(i / P) * bsstride + (long)(i % P) * Cout + c];
        sh[ln][ol] = s;
        __syncthreads();
        if (ln == 0 && c < Cout) {
            float t = 0.f;
#pragma unroll
            for (int r = 0; r < 8; ++r) t += sh[r][ol];
            gb[c] += scale * t;
        }
    }
}

// The plain (not phase-folded) weight blocks with 16-byte loads (round 5): a workgroup owns 128 consecutive partial elements - lane ol
// the float4 at bx * 128 + 4 ol, the S partial planes split over the 8 lane groups as above, four planes in flight per thread - and
// 128 threads add the eight lane sums in the same order and scatter them to the canonical layout.  Same bits as the 32-element form
// (per element: lane group ln adds planes ln, ln + 8, ..., then groups 0..7 in order); a quarter of the workgroups and load instructions.
__device__ __forceinline__ void wgrad_reduce_small_body_v4(float* shf, int bx, int group, const float* part, const RedPtrs& rp, int Cin,
                                                           int Cout, int KK, int S, float scale, long sstride, long gstride) {
    float4* sh4 = reinterpret_cast<float4*>(shf);     // [8][32]
    const int ol = threadIdx.x & 31, ln = threadIdx.x >> 5;
    const long total = (long)KK * Cin * Cout;
    const long i = bx * 128L + 4 * ol;
    const float* pg = part + (long)group * gstride + i;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < total) {
        int sp = ln;
        for (; sp + 24 < S; sp += 32) {
            const float4 p0 = ld4(pg + (long)sp * sstride), p1 = ld4(pg + (long)(sp + 8) * sstride);
            const float4 p2 = ld4(pg + (long)(sp + 16) * sstride), p3 = ld4(pg + (long)(sp + 24) * sstride);
            s.x += p0.x; s.y += p0.y; s.z += p0.z; s.w += p0.w;
            s.x += p1.x; s.y += p1.y; s.z += p1.z; s.w += p1.w;
            s.x += p2.x; s.y += p2.y; s.z += p2.z; s.w += p2.w;
            s.x += p3.x; s.y += p3.y; s.z += p3.z; s.w += p3.w;
        }
        for (; sp < S; sp += 8) {
            const float4 p0 = ld4(pg + (long)sp * sstride);
            s.x += p0.x; s.y += p0.y; s.z += p0.z; s.w += p0.w;
        }
    }
    sh4[ln * 32 + ol] = s;
    __syncthreads();
    if (ln < 4 && i < total) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) t += f4c(sh4[r * 32 + ol], ln);
        const long e = i + ln;                      // element (tap, ci, co) of the partial plane
        const int co = (int)(e % Cout);
        const long rr = e / Cout;
        const int ci = (int)(rr % Cin), tap = (int)(rr / Cin);
        float* gw = rp.gws ? rp.gw0 + (long)group * rp.gws : sel4(group, rp.gw0, rp.gw1, rp.gw2, rp.gw3);
        gw[((long)co * Cin + ci) * KK + tap] += scale * t;
    }
}

template <bool UPS>
__global__ __launch_bounds__(256) void wgrad_reduce_small_kernel(const float* part, const float* bias_part, RedPtrs rp, int Cin,
                                                                 int Cout, int k, int KK, int pad, int kp, int S, int P,
                                                                 float scale, long sstride, long gstride, long bsstride,
                                                                 int wblocks, int nblocks, int v4) {
    __shared__ __attribute__((aligned(16))) float sh[32][33];
    // blockIdx.x walks the group's wblocks + bblocks blocks with stride gridDim.x (= all of them for short launches)
    for (int bx = (int)blockIdx.x; bx < nblocks; bx += (int)gridDim.x) {
        if (!UPS && v4 && bx < wblocks)
            wgrad_reduce_small_body_v4(&sh[0][0], bx, (int)blockIdx.y, part, rp, Cin, Cout, KK, S, scale, sstride, gstride);
        else
            wgrad_reduce_small_body<UPS>(sh, bx, (int)blockIdx.y, part, bias_part, rp, Cin, Cout, k, KK, pad, kp, S, P, scale,
                                         sstride, gstride, bsstride, wblocks);
        __syncthreads();
    }
}

// KK == 1 (linear layers, the Winograd-domain products of winograd.hip): canonical gw[co][ci] += scale * sum_s part[s][ci][co] is a
// TRANSPOSE of the partial planes.  The generic kernels above give a workgroup 32 consecutive partial elements and scatter them to
// addresses Cin floats apart: 65 536 workgroups of 1 KB for the 16 Winograd products - 69 us alone, 295 us beside the data-gradient
// chain (profiles/r04_eager_breakdown.txt).  Here: 32 x 32 tiles through LDS, 128-byte rows both ways, splits added in order.
__device__ __forceinline__ void wgrad_reduce_t_tile(float (*t)[33], const float* __restrict__ pg, float* __restrict__ gw, int ci0, int co0,
                                                    int Cin, int Cout, int S, long sstride, float scale) {
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    if ((Cout & 3) == 0 && (sstride & 3) == 0 && ((uintptr_t)pg & 15) == 0) {
        // 16-byte loads (round 5): thread = (row r = tid / 8, four columns), the S planes in order, four in flight
        const int r = threadIdx.x >> 3, c4 = threadIdx.x & 7;
        const float* src = pg + (long)(ci0 + r) * Cout + co0 + 4 * c4;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        int sp = 0;
        for (; sp + 4 <= S; sp += 4) {
            const float4 p0 = ld4(src + (long)sp * sstride), p1 = ld4(src + (long)(sp + 1) * sstride);
            const float4 p2 = ld4(src + (long)(sp + 2) * sstride), p3 = ld4(src + (long)(sp + 3) * sstride);
            s.x += p0.x; s.y += p0.y; s.z += p0.z; s.w += p0.w;
            s.x += p1.x; s.y += p1.y; s.z += p1.z; s.w += p1.w;
            s.x += p2.x; s.y += p2.y; s.z += p2.z; s.w += p2.w;
            s.x += p3.x; s.y += p3.y; s.z += p3.z; s.w += p3.w;
        }
        for (; sp < S; ++sp) {
            const float4 p0 = ld4(src + (long)sp * sstride);
            s.x += p0.x; s.y += p0.y; s.z += p0.z; s.w += p0.w;
        }
        t[r][4 * c4] = s.x; t[r][4 * c4 + 1] = s.y; t[r][4 * c4 + 2] = s.z; t[r][4 * c4 + 3] = s.w;
    } else
#pragma unroll
    for (int r = ty; r < 32; r += 8) {
        const long off = (long)(ci0 + r) * Cout + co0 + tx;
        float s = 0.f;
        for (int sp = 0; sp < S; ++sp) s += pg[(long)sp * sstride + off];      // fixed order
        t[r][tx] = s;
    }
    __syncthreads();
#pragma unroll
    for (int r = ty; r < 32; r += 8) gw[(long)(co0 + r) * Cin + ci0 + tx] += scale * t[tx][r];
}
__global__ __launch_bounds__(256) void wgrad_reduce_t_kernel(const float* __restrict__ part, float* __restrict__ gw0, long gws, int Cin,
                                                             int Cout, int S, long sstride, long gstride, float scale) {
    __shared__ float t[32][33];
    const int group = blockIdx.z;
    wgrad_reduce_t_tile(t, part + (long)group * gstride, gw0 + (long)group * gws, blockIdx.x * 32, blockIdx.y * 32, Cin, Cout, S, sstride, scale);
}

// The same reduction for SEVERAL layers in one launch (cg_conv2d_wgrad_flush): the deferred form of cg_conv2d_wgrad* runs only
// its GEMM and queues one RedJob; a dozen ~10 us reductions, each a short dependent chain, then overlap instead of queueing
// up behind one another.  Workgroup b belongs to the job whose [block0, block0 + ngroups * (wblocks + bblocks)) holds b.
struct RedJob {
    const float* part; const float* bias_part;
    RedPtrs rp;
    long sstride, gstride, bsstride;
    int Cin, Cout, k, KK, pad, kp, S, P, ups, wblocks, bblocks, block0;
    float scale;
    int tr;      // KK == 1, planes multiples of 32: the weight blocks are 32 x 32 transpose tiles (wgrad_reduce_t_tile), not 32-element runs
    int v4;      // plain layout, Cout % 4 == 0, 16-byte aligned planes: the weight blocks are 128-element runs (wgrad_reduce_small_body_v4)
};
constexpr int kRedJobs = 20;
struct RedJobTable { RedJob j[kRedJobs]; int n; };

// A workgroup walks blocks b, b + gridDim.x, ... of the `total` 32-element blocks (round 4: the launch used to have one workgroup per
// block - 31 000 of them for D's flush, each waiting for a slot of its own beside the GEMMs of the other queues; see ew_grid)
__global__ __launch_bounds__(256) void wgrad_reduce_batch_kernel(RedJobTable t, int total) {
    __shared__ __attribute__((aligned(16))) float sh[32][33];
    for (int b = (int)blockIdx.x; b < total; b += (int)gridDim.x) {
        int ji = 0;
        for (int q = 1; q < t.n; ++q)
            if (b >= t.j[q].block0) ji = q;
        const RedJob& J = t.j[ji];
        const int per = J.wblocks + J.bblocks;
        const int group = (b - J.block0) / per, bx = (b - J.block0) - group * per;
        if (J.tr && bx < J.wblocks) {
            const int nci = J.Cin / 32;
            float* gw = J.rp.gws ? J.rp.gw0 + (long)group * J.rp.gws : sel4(group, J.rp.gw0, J.rp.gw1, J.rp.gw2, J.rp.gw3);
            wgrad_reduce_t_tile(sh, J.part + (long)group * J.gstride, gw, (bx % nci) * 32, (bx / nci) * 32, J.Cin, J.Cout, J.S, J.sstride, J.scale);
        } else if (J.v4 && bx < J.wblocks)
            wgrad_reduce_small_body_v4(&sh[0][0], bx, group, J.part, J.rp, J.Cin, J.Cout, J.KK, J.S, J.scale, J.sstride, J.gstride);
        else if (J.ups)
            wgrad_reduce_small_body<true>(sh, bx, group, J.part, J.bias_part, J.rp, J.Cin, J.Cout, J.k, J.KK, J.pad, J.kp, J.S, J.P, J.scale,
                                          J.sstride, J.gstride, J.bsstride, J.wblocks);
        else
            wgrad_reduce_small_body<false>(sh, bx, group, J.part, J.bias_part, J.rp, J.Cin, J.Cout, J.k, J.KK, J.pad, J.kp, J.S, J.P, J.scale,
                                           J.sstride, J.gstride, J.bsstride, J.wblocks);
        __syncthreads();   // the block's sums have been read out of `sh` before the next block overwrites them
    }
}

// gb[c] += scale * sum_{s,p} bias_part[s*P+p][c]; 32 channels x 8 partial-sum lanes per workgroup
// (S splits x P phases of one group; consecutive splits are `sstride` floats apart)
__global__ __launch_bounds__(256) void bias_part_reduce_kernel(const float* bp, float* gb, int S, int P, long sstride,
                                                               int Cout, float scale) {
    __shared__ float sh[8][33];
    const int cl = threadIdx.x & 31, ln = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    float s = 0.f;
    if (c < Cout)
        for (int i = ln; i < S * P; i += 8) s += bp[(long)(i / P) * sstride + (long)(i % P) * Cout + c];
    sh[ln][cl] = s;
    __syncthreads();
    if (ln == 0 && c < Cout) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) t += sh[r][cl];
        gb[c] += scale * t;
    }
}

// canonical [Cout][Cin][KK] -> wf[tap*Cin+ci][Cout], wb[(KK-1-tap)*Cout+co][Cin]
// (the parameters change every step, so this runs every step).  One workgroup owns a 32 ci x 32 co block: thread
// (ci_l, co_l + 8j) reads its canonical tap run, writes wb directly (ci-fastest, coalesced) and transposes the
// 32x32 plane of each tap through LDS for wf (co-fastest).
// wb_map != 0: wb receives the row-permuted canonical matrix wbT[co][tap*Cin + ci] instead of the flipped-tap data-gradient
// operand - the backward operand of a LINEAR layer that consumes an NHWC map directly (its canonical [out][C*H*W] weight is
// the canonical weight of a k = H x W convolution on that map; see cg_pack_conv_weight_map).
__global__ __launch_bounds__(256) void pack_weight_kernel(const float* w, float* wf, float* wb, int Cout, int Cin, int KK,
                                                          int wb_map) {
    __shared__ float sh[32][33];
    const int lo = threadIdx.x & 31, hi = threadIdx.x >> 5;
    const int ci0 = blockIdx.x * 32, co0 = blockIdx.y * 32;
    {
        const int tap = blockIdx.z;  // one tap plane per workgroup: KK independent workgroups instead of KK serial rounds
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ci = ci0 + lo, co_l = hi + 8 * j, co = co0 + co_l;
            float v = 0.f;
            if (ci < Cin && co < Cout) {
                v = w[((long)co * Cin + ci) * KK + tap];
                if (wb) wb[wb_map ? ((long)co * KK + tap) * Cin + ci : ((long)(KK - 1 - tap) * Cout + co) * Cin + ci] = v;
            }
            sh[lo][co_l] = v;
        }
        if (wf) {
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int co = co0 + lo, ci_l = hi + 8 * j, ci = ci0 + ci_l;
                if (ci < Cin && co < Cout) wf[((long)tap * Cin + ci) * Cout + co] = sh[ci_l][lo];
            }
        }
    }
}

// The same for up to kPackBatch layers in ONE launch (all of D32_st3's 28 convolution / linear layers re-pack after every
// Adam step: 28 launches of a few microseconds each otherwise).  A workgroup finds its layer by its block index.
constexpr int kPackBatch = 48;
struct PackBatch {
    const float* w[kPackBatch]; float* wf[kPackBatch]; float* wb[kPackBatch];
    int Cout[kPackBatch], Cin[kPackBatch], KK[kPackBatch], first[kPackBatch + 1];
    unsigned long long map_mask;   // bit e: layer e wants the wbT layout (see pack_weight_kernel)
    int n;
};
__global__ __launch_bounds__(256) void pack_weight_batch_kernel(PackBatch b) {
    __shared__ float sh[32][33];
    int e = 0;
    while (e + 1 < b.n && (int)blockIdx.x >= b.first[e + 1]) ++e;
    const float* w = b.w[e]; float* wf = b.wf[e]; float* wb = b.wb[e];
    const int Cout = b.Cout[e], Cin = b.Cin[e], KK = b.KK[e];
    const bool wb_map = (b.map_mask >> e) & 1ull;
    const int nbx = (Cin + 31) / 32, nby = (Cout + 31) / 32;
    int r = (int)blockIdx.x - b.first[e];
    const int bx = r % nbx; r /= nbx;
    const int by = r % nby;
    const int tap = r / nby;
    const int lo = threadIdx.x & 31, hi = threadIdx.x >> 5;
    const int ci0 = bx * 32, co0 = by * 32;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int ci = ci0 + lo, co_l = hi + 8 * j, co = co0 + co_l;
        float v = 0.f;
        if (ci < Cin && co < Cout) {
            v = w[((long)co * Cin + ci) * KK + tap];
            if (wb) wb[wb_map ? ((long)co * KK + tap) * Cin + ci : ((long)(KK - 1 - tap) * Cout + co) * Cin + ci] = v;
        }
        sh[lo][co_l] = v;
    }
    if (wf) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int co = co0 + lo, ci_l = hi + 8 * j, ci = ci0 + ci_l;
            if (ci < Cin && co < Cout) wf[((long)tap * Cin + ci) * Cout + co] = sh[ci_l][lo];
        }
    }
}

// phase-summed weights, k in {3, 5} (the generator's layers): same 32x32 blocking as pack_weight_kernel, the tap
// sums fully unrolled so every accumulator index is a compile-time constant.
template <int K>
__global__ __launch_bounds__(256) void pack_weight_ups2_fast_kernel(const float* w, float* wf, float* wb, int Cout, int Cin) {
    constexpr int KK = K * K, PAD = (K - 1) / 2, KPD = K == 3 ? 2 : 3, KP = KPD * KPD;
    __shared__ float sh[KP][32][33];
    const int lo = threadIdx.x & 31, hi = threadIdx.x >> 5;
    const int ci0 = blockIdx.x * 32, co0 = blockIdx.y * 32;
    float v[4][KK];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int ci = ci0 + lo, co = co0 + hi + 8 * j;
        const bool ok = ci < Cin && co < Cout;
        const float* src = w + ((long)co * Cin + ci) * KK;
#pragma unroll
        for (int t = 0; t < KK; ++t) v[j][t] = ok ? src[t] : 0.f;
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float acc[KP];
#pragma unroll
            for (int t = 0; t < KP; ++t) acc[t] = 0.f;
#pragma unroll
            for (int dy = 0; dy < K; ++dy)
#pragma unroll
                for (int dx = 0; dx < K; ++dx)
                    acc[phase_map(p >> 1, dy, PAD) * KPD + phase_map(p & 1, dx, PAD)] += v[j][dy * K + dx];
            const int ci = ci0 + lo, co_l = hi + 8 * j, co = co0 + co_l;
#pragma unroll
            for (int t = 0; t < KP; ++t) {
                if (wb && ci < Cin && co < Cout) wb[(((long)p * KP + t) * Cout + co) * Cin + ci] = acc[t];
                sh[t][lo][co_l] = acc[t];
            }
        }
        if (wf) {
            __syncthreads();
#pragma unroll
            for (int t = 0; t < KP; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int co = co0 + lo, ci_l = hi + 8 * j, ci = ci0 + ci_l;
                    if (ci < Cin && co < Cout) wf[(((long)p * KP + t) * Cin + ci) * Cout + co] = sh[t][ci_l][lo];
                }
            __syncthreads();
        }
    }
}

// phase-summed weights for upsample(2) -> conv k x k:
//   wf[p][(t'*Cin+ci)][co] = sum_{(dy,dx) -> t' under phase p} w[co][ci][dy][dx]
//   wb[((p*kp*kp + t')*Cout + co)][ci] = the same value (data-gradient operand)
// Same LDS staging as pack_weight_kernel.
__global__ __launch_bounds__(256) void pack_weight_ups2_kernel(const float* w, float* wf, float* wb, int Cout, int Cin, int k,
                                                               int pad, int kp, int CI_T) {
    extern __shared__ float sh[];  // [32][CI_T*KK + 1]
    const int KK = k * k, KP = kp * kp;
    const int run = CI_T * KK, ld = run + 1;
    const int ci0 = blockIdx.x * CI_T, co0 = blockIdx.y * 32;
    const int cis = min(CI_T, Cin - ci0);
    for (int idx = threadIdx.x; idx < 32 * run; idx += 256) {
        const int j = idx % run, co_l = idx / run;
        if (co0 + co_l < Cout && j < cis * KK) sh[co_l * ld + j] = w[((long)(co0 + co_l) * Cin + ci0) * KK + j];
    }
    __syncthreads();
    const int n = 4 * KP * CI_T * 32;
    for (int pass = 0; pass < 2; ++pass) {
        if (pass == 0 ? wf == nullptr : wb == nullptr) continue;
        for (int idx = threadIdx.x; idx < n; idx += 256) {
            int co_l, ci_l, q;
            if (pass == 0) { co_l = idx & 31; const int r = idx >> 5; ci_l = r % CI_T; q = r / CI_T; }
            else { ci_l = idx % CI_T; const int r = idx / CI_T; co_l = r & 31; q = r >> 5; }
            if (co0 + co_l >= Cout || ci_l >= cis) continue;
            const int tp = q % KP, p = q / KP;
            const int ry = tp / kp, rx = tp - ry * kp;
            const float* src = sh + co_l * ld + ci_l * KK;
            float s = 0.f;
            for (int dy = 0; dy < k; ++dy) {
                if (phase_map(p >> 1, dy, pad) != ry) continue;
                for (int dx = 0; dx < k; ++dx)
                    if (phase_map(p & 1, dx, pad) == rx) s += src[dy * k + dx];
            }
            if (pass == 0) wf[(((long)p * KP + tp) * Cin + ci0 + ci_l) * Cout + co0 + co_l] = s;
            else wb[(((long)p * KP + tp) * Cout + co0 + co_l) * Cin + ci0 + ci_l] = s;
        }
    }
}

// ---------------------------------------------------------------------------
// Skinny 3x3 convolution, Cout <= 4 (G's last layer 128 -> 3, and D's first layer seen from its data gradient
// 64 -> 3): 2.3 kFLOP per 512-byte pixel, i.e. HBM-bound; an MFMA tile would spend >= 90 % of its columns on
// padding.  LP lanes share a pixel, each lane owns one channel quad (Cin = 4*LP) whose 9 x 4 x CO weights
// (weight-gradient: accumulators) live in registers; pixel loads are coalesced float4 rows.
// ---------------------------------------------------------------------------
template <int CO, int LP>
__global__ __launch_bounds__(256) void skinny_conv3x3_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ bias, float* __restrict__ y, int N,
                                                             int H, int W) {
    constexpr int Cin = 4 * LP;
    const int sub = threadIdx.x & (LP - 1);
    float wr[9][4][CO];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int co = 0; co < CO; ++co) wr[t][j][co] = w[((long)t * Cin + sub * 4 + j) * CO + co];
    const long npix = (long)N * H * W;
    const long grp0 = (blockIdx.x * 256L + threadIdx.x) / LP;
    const long ngrp = gridDim.x * 256L / LP;
    const long iters = (npix + ngrp - 1) / ngrp;   // uniform trip count: the shuffles below need every lane
    for (long it = 0; it < iters; ++it) {
        const long pix = grp0 + it * ngrp;
        const bool live = pix < npix;
        const unsigned pc = live ? (unsigned)pix : 0u;   // npix < 2^31 (checked by the host): 32-bit divisions
        const unsigned row = pc / (unsigned)W;
        const int ox = (int)(pc - row * (unsigned)W), oy = (int)(row % (unsigned)H);
        const long n = row / (unsigned)H;
        float acc[CO];
#pragma unroll
        for (int co = 0; co < CO; ++co) acc[co] = 0.f;
        float4 v[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int iy = oy + t / 3 - 1, ix = ox + t % 3 - 1;
            const bool ok = live && iy >= 0 && iy < H && ix >= 0 && ix < W;
            const int cy = min(max(iy, 0), H - 1), cx = min(max(ix, 0), W - 1);
            const float4 q = ld4(x + ((n * H + cy) * (long)W + cx) * Cin + sub * 4);
            v[t] = ok ? q : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int co = 0; co < CO; ++co)
                acc[co] += v[t].x * wr[t][0][co] + v[t].y * wr[t][1][co] + v[t].z * wr[t][2][co] + v[t].w * wr[t][3][co];
#pragma unroll
        for (int off = LP >> 1; off > 0; off >>= 1)
#pragma unroll
            for (int co = 0; co < CO; ++co) acc[co] += __shfl_down(acc[co], off, LP);
        if (live && sub == 0) {
#pragma unroll
            for (int co = 0; co < CO; ++co) y[pix * CO + co] = acc[co] + (bias ? bias[co] : 0.f);
        }
    }
}

// weight/bias gradient partials of the same layer: part[block][(t*Cin+ci)][co], bias_part[block][co]
template <int CO, int LP>
__global__ __launch_bounds__(256) void skinny_wgrad3x3_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                              float* __restrict__ part, float* __restrict__ bias_part, int N,
                                                              int H, int W, long pix_per_block) {
    constexpr int Cin = 4 * LP, NGL = 256 / LP, PER = 36 * CO;
    extern __shared__ float sk_sh[];   // [4 waves][LP][PER + 1]
    const int sub = threadIdx.x & (LP - 1), gl = threadIdx.x / LP;
    const long npix = (long)N * H * W;
    const long p0 = blockIdx.x * pix_per_block, p1 = min(npix, p0 + pix_per_block);
    float acc[9][4][CO];
    float bacc[CO];
#pragma unroll
    for (int co = 0; co < CO; ++co) {
        bacc[co] = 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[t][j][co] = 0.f;
    }
    for (long pix = p0 + gl; pix < p1; pix += NGL) {
        const unsigned row = (unsigned)pix / (unsigned)W;
        const int ox = (int)((unsigned)pix - row * (unsigned)W), oy = (int)(row % (unsigned)H);
        const long n = row / (unsigned)H;
        float d[CO];
#pragma unroll
        for (int co = 0; co < CO; ++co) { d[co] = dy[pix * CO + co]; bacc[co] += d[co]; }
        float4 v[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int iy = oy + t / 3 - 1, ix = ox + t % 3 - 1;
            const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
            const int cy = min(max(iy, 0), H - 1), cx = min(max(ix, 0), W - 1);
            const float4 q = ld4(x + ((n * H + cy) * (long)W + cx) * Cin + sub * 4);
            v[t] = ok ? q : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int co = 0; co < CO; ++co) {
                acc[t][0][co] += v[t].x * d[co]; acc[t][1][co] += v[t].y * d[co];
                acc[t][2][co] += v[t].z * d[co]; acc[t][3][co] += v[t].w * d[co];
            }
    }
    // fold the pixel groups of a wave (lanes with equal sub), then the four waves through LDS
#pragma unroll
    for (int off = LP; off < 64; off <<= 1) {
#pragma unroll
        for (int co = 0; co < CO; ++co) {
            bacc[co] += __shfl_xor(bacc[co], off, 64);
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[t][j][co] += __shfl_xor(acc[t][j][co], off, 64);
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane < LP) {
        float* dst = sk_sh + ((long)wave * LP + sub) * (PER + CO + 1);
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int co = 0; co < CO; ++co) dst[(t * 4 + j) * CO + co] = acc[t][j][co];
#pragma unroll
        for (int co = 0; co < CO; ++co) dst[PER + co] = bacc[co];
    }
    __syncthreads();
    float* pout = part + (long)blockIdx.x * 9 * Cin * CO;
    for (int idx = threadIdx.x; idx < LP * PER; idx += 256) {
        const int sb = idx / PER, r = idx % PER;           // r = (t*4+j)*CO+co
        const int co = r % CO, tj = r / CO, t = tj >> 2, j = tj & 3;
        float s = 0.f;
#pragma unroll
        for (int wv = 0; wv < 4; ++wv) s += sk_sh[((long)wv * LP + sb) * (PER + CO + 1) + r];
        pout[((long)t * Cin + sb * 4 + j) * CO + co] = s;
    }
    if (bias_part && threadIdx.x < CO) {
        float s = 0.f;
        // every lane group saw the same dy values of its own pixels; sub == 0 of each wave holds the wave total
#pragma unroll
        for (int wv = 0; wv < 4; ++wv) s += sk_sh[((long)wv * LP + 0) * (PER + CO + 1) + PER + threadIdx.x];
        bias_part[(long)blockIdx.x * CO + threadIdx.x] = s;
    }
}

// ---- host-side dispatch -----------------------------------------------------
struct TileCfg { int bm, bn; };
// swept in rounds 2-5 (profiles/r0N_sweeps.txt: flat or worse around these), constants since round 6
constexpr int kSplitTarget = 1, kSplitMinK = 8;      // forward / data gradient: split K until one workgroup per CU, >= 8 K iterations per split
constexpr int kTnSplitMax = 256, kTnTarget = 3;      // weight gradients: pixel splits up to 256, three workgroups per CU

// pick the block tile: BN covers Cout with least padding, BM chosen so the grid
// fills the 256 CUs at >= ~2 workgroups each when the problem allows it.
static bool tile_forced(cg::Opt o, TileCfg& tc) {
    // CG_NN_TILE / CG_TN_TILE = bm*1000 + bn forces one of the compiled block tiles (parity tests walk all of them)
    const long v = cg::opt(o);
    if (v <= 0) return false;
    const int bm = (int)(v / 1000), bn = (int)(v % 1000);
    if (!((bm == 128 || bm == 64) && (bn == 128 || bn == 64)) && !(bm == 128 && bn == 32)) return false;
    tc = {bm, bn};
    return true;
}

static TileCfg pick_tile(long M, int Cout, int nphase) {
    TileCfg f;
    if (tile_forced(cg::OPT_NN_TILE, f)) return f;
    int bn = Cout > 64 ? 128 : (Cout > 32 ? 64 : 32);
    int bm = 128;
    if (bn == 128 || bn == 64) {
        const long blocks128 = ((M + 127) / 128) * ((Cout + bn - 1) / bn) * nphase;
        if (blocks128 < 2 * cg::kNumCU) bm = 64;   // swept 256 .. 2048 workgroups (r02): flat within 0.3 % from 384 up
    }
    return {bm, bn};
}

static int pick_splits(long tiles, long kiters) {
    // Small grids are latency-bound (one global-load round trip per K tile, nothing to overlap it with), so
    // split K until there are kSplitTarget workgroups per CU, keeping >= kSplitMinK K-iterations per split.
    const int target = kSplitTarget, mink = kSplitMinK;   // swept in rounds 2-5 (profiles/r0N_sweeps.txt): constants since round 6
    const int forced = (int)cg::opt(cg::OPT_NN_SPLITS);
    if (forced > 0) return (int)std::min<long>(forced, std::max<long>(1, kiters / 2));
    const long want = target < 16 ? (long)target * cg::kNumCU : (long)target;
    int s = 1;
    while (tiles * s < want && kiters / (s * 2) >= mink && s < 64) s *= 2;
    return s;
}

// position-major rows + skipped padding taps (Geom::pmn): the 64-row LDS-direct tiles at K step 32 only
template <int BM, int BN, int WM, int WN>
static void launch_nn_pm(const NNArgs& a, dim3 grid, hipStream_t st) {
    if constexpr (BM == 64 && (BN == 128 || BN == 64)) hipLaunchKernelGGL((igemm_nng_kernel<BM, BN, WM, WN, 32, true>), grid, dim3(256), 0, st, a);
}

template <int BM, int BN, int WM, int WN>
static void launch_nn(const NNArgs& a, dim3 grid, hipStream_t st, bool fast, bool vecb, bool bk32) {
    if (a.g.pmn) { launch_nn_pm<BM, BN, WM, WN>(a, grid, st); return; }
    if (fast && vecb) {
        // LDS-direct loads (igemm_nng_kernel).  CG_NN_GLDS = 1: the 64-row tiles at K step 32 (a pixel's 128 consecutive
        // bytes per 8 lanes; measured inside the replayed step 2.43 -> 1.91 ms for the 64x128 tile), 2: K step 32 for every
        // tile whose Cin allows it, 3: K step 16 as well (64-byte runs per pixel: slower than the register path in the step)
        const long glds = cg::opt(cg::OPT_NN_GLDS);
        if (glds && bk32 && (BM == 64 || glds >= 2)) {
            hipLaunchKernelGGL((igemm_nng_kernel<BM, BN, WM, WN, 32>), grid, dim3(256), 0, st, a);
            return;
        }
        if (glds >= 3) {
            hipLaunchKernelGGL((igemm_nng_kernel<BM, BN, WM, WN, 16>), grid, dim3(256), 0, st, a);
            return;
        }
        // 64-row tiles do only 8-16 MFMAs per wave per K step of 16: give them 32 so the per-tile work
        // (address VALU, LDS stores, barrier) is amortised like in the 128x128 tile
        if (BM == 64 && bk32) hipLaunchKernelGGL((igemm_nn_kernel<BM, BN, WM, WN, true, true, (BM == 64 ? 32 : 16)>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((igemm_nn_kernel<BM, BN, WM, WN, true, true, 16>), grid, dim3(256), 0, st, a);
    }
    else if (fast) hipLaunchKernelGGL((igemm_nn_kernel<BM, BN, WM, WN, true, false, 16>), grid, dim3(256), 0, st, a);
    else if (vecb) hipLaunchKernelGGL((igemm_nn_kernel<BM, BN, WM, WN, false, true, 16>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((igemm_nn_kernel<BM, BN, WM, WN, false, false, 16>), grid, dim3(256), 0, st, a);
}

template <int BM, int BN, int WM, int WN>
static void launch_tn(const TNArgs& a, dim3 grid, hipStream_t st, bool veca, bool vecb) {
    if (veca && vecb) hipLaunchKernelGGL((igemm_tn_kernel<BM, BN, WM, WN, true, true>), grid, dim3(256), 0, st, a);
    else if (veca) hipLaunchKernelGGL((igemm_tn_kernel<BM, BN, WM, WN, true, false>), grid, dim3(256), 0, st, a);
    else if (vecb) hipLaunchKernelGGL((igemm_tn_kernel<BM, BN, WM, WN, false, true>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((igemm_tn_kernel<BM, BN, WM, WN, false, false>), grid, dim3(256), 0, st, a);
}

// LDS-direct weight-gradient kernel (igemm_tng_kernel, K step 16): lean addressing only
static bool tng_ok(const Geom& g, int pchunk, bool veca, bool vecb) {
    if (!cg::opt(cg::OPT_TN_GLDS) || !veca || !vecb) return false;
    const bool lin_src = g.td.ss == 1 && g.Hs == g.Hg && g.Ws == g.Wg;
    const bool lin_out = g.so == 1 && g.nphase == 1;
    if (!lin_src || g.M % 16 || pchunk % 16) return false;
    const bool flat = g.ntaps == 1 && g.td.r0y0 == 0 && g.td.r0x0 == 0 && lin_out;
    return flat || (g.lgW >= 0 && g.lgHW >= 0 && g.Hg * g.Wg >= 16 && (lin_out || g.so == 2));
}
static bool tng_flat(const Geom& g) { return g.ntaps == 1 && g.td.r0y0 == 0 && g.td.r0x0 == 0 && g.so == 1 && g.nphase == 1; }

// Position-major K tiles (igemm_tng_kernel mode 2): the share of the layer's MACs that multiply zero padding in per cent, -1 = not this
// geometry (a plain stride-1 convolution with a tap at offset (0, 0) that is valid at every position, whole K tiles of images).
static int tn_pm_share(const Geom& g) {
    if (g.nphase != 1 || g.so != 1 || g.td.ss != 1 || g.td.ngroups != 1 || g.td.sgn != 1 || g.ntaps < 2) return -1;
    if (g.Hout != g.Hg || g.Wout != g.Wg || g.Hg > g.Hv || g.Wg > g.Wv || g.Cin % 4 || g.Cout % 4) return -1;
    const int hw = g.Hg * g.Wg, N = g.M / hw, kh = g.td.kk / g.td.kw;
    if ((long)N * hw != g.M || N % BK) return -1;
    if (g.td.r0y0 > 0 || g.td.r0x0 > 0 || -g.td.r0y0 >= kh || -g.td.r0x0 >= g.td.kw) return -1;
    if ((long)N * g.Hs * g.Ws * g.Cin * 4L >= 0x7fffffffL || (long)g.M * g.Cout * 4L >= 0x7fffffffL) return -1;
    long valid = 0;
    for (int t = 0; t < g.ntaps; ++t) {
        int ty, tx, off;
        tap_decode(g, t, 0, 0, ty, tx, off);
        const int rh = std::min(g.Hg, g.Hv - ty) - std::max(0, -ty), rw = std::min(g.Wg, g.Wv - tx) - std::max(0, -tx);
        if (rh <= 0 || rw <= 0) return -1;
        valid += (long)rh * rw;
    }
    return (int)(100 - 100 * valid / ((long)g.ntaps * hw));
}
// ... used when the padding is at least HALF of CG_PAD_SKIP's share (default 20 -> 10 %: D32_st3's 7x7 at 8x8 38 %, 3x3 at 8x8 16 %, 5x5 at
// 16x16 14 %), or when the image-major tiles cannot run at all (fewer than 16 pixels per image: the View -> Linear head's 1 x 1 grid)
static bool tn_pm_use(const Geom& g, bool image_major_ok) {
    const long thr = cg::opt(cg::OPT_PAD_SKIP);
    if (thr <= 0 || !cg::opt(cg::OPT_TN_GLDS)) return false;
    const int sh = tn_pm_share(g);
    return sh >= 0 && (2 * sh >= thr || !image_major_ok);
}

template <int BM, int BN, int WM, int WN>
static void launch_tng(const TNArgs& a, dim3 grid, hipStream_t st) {
    // (round 5: capping the workgroups per CU with unused dynamic LDS - 3, 2, 1 instead of 4 - to leave slots for the other queues'
    // memory-bound kernels measured 6.05 / 5.95 / 6.10 ms per step against 5.93: profiles/r05_sweeps.txt)
    hipLaunchKernelGGL((igemm_tng_kernel<BM, BN, WM, WN>), grid, dim3(256), 0, st, a, a.pm_n ? 2 : (tng_flat(a.g) ? 1 : 0));
}

static int ilog2_exact(int v) {
    if (v <= 0 || (v & (v - 1))) return -1;
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}

static int finish_geom(Geom& g, long N) {
    const long M = N * g.Hg * g.Wg;
    if (g.Hg <= 0 || g.Wg <= 0) return cg::fail("conv2d: empty output %dx%d", g.Hg, g.Wg);
    if (M > 0x7fffffffL || N * (long)g.Hs * g.Ws * g.Cin > 0x1fffffffL || N * (long)g.Hout * g.Wout * g.Cout > 0x1fffffffL)
        return cg::fail("conv2d: tensor too large for 32-bit element offsets");
    g.M = (int)M;
    const int lw = ilog2_exact(g.Wg), lh = ilog2_exact(g.Hg);
    g.lgW = (lw >= 0 && lh >= 0) ? lw : -1;
    g.lgHW = (lw >= 0 && lh >= 0) ? lw + lh : -1;
    g.Ktot = g.ntaps * g.Cin;
    if (g.ntaps > 64) return cg::fail("conv2d: more than 64 taps");
    int mo[4] = {0, 0, 0, 0};
    for (int ph = 0; ph < g.nphase; ++ph)
        for (int t = 0; t < g.ntaps; ++t) {
            int ty, tx, off;
            tap_decode(g, t, ph >> 1, ph & 1, ty, tx, off);
            mo[ph] = std::min(mo[ph], off);
        }
    g.minoff0 = mo[0]; g.minoff1 = mo[1]; g.minoff2 = mo[2]; g.minoff3 = mo[3];
    return 0;
}

static int check_dims(int N, int Hp, int Wp, int Cin, int Cout, int kH, int kW, int padH, int padW, int ups) {
    if (N <= 0 || Hp <= 0 || Wp <= 0 || Cin <= 0 || Cout <= 0 || kH <= 0 || kW <= 0 || padH < 0 || padW < 0 ||
        (ups != 0 && ups != 1))
        return cg::fail("conv2d: bad geometry N=%d H=%d W=%d Cin=%d Cout=%d k=%dx%d pad=%d,%d ups=%d", N, Hp, Wp, Cin,
                        Cout, kH, kW, padH, padW, ups);
    if (ups && (kH != kW || (kH & 1) == 0 || padH != (kH - 1) / 2 || padW != padH))
        return cg::fail("conv2d: folded upsampling needs an odd square kernel with pad=(k-1)/2 (got %dx%d pad %d,%d)", kH,
                        kW, padH, padW);
    return 0;
}

// plain stride-1 convolution: grid = output pixels, source = the input tensor
static int geom_plain(Geom& g, int N, int Hp, int Wp, int Cin, int Cout, int kH, int kW, int padH, int padW) {
    memset(&g, 0, sizeof(g));
    g.Hg = Hp + 2 * padH - kH + 1; g.Wg = Wp + 2 * padW - kW + 1;
    g.Hv = Hp; g.Wv = Wp; g.Hs = Hp; g.Ws = Wp;
    g.Cin = Cin; g.Cout = Cout;
    g.so = 1; g.Hout = g.Hg; g.Wout = g.Wg;
    g.nphase = 1; g.ntaps = kH * kW;
    g.td.kw = kW; g.td.kk = kH * kW; g.td.ngroups = 1; g.td.sgn = 1; g.td.ss = 1;
    g.td.r0y0 = g.td.r0y1 = -padH; g.td.r0x0 = g.td.r0x1 = -padW;
    return finish_geom(g, N);
}

static int phase_kp(int k, int pad) { return phase_map(0, k - 1, pad) + 1; }

// upsample(2) -> conv, forward / weight-gradient view: 4 phases over the low-res grid
static int geom_phase_fwd(Geom& g, int N, int Hp, int Wp, int Cin, int Cout, int k, int pad) {
    memset(&g, 0, sizeof(g));
    const int kp = phase_kp(k, pad);
    g.Hg = Hp; g.Wg = Wp; g.Hv = Hp; g.Wv = Wp; g.Hs = Hp; g.Ws = Wp;
    g.Cin = Cin; g.Cout = Cout;
    g.so = 2; g.Hout = 2 * Hp; g.Wout = 2 * Wp;
    g.nphase = 4; g.ntaps = kp * kp;
    g.td.kw = kp; g.td.kk = kp * kp; g.td.ngroups = 1; g.td.sgn = 1; g.td.ss = 1;
    g.td.r0y0 = g.td.r0x0 = (0 - pad) >> 1;
    g.td.r0y1 = g.td.r0x1 = (1 - pad) >> 1;
    return finish_geom(g, N);
}

// data gradient w.r.t. the low-res input: one GEMM whose taps run over (phase, ry', rx') of dy [N,2Hp,2Wp,CoutF]
static int geom_phase_dgrad(Geom& g, int N, int Hp, int Wp, int CinF, int CoutF, int k, int pad) {
    memset(&g, 0, sizeof(g));
    const int kp = phase_kp(k, pad);
    g.Hg = Hp; g.Wg = Wp; g.Hv = Hp; g.Wv = Wp; g.Hs = 2 * Hp; g.Ws = 2 * Wp;
    g.Cin = CoutF; g.Cout = CinF;
    g.so = 1; g.Hout = Hp; g.Wout = Wp;
    g.nphase = 1; g.ntaps = 4 * kp * kp;
    g.td.kw = kp; g.td.kk = kp * kp; g.td.ngroups = 4; g.td.sgn = -1; g.td.ss = 2;
    g.td.r0y0 = g.td.r0x0 = (0 - pad) >> 1;
    g.td.r0y1 = g.td.r0x1 = (1 - pad) >> 1;
    return finish_geom(g, N);
}

struct NNPlan { TileCfg tc; int splits; int kchunk; bool pm; };
static long pad_skip_valid(const Geom& g, const TileCfg& tc);
static NNPlan plan_nn(const Geom& g, int ngroups, bool allow_pm = true) {
    NNPlan p;
    const int zdim = g.nphase * ngroups;
    p.tc = pick_tile(g.M, g.Cout, zdim);
    p.pm = false;
    const long valid = (allow_pm && ngroups == 1 && g.Cin % 32 == 0) ? pad_skip_valid(g, p.tc) : 0;
    if (valid > 0) {
        // position-major tiles (Geom::pmn): K units of equal length cut out of every tile's VALID taps; gridDim.y = the units of an interior
        // tile.  Round 5 aimed at one unit per CU (61 K tiles for D's 7x7 layer at batch 128: 312 workgroups, 1.2 per CU, the short last
        // units of every tile in the way); round 6 measured the unit length itself: 28 K tiles (592 workgroups) run the layer in 96 us
        // alone against 122 (61), 109 (49 / 33 / 25 / 20), 123 (40), and the step at 5.635 / 11.27 / 8.33 ms (configs 2 / 5 / 3) against
        // 5.678 / 11.36 / 8.42; 20 and 33 sit between (profiles/r06_sweeps.txt)
        constexpr long kPmUnit = 28;
        long unit = kPmUnit;
        const int forced = (int)cg::opt(cg::OPT_NN_SPLITS);
        if (forced > 0) unit = cg::cdiv(g.Ktot / 32, forced);
        p.kchunk = (int)std::min<long>(unit * 32, g.Ktot);
        p.splits = cg::cdiv(g.Ktot, p.kchunk);
        p.pm = true;
        return p;
    }
    const long tiles = (long)cg::cdiv(g.M, p.tc.bm) * cg::cdiv(g.Cout, p.tc.bn) * zdim;
    const long kiters = cg::cdiv(g.Ktot, BK);
    p.splits = pick_splits(tiles, kiters);
    p.kchunk = cg::cdiv(cg::cdiv(kiters, p.splits) * BK, 32) * 32;
    p.splits = cg::cdiv(g.Ktot, p.kchunk);
    return p;
}
static size_t nn_ws_bytes(const Geom& g, const NNPlan& p, int ngroups) {
    size_t b = p.splits > 1 ? (size_t)p.splits * ngroups * g.nphase * g.M * g.Cout * sizeof(float) : 0;
    if (p.pm) b = std::max(b, nn_ws_bytes(g, plan_nn(g, ngroups, false), ngroups));   // run_nn may fall back (unaligned operands, statistics)
    return b;
}

struct TNPlan { TileCfg tc; int splits; int pchunk; bool pm; };
static bool tn_pm_use(const Geom& g, bool image_major_ok);
static bool tng_ok(const Geom& g, int pchunk, bool veca, bool vecb);
static int tn_pm_share(const Geom& g);
static TNPlan plan_tn(const Geom& g, int ngroups, bool allow_pm = true) {
    TNPlan p;
    int bn = g.Cout > 64 ? 128 : (g.Cout > 32 ? 64 : 32);
    int bm = g.Ktot > 64 ? 128 : 64;
    if (bn == 32) bm = 128;
    p.tc = {bm, bn};
    tile_forced(cg::OPT_TN_TILE, p.tc);
    bm = p.tc.bm; bn = p.tc.bn;
    const long tiles = (long)cg::cdiv(g.Ktot, bm) * cg::cdiv(g.Cout, bn) * g.nphase * ngroups;
    const long piters = cg::cdiv(g.M, BK);
    const int smax = kTnSplitMax, tgt = kTnTarget;
    const int forced = (int)cg::opt(cg::OPT_TN_SPLITS);
    p.pm = allow_pm && tn_pm_use(g, tng_ok(g, BK, true, true));
    if (p.pm) {
        // position-major tiles: a split is a share of every workgroup's OWN tile list, so any count works - the one that fills tgt
        // workgroups per CU most exactly (D32_st3's 5x5 layer: 13 tiles x 59 = 767 workgroups 119 us, x 64 = 832 130 us), with >= 8 K
        // tiles per workgroup on average (profiles/r06_sweeps.txt)
        const long kt = std::max<long>(1, piters * (100 - tn_pm_share(g)) / 100);
        long sp = forced > 0 ? forced : std::min<long>((long)tgt * cg::kNumCU / std::max<long>(1, tiles), kt / 8);
        p.splits = (int)std::max<long>(1, std::min<long>(sp, smax));
        p.pchunk = 0;
        return p;
    }
    int s = 1;
    if (forced > 0) s = (int)std::min<long>(forced, std::max<long>(1, piters));
    else
    while (tiles * s < (long)tgt * cg::kNumCU && piters / (s * 2) >= 8 && s < smax) s *= 2;
    p.pchunk = cg::cdiv(piters, s) * BK;
    p.splits = cg::cdiv(g.M, p.pchunk);
    return p;
}
static size_t tn_ws_bytes(const Geom& g, const TNPlan& p, int ngroups) {
    // a position-major plan falls back to the image-major one for unaligned operands: room for either
    const int splits = p.pm ? std::max(p.splits, plan_tn(g, ngroups, false).splits) : p.splits;
    return (size_t)splits * ngroups * g.nphase * ((size_t)g.Ktot + 1) * g.Cout * sizeof(float);
}

// ---- skinny 3x3 path (see skinny_conv3x3_kernel) ----
constexpr int kSkinnyWgradBlocks = 512;
static bool skinny_ok(int ngroups, int Cin, int Cout, int kH, int kW, int padH, int padW, int ups) {
    const bool on = cg::opt(cg::OPT_SKINNY) != 0;
    return on && ngroups == 1 && ups == 0 && kH == 3 && kW == 3 && padH == 1 && padW == 1 && (Cout == 3 || Cout == 1) &&
           (Cin == 64 || Cin == 128);
}
static size_t skinny_wgrad_ws_bytes(int Cin, int Cout) {
    return (size_t)kSkinnyWgradBlocks * (9 * (size_t)Cin + 1) * Cout * sizeof(float);
}
template <int CO>
static void skinny_forward_launch(hipStream_t st, const float* x, const float* w, const float* bias, float* y, int N, int H,
                                  int W, int Cin) {
    const long npix = (long)N * H * W;
    const int lp = Cin / 4;
    long blocks = (npix * lp + 255) / 256;
    if (blocks > cg::kNumCU * 2) blocks = cg::kNumCU * 2;
    if (lp == 32) hipLaunchKernelGGL((skinny_conv3x3_kernel<CO, 32>), dim3((unsigned)blocks), dim3(256), 0, st, x, w, bias, y, N, H, W);
    else hipLaunchKernelGGL((skinny_conv3x3_kernel<CO, 16>), dim3((unsigned)blocks), dim3(256), 0, st, x, w, bias, y, N, H, W);
}
template <int CO>
static void skinny_wgrad_launch(hipStream_t st, const float* x, const float* dy, float* part, float* bias_part, int N, int H,
                                int W, int Cin, int blocks, long ppb) {
    const int lp = Cin / 4;
    const size_t shb = (size_t)4 * lp * (36 * CO + CO + 1) * sizeof(float);
    if (lp == 32) hipLaunchKernelGGL((skinny_wgrad3x3_kernel<CO, 32>), dim3(blocks), dim3(256), shb, st, x, dy, part, bias_part, N, H, W, ppb);
    else hipLaunchKernelGGL((skinny_wgrad3x3_kernel<CO, 16>), dim3(blocks), dim3(256), shb, st, x, dy, part, bias_part, N, H, W, ppb);
}

struct Epi { int act; float slope; const float* const* alpha; float* const* y_act; float* stats; };

// Position-major rows with the zero-padding taps skipped (Geom::pmn): a plain stride-1 convolution whose padding is a real share of
// its MACs (CG_PAD_SKIP = the least share in per cent, 0 = off), batch a power of two and a multiple of the 64-row tile.  Returns the
// number of (pixel position, tap) pairs inside the image, 0 = not this launch.
static long pad_skip_valid(const Geom& g, const TileCfg& tc) {
    const long thr = cg::opt(cg::OPT_PAD_SKIP);
    if (thr <= 0 || !cg::opt(cg::OPT_NN_GLDS) || !cg::opt(cg::OPT_GEMM_BK32)) return 0;
    if (g.nphase != 1 || g.so != 1 || g.td.ngroups != 1 || g.td.ss != 1 || g.ntaps < 9 || tc.bm != 64 || !(tc.bn == 128 || tc.bn == 64)) return 0;
    const int hw = g.Hg * g.Wg, N = g.M / hw;
    if (N * hw != g.M || N % 64 || ilog2_exact(N) < 0 || g.Cout % tc.bn || (long)g.M * g.Cout * 4L >= 0x7fffffffL) return 0;
    long valid = 0;
    for (int oy = 0; oy < g.Hg; ++oy)
        for (int ox = 0; ox < g.Wg; ++ox)
            for (int t = 0; t < g.ntaps; ++t) {
                int ty, tx, off;
                tap_decode(g, t, 0, 0, ty, tx, off);
                valid += (unsigned)(oy + ty) < (unsigned)g.Hv && (unsigned)(ox + tx) < (unsigned)g.Wv;
            }
    return (long)hw * g.ntaps * (100 - thr) >= valid * 100 ? valid : 0;
}

static int run_nn(hipStream_t st, const Geom& g, int ngroups, const float* const* x, const float* const* w,
                  const float* const* bias, float* const* y, void* ws, size_t ws_bytes, const char* who,
                  const Epi* ep = nullptr) {
    CG_REQUIRE(ngroups >= 1 && ngroups <= MAXG, "%s: 1..%d groups per launch", who, MAXG);
    NNPlan p = plan_nn(g, ngroups);
    {
        bool al16 = true;
        for (int i = 0; i < ngroups; ++i) al16 = al16 && x[i] && w[i] && (uintptr_t)x[i] % 16 == 0 && (uintptr_t)w[i] % 16 == 0;
        if (p.pm && (!al16 || g.Cout % 4 || (ep && ep->stats))) p = plan_nn(g, ngroups, false);   // the LDS-direct kernel cannot: image-major plan
    }
    const size_t need = nn_ws_bytes(g, p, ngroups);
    CG_REQUIRE(need == 0 || (ws && ws_bytes >= need), "%s: workspace too small (%zu < %zu)", who, ws_bytes, need);
    NNArgs a;
    memset(&a, 0, sizeof(a));
    const float* xs[MAXG] = {nullptr, nullptr, nullptr, nullptr};
    const float* wsv[MAXG] = {nullptr, nullptr, nullptr, nullptr};
    const float* bs[MAXG] = {nullptr, nullptr, nullptr, nullptr};
    float* ys[MAXG] = {nullptr, nullptr, nullptr, nullptr};
    bool al = true, alw = true;
    for (int i = 0; i < ngroups; ++i) {
        CG_REQUIRE(x[i] && w[i] && y[i], "%s: null pointer (group %d)", who, i);
        xs[i] = x[i]; wsv[i] = w[i]; bs[i] = bias ? bias[i] : nullptr; ys[i] = y[i];
        al = al && ((uintptr_t)x[i] % 16 == 0);
        alw = alw && ((uintptr_t)w[i] % 16 == 0);
    }
    a.x0 = xs[0]; a.x1 = xs[1]; a.x2 = xs[2]; a.x3 = xs[3];
    a.w0 = wsv[0]; a.w1 = wsv[1]; a.w2 = wsv[2]; a.w3 = wsv[3];
    a.b0 = bs[0]; a.b1 = bs[1]; a.b2 = bs[2]; a.b3 = bs[3];
    a.y0 = ys[0]; a.y1 = ys[1]; a.y2 = ys[2]; a.y3 = ys[3];
    a.part = (float*)ws; a.ngroups = ngroups; a.g = g;
    a.kchunk = p.kchunk; a.nsplit = p.splits;
    a.xcd_swizzle = (int)cg::opt(cg::OPT_XCD_SWIZZLE);
    if (ep && ep->act) {
        CG_REQUIRE(ep->act == 1 || ep->act == 2, "%s: unknown activation %d", who, ep->act);
        CG_REQUIRE(ep->y_act, "%s: fused activation needs y_act", who);
        const float* als[MAXG] = {nullptr, nullptr, nullptr, nullptr};
        float* zs[MAXG] = {nullptr, nullptr, nullptr, nullptr};
        for (int i = 0; i < ngroups; ++i) {
            CG_REQUIRE(ep->y_act[i], "%s: null y_act (group %d)", who, i);
            zs[i] = ep->y_act[i];
            if (ep->act == 1) {
                CG_REQUIRE(ep->alpha && ep->alpha[i], "%s: PReLU epilogue needs the slope pointer (group %d)", who, i);
                als[i] = ep->alpha[i];
            }
        }
        a.act = ep->act; a.slope = ep->slope;
        a.al0 = als[0]; a.al1 = als[1]; a.al2 = als[2]; a.al3 = als[3];
        a.z0 = zs[0]; a.z1 = zs[1]; a.z2 = zs[2]; a.z3 = zs[3];
    }
    if (ep && ep->stats) {
        CG_REQUIRE(ngroups == 1 && p.splits == 1, "%s: epilogue statistics need a single-group, unsplit launch", who);
        a.stats = ep->stats;
    }
    const bool fast = (g.Cin % BK == 0) && al;
    const bool vecb = (g.Cout % 4 == 0) && alw;
    const bool bk32 = cg::opt(cg::OPT_GEMM_BK32) != 0 && (g.Cin % 32 == 0) && (p.kchunk % 32 == 0);
    if (a.xcd_swizzle & 4) {
        // weights-stationary XCDs only where they move fewer bytes: an r x c XCD grid over (row tiles) x (column tiles) fetches
        // c |x| + r |w| per phase - row ranges (r = 8, c = 1) against (8 / ntn, ntn).  conv1's forward: |x| = |w| = 4.2 MB -> stationary
        // (161 -> 68 MB measured); conv2's direct data gradient: |x| = 33.5, |w| = 8.4 MB -> row ranges (118 MB; 258 MB stationary)
        const int ntn = cg::cdiv(g.Cout, p.tc.bn);
        const double xb = (double)(g.M / (g.Hg * g.Wg)) * g.Hs * g.Ws * g.Cin * 4.0, wb = (double)g.Ktot * g.Cout * 4.0;
        if (!(ntn == 2 || ntn == 4 || ntn == 8) || ntn * xb + (8 / ntn) * wb >= xb + 8 * wb) a.xcd_swizzle &= ~4;
    }
    if (p.pm) {
        a.g.pmn = g.M / (g.Hg * g.Wg);
        a.g.pm_lg = ilog2_exact(a.g.pmn);
    }
    dim3 grid(cg::cdiv(g.M, p.tc.bm) * cg::cdiv(g.Cout, p.tc.bn), p.splits, g.nphase * ngroups);
    if (p.tc.bm == 128 && p.tc.bn == 128) launch_nn<128, 128, 2, 2>(a, grid, st, fast, vecb, bk32);
    else if (p.tc.bm == 64 && p.tc.bn == 128) launch_nn<64, 128, 2, 2>(a, grid, st, fast, vecb, bk32);
    else if (p.tc.bm == 128 && p.tc.bn == 64) launch_nn<128, 64, 2, 2>(a, grid, st, fast, vecb, bk32);
    else if (p.tc.bm == 64 && p.tc.bn == 64) launch_nn<64, 64, 2, 2>(a, grid, st, fast, vecb, bk32);
    else launch_nn<128, 32, 4, 1>(a, grid, st, fast, vecb, bk32);
    CG_LAUNCH_CHECK();
    if (p.splits > 1) {
        const long PMN = (long)ngroups * g.nphase * g.M * g.Cout;
        bool v4 = g.Cout % 4 == 0 && (uintptr_t)ws % 16 == 0;
        for (int i = 0; i < ngroups; ++i)
            v4 = v4 && (uintptr_t)ys[i] % 16 == 0 && (!bs[i] || (uintptr_t)bs[i] % 16 == 0) && (!a.act || (uintptr_t)ep->y_act[i] % 16 == 0);
        if ((PMN >= 256L * 1024 || p.splits < 8) && v4)
            hipLaunchKernelGGL((nn_splitk_reduce_kernel<1, true>), dim3(cg::ew_grid(PMN / 4)), dim3(256), 0, st, a, p.splits);
        else if (PMN >= 256L * 1024 || p.splits < 8)
            hipLaunchKernelGGL(nn_splitk_reduce_kernel<1>, dim3(cg::ew_grid(PMN)), dim3(256), 0, st, a, p.splits);
        else
            hipLaunchKernelGGL(nn_splitk_reduce_kernel<8>, dim3((unsigned)cg::cdiv(PMN, 32)), dim3(256), 0, st, a, p.splits);
        CG_LAUNCH_CHECK();
    }
    return 0;
}

static int conv_geom(Geom& g, int N, int Hp, int Wp, int Cin, int Cout, int kH, int kW, int padH, int padW, int ups) {
    if (check_dims(N, Hp, Wp, Cin, Cout, kH, kW, padH, padW, ups)) return 1;
    return ups ? geom_phase_fwd(g, N, Hp, Wp, Cin, Cout, kH, padH) : geom_plain(g, N, Hp, Wp, Cin, Cout, kH, kW, padH, padW);
}

}  // namespace

#ifdef CG_TRACE
extern "C" int cg_debug_set_trace(void* p) {
    unsigned long long* q = (unsigned long long*)p;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_trace), &q, sizeof(q)) == hipSuccess ? 0 : 1;
}
#endif

extern "C" {

size_t cg_conv2d_workspace_bytes_grouped(int ngroups, int N, int Hp, int Wp, int Cin, int Cout, int kH, int kW, int padH,
                                         int padW, int ups) {
    Geom g;
    if (ngroups < 1 || ngroups > MAXG || conv_geom(g, N, Hp, Wp, Cin, Cout, kH, kW, padH, padW, ups)) return 0;
    return nn_ws_bytes(g, plan_nn(g, ngroups), ngroups);
}
size_t cg_conv2d_workspace_bytes(int N, int Hp, int Wp, int Cin, int Cout, int kH, int kW, int padH, int padW, int ups) {
    return cg_conv2d_workspace_bytes_grouped(1, N, Hp, Wp, Cin, Cout, kH, kW, padH, padW, ups);
}

// rows of the statistics buffer an epilogue-statistics launch writes (0: this geometry runs split-K or skinny, take the
// statistics with cg_bn_stats instead)
size_t cg_conv2d_stats_rows(int N, int Hp, int Wp, int Cin, int Cout, int kH, int kW, int padH, int padW, int ups) {
    Geom g;
    if (conv_geom(g, N, Hp, Wp, Cin, Cout, kH, kW, padH, padW, ups)) return 0;
    if (skinny_ok(1, Cin, Cout, kH, kW, padH, padW, ups)) return 0;
    const NNPlan p = plan_nn(g, 1, false);   // run_nn launches the image-major plan whenever statistics are asked for: query THAT plan
    if (p.splits != 1) return 0;
    const int wm = p.tc.bn == 32 ? 4 : 2;
    return (size_t)g.nphase * cg::cdiv(g.M, p.tc.bm) * wm;
}

int cg_conv2d_forward_ex(void* stream, int ngroups, const float* const* x, const float* const* wpk, const float* const* bias,
                         float* const* y, int N, int Hp, int Wp, int Cin, int Cout, int kH, int kW, int padH, int padW,
                         int ups, int act, float slope, const float* const* alpha, float* const* y_act, float* stats,
                         void* ws, size_t ws_bytes) {
    CG_REQUIRE(x && wpk && y, "cg_conv2d_forward_ex: null pointer");
    Geom g;
    if (conv_geom(g, N, Hp, Wp, Cin, Cout, kH, kW, padH, padW, ups)) return 1;
    CG_REQUIRE(!skinny_ok(ngroups, Cin, Cout, kH, kW, padH, padW, ups) || (act == 0 && !stats),
               "cg_conv2d_forward_ex: no fused epilogue on the skinny (<= 4 output planes) path");
    if (act == 0 && !stats)
        return cg_conv2d_forward_grouped(stream, ngroups, x, wpk, bias, y, N, Hp, Wp, Cin, Cout, kH, kW, padH, padW, ups, ws,
                                         ws_bytes);
    Epi ep{act, slope, alpha, y_act, stats};
    return run_nn(cg::S(stream), g, ngroups, x, wpk, bias, y, ws, ws_bytes, "cg_conv2d_forward_ex", &ep);
}

int cg_conv2d_forward_grouped(void* stream, int ngroups, const float* const* x, const float* const* wpk,
                              const float* const* bias, float* const* y, int N, int Hp, int Wp, int Cin, int Cout, int kH,
                              int kW, int padH, int padW, int ups, void* ws, size_t ws_bytes) {
    CG_REQUIRE(x && wpk && y, "cg_conv2d_forward_grouped: null pointer");
    Geom g;
    if (conv_geom(g, N, Hp, Wp, Cin, Cout, kH, kW, padH, padW, ups)) return 1;
    if (skinny_ok(ngroups, Cin, Cout, kH, kW, padH, padW, ups) && x[0] && wpk[0] && y[0] && (uintptr_t)x[0] % 16 == 0 &&
        (long)N * Hp * Wp < 0x7fffffffL) {
        const float* b0 = bias ? bias[0] : nullptr;
        if (cg::opt(cg::OPT_SKINNY) == 1 && cg::skinny_mfma_ok(Cin, Cout, Hp, Wp)) {
            if (cg::skinny_mfma_forward(cg::S(stream), x[0], wpk[0], b0, y[0], N, Hp, Wp, Cin, Cout)) return 1;
        }
        else if (Cout == 3) skinny_forward_launch<3>(cg::S(stream), x[0], wpk[0], b0, y[0], N, Hp, Wp, Cin);
        else skinny_forward_launch<1>(cg::S(stream), x[0], wpk[0], b0, y[0], N, Hp, Wp, Cin);
        CG_LAUNCH_CHECK();
        return 0;
    }
    if (cg::wino3_geom_ok(ngroups, N, Hp, Wp, Cin, Cout, kH, kW, padH, padW, ups)) {   // fused F(2x2,3x3): csrc/wino3.hip
        const int rc = cg::wino3_forward(cg::S(stream), ngroups, x, wpk, bias, y, N, Hp, Wp);
        if (rc) return rc < 0 ? 1 : 0;
    }
    return run_nn(cg::S(stream), g, ngroups, x, wpk, bias, y, ws, ws_bytes, "cg_conv2d_forward_grouped");
}

int cg_conv2d_forward(void* stream, const float* x, const float* wpk, const float* bias, float* y, int N, int Hp, int Wp,
                      int Cin, int Cout, int kH, int kW, int padH, int padW, int ups, void* ws, size_t ws_bytes) {
    return cg_conv2d_forward_grouped(stream, 1, &x, &wpk, bias ? &bias : nullptr, &y, N, Hp, Wp, Cin, Cout, kH, kW, padH, padW,
                                     ups, ws, ws_bytes);
}

size_t cg_conv2d_dgrad_ups2_workspace_bytes(int N, int Hp, int Wp, int Cin, int Cout, int k, int pad) {
    Geom g;
    if (check_dims(N, Hp, Wp, Cin, Cout, k, k, pad, pad, 1)) return 0;
    if (geom_phase_dgrad(g, N, Hp, Wp, Cin, Cout, k, pad)) return 0;
    return nn_ws_bytes(g, plan_nn(g, 1), 1);
}

int cg_conv2d_dgrad_ups2(void* stream, const float* dy, const float* wb_ph, float* dx_lo, int N, int Hp, int Wp, int Cin,
                         int Cout, int k, int pad, void* ws, size_t ws_bytes) {
    CG_REQUIRE(dy && wb_ph && dx_lo, "cg_conv2d_dgrad_ups2: null pointer");
    Geom g;
    if (check_dims(N, Hp, Wp, Cin, Cout, k, k, pad, pad, 1)) return 1;
    if (geom_phase_dgrad(g, N, Hp, Wp, Cin, Cout, k, pad)) return 1;
    return run_nn(cg::S(stream), g, 1, &dy, &wb_ph, nullptr, &dx_lo, ws, ws_bytes, "cg_conv2d_dgrad_ups2");
}

static size_t wgrad_ws_bytes_groups(int ngroups, int maxg, int N, int Hp, int Wp, int Cin, int Cout, int kH, int kW, int padH, int padW, int ups) {
    Geom g;
    if (ngroups < 1 || ngroups > maxg || conv_geom(g, N, Hp, Wp, Cin, Cout, kH, kW, padH, padW, ups)) return 0;
    const size_t reg = tn_ws_bytes(g, plan_tn(g, ngroups), ngroups);
    return skinny_ok(ngroups, Cin, Cout, kH, kW, padH, padW, ups) ? std::max(reg, skinny_wgrad_ws_bytes(Cin, Cout)) : reg;
}
// The size query doubles as the capability check of the entry point it belongs to (0 = this launch does not exist): <= MAXG groups of
// separate tensors for cg_conv2d_wgrad_grouped / _deferred, <= kMaxStridedGroups equally spaced ones for cg_conv2d_wgrad_strided.
size_t cg_conv2d_wgrad_workspace_bytes_grouped(int ngroups, int N, int Hp, int Wp, int Cin, int Cout, int kH, int kW,
                                               int padH, int padW, int ups) {
    return wgrad_ws_bytes_groups(ngroups, MAXG, N, Hp, Wp, Cin, Cout, kH, kW, padH, padW, ups);
}
size_t cg_conv2d_wgrad_workspace_bytes_strided(int ngroups, int N, int Hp, int Wp, int Cin, int Cout, int kH, int kW,
                                               int padH, int padW, int ups) {
    return wgrad_ws_bytes_groups(ngroups, kMaxStridedGroups, N, Hp, Wp, Cin, Cout, kH, kW, padH, padW, ups);
}
size_t cg_conv2d_wgrad_workspace_bytes(int N, int Hp, int Wp, int Cin, int Cout, int kH, int kW, int padH, int padW,
                                       int ups) {
    return cg_conv2d_wgrad_workspace_bytes_grouped(1, N, Hp, Wp, Cin, Cout, kH, kW, padH, padW, ups);
}

namespace {
// reductions queued by the deferred entry points, per stream (one host thread per stream, but several streams may queue)
std::mutex g_red_mu;
std::unordered_map<hipStream_t, std::vector<RedJob>> g_red_queue;

// Workgroups of a partial-sum reduction launch: one per 32-element block.  Unlike the grid-stride element-wise kernels (cg::ew_grid)
// these do NOT gain from fewer, longer-lived workgroups (round 4, same box: 6.32 / 6.38 / 6.40 ms per step at 0 / 8 / 32 per CU): a
// block is a short dependent chain (strided partial loads -> LDS -> one add), so walking blocks serialises latency.
long red_cap() { return 0x7fffffffL; }

int small_reduce(hipStream_t st, bool defer, const RedJob& job, int ngroups) {
    if (defer) {
        std::lock_guard<std::mutex> lk(g_red_mu);
        g_red_queue[st].push_back(job);
        g_red_queue[st].back().block0 = ngroups;   // the group count rides here until the flush lays the blocks out
        return 0;
    }
    const int nblocks = job.wblocks + job.bblocks;
    const long cap = std::max<long>(1, red_cap() / std::max(1, ngroups));   // workgroups per group
    dim3 sgrid((unsigned)std::min<long>(nblocks, cap), ngroups);
    if (job.ups)
        hipLaunchKernelGGL(wgrad_reduce_small_kernel<true>, sgrid, dim3(256), 0, st, job.part, job.bias_part, job.rp, job.Cin, job.Cout,
                           job.k, job.KK, job.pad, job.kp, job.S, job.P, job.scale, job.sstride, job.gstride, job.bsstride, job.wblocks, nblocks, 0);
    else
        hipLaunchKernelGGL(wgrad_reduce_small_kernel<false>, sgrid, dim3(256), 0, st, job.part, job.bias_part, job.rp, job.Cin, job.Cout,
                           job.k, job.KK, job.pad, job.kp, job.S, job.P, job.scale, job.sstride, job.gstride, job.bsstride, job.wblocks, nblocks, job.v4);
    CG_LAUNCH_CHECK();
    return 0;
}

int wgrad_impl(void* stream, int ngroups, const float* const* x, const float* const* dy, float* const* gw, float* const* gb, int N,
               int Hp, int Wp, int Cin, int Cout, int kH, int kW, int padH, int padW, int ups, float scale, void* ws, size_t ws_bytes,
               bool defer, long xgs = 0, long dgs = 0, long gws = 0, bool gemm_only = false);
}  // namespace

int cg_conv2d_wgrad_grouped(void* stream, int ngroups, const float* const* x, const float* const* dy, float* const* gw,
                            float* const* gb, int N, int Hp, int Wp, int Cin, int Cout, int kH, int kW, int padH, int padW,
                            int ups, float scale, void* ws, size_t ws_bytes) {
    return wgrad_impl(stream, ngroups, x, dy, gw, gb, N, Hp, Wp, Cin, Cout, kH, kW, padH, padW, ups, scale, ws, ws_bytes, false);
}

int cg_conv2d_wgrad_grouped_deferred(void* stream, int ngroups, const float* const* x, const float* const* dy, float* const* gw,
                                     float* const* gb, int N, int Hp, int Wp, int Cin, int Cout, int kH, int kW, int padH, int padW,
                                     int ups, float scale, void* ws, size_t ws_bytes) {
    return wgrad_impl(stream, ngroups, x, dy, gw, gb, N, Hp, Wp, Cin, Cout, kH, kW, padH, padW, ups, scale, ws, ws_bytes, true);
}

// ngroups (<= 16) weight gradients of ONE geometry whose tensors are equally spaced in memory - group g reads x + g*x_stride and
// dy + g*dy_stride (floats) and accumulates into gw + g*gw_stride - in one GEMM launch and one reduction.  What the Winograd-domain
// weight gradient of winograd.hip is: 16 independent [tiles x Cin]^T [tiles x 4 Cout] products, one per transform position
// (four launches of four groups through cg_conv2d_wgrad_grouped before round 4).  No bias gradients.  Workspace:
// cg_conv2d_wgrad_workspace_bytes_strided(ngroups, ...).
int cg_conv2d_wgrad_strided(void* stream, int ngroups, const float* x, long x_stride, const float* dy, long dy_stride, float* gw,
                            long gw_stride, int N, int Hp, int Wp, int Cin, int Cout, int kH, int kW, int padH, int padW, int ups,
                            float scale, void* ws, size_t ws_bytes) {
    CG_REQUIRE(x && dy && gw && x_stride > 0 && dy_stride > 0 && gw_stride > 0, "cg_conv2d_wgrad_strided: bad arguments");
    CG_REQUIRE(ngroups >= 1 && ngroups <= kMaxStridedGroups, "cg_conv2d_wgrad_strided: 1..%d groups per launch", kMaxStridedGroups);
    return wgrad_impl(stream, ngroups, &x, &dy, &gw, nullptr, N, Hp, Wp, Cin, Cout, kH, kW, padH, padW, ups, scale, ws, ws_bytes, false,
                      x_stride, dy_stride, gw_stride);
}

// The weight-gradient GEMM alone: the split partial sums [splits][phases][taps*Cin][Cout] are left in ws, nothing is reduced into a
// gradient (exported so that the kernel can be timed / profiled in isolation - bench.py's roofline entry; not part of a training step).
int cg_conv2d_wgrad_gemm(void* stream, const float* x, const float* dy, int N, int Hp, int Wp, int Cin, int Cout, int kH, int kW, int padH,
                         int padW, int ups, void* ws, size_t ws_bytes) {
    CG_REQUIRE(x && dy && ws, "cg_conv2d_wgrad_gemm: null pointer");
    CG_REQUIRE(!skinny_ok(1, Cin, Cout, kH, kW, padH, padW, ups), "cg_conv2d_wgrad_gemm: this geometry runs the skinny kernel, not the GEMM");
    float* gw = (float*)ws;   // never written: the reduction is skipped
    return wgrad_impl(stream, 1, &x, &dy, &gw, nullptr, N, Hp, Wp, Cin, Cout, kH, kW, padH, padW, ups, 1.f, ws, ws_bytes, false, 0, 0, 0, true);
}

int cg_conv2d_wgrad_pending(void* stream, int* njobs) {
    CG_REQUIRE(njobs, "cg_conv2d_wgrad_pending: null pointer");
    std::lock_guard<std::mutex> lk(g_red_mu);
    auto it = g_red_queue.find(cg::S(stream));
    *njobs = it == g_red_queue.end() ? 0 : (int)it->second.size();
    return 0;
}

int cg_conv2d_wgrad_flush(void* stream) {
    hipStream_t st = cg::S(stream);
    std::vector<RedJob> jobs;
    {
        std::lock_guard<std::mutex> lk(g_red_mu);
        auto it = g_red_queue.find(st);
        if (it == g_red_queue.end() || it->second.empty()) return 0;
        jobs.swap(it->second);
    }
    for (size_t j0 = 0; j0 < jobs.size(); j0 += kRedJobs) {
        RedJobTable t;
        memset(&t, 0, sizeof(t));
        t.n = (int)std::min<size_t>(kRedJobs, jobs.size() - j0);
        int blocks = 0;
        for (int q = 0; q < t.n; ++q) {
            t.j[q] = jobs[j0 + q];
            const int ngroups = t.j[q].block0;
            t.j[q].block0 = blocks;
            blocks += ngroups * (t.j[q].wblocks + t.j[q].bblocks);
        }
        const int wgs = (int)std::min<long>(blocks, red_cap());
        hipLaunchKernelGGL(wgrad_reduce_batch_kernel, dim3(wgs), dim3(256), 0, st, t, blocks);
        CG_LAUNCH_CHECK();
    }
    return 0;
}

namespace {
int wgrad_impl(void* stream, int ngroups, const float* const* x, const float* const* dy, float* const* gw, float* const* gb, int N,
               int Hp, int Wp, int Cin, int Cout, int kH, int kW, int padH, int padW, int ups, float scale, void* ws, size_t ws_bytes,
               bool defer, long xgs, long dgs, long gws_, bool gemm_only) {
    CG_REQUIRE(x && dy && gw, "cg_conv2d_wgrad: null pointer");
    const bool strided = xgs != 0;     // equally spaced groups: only x[0] / dy[0] / gw[0] are pointers
    CG_REQUIRE(ngroups >= 1 && ngroups <= (strided ? kMaxStridedGroups : MAXG), "cg_conv2d_wgrad: 1..%d groups per launch", strided ? kMaxStridedGroups : MAXG);
    CG_REQUIRE(!strided || (!gb && !defer), "cg_conv2d_wgrad: strided groups carry no bias gradient and are not deferred");
    TNArgs a;
    memset(&a, 0, sizeof(a));
    if (conv_geom(a.g, N, Hp, Wp, Cin, Cout, kH, kW, padH, padW, ups)) return 1;
    const Geom& g = a.g;
    if (!strided && skinny_ok(ngroups, Cin, Cout, kH, kW, padH, padW, ups) && x[0] && dy[0] && gw[0] && (uintptr_t)x[0] % 16 == 0 && ws &&
        ws_bytes >= skinny_wgrad_ws_bytes(Cin, Cout) && (long)N * Hp * Wp < 0x7fffffffL) {
        hipStream_t st = cg::S(stream);
        const long npix = (long)N * Hp * Wp;
        const long ppb = (npix + kSkinnyWgradBlocks - 1) / kSkinnyWgradBlocks;
        int blocks = (int)((npix + ppb - 1) / ppb);
        float* part = (float*)ws;
        float* bpart = part + (size_t)kSkinnyWgradBlocks * 9 * Cin * Cout;
        float* gb0 = gb ? gb[0] : nullptr;
        if (cg::opt(cg::OPT_SKINNY) == 1 && cg::skinny_mfma_ok(Cin, Cout, Hp, Wp)) {
            // 256 partial planes: the kernel itself measures the same with 512 (18.0 / 18.8 us at batch 128), the reduction behind it 18 -> 12.6 us
            blocks = cg::skinny_mfma_wgrad(st, x[0], dy[0], part, gb0 ? bpart : nullptr, N, Hp, Wp, Cin, Cout, kSkinnyWgradBlocks / 2);
            if (blocks < 0) return 1;
        }
        else if (Cout == 3) skinny_wgrad_launch<3>(st, x[0], dy[0], part, gb0 ? bpart : nullptr, N, Hp, Wp, Cin, blocks, ppb);
        else skinny_wgrad_launch<1>(st, x[0], dy[0], part, gb0 ? bpart : nullptr, N, Hp, Wp, Cin, blocks, ppb);
        CG_LAUNCH_CHECK();
        RedJob job;
        memset(&job, 0, sizeof(job));
        job.part = part; job.bias_part = bpart;
        job.rp.gw0 = gw[0]; job.rp.gb0 = gb0;
        job.Cin = Cin; job.Cout = Cout; job.k = 3; job.KK = 9; job.pad = 1; job.S = blocks; job.P = 1; job.scale = scale;
        job.sstride = 9L * Cin * Cout; job.gstride = 0; job.bsstride = Cout;
        job.wblocks = cg::cdiv(9L * Cin * Cout, 32); job.bblocks = gb0 ? cg::cdiv(Cout, 32) : 0;
        return small_reduce(st, defer, job, 1);
    }
    // nn.View -> nn.Linear (an H x W kernel on the H x W map, 1 x 1 grid): one kernel straight into gradWeight (headwg.hip)
    if (!strided && !gemm_only && ngroups == 1 && !ups && kH == Hp && kW == Wp && padH == 0 && padW == 0 && cg::opt(cg::OPT_PAD_SKIP) > 0 &&
        cg::opt(cg::OPT_TN_GLDS) && cg::head_wgrad_ok(N, kH * kW, Cin, Cout) && x[0] && dy[0] && gw[0] && (uintptr_t)x[0] % 16 == 0 &&
        (uintptr_t)dy[0] % 16 == 0 && (uintptr_t)gw[0] % 4 == 0)
        return cg::head_wgrad(cg::S(stream), x[0], dy[0], gw[0], gb ? gb[0] : nullptr, N, kH * kW, Cin, Cout, scale) != 0;
    TNPlan p = plan_tn(g, ngroups);
    const size_t need = tn_ws_bytes(g, p, ngroups);
    CG_REQUIRE(ws && ws_bytes >= need, "cg_conv2d_wgrad: workspace too small (%zu < %zu)", ws_bytes, need);
    const float* xs[MAXG] = {nullptr, nullptr, nullptr, nullptr};
    const float* ds[MAXG] = {nullptr, nullptr, nullptr, nullptr};
    bool veca = Cin % 4 == 0, vecb = Cout % 4 == 0, any_gb = false;
    for (int i = 0; i < (strided ? 1 : ngroups); ++i) {
        CG_REQUIRE(x[i] && dy[i] && gw[i], "cg_conv2d_wgrad: null pointer (group %d)", i);
        xs[i] = x[i]; ds[i] = dy[i];
        veca = veca && ((uintptr_t)x[i] % 16 == 0);
        vecb = vecb && ((uintptr_t)dy[i] % 16 == 0);
        any_gb = any_gb || (gb && gb[i]);
    }
    if (strided) { veca = veca && xgs % 4 == 0; vecb = vecb && dgs % 4 == 0; }
    if (p.pm && !(veca && vecb)) p = plan_tn(g, ngroups, false);   // unaligned operands: the image-major plan (the workspace holds either)
    a.xgs = xgs; a.dgs = dgs;
    a.x0 = xs[0]; a.x1 = xs[1]; a.x2 = xs[2]; a.x3 = xs[3];
    a.d0 = ds[0]; a.d1 = ds[1]; a.d2 = ds[2]; a.d3 = ds[3];
    a.ngroups = ngroups; a.part = (float*)ws; a.pchunk = p.pchunk;
    a.xcd_swizzle = (int)cg::opt(cg::OPT_XCD_SWIZZLE);
    const int ZP = ngroups * g.nphase;
    const long wplane = (long)g.Ktot * g.Cout;              // one (group, phase) slab of weight partials
    a.bias_part = any_gb ? (float*)ws + (size_t)p.splits * ZP * wplane : nullptr;
    hipStream_t st = cg::S(stream);
    dim3 grid(cg::cdiv(g.Ktot, p.tc.bm) * cg::cdiv(g.Cout, p.tc.bn), p.splits, ZP);
    if (p.pm && veca && vecb) a.pm_n = g.M / (g.Hg * g.Wg);
    if (a.pm_n || tng_ok(g, p.pchunk, veca, vecb)) {
        if (p.tc.bm == 128 && p.tc.bn == 128) launch_tng<128, 128, 2, 2>(a, grid, st);
        else if (p.tc.bm == 64 && p.tc.bn == 128) launch_tng<64, 128, 2, 2>(a, grid, st);
        else if (p.tc.bm == 128 && p.tc.bn == 64) launch_tng<128, 64, 2, 2>(a, grid, st);
        else if (p.tc.bm == 64 && p.tc.bn == 64) launch_tng<64, 64, 2, 2>(a, grid, st);
        else launch_tng<128, 32, 4, 1>(a, grid, st);
    }
    else if (p.tc.bm == 128 && p.tc.bn == 128) launch_tn<128, 128, 2, 2>(a, grid, st, veca, vecb);
    else if (p.tc.bm == 64 && p.tc.bn == 128) launch_tn<64, 128, 2, 2>(a, grid, st, veca, vecb);
    else if (p.tc.bm == 128 && p.tc.bn == 64) launch_tn<128, 64, 2, 2>(a, grid, st, veca, vecb);
    else if (p.tc.bm == 64 && p.tc.bn == 64) launch_tn<64, 64, 2, 2>(a, grid, st, veca, vecb);
    else launch_tn<128, 32, 4, 1>(a, grid, st, veca, vecb);
    CG_LAUNCH_CHECK();
    if (gemm_only) return 0;   // cg_conv2d_wgrad_gemm: the partial sums stay in ws
    const int KK = kH * kW;
    int ci_t = KK == 1 ? 64 : 8;
    while (ci_t > 1 && (size_t)KK * ci_t * 33 * sizeof(float) > 60000) ci_t >>= 1;
    const size_t shb = (size_t)KK * ci_t * 33 * sizeof(float);
    dim3 rgrid(cg::cdiv(Cin, ci_t), cg::cdiv(Cout, 32));
    const long relems = (long)KK * Cin * Cout;
    const long sstride = (long)ZP * wplane;                 // floats between consecutive splits
    const int kp = ups ? phase_kp(kH, padH) : 0;
    if (strided && KK == 1 && !ups && !any_gb && Cin % 32 == 0 && Cout % 32 == 0 && g.nphase == 1) {
        hipLaunchKernelGGL(wgrad_reduce_t_kernel, dim3(Cin / 32, Cout / 32, ngroups), dim3(256), 0, st, (const float*)ws, gw[0], gws_, Cin, Cout,
                           p.splits, sstride, (long)g.nphase * wplane, scale);
        CG_LAUNCH_CHECK();
        return 0;
    }
    if ((long)rgrid.x * rgrid.y < cg::kNumCU) {
        RedJob job;
        memset(&job, 0, sizeof(job));
        float* gws[MAXG] = {nullptr, nullptr, nullptr, nullptr};
        float* gbs[MAXG] = {nullptr, nullptr, nullptr, nullptr};
        for (int gi = 0; gi < (strided ? 1 : ngroups); ++gi) { gws[gi] = gw[gi]; gbs[gi] = gb ? gb[gi] : nullptr; }
        job.rp.gws = strided ? gws_ : 0;
        job.rp.gw0 = gws[0]; job.rp.gw1 = gws[1]; job.rp.gw2 = gws[2]; job.rp.gw3 = gws[3];
        job.rp.gb0 = gbs[0]; job.rp.gb1 = gbs[1]; job.rp.gb2 = gbs[2]; job.rp.gb3 = gbs[3];
        job.part = (const float*)ws; job.bias_part = a.bias_part;
        job.Cin = Cin; job.Cout = Cout; job.k = kH; job.KK = KK; job.pad = padH; job.kp = ups ? kp : 0; job.S = p.splits; job.P = g.nphase;
        job.ups = ups ? 1 : 0; job.scale = scale;
        job.sstride = sstride; job.gstride = (long)g.nphase * wplane; job.bsstride = (long)ZP * Cout;
        job.wblocks = cg::cdiv(relems, 32); job.bblocks = any_gb ? cg::cdiv(Cout, 32) : 0;
        // a queued KK == 1 job (nn.Linear; D's 20480 -> 256 head is 5.2 M elements = 164 000 32-element blocks) reduces as transpose tiles
        if (defer && KK == 1 && !ups && Cin % 32 == 0 && Cout % 32 == 0 && g.nphase == 1) {
            job.tr = 1; job.wblocks = (Cin / 32) * (Cout / 32);
        } else if (!ups && Cout % 4 == 0 && (uintptr_t)ws % 16 == 0 && sstride % 4 == 0 && job.gstride % 4 == 0) {
            job.v4 = 1; job.wblocks = cg::cdiv(relems, 128);
        }
        return small_reduce(st, defer, job, ngroups);
    }
    for (int gi = 0; gi < ngroups; ++gi) {
        const float* pg = (const float*)ws + (long)gi * g.nphase * wplane;
        float* gwi = strided ? gw[0] + (long)gi * gws_ : gw[gi];
        if (ups)
            hipLaunchKernelGGL(wgrad_reduce_kernel<true>, rgrid, dim3(256), shb, st, pg, gwi, Cin, Cout, kH, KK, padH, kp,
                               p.splits, scale, ci_t, sstride);
        else
            hipLaunchKernelGGL(wgrad_reduce_kernel<false>, rgrid, dim3(256), shb, st, pg, gwi, Cin, Cout, kH, KK, padH,
                               0, p.splits, scale, ci_t, sstride);
        CG_LAUNCH_CHECK();
        if (gb && gb[gi]) {
            hipLaunchKernelGGL(bias_part_reduce_kernel, dim3(cg::cdiv(Cout, 32)), dim3(256), 0, st,
                               (const float*)a.bias_part + (long)gi * g.nphase * Cout, gb[gi], p.splits, g.nphase,
                               (long)ZP * Cout, Cout, scale);
            CG_LAUNCH_CHECK();
        }
    }
    return 0;
}

}  // namespace

int cg_conv2d_wgrad(void* stream, const float* x, const float* dy, float* gw, float* gb, int N, int Hp, int Wp, int Cin,
                    int Cout, int kH, int kW, int padH, int padW, int ups, float scale, void* ws, size_t ws_bytes) {
    CG_REQUIRE(x && dy && gw, "cg_conv2d_wgrad: null pointer");
    return cg_conv2d_wgrad_grouped(stream, 1, &x, &dy, &gw, &gb, N, Hp, Wp, Cin, Cout, kH, kW, padH, padW, ups, scale, ws,
                                   ws_bytes);
}

// input channels per pack workgroup: the largest power of two with ci_t*KK <= 448 floats per output channel (57 KB of LDS)
static int pack_ci_tile(int Cin, int KK) {
    int t = 1;
    while (t * 2 * KK <= 448 && t < Cin) t *= 2;
    return t;
}

int cg_pack_conv_weight(void* stream, const float* w, float* wf, float* wb, int Cout, int Cin, int kH, int kW) {
    CG_REQUIRE(w && (wf || wb), "cg_pack_conv_weight: null pointer");
    CG_REQUIRE(Cout > 0 && Cin > 0 && kH > 0 && kW > 0, "cg_pack_conv_weight: bad dims");
    const int KK = kH * kW;
    hipLaunchKernelGGL(pack_weight_kernel, dim3(cg::cdiv(Cin, 32), cg::cdiv(Cout, 32), KK), dim3(256), 0, cg::S(stream), w,
                       wf, wb, Cout, Cin, KK, 0);
    CG_LAUNCH_CHECK();
    return cg::wino3_note_pack(cg::S(stream), 1, &w, &wf, &wb, &Cout, &Cin, &kH, &kW, nullptr);
}

// nn.View(C*H*W) -> nn.Linear(C*H*W -> Cout) on an NHWC map (models.lua:696-697, 849-850) without materialising the NCHW
// view: the canonical [Cout][C*H*W] weight IS the canonical weight of a convolution C -> Cout with an H x W kernel, no
// padding, on the H x W map (output 1 x 1).  wf (forward / weight-gradient view) is cg_pack_conv_weight's;
// wbT[co][(h*W+w)*C + c] = w[co][c][h][w] is the [K = Cout][N = H*W*C] operand of the data gradient, run as a linear layer:
//   cg_conv2d_forward(dy [N][Cout], wbT, NULL, dx [N][H*W*C] = NHWC, N, 1, 1, Cout, H*W*C, 1, 1, 0, 0, 0)
int cg_pack_conv_weight_map(void* stream, const float* w, float* wf, float* wbT, int Cout, int Cin, int kH, int kW) {
    CG_REQUIRE(w && (wf || wbT), "cg_pack_conv_weight_map: null pointer");
    CG_REQUIRE(Cout > 0 && Cin > 0 && kH > 0 && kW > 0, "cg_pack_conv_weight_map: bad dims");
    const int KK = kH * kW;
    hipLaunchKernelGGL(pack_weight_kernel, dim3(cg::cdiv(Cin, 32), cg::cdiv(Cout, 32), KK), dim3(256), 0, cg::S(stream), w,
                       wf, wbT, Cout, Cin, KK, 1);
    CG_LAUNCH_CHECK();
    return 0;
}

int cg_pack_conv_weight_batch(void* stream, int n, const float* const* w_canonical, float* const* wf, float* const* wb,
                              const int* Cout, const int* Cin, const int* kH, const int* kW, const int* wb_map) {
    CG_REQUIRE(n >= 0 && w_canonical && wf && wb && Cout && Cin && kH && kW, "cg_pack_conv_weight_batch: null pointer");
    for (int i0 = 0; i0 < n; i0 += kPackBatch) {
        PackBatch b;
        memset(&b, 0, sizeof(b));
        b.n = std::min(kPackBatch, n - i0);
        int blocks = 0;
        for (int e = 0; e < b.n; ++e) {
            const int i = i0 + e;
            CG_REQUIRE(w_canonical[i] && (wf[i] || wb[i]) && Cout[i] > 0 && Cin[i] > 0 && kH[i] > 0 && kW[i] > 0,
                       "cg_pack_conv_weight_batch: bad layer %d", i);
            b.w[e] = w_canonical[i]; b.wf[e] = wf[i]; b.wb[e] = wb[i];
            b.Cout[e] = Cout[i]; b.Cin[e] = Cin[i]; b.KK[e] = kH[i] * kW[i];
            if (wb_map && wb_map[i]) b.map_mask |= 1ull << e;
            b.first[e] = blocks;
            blocks += cg::cdiv(Cin[i], 32) * cg::cdiv(Cout[i], 32) * b.KK[e];
        }
        b.first[b.n] = blocks;
        hipLaunchKernelGGL(pack_weight_batch_kernel, dim3(blocks), dim3(256), 0, cg::S(stream), b);
        CG_LAUNCH_CHECK();
    }
    return cg::wino3_note_pack(cg::S(stream), n, w_canonical, wf, wb, Cout, Cin, kH, kW, wb_map);
}

size_t cg_pack_conv_weight_ups2_floats(int Cout, int Cin, int k, int pad) {
    if (Cout <= 0 || Cin <= 0 || k <= 0 || (k & 1) == 0 || pad != (k - 1) / 2) return 0;
    const int kp = phase_kp(k, pad);
    return (size_t)4 * kp * kp * Cin * Cout;
}

int cg_pack_conv_weight_ups2(void* stream, const float* w, float* wf_ph, float* wb_ph, int Cout, int Cin, int k, int pad) {
    CG_REQUIRE(w && (wf_ph || wb_ph), "cg_pack_conv_weight_ups2: null pointer");
    CG_REQUIRE(Cout > 0 && Cin > 0 && k > 0 && (k & 1) == 1 && pad == (k - 1) / 2,
               "cg_pack_conv_weight_ups2: needs an odd kernel with pad=(k-1)/2");
    const int kp = phase_kp(k, pad);
    CG_REQUIRE(k * k <= 448, "cg_pack_conv_weight_ups2: kernel too large (%d taps)", k * k);
    const dim3 fgrid(cg::cdiv(Cin, 32), cg::cdiv(Cout, 32));
    if (k == 3 || k == 5) {
        if (k == 3)
            hipLaunchKernelGGL(pack_weight_ups2_fast_kernel<3>, fgrid, dim3(256), 0, cg::S(stream), w, wf_ph, wb_ph, Cout, Cin);
        else
            hipLaunchKernelGGL(pack_weight_ups2_fast_kernel<5>, fgrid, dim3(256), 0, cg::S(stream), w, wf_ph, wb_ph, Cout, Cin);
        CG_LAUNCH_CHECK();
        return 0;
    }
    const int ci_t = pack_ci_tile(Cin, k * k);
    hipLaunchKernelGGL(pack_weight_ups2_kernel, dim3(cg::cdiv(Cin, ci_t), cg::cdiv(Cout, 32)), dim3(256),
                       (size_t)32 * (ci_t * k * k + 1) * sizeof(float), cg::S(stream), w, wf_ph, wb_ph, Cout, Cin, k, pad, kp,
                       ci_t);
    CG_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"

void cg::wgrad_discard_all() {
    std::lock_guard<std::mutex> lk(g_red_mu);
    for (auto& kv : g_red_queue) kv.second.clear();
}
