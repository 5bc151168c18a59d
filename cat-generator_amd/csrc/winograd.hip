// Winograd F(2x2,3x3) for nn.SpatialUpSamplingNearest(2) -> 5x5 convolution (models.lua:217-218, G's
// 256 -> 128 layer: 60 % of the generator's FLOPs).
//
// After the phase folding of gemm.hip, output phase (a,b) of that layer is a plain 3x3 / pad-1 convolution of the
// LOW-RESOLUTION input with the phase-summed kernel g_p (both phases of a 5-tap axis reach rows i-1..i+1).  All
// four phases therefore share their 4x4 input tiles, and each is computed as
//     Y_p = A^T [ (G g_p G^T) .* (B^T d B) ] A            (Lavin & Gray, F(2x2,3x3): 16 multiplies per 4 outputs
// instead of 36, i.e. 2.25x fewer MFMA FLOPs on top of the 2.78x of the phase folding)
// as 16 GEMMs  M_xi[tile][co] = sum_ci V_xi[tile][ci] U_p,xi[ci][co]  whose output transform happens in registers:
// one workgroup walks xi = 0..15 for its (tile block, co block, phase), folding every finished M_xi into the four
// output accumulators with the +-1/0 coefficients of A^T (x) A^T.  Nothing Winograd-domain but V is ever written.
//
// The data gradient w.r.t. the low-res input is the same machinery on dy: one 3x3 convolution with 4*Cout input
// planes (the four phase sub-lattices of dy) and flipped kernels.  The weight gradient is taken in the Winograd
// domain (dU_xi = V_xi^T (A dY A^T)_xi, 16 plain TN GEMMs through cg_conv2d_wgrad) and brought back by G^T . G.
//
// Numerics: exact-fp32 MFMA as everywhere; the transforms add a few ulp (entries 0, +-1, +-1/2), the parity
// tests hold the layer to the same 2e-5*sqrt(K/1024) bound as the direct path.
#include "common.h"
#include <algorithm>
#include <type_traits>
#include <stdlib.h>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float wf4c(const float4& v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : (i == 2 ? v.z : v.w)); }
__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4sub(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }

// ---------------------------------------------------------------------------
// V[xi][tile][c] = (B^T d B)[xi] of the 4x4 patch around tile (n, ti, tj): rows 2ti-1 .. 2ti+2.
// MODE 0: d from a plain NHWC tensor [N][Hl][Wl][C].
// MODE 1: d from the phase sub-lattices of dy [N][2Hl][2Wl][Cs]: channel c = p*Cs + co reads dy[2i+a][2j+b][co].
// One thread per (tile, channel quad): 16 float4 loads, 16 float4 stores, coalesced along the channels.
// ---------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256) void wino_input_transform_kernel(const float* __restrict__ src, float* __restrict__ V,
                                                                   int N, int Hl, int Wl, int C, int Cs) {
    const int cq_n = C >> 2;
    const int tH = Hl >> 1, tW = Wl >> 1;
    const long T = (long)N * tH * tW;
    const long total = T * cq_n;
    for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += gridDim.x * 256L) {
        const int cq = (int)(idx % cq_n);
        const long tile = idx / cq_n;
        const int tj = (int)(tile % tW);
        const int ti = (int)((tile / tW) % tH);
        const long n = tile / ((long)tW * tH);
        int pa = 0, pb = 0, cs = cq * 4;
        if (MODE == 1) {
            const int p = cs / Cs;
            cs -= p * Cs;
            pa = p >> 1; pb = p & 1;
        }
        float4 d[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int iy = 2 * ti - 1 + r, ix = 2 * tj - 1 + c;
                const bool ok = iy >= 0 && iy < Hl && ix >= 0 && ix < Wl;
                const int cy = min(max(iy, 0), Hl - 1), cx = min(max(ix, 0), Wl - 1);
                const float* ptr = MODE == 0 ? src + ((n * Hl + cy) * (long)Wl + cx) * C + cs
                                             : src + ((n * 2 * Hl + 2 * cy + pa) * (long)(2 * Wl) + 2 * cx + pb) * Cs + cs;
                const float4 q = ld4(ptr);
                d[r][c] = ok ? q : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        float4 t[4][4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {  // B^T d
            t[0][c] = f4sub(d[0][c], d[2][c]);
            t[1][c] = f4add(d[1][c], d[2][c]);
            t[2][c] = f4sub(d[2][c], d[1][c]);
            t[3][c] = f4sub(d[1][c], d[3][c]);
        }
        float* out = V + tile * C + cq * 4;
        const long xs = T * C;
#pragma unroll
        for (int r = 0; r < 4; ++r) {  // (.) B
            *reinterpret_cast<float4*>(out + (r * 4 + 0) * xs) = f4sub(t[r][0], t[r][2]);
            *reinterpret_cast<float4*>(out + (r * 4 + 1) * xs) = f4add(t[r][1], t[r][2]);
            *reinterpret_cast<float4*>(out + (r * 4 + 2) * xs) = f4sub(t[r][2], t[r][1]);
            *reinterpret_cast<float4*>(out + (r * 4 + 3) * xs) = f4sub(t[r][1], t[r][3]);
        }
    }
}

// ---------------------------------------------------------------------------
// U = G g G^T of the phase kernels.
// FWD: src = wf_ph [p][tp][ci][co]  ->  U[p][xi][ci][co]
// BWD: src = wb_ph [p][tp][co][ci]  ->  U[xi][p*Cout+co][ci] of the FLIPPED kernel (data gradient operand)
// One thread per (p, row, column quad) of the [rows][cols] plane (rows x cols = Cin x Cout or Cout x Cin).
// ---------------------------------------------------------------------------
template <bool BWD>
__global__ __launch_bounds__(256) void wino_filter_transform_kernel(const float* __restrict__ src, float* __restrict__ U,
                                                                    int rows, int cols) {
    const int cqn = cols >> 2;
    const long plane = (long)rows * cols;
    const long total = 4L * rows * cqn;
    for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += gridDim.x * 256L) {
        const int cq = (int)(idx % cqn);
        const int row = (int)((idx / cqn) % rows);
        const int p = (int)(idx / ((long)cqn * rows));
        float4 g[3][3];
#pragma unroll
        for (int ry = 0; ry < 3; ++ry)
#pragma unroll
            for (int rx = 0; rx < 3; ++rx) {
                const int tp = BWD ? (2 - ry) * 3 + (2 - rx) : ry * 3 + rx;
                g[ry][rx] = ld4(src + ((long)p * 9 + tp) * plane + (long)row * cols + cq * 4);
            }
        float4 t[4][3];  // G g
#pragma unroll
        for (int rx = 0; rx < 3; ++rx) {
            const float4 s = f4add(g[0][rx], g[2][rx]);
            t[0][rx] = g[0][rx];
            t[1][rx] = make_float4(0.5f * (s.x + g[1][rx].x), 0.5f * (s.y + g[1][rx].y), 0.5f * (s.z + g[1][rx].z), 0.5f * (s.w + g[1][rx].w));
            t[2][rx] = make_float4(0.5f * (s.x - g[1][rx].x), 0.5f * (s.y - g[1][rx].y), 0.5f * (s.z - g[1][rx].z), 0.5f * (s.w - g[1][rx].w));
            t[3][rx] = g[2][rx];
        }
#pragma unroll
        for (int xy = 0; xy < 4; ++xy) {
            const float4 s = f4add(t[xy][0], t[xy][2]);
            float4 o[4];
            o[0] = t[xy][0];
            o[1] = make_float4(0.5f * (s.x + t[xy][1].x), 0.5f * (s.y + t[xy][1].y), 0.5f * (s.z + t[xy][1].z), 0.5f * (s.w + t[xy][1].w));
            o[2] = make_float4(0.5f * (s.x - t[xy][1].x), 0.5f * (s.y - t[xy][1].y), 0.5f * (s.z - t[xy][1].z), 0.5f * (s.w - t[xy][1].w));
            o[3] = t[xy][2];
#pragma unroll
            for (int xx = 0; xx < 4; ++xx) {
                const int xi = xy * 4 + xx;
                float* dst = BWD ? U + ((long)xi * 4 * rows + (long)p * rows + row) * cols + cq * 4
                                 : U + (((long)p * 16 + xi) * rows + row) * cols + cq * 4;
                *reinterpret_cast<float4*>(dst) = o[xx];
            }
        }
    }
}

// ---------------------------------------------------------------------------
// F(2x2,2x2) for nn.SpatialUpSamplingNearest(2) -> 3x3 convolution (models.lua:211-212; round 4).  After the phase folding, output
// phase (a,b) is a 2x2-tap convolution of the low-res input whose window starts at row -1 + a, column -1 + b: per 2x2 tile of low-res
// outputs and phase, 16 multiplies per channel pair; Winograd's minimal form needs 9 (y0 = (d0-d1) g0 + d1 (g0+g1), y1 = d1 (g0+g1) +
// (d2-d1) g1 per axis: B^T = [1 -1 0; 0 1 0; 0 -1 1], G = [1 0; 1 1; 0 1], A^T = [1 1 0; 0 1 1] - all 0 / +-1, exact up to the
// additions).  The four phases' 3x3 windows are the four 3x3 corners of ONE 4x4 patch (rows 2ti-1 .. 2ti+2), so the input transform
// reads the patch once and writes 4 x 9 planes:  V[(p*9 + xi)][tile][c].  The GEMMs and the output transform are wino_gemm_g_kernel's
// with 9 positions and V taken per phase.
// One thread per (tile, channel quad): 16 float4 loads, 36 float4 stores.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void wino22_input_transform_kernel(const float* __restrict__ src, float* __restrict__ V, int N, int Hl,
                                                                     int Wl, int C) {
    const int cq_n = C >> 2;
    const int tH = Hl >> 1, tW = Wl >> 1;
    const long T = (long)N * tH * tW;
    const long total = T * cq_n;
    for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += gridDim.x * 256L) {
        const int cq = (int)(idx % cq_n);
        const long tile = idx / cq_n;
        const int tj = (int)(tile % tW);
        const int ti = (int)((tile / tW) % tH);
        const long n = tile / ((long)tW * tH);
        float4 d[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int iy = 2 * ti - 1 + r, ix = 2 * tj - 1 + c;
                const bool ok = iy >= 0 && iy < Hl && ix >= 0 && ix < Wl;
                const int cy = min(max(iy, 0), Hl - 1), cx = min(max(ix, 0), Wl - 1);
                const float4 q = ld4(src + ((n * Hl + cy) * (long)Wl + cx) * C + cq * 4);
                d[r][c] = ok ? q : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        float* out = V + tile * C + cq * 4;
        const long xs = T * C;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            float4 t[3][4];   // B^T applied to rows a .. a+2, all four columns
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                t[0][c] = f4sub(d[a][c], d[a + 1][c]);
                t[1][c] = d[a + 1][c];
                t[2][c] = f4sub(d[a + 2][c], d[a + 1][c]);
            }
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const long pbase = (long)(a * 2 + b) * 9;
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    *reinterpret_cast<float4*>(out + (pbase + i * 3 + 0) * xs) = f4sub(t[i][b], t[i][b + 1]);
                    *reinterpret_cast<float4*>(out + (pbase + i * 3 + 1) * xs) = t[i][b + 1];
                    *reinterpret_cast<float4*>(out + (pbase + i * 3 + 2) * xs) = f4sub(t[i][b + 2], t[i][b + 1]);
                }
            }
        }
    }
}

// Data gradient of the same layers.  dx_lo[u][v] = sum_p sum_r dy_p[u - a + ry][v - b + rx] . g_p[1-ry][1-rx] over the phase
// sub-lattices dy_p[i][j] = dy[2i+a][2j+b]: per phase a 2x2-tap correlation with the flipped kernel whose window starts at row u - a,
// the four phases summed - one F(2x2,2x2) problem with the phases side by side along K (K = 4 Cout).
// V[xi][tile][p*Cs + co] = B^T d B of the 3x3 patch (rows 2ti - a .. 2ti - a + 2 of sub-lattice p).  One thread per (tile, channel
// quad): 9 float4 loads, 9 float4 stores.
__global__ __launch_bounds__(256) void wino22_dy_input_transform_kernel(const float* __restrict__ dy, float* __restrict__ V, int N, int Hl,
                                                                        int Wl, int Cs) {
    const int C = 4 * Cs, cq_n = C >> 2;
    const int tH = Hl >> 1, tW = Wl >> 1;
    const long T = (long)N * tH * tW;
    const long total = T * cq_n;
    for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += gridDim.x * 256L) {
        const int cq = (int)(idx % cq_n);
        const long tile = idx / cq_n;
        const int tj = (int)(tile % tW);
        const int ti = (int)((tile / tW) % tH);
        const long n = tile / ((long)tW * tH);
        const int p = (cq * 4) / Cs, cs = cq * 4 - p * Cs;
        const int pa = p >> 1, pb = p & 1;
        float4 d[3][3];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int iy = 2 * ti - pa + r, ix = 2 * tj - pb + c;
                const bool ok = iy >= 0 && iy < Hl && ix >= 0 && ix < Wl;
                const int cy = min(max(iy, 0), Hl - 1), cx = min(max(ix, 0), Wl - 1);
                const float4 q = ld4(dy + ((n * 2 * Hl + 2 * cy + pa) * (long)(2 * Wl) + 2 * cx + pb) * Cs + cs);
                d[r][c] = ok ? q : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        float4 t[3][3];
#pragma unroll
        for (int c = 0; c < 3; ++c) { t[0][c] = f4sub(d[0][c], d[1][c]); t[1][c] = d[1][c]; t[2][c] = f4sub(d[2][c], d[1][c]); }
        float* out = V + tile * C + cq * 4;
        const long xs = T * C;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            *reinterpret_cast<float4*>(out + (i * 3 + 0) * xs) = f4sub(t[i][0], t[i][1]);
            *reinterpret_cast<float4*>(out + (i * 3 + 1) * xs) = t[i][1];
            *reinterpret_cast<float4*>(out + (i * 3 + 2) * xs) = f4sub(t[i][2], t[i][1]);
        }
    }
}

// U = G g_p G^T of the 2x2 phase kernels.
// FWD: src = wf_ph [p][tp = ry*2 + rx][ci][co] -> U[p][xi = i*3 + j][ci][co].
// BWD: src = wb_ph [p][tp][co][ci] -> U[xi][p*Cout + co][ci] of the FLIPPED kernel (data-gradient operand: the four phases side by side along K).
template <bool BWD>
__global__ __launch_bounds__(256) void wino22_filter_transform_kernel(const float* __restrict__ src, float* __restrict__ U, int rows,
                                                                      int cols) {
    const int cqn = cols >> 2;
    const long plane = (long)rows * cols;
    const long total = 4L * rows * cqn;
    for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += gridDim.x * 256L) {
        const int cq = (int)(idx % cqn);
        const int row = (int)((idx / cqn) % rows);
        const int p = (int)(idx / ((long)cqn * rows));
        float4 g[2][2];
#pragma unroll
        for (int ry = 0; ry < 2; ++ry)
#pragma unroll
            for (int rx = 0; rx < 2; ++rx) {
                const int tp = BWD ? (1 - ry) * 2 + (1 - rx) : ry * 2 + rx;
                g[ry][rx] = ld4(src + ((long)p * 4 + tp) * plane + (long)row * cols + cq * 4);
            }
        float4 t[3][2];   // G g
#pragma unroll
        for (int rx = 0; rx < 2; ++rx) { t[0][rx] = g[0][rx]; t[1][rx] = f4add(g[0][rx], g[1][rx]); t[2][rx] = g[1][rx]; }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            float* dst = BWD ? U + ((long)(i * 3) * 4 * rows + (long)p * rows + row) * cols + cq * 4
                             : U + (((long)p * 9 + i * 3) * rows + row) * cols + cq * 4;
            const long xstep = BWD ? 4 * plane : plane;   // distance between neighbouring positions xi
            *reinterpret_cast<float4*>(dst) = t[i][0];
            *reinterpret_cast<float4*>(dst + xstep) = f4add(t[i][0], t[i][1]);
            *reinterpret_cast<float4*>(dst + 2 * xstep) = t[i][1];
        }
    }
}

// ---------------------------------------------------------------------------
// The 16 GEMMs + output transform.  Block tile 64 tiles x 128 columns, K step 16, 4 waves of 32 x 64.
// grid = (T/64 * Nc/128, 1, P).  LDS: A [k][m] XOR-swizzled, B [k][n], double buffered (same scheme as gemm.hip).
// ---------------------------------------------------------------------------
struct WinoArgs {
    const float* V;      // [16][T][K]
    const float* U;      // [P][16][K][Nc]
    const float* bias;   // [Nc] or null
    float* y;            // [N][Ho][Wo][Nc]
    int T, K, Nc;
    int tH, tW;          // tiles per image
    int so;              // 2: output pixel (2i+a, 2j+b) of phase (a,b) = blockIdx.z; 1: pixel (i, j)
    int Ho, Wo;
    float* stats;        // != null: column sums of y and y^2 per (phase, tile block, wave row): [rows][2][Nc] (see gemm.hip)
    int xcd;             // XCD-aware placement of the workgroups that share a V block (CG_XCD_SWIZZLE)
    int vpp;             // 1: V holds NPOS planes PER PHASE ([P][NPOS][T][K], F(2x2,2x2)); 0: one set shared by the phases ([NPOS][T][K])
    int kz;              // > 0 (wino_gemm_g_kernel, so == 1): blockIdx.z is a K slice of kz rows, its partial result goes to y + z * T*4*Nc
    int lg_tw, lg_th;    // log2(tW), log2(tH) when both are powers of two and the output is below 2 GB (wino_gemm_g_kernel's lean epilogue), else -1
};

// The register-staged form (8 waves, wave tile 32 x 32; BK = K step): the fallback of wino_gemm_g_kernel below for tensors whose
// planes a buffer descriptor cannot span, and the reference the LDS-direct kernels are tested against.  (The 4-wave and k-quad
// instances of rounds 2-4 never won an A/B and were removed in round 5.)
template <int NW, int BK>
__global__ __launch_bounds__(64 * NW, 4) void wino_gemm_kernel(WinoArgs a) {
    static_assert(NW == 8, "8 waves per workgroup");
    constexpr int BM = 64, BN = 128, NI = 8 / NW, NT = 64 * NW;
    constexpr int KV = BK / 4;                 // float4 per A row per K tile
    constexpr int BQ = BK * 32 / NT;           // B float4 per thread (1 or 2), NT/32 rows apart
    static_assert(BM * KV <= NT && (BQ == 1 || BQ == 2), "staging layout");
    constexpr int A_TILE = BK * BM, B_TILE = BK * BN;
    __shared__ __attribute__((aligned(16))) float smem[2 * A_TILE + 2 * B_TILE];
    float* As = smem;
    float* Bs = smem + 2 * A_TILE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wm0 = (wave & 1) * 32, wn0 = (wave >> 1) * (32 * NI);
    const int ntn = a.Nc / BN;
    // The ntn * P workgroups that read the same 64-tile block of V (all column blocks, all phases) are placed on ONE XCD in
    // consecutive dispatch slots, so that V comes out of that XCD's L2 after the first of them has fetched it (workgroup L of
    // the dispatch order runs on XCD L % 8).  Needs a multiple of 8 tile blocks; else the plain (x, z) order.
    int tn = blockIdx.x % ntn, tm = blockIdx.x / ntn, phase = blockIdx.z;
    const int tiles_m = gridDim.x / ntn;
    if (a.xcd && (tiles_m & 7) == 0) {
        const int share = ntn * (int)gridDim.z;
        const int L = (int)blockIdx.z * (int)gridDim.x + (int)blockIdx.x;
        const int slot = L >> 3, member = slot % share;
        tm = (slot / share) * 8 + (L & 7);
        phase = member / ntn;
        tn = member - phase * ntn;
    }
    const int m0 = tm * BM, n0 = tn * BN;
    const int KT = CG_PROBE_HALF(4, a.K / BK);

    // staging: A one float4 per thread (row a_r, k quad a_kv); B two float4 per thread
    const int a_kv = tid % KV, a_r = (tid / KV) % BM;
    const bool a_thr = tid < BM * KV;             // 8 waves, K step 16: the upper four waves stage only B
    const int a_row = min(m0 + a_r, a.T - 1);
    const int b_kr = tid >> 5, b_nv = tid & 31;   // k rows b_kr (and b_kr + NT/32 when BQ == 2)
    const float* Ap = a.V + (long)a_row * a.K + 4 * a_kv;
    const float* Bp = a.U + (long)phase * 16 * a.K * a.Nc + (long)b_kr * a.Nc + n0 + 4 * b_nv;
    const long a_xi = (long)a.T * a.K;            // V stride between xi
    const long b_half = (long)(NT / 32) * a.Nc;   // second B float4: k row + NT/32
    float4 areg, breg0, breg1;
    // running operand pointers: U is contiguous over (xi, k), V jumps to the next xi plane after the last K tile
    const float* Ac = Ap;
    const float* Bc = Bp;
    const long b_step = (long)BK * a.Nc;
    const long a_jump = a_xi - a.K + BK;
    auto load_next = [&](bool last_kt) {
        Ac += last_kt ? a_jump : (long)BK;
        Bc += b_step;
        if (BM * KV == NT || a_thr) areg = ld4(Ac);
        breg0 = ld4(Bc);
        if (BQ == 2) breg1 = ld4(Bc + b_half);
    };
    const int a_so = a_r ^ (BK == 32 ? ((a_kv & 7) << 2) : ((a_kv & 3) << 3));   // see a_swizzle in gemm.hip
    auto store_tile = [&](int buf) {
        float* A = As + buf * A_TILE;
        float* B = Bs + buf * B_TILE;
        if (BM * KV == NT || a_thr) {
            A[(4 * a_kv + 0) * BM + a_so] = areg.x;
            A[(4 * a_kv + 1) * BM + a_so] = areg.y;
            A[(4 * a_kv + 2) * BM + a_so] = areg.z;
            A[(4 * a_kv + 3) * BM + a_so] = areg.w;
        }
        *reinterpret_cast<float4*>(B + b_kr * BN + 4 * b_nv) = breg0;
        if (BQ == 2) *reinterpret_cast<float4*>(B + (b_kr + NT / 32) * BN + 4 * b_nv) = breg1;
    };

    f32x16 accM[NI], accY[4][NI];
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            accM[j][r] = 0.f;
#pragma unroll
            for (int o = 0; o < 4; ++o) accY[o][j][r] = 0.f;
        }

    areg = ld4(Ac);
    breg0 = ld4(Bc);
    breg1 = ld4(Bc + b_half);
    store_tile(0);
    __syncthreads();
    // one K tile from LDS buffer BUF (compile-time, so every ds_read offset is an immediate)
    auto k_tile = [&](auto bufc, bool more, bool last_kt) {
        constexpr int BUF = decltype(bufc)::value;
        if (more) load_next(last_kt);
        const float* A = As + BUF * A_TILE + wm0;
        const float* B = Bs + BUF * B_TILE + wn0 + l31;
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            const float av = A[(kk + h) * BM + (l31 ^ (BK == 32 ? (((kk >> 2) & 7) << 2) : (((kk >> 2) & 3) << 3)))];
#pragma unroll
            for (int j = 0; j < NI; ++j)
                accM[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, B[(kk + h) * BN + j * 32], accM[j], 0, 0, 0);
        }
        if (more) store_tile(BUF ^ 1);
        __syncthreads();
    };
    for (int xi = 0; xi < 16; ++xi) {
        for (int kt = 0; kt < KT; kt += 2) {   // KT is even (host checks K % 32 == 0)
            k_tile(std::integral_constant<int, 0>{}, true, false);
            const bool last = kt + 2 == KT;
            k_tile(std::integral_constant<int, 1>{}, !(last && xi == 15), last);
        }
        // M_xi complete: fold it into the four outputs, A^T = [1 1 1 0; 0 1 -1 -1]
        const int xy = xi >> 2, xx = xi & 3;
        const float cy0 = xy < 3 ? 1.f : 0.f, cy1 = xy == 0 ? 0.f : (xy == 1 ? 1.f : -1.f);
        const float cx0 = xx < 3 ? 1.f : 0.f, cx1 = xx == 0 ? 0.f : (xx == 1 ? 1.f : -1.f);
        const float c00 = cy0 * cx0, c01 = cy0 * cx1, c10 = cy1 * cx0, c11 = cy1 * cx1;
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float m = accM[j][r];
                accY[0][j][r] += c00 * m; accY[1][j][r] += c01 * m;
                accY[2][j][r] += c10 * m; accY[3][j][r] += c11 * m;
                accM[j][r] = 0.f;
            }
    }

    // ---- epilogue: D[i][j], i = (r&3) + 8*(r>>2) + 4*h (tile), j = l31 (column)
    const int pa = a.so == 2 ? (phase >> 1) : 0, pb = a.so == 2 ? (phase & 1) : 0;
    float st1[NI], st2[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) { st1[j] = 0.f; st2[j] = 0.f; }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm0 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (m >= a.T) continue;
        const int tj = m % a.tW;
        const int ti = (m / a.tW) % a.tH;
        const long n = m / (a.tW * a.tH);
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            const int oy = (2 * ti + (o >> 1)) * a.so + pa, ox = (2 * tj + (o & 1)) * a.so + pb;
            float* row = a.y + ((n * a.Ho + oy) * (long)a.Wo + ox) * a.Nc + n0 + wn0 + l31;
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const float v = accY[o][j][r] + (a.bias ? a.bias[n0 + wn0 + j * 32 + l31] : 0.f);
                row[j * 32] = v;
                if (a.stats) { st1[j] += v; st2[j] += v * v; }
            }
        }
    }
    if (a.stats) {
        const int srow = (phase * (int)((a.T + BM - 1) / BM) + tm) * 2 + (wave & 1);
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const float t1 = st1[j] + __shfl_xor(st1[j], 32, 64), t2 = st2[j] + __shfl_xor(st2[j], 32, 64);
            if (h == 0) {
                a.stats[((long)srow * 2 + 0) * a.Nc + n0 + wn0 + j * 32 + l31] = t1;
                a.stats[((long)srow * 2 + 1) * a.Nc + n0 + wn0 + j * 32 + l31] = t2;
            }
        }
    }
}

// ---------------------------------------------------------------------------
// The same 16 GEMMs + output transform with LDS-direct loads (buffer_load_dwordx4 ... lds; igemm_nng_kernel in gemm.hip has
// the layouts and the reasoning): 8 waves, wave tile 32 x 32.  A = 64 rows of V (a tile's BK consecutive k = 8 or 4 quads,
// row-major, quad index XOR-ed with swz(row) so that the ds_read_b128 fragment reads are conflict-free), B = BK rows of U
// ([k][n], ds_read_b32 fragments).  A wave instruction fills 1 KB: BK 32 -> every wave brings 8 rows of V and 2 x 2 rows
// of U per K tile; BK 16 -> waves 0..3 bring 16 rows of V each, every wave 2 rows of U.  Rows past the last tile are
// clamped to it (their results are never stored), as in wino_gemm_kernel.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void wino_glds16(__amdgpu_buffer_rsrc_t r, float* lds, unsigned voff, int soff) {
#if defined(__HIP_DEVICE_COMPILE__)   // device pass only: the host pass would drop the kernel's launch stub (see gemm.hip)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 16, voff, soff, 0, 0);
#endif
}

template <int BK, int NPOS = 16>
__global__ __launch_bounds__(512, 4) void wino_gemm_g_kernel(WinoArgs a) {
    constexpr int BM = 64, BN = 128;
    constexpr int KV = BK / 4, RPW = 64 / KV, G2 = KV / 2;
    constexpr int A_TILE = BK * BM, B_TILE = BK * BN;
    constexpr int AW = BM / RPW;          // waves that bring A rows (8 at BK 32, 4 at BK 16)
    constexpr int BQ = BK / 16;           // B instructions per wave per K tile (16 rows per pass of the 8 waves)
    __shared__ __attribute__((aligned(16))) float As0[A_TILE];
    __shared__ __attribute__((aligned(16))) float As1[A_TILE];
    __shared__ __attribute__((aligned(16))) float Bs0[B_TILE];
    __shared__ __attribute__((aligned(16))) float Bs1[B_TILE];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int wm0 = (wave & 1) * 32, wn0 = (wave >> 1) * 32;
    const int ntn = a.Nc / BN;
    int tn = blockIdx.x % ntn, tm = blockIdx.x / ntn, phase = blockIdx.z;
    const int tiles_m = gridDim.x / ntn;
    if (a.xcd && (tiles_m & 7) == 0) {   // the workgroups sharing a V block on one XCD, see wino_gemm_kernel
        const int share = ntn * (int)gridDim.z;
        const int L = (int)blockIdx.z * (int)gridDim.x + (int)blockIdx.x;
        const int slot = L >> 3, member = slot % share;
        tm = (slot / share) * 8 + (L & 7);
        phase = member / ntn;
        tn = member - phase * ntn;
    }
    const int m0 = tm * BM, n0 = tn * BN;
    const int Kext = a.kz ? a.kz : a.K;                           // K rows this workgroup walks, from row koff
    const int koff = a.kz ? phase * a.kz : 0;
    float* const yout = a.y + (a.kz ? (long)phase * a.T * 4 * a.Nc : 0L);
    if (a.kz) phase = 0;
    const int KT = CG_PROBE_HALF(4, Kext / BK);

    // ---- staging: lane -> (row, LDS quad position) of A, (k row, column quad) of B
    const int a_row_l = (wave % AW) * RPW + lane / KV;            // row of the tile block this lane brings (waves >= AW: unused)
    const int a_pos = lane % KV;
    const int a_sw = KV == 8 ? ((a_row_l >> 1) & 7) : ((a_row_l >> 2) & 3);
    const int a_row = min(m0 + a_row_l, a.T - 1);
    const unsigned a_voff = (unsigned)(a_row * a.K + 4 * (a_pos ^ a_sw)) * 4u;
    const int b_kr = wave * 2 + (lane >> 5);                      // + 16 q
    const unsigned b_voff = (unsigned)(b_kr * a.Nc + n0 + 4 * (lane & 31)) * 4u;
    const float* Up = a.U + (long)phase * NPOS * a.K * a.Nc;
    __amdgpu_buffer_rsrc_t rsu = __builtin_amdgcn_make_buffer_rsrc((void*)Up, 0, 0x7fffffff, 0x00020000);
    const long a_plane = (long)a.T * a.K;                         // V stride between xi (one plane < 2 GB: host check)
    const float* Vp = a.V + (a.vpp ? (long)phase * NPOS * a_plane : 0L);

    // the next tile to request: (xi, k0), advanced by every dma_tile
    int nxi = 0, nk0 = 0;
    auto dma_tile = [&](auto bufc) {
        constexpr int buf = decltype(bufc)::value;
        float* A = buf ? As1 : As0;
        float* B = buf ? Bs1 : Bs0;
        if (AW == 8 || wave < AW) {
            __amdgpu_buffer_rsrc_t rsv = __builtin_amdgcn_make_buffer_rsrc((void*)(Vp + nxi * a_plane), 0, 0x7fffffff, 0x00020000);
            wino_glds16(rsv, A + (wave % AW) * RPW * BK, a_voff, (koff + nk0) * 4);
        }
        const int sb = (nxi * a.K + koff + nk0) * a.Nc * 4;
#pragma unroll
        for (int q = 0; q < BQ; ++q) wino_glds16(rsu, B + (16 * q + wave * 2) * BN, b_voff + (unsigned)(16 * q * a.Nc) * 4u, sb);
        nk0 += BK;
        if (nk0 >= Kext) { nk0 = 0; ++nxi; }
    };

    f32x16 accM, accY[4];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        accM[r] = 0.f;
#pragma unroll
        for (int o = 0; o < 4; ++o) accY[o][r] = 0.f;
    }

    dma_tile(std::integral_constant<int, 0>{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const int f_sw = KV == 8 ? ((l31 >> 1) & 7) : ((l31 >> 2) & 3);
    int qa_rd[G2];
#pragma unroll
    for (int gq = 0; gq < G2; ++gq) qa_rd[gq] = (wm0 + l31) * KV + ((2 * gq + h) ^ f_sw);
    const int qb_rd = 4 * h * BN + wn0 + l31;

    auto k_tile = [&](auto bufc, bool more) {
        constexpr int BUF = decltype(bufc)::value;
        if (more) dma_tile(std::integral_constant<int, BUF ^ 1>{});
        const float4* A4 = reinterpret_cast<const float4*>(BUF ? As1 : As0);
        const float* B = (BUF ? Bs1 : Bs0) + qb_rd;
#pragma unroll
        for (int gq = 0; gq < G2; ++gq) {
            const float4 af = A4[qa_rd[gq]];
#pragma unroll
            for (int sidx = 0; sidx < 4; ++sidx)
                accM = __builtin_amdgcn_mfma_f32_32x32x2f32(wf4c(af, sidx), B[(8 * gq + sidx) * BN], accM, 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };
    for (int xi = 0; xi < NPOS; ++xi) {
        for (int kt = 0; kt < KT; kt += 2) {   // KT is even (host checks K % (2 BK) == 0)
            k_tile(std::integral_constant<int, 0>{}, true);
            k_tile(std::integral_constant<int, 1>{}, !(kt + 2 == KT && xi == NPOS - 1));
        }
        // M_xi complete: fold it into the four outputs.  F(2x2,3x3): A^T = [1 1 1 0; 0 1 -1 -1];  F(2x2,2x2): A^T = [1 1 0; 0 1 1]
        const int xy = NPOS == 16 ? xi >> 2 : xi / 3, xx = NPOS == 16 ? xi & 3 : xi - 3 * (xi / 3);
        const float cy0 = NPOS == 16 ? (xy < 3 ? 1.f : 0.f) : (xy < 2 ? 1.f : 0.f);
        const float cy1 = NPOS == 16 ? (xy == 0 ? 0.f : (xy == 1 ? 1.f : -1.f)) : (xy > 0 ? 1.f : 0.f);
        const float cx0 = NPOS == 16 ? (xx < 3 ? 1.f : 0.f) : (xx < 2 ? 1.f : 0.f);
        const float cx1 = NPOS == 16 ? (xx == 0 ? 0.f : (xx == 1 ? 1.f : -1.f)) : (xx > 0 ? 1.f : 0.f);
        const float c00 = cy0 * cx0, c01 = cy0 * cx1, c10 = cy1 * cx0, c11 = cy1 * cx1;
        // the coefficients are 0 / +-1 and wave-uniform (xi is): a zero term is skipped by a scalar branch - 36 of the 64 products of
        // F(2x2,3x3) (16 of 36 for F(2x2,2x2)) are non-zero, and fp32 MFMA time is VALU time (round 5)
        if (c00 != 0.f) {
#pragma unroll
            for (int r = 0; r < 16; ++r) accY[0][r] += c00 * accM[r];
        }
        if (c01 != 0.f) {
#pragma unroll
            for (int r = 0; r < 16; ++r) accY[1][r] += c01 * accM[r];
        }
        if (c10 != 0.f) {
#pragma unroll
            for (int r = 0; r < 16; ++r) accY[2][r] += c10 * accM[r];
        }
        if (c11 != 0.f) {
#pragma unroll
            for (int r = 0; r < 16; ++r) accY[3][r] += c11 * accM[r];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) accM[r] = 0.f;
    }

    // ---- epilogue (as wino_gemm_kernel, NI = 1): D[i][j], i = (r&3) + 8*(r>>2) + 4*h (tile), j = l31 (column)
    const int pa = a.so == 2 ? (phase >> 1) : 0, pb = a.so == 2 ? (phase & 1) : 0;
    float st1 = 0.f, st2 = 0.f;
    const float bcol = a.bias ? a.bias[n0 + wn0 + l31] : 0.f;
    if (a.lg_tw >= 0) {
        // lean path (round 5): power-of-two tile grids and an output below 2 GB - the tile decode is shifts and masks, the pixel's byte
        // offset one 32-bit value per accumulator row, the four outputs of a tile sit at wave-uniform distances (an SGPR offset of the
        // buffer store), and a row past the last tile stores to the out-of-range offset.  The generic loop below (two integer divisions
        // per row, a 64-bit address per value) was ~1.5 VALU per MFMA of the whole launch.
        const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)yout, 0, 0x7fffffff, 0x00020000);
        const int colb = (n0 + wn0 + l31) * 4;
        const int rowp = a.Wo * a.Nc * 4;                      // bytes per output row of pixels
        int soffs[4];
#pragma unroll
        for (int o = 0; o < 4; ++o) soffs[o] = ((o >> 1) * a.so) * rowp + ((o & 1) * a.so) * a.Nc * 4;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm0 + (r & 3) + 8 * (r >> 2) + 4 * h;
            const int tj = m & (a.tW - 1), ti = (m >> a.lg_tw) & (a.tH - 1), n = m >> (a.lg_tw + a.lg_th);
            const int oy = 2 * ti * a.so + pa, ox = 2 * tj * a.so + pb;
            const unsigned vo = m < a.T ? (unsigned)(((n * a.Ho + oy) * a.Wo + ox) * a.Nc * 4 + colb) : 0x80000000u;
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                const float v = accY[o][r] + bcol;
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), ry, (int)vo, soffs[o], 0);
                if (a.stats && m < a.T) { st1 += v; st2 += v * v; }
            }
        }
    } else
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm0 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (m >= a.T) continue;
        const int tj = m % a.tW;
        const int ti = (m / a.tW) % a.tH;
        const long n = m / (a.tW * a.tH);
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            const int oy = (2 * ti + (o >> 1)) * a.so + pa, ox = (2 * tj + (o & 1)) * a.so + pb;
            const float v = accY[o][r] + bcol;
            yout[((n * a.Ho + oy) * (long)a.Wo + ox) * a.Nc + n0 + wn0 + l31] = v;
            if (a.stats) { st1 += v; st2 += v * v; }
        }
    }
    if (a.stats) {
        const int srow = (phase * (int)((a.T + BM - 1) / BM) + tm) * 2 + (wave & 1);
        const float t1 = st1 + __shfl_xor(st1, 32, 64), t2 = st2 + __shfl_xor(st2, 32, 64);
        if (h == 0) {
            a.stats[((long)srow * 2 + 0) * a.Nc + n0 + wn0 + l31] = t1;
            a.stats[((long)srow * 2 + 1) * a.Nc + n0 + wn0 + l31] = t2;
        }
    }
}

// ---------------------------------------------------------------------------
// Weight gradient, Winograd domain.  dM = A dY_p A^T per tile and phase (A = [1 0; 1 1; 1 -1; 0 -1]):
//   Mdy[xi][tile][p*Cout+co]; then dU_xi = V_xi^T Mdy_xi (plain TN GEMMs), and G^T dU G maps back to the 3x3 phase
//   kernels, whose taps are scattered onto the canonical 5x5 taps exactly like the direct path's reduce.
// ---------------------------------------------------------------------------
// bpart != null (host: the grid's stride is a multiple of the channel quads, so a thread keeps its quad): the kernel also leaves the column
// sums of the dy it reads - the gradBias partials - per workgroup in bpart[gridDim.x][4*Cout] (wino_bias_finish_kernel adds them in a fixed
// order): dy is read once for both, where cg_bias_grad was a second 67 MB pass on the weight-gradient stream (59 us in the step).
__global__ __launch_bounds__(256) void wino_dy_transform_kernel(const float* __restrict__ dy, float* __restrict__ Mdy, int N,
                                                                int Hl, int Wl, int Cout, float* __restrict__ bpart) {
    const int C = 4 * Cout, cq_n = C >> 2;
    float4 bacc = make_float4(0.f, 0.f, 0.f, 0.f);
    const int tH = Hl >> 1, tW = Wl >> 1;
    const long T = (long)N * tH * tW;
    const long total = T * cq_n;
    for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += gridDim.x * 256L) {
        const int cq = (int)(idx % cq_n);
        const long tile = idx / cq_n;
        const int tj = (int)(tile % tW);
        const int ti = (int)((tile / tW) % tH);
        const long n = tile / ((long)tW * tH);
        const int p = (cq * 4) / Cout, co = cq * 4 - p * Cout;
        const int pa = p >> 1, pb = p & 1;
        float4 e[2][2];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int v = 0; v < 2; ++v)
                e[u][v] = ld4(dy + ((n * 2 * Hl + 2 * (2 * ti + u) + pa) * (long)(2 * Wl) + 2 * (2 * tj + v) + pb) * Cout + co);
        bacc = f4add(bacc, f4add(f4add(e[0][0], e[0][1]), f4add(e[1][0], e[1][1])));
        float4 r[4][2];
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            r[0][v] = e[0][v];
            r[1][v] = f4add(e[0][v], e[1][v]);
            r[2][v] = f4sub(e[0][v], e[1][v]);
            r[3][v] = make_float4(-e[1][v].x, -e[1][v].y, -e[1][v].z, -e[1][v].w);
        }
        float* out = Mdy + tile * C + cq * 4;
        const long xs = T * C;
#pragma unroll
        for (int xy = 0; xy < 4; ++xy) {
            *reinterpret_cast<float4*>(out + (xy * 4 + 0) * xs) = r[xy][0];
            *reinterpret_cast<float4*>(out + (xy * 4 + 1) * xs) = f4add(r[xy][0], r[xy][1]);
            *reinterpret_cast<float4*>(out + (xy * 4 + 2) * xs) = f4sub(r[xy][0], r[xy][1]);
            *reinterpret_cast<float4*>(out + (xy * 4 + 3) * xs) = make_float4(-r[xy][1].x, -r[xy][1].y, -r[xy][1].z, -r[xy][1].w);
        }
    }
    if (bpart) {
        // threads t and t + cq_n of a workgroup hold the same channel quad when cq_n < 256 (256 % cq_n == 0: host check)
        __shared__ float4 shb[256];
        shb[threadIdx.x] = bacc;
        __syncthreads();
        const int cq = (int)((blockIdx.x * 256L + threadIdx.x) % cq_n);
        if (cq_n >= 256) *reinterpret_cast<float4*>(bpart + (long)blockIdx.x * C + cq * 4) = bacc;
        else if ((int)threadIdx.x < cq_n) {
            float4 t = shb[threadIdx.x];
            for (int j = threadIdx.x + cq_n; j < 256; j += cq_n) t = f4add(t, shb[j]);
            *reinterpret_cast<float4*>(bpart + (long)blockIdx.x * C + cq * 4) = t;
        }
    }
}

// gb[co] += scale * sum over workgroups b and phases p of bpart[b][p*Cout + co]: one workgroup per 32 channels, 8 row lanes, fp64, fixed order
__global__ __launch_bounds__(256) void wino_bias_finish_kernel(const float* __restrict__ bpart, int nblocks, int Cout, float* __restrict__ gb,
                                                               float scale) {
    __shared__ double sh[8][33];
    const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int co = blockIdx.x * 32 + cl;
    double a = 0.0;
    if (co < Cout)
        for (int b = rl; b < nblocks; b += 8) {
            const float* row = bpart + (long)b * 4 * Cout + co;
            a += ((double)row[0] + (double)row[Cout]) + ((double)row[2 * Cout] + (double)row[3 * Cout]);
        }
    sh[rl][cl] = a;
    __syncthreads();
    if (rl == 0 && co < Cout) {
        double t = 0.0;
#pragma unroll
        for (int r = 0; r < 8; ++r) t += sh[r][cl];
        gb[co] += scale * (float)t;
    }
}

__device__ __host__ __forceinline__ int wino_phase_map(int a, int d) { return ((a + d - 2) >> 1) - ((a - 2) >> 1); }

// gw[co][ci][5][5] += scale * sum_p (G^T dU_p G)[map_p(dy,dx)];  dUt layout [xi][p*Cout+co][ci].  One thread per (co, ci).
__global__ __launch_bounds__(256) void wino_wgrad_finish_kernel(const float* __restrict__ dUt, float* __restrict__ gw, int Cout,
                                                                int Cin, float scale) {
    const long total = (long)Cout * Cin;
    const long xs = 4L * Cout * Cin;
    for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += gridDim.x * 256L) {
        const int ci = (int)(idx % Cin);
        const int co = (int)(idx / Cin);
        float acc[25];
#pragma unroll
        for (int t = 0; t < 25; ++t) acc[t] = 0.f;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            float u[4][4];
#pragma unroll
            for (int xi = 0; xi < 16; ++xi) u[xi >> 2][xi & 3] = dUt[xi * xs + ((long)p * Cout + co) * Cin + ci];
            float t[3][4];  // G^T u, G^T = [1 .5 .5 0; 0 .5 -.5 0; 0 .5 .5 1]
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                t[0][c] = u[0][c] + 0.5f * (u[1][c] + u[2][c]);
                t[1][c] = 0.5f * (u[1][c] - u[2][c]);
                t[2][c] = 0.5f * (u[1][c] + u[2][c]) + u[3][c];
            }
            float g[3][3];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                g[r][0] = t[r][0] + 0.5f * (t[r][1] + t[r][2]);
                g[r][1] = 0.5f * (t[r][1] - t[r][2]);
                g[r][2] = 0.5f * (t[r][1] + t[r][2]) + t[r][3];
            }
#pragma unroll
            for (int dy = 0; dy < 5; ++dy)
#pragma unroll
                for (int dx = 0; dx < 5; ++dx) acc[dy * 5 + dx] += g[wino_phase_map(p >> 1, dy)][wino_phase_map(p & 1, dx)];
        }
        float* dst = gw + idx * 25;
#pragma unroll
        for (int t = 0; t < 25; ++t) dst[t] += scale * acc[t];
    }
}


__device__ __host__ __forceinline__ int wino22_phase_map(int a, int d) { return ((a + d - 1) >> 1) - ((a - 1) >> 1); }


static bool wino_dims_ok(int N, int Hp, int Wp, int Cin, int Cout) {
    return N > 0 && Hp > 0 && Wp > 0 && (Hp & 1) == 0 && (Wp & 1) == 0 && Cin > 0 && Cout > 0 && Cin % 128 == 0 &&
           Cout % 128 == 0 /* K % 32, Nc % 128 for both GEMM geometries */ && (long)N * Hp * Wp * std::max(Cin, 4 * Cout) * 4L < (1L << 40);
}

}  // namespace

extern "C" {

// 1 if the Winograd path covers upsample2 -> conv 5x5 (pad 2) at these dimensions
size_t cg_conv2d_ups2_wino_supported(int N, int Hp, int Wp, int Cin, int Cout, int k, int pad) {
    return k == 5 && pad == 2 && wino_dims_ok(N, Hp, Wp, Cin, Cout) ? 1 : 0;
}
size_t cg_conv2d_ups2_wino_v_floats(int N, int Hp, int Wp, int C) {  // C = Cin (forward) or 4*Cout (data gradient)
    return (size_t)16 * ((size_t)N * (Hp / 2) * (Wp / 2)) * C;
}
size_t cg_conv2d_ups2_wino_u_floats(int Cin, int Cout) { return (size_t)4 * 16 * Cin * Cout; }

int cg_conv2d_ups2_wino_pack(void* stream, const float* wf_ph, const float* wb_ph, float* u_fwd, float* u_bwd, int Cout,
                             int Cin) {
    CG_REQUIRE((wf_ph && u_fwd) || (wb_ph && u_bwd), "cg_conv2d_ups2_wino_pack: null pointer");
    CG_REQUIRE(Cin % 4 == 0 && Cout % 4 == 0, "cg_conv2d_ups2_wino_pack: channel counts must be multiples of 4");
    if (wf_ph && u_fwd) {
        const long total = 4L * Cin * (Cout / 4);
        hipLaunchKernelGGL(wino_filter_transform_kernel<false>, dim3(cg::ew_grid(total)), dim3(256), 0, cg::S(stream), wf_ph,
                           u_fwd, Cin, Cout);
        CG_LAUNCH_CHECK();
    }
    if (wb_ph && u_bwd) {
        const long total = 4L * Cout * (Cin / 4);
        hipLaunchKernelGGL(wino_filter_transform_kernel<true>, dim3(cg::ew_grid(total)), dim3(256), 0, cg::S(stream), wb_ph,
                           u_bwd, Cout, Cin);
        CG_LAUNCH_CHECK();
    }
    return 0;
}

// The 16 GEMMs + output transform alone, on an already transformed input (what the two entry points below launch after
// their input transform; exported so that it can be timed / profiled in isolation).
// dgrad == 0: v [16][T][Cin], u = u_fwd, y [N][2Hp][2Wp][Cout];  dgrad == 1: v [16][T][4*Cout], u = u_bwd, y [N][Hp][Wp][Cin].
static int wino_gemm_launch(void* stream, const float* v, const float* u, const float* bias, float* y, int N, int Hp, int Wp,
                            int Cin, int Cout, int dgrad, float* stats, int npos = 16, int kslices = 1);

int cg_conv2d_ups2_wino_gemm(void* stream, const float* v, const float* u, const float* bias, float* y, int N, int Hp, int Wp,
                             int Cin, int Cout, int dgrad) {
    return wino_gemm_launch(stream, v, u, bias, y, N, Hp, Wp, Cin, Cout, dgrad, nullptr);
}

// rows of the [rows][2][Cout] statistics buffer cg_conv2d_ups2_wino_forward_stats fills
size_t cg_conv2d_ups2_wino_stats_rows(int N, int Hp, int Wp, int Cin, int Cout) {
    if (!wino_dims_ok(N, Hp, Wp, Cin, Cout)) return 0;
    return (size_t)4 * cg::cdiv((long)N * (Hp / 2) * (Wp / 2), 64) * 2;
}

static int wino_gemm_launch(void* stream, const float* v, const float* u, const float* bias, float* y, int N, int Hp, int Wp,
                            int Cin, int Cout, int dgrad, float* stats, int npos, int kslices) {
    CG_REQUIRE(v && u && y, "cg_conv2d_ups2_wino_gemm: null pointer");
    CG_REQUIRE(wino_dims_ok(N, Hp, Wp, Cin, Cout), "cg_conv2d_ups2_wino_gemm: unsupported dimensions");
    const int T = N * (Hp / 2) * (Wp / 2);
    WinoArgs a;
    a.V = v; a.U = u; a.bias = bias; a.y = y; a.T = T; a.tH = Hp / 2; a.tW = Wp / 2; a.stats = stats;
    a.xcd = (int)cg::opt(cg::OPT_XCD_SWIZZLE);
    a.vpp = npos == 9 && !dgrad ? 1 : 0;
    a.kz = 0;
    a.lg_tw = a.lg_th = -1;
    CG_REQUIRE(!stats || !dgrad, "wino_gemm: statistics only on the forward launch");
    if (dgrad) { a.K = 4 * Cout; a.Nc = Cin; a.so = 1; a.Ho = Hp; a.Wo = Wp; }
    else { a.K = Cin; a.Nc = Cout; a.so = 2; a.Ho = 2 * Hp; a.Wo = 2 * Wp; }
    const int bk = (int)cg::opt(cg::OPT_WINO_BK);
    CG_REQUIRE(kslices <= 1 || (dgrad && a.K % kslices == 0 && (a.K / kslices) % 64 == 0), "wino_gemm: K slices only on the data gradient, in multiples of 64 rows");
    if (kslices > 1) a.kz = a.K / kslices;
    {
        auto lg = [](int v) { int l = 0; while ((1 << l) < v) ++l; return (1 << l) == v ? l : -1; };
        const long ybytes = (long)N * a.Ho * a.Wo * a.Nc * 4L;
        if (lg(a.tW) >= 0 && lg(a.tH) >= 0 && ybytes < 0x7fffffffL) { a.lg_tw = lg(a.tW); a.lg_th = lg(a.tH); }
    }
    const dim3 grid(cg::cdiv(T, 64) * (a.Nc / 128), 1, dgrad ? (kslices > 1 ? kslices : 1) : 4);
    // K step 32 (half the barriers per MFMA) pays when the launch is at most ~one workgroup per CU - the data-gradient
    // geometry at batch 128 (0.53 -> 0.41 ms) - and costs 6 % when two workgroups per CU already hide each other's barriers
    const bool k32 = bk == 32 || (bk == 0 && (long)grid.x * grid.z <= cg::kNumCU * 3 / 2);
    // LDS-direct loads (CG_WINO_GLDS): one xi plane of V must stay below the 2 GB a buffer offset reaches
    const bool glds = cg::opt(cg::OPT_WINO_GLDS) != 0 && (long)T * a.K * 4L < 0x7fffffffL &&
                      16L * a.K * a.Nc * 4L < 0x7fffffffL;
    if (npos == 9) {     // F(2x2,2x2): the LDS-direct-load kernel only (cg_conv2d_ups2_wino22_supported checks the same conditions)
        CG_REQUIRE(glds, "wino_gemm: the 9-position form needs the LDS-direct-load kernel (CG_WINO_GLDS)");
        if (k32 && (a.kz ? a.kz : a.K) % 64 == 0) hipLaunchKernelGGL((wino_gemm_g_kernel<32, 9>), grid, dim3(512), 0, cg::S(stream), a);
        else hipLaunchKernelGGL((wino_gemm_g_kernel<16, 9>), grid, dim3(512), 0, cg::S(stream), a);
        CG_LAUNCH_CHECK();
        return 0;
    }
    CG_REQUIRE(kslices <= 1 || glds, "wino_gemm: K slices need the LDS-direct-load kernel (CG_WINO_GLDS)");
    if (glds && k32 && (a.kz ? a.kz : a.K) % 64 == 0) { hipLaunchKernelGGL((wino_gemm_g_kernel<32>), grid, dim3(512), 0, cg::S(stream), a); CG_LAUNCH_CHECK(); return 0; }
    if (glds) { hipLaunchKernelGGL((wino_gemm_g_kernel<16>), grid, dim3(512), 0, cg::S(stream), a); CG_LAUNCH_CHECK(); return 0; }
    if (k32 && a.K % 64 == 0) hipLaunchKernelGGL((wino_gemm_kernel<8, 32>), grid, dim3(512), 0, cg::S(stream), a);
    else hipLaunchKernelGGL((wino_gemm_kernel<8, 16>), grid, dim3(512), 0, cg::S(stream), a);
    CG_LAUNCH_CHECK();
    return 0;
}

// y[N][2Hp][2Wp][Cout] = bias + conv5x5(upsample2(x_lo)); v (cg_conv2d_ups2_wino_v_floats(..., Cin) floats) receives
// the transformed input and is what the weight gradient consumes later.
int cg_conv2d_ups2_wino_forward_stats(void* stream, const float* x_lo, const float* u_fwd, const float* bias, float* y, float* v,
                                      int N, int Hp, int Wp, int Cin, int Cout, float* stats) {
    CG_REQUIRE(x_lo && u_fwd && y && v, "cg_conv2d_ups2_wino_forward: null pointer");
    CG_REQUIRE(wino_dims_ok(N, Hp, Wp, Cin, Cout), "cg_conv2d_ups2_wino_forward: unsupported dimensions");
    hipStream_t st = cg::S(stream);
    const int T = N * (Hp / 2) * (Wp / 2);
    hipLaunchKernelGGL(wino_input_transform_kernel<0>, dim3(cg::ew_grid((long)T * (Cin / 4))), dim3(256), 0, st, x_lo, v, N, Hp,
                       Wp, Cin, Cin);
    CG_LAUNCH_CHECK();
    return wino_gemm_launch(stream, v, u_fwd, bias, y, N, Hp, Wp, Cin, Cout, 0, stats);
}

int cg_conv2d_ups2_wino_forward(void* stream, const float* x_lo, const float* u_fwd, const float* bias, float* y, float* v,
                                int N, int Hp, int Wp, int Cin, int Cout) {
    return cg_conv2d_ups2_wino_forward_stats(stream, x_lo, u_fwd, bias, y, v, N, Hp, Wp, Cin, Cout, nullptr);
}

// ---- F(2x2,2x2): upsample2 -> conv 3x3 (pad 1), forward ----
size_t cg_conv2d_ups2_wino22_supported(int N, int Hp, int Wp, int Cin, int Cout) {
    if (!wino_dims_ok(N, Hp, Wp, Cin, Cout)) return 0;
    const long T = (long)N * (Hp / 2) * (Wp / 2);
    const bool glds = cg::opt(cg::OPT_WINO_GLDS) != 0 && T * Cin * 4L < 0x7fffffffL &&
                      16L * Cin * Cout * 4L < 0x7fffffffL;
    return glds ? 1 : 0;
}
// the limits cg_conv2d_ups2_wino22_dgrad / _wgrad apply on top of the forward's (a planner that only asked the forward's
// question would compile the forward and fail in Module:backward for a large batch x Cout)
size_t cg_conv2d_ups2_wino22_dgrad_supported(int N, int Hp, int Wp, int Cin, int Cout) {
    return cg_conv2d_ups2_wino22_supported(N, Hp, Wp, Cin, Cout) && (long)N * (Hp / 2) * (Wp / 2) * Cout * 16L < 0x7fffffffL ? 1 : 0;
}
size_t cg_conv2d_ups2_wino22_v_floats(int N, int Hp, int Wp, int Cin) { return (size_t)36 * ((size_t)N * (Hp / 2) * (Wp / 2)) * Cin; }
size_t cg_conv2d_ups2_wino22_u_floats(int Cin, int Cout) { return (size_t)4 * 9 * Cin * Cout; }

int cg_conv2d_ups2_wino22_pack(void* stream, const float* wf_ph, const float* wb_ph, float* u_fwd, float* u_bwd, int Cout, int Cin) {
    CG_REQUIRE((wf_ph && u_fwd) || (wb_ph && u_bwd), "cg_conv2d_ups2_wino22_pack: null pointer");
    CG_REQUIRE(Cin % 4 == 0 && Cout % 4 == 0, "cg_conv2d_ups2_wino22_pack: channel counts must be multiples of 4");
    if (wf_ph && u_fwd) {
        hipLaunchKernelGGL(wino22_filter_transform_kernel<false>, dim3(cg::ew_grid(4L * Cin * (Cout / 4))), dim3(256), 0, cg::S(stream), wf_ph,
                           u_fwd, Cin, Cout);
        CG_LAUNCH_CHECK();
    }
    if (wb_ph && u_bwd) {
        hipLaunchKernelGGL(wino22_filter_transform_kernel<true>, dim3(cg::ew_grid(4L * Cout * (Cin / 4))), dim3(256), 0, cg::S(stream), wb_ph,
                           u_bwd, Cout, Cin);
        CG_LAUNCH_CHECK();
    }
    return 0;
}

// y[N][2Hp][2Wp][Cout] = bias + conv3x3(upsample2(x_lo)); v: scratch of cg_conv2d_ups2_wino22_v_floats floats; stats as
// cg_conv2d_ups2_wino_forward_stats (same rows: cg_conv2d_ups2_wino_stats_rows).
int cg_conv2d_ups2_wino22_forward_stats(void* stream, const float* x_lo, const float* u22, const float* bias, float* y, float* v,
                                        int N, int Hp, int Wp, int Cin, int Cout, float* stats) {
    CG_REQUIRE(x_lo && u22 && y && v, "cg_conv2d_ups2_wino22_forward: null pointer");
    CG_REQUIRE(cg_conv2d_ups2_wino22_supported(N, Hp, Wp, Cin, Cout), "cg_conv2d_ups2_wino22_forward: unsupported dimensions / options");
    const int T = N * (Hp / 2) * (Wp / 2);
    hipLaunchKernelGGL(wino22_input_transform_kernel, dim3(cg::ew_grid((long)T * (Cin / 4))), dim3(256), 0, cg::S(stream), x_lo, v, N, Hp, Wp, Cin);
    CG_LAUNCH_CHECK();
    return wino_gemm_launch(stream, v, u22, bias, y, N, Hp, Wp, Cin, Cout, 0, stats, 9);
}

// The data gradient's launch has T/64 x Cin/128 workgroups - 128 on G's 512 -> 256 layer at batch 128, half the chip.  Below ~one
// workgroup per CU the four phases (K slices of Cout rows) go to blockIdx.z, each writing a partial dx_lo behind V in the scratch, and
// wino22_sum_kernel adds them in a fixed order.
static int wino22_dgrad_slices(int N, int Hp, int Wp, int Cin, int Cout) {   // 1 (unsplit) or 4 K slices
    if (Cout % 64 != 0) return 1;
    return (long)cg::cdiv(N * (Hp / 2) * (Wp / 2), 64) * (Cin / 128) < cg::kNumCU ? 4 : 1;
}

extern "C++" template <int S>
__global__ __launch_bounds__(256) void wino22_sum_kernel(const float* __restrict__ part, float* __restrict__ out, long n4, long stride) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += gridDim.x * 256L) {
        const float4 a = ld4(part + 4 * i), b = ld4(part + stride + 4 * i);
        float4 r = f4add(a, b);
        if (S == 4) r = f4add(r, f4add(ld4(part + 2 * stride + 4 * i), ld4(part + 3 * stride + 4 * i)));
        *reinterpret_cast<float4*>(out + 4 * i) = r;
    }
}

size_t cg_conv2d_ups2_wino22_dgrad_v_floats(int N, int Hp, int Wp, int Cin, int Cout) {
    const size_t T = (size_t)N * (Hp / 2) * (Wp / 2);
    const int sl = wino22_dgrad_slices(N, Hp, Wp, Cin, Cout);
    return 36 * T * Cout + (sl > 1 ? (size_t)sl * 4 * T * Cin : 0);
}

// dx_lo[N][Hp][Wp][Cin] = gradient of upsample2 -> conv3x3 w.r.t. the low-res input; v_dy: scratch of cg_conv2d_ups2_wino22_dgrad_v_floats()
// floats (9 planes x T x 4 Cout, + the four partial results when the K slices are split); u_bwd from cg_conv2d_ups2_wino22_pack.
int cg_conv2d_ups2_wino22_dgrad(void* stream, const float* dy, const float* u_bwd, float* dx_lo, float* v_dy, int N, int Hp, int Wp, int Cin,
                                int Cout) {
    CG_REQUIRE(dy && u_bwd && dx_lo && v_dy, "cg_conv2d_ups2_wino22_dgrad: null pointer");
    CG_REQUIRE(cg_conv2d_ups2_wino22_dgrad_supported(N, Hp, Wp, Cin, Cout), "cg_conv2d_ups2_wino22_dgrad: unsupported dimensions / options");
    const int T = N * (Hp / 2) * (Wp / 2);
    hipLaunchKernelGGL(wino22_dy_input_transform_kernel, dim3(cg::ew_grid((long)T * Cout)), dim3(256), 0, cg::S(stream), dy, v_dy, N, Hp, Wp, Cout);
    CG_LAUNCH_CHECK();
    const int sl = wino22_dgrad_slices(N, Hp, Wp, Cin, Cout);
    if (sl == 1) return wino_gemm_launch(stream, v_dy, u_bwd, nullptr, dx_lo, N, Hp, Wp, Cin, Cout, 1, nullptr, 9);
    float* part = v_dy + (size_t)36 * T * Cout;
    if (wino_gemm_launch(stream, v_dy, u_bwd, nullptr, part, N, Hp, Wp, Cin, Cout, 1, nullptr, 9, sl)) return 1;
    const long n = (long)T * 4 * Cin;
    if (sl == 4) hipLaunchKernelGGL(wino22_sum_kernel<4>, dim3(cg::ew_grid(n / 4)), dim3(256), 0, cg::S(stream), part, dx_lo, n / 4, n);
    else hipLaunchKernelGGL(wino22_sum_kernel<2>, dim3(cg::ew_grid(n / 4)), dim3(256), 0, cg::S(stream), part, dx_lo, n / 4, n);
    CG_LAUNCH_CHECK();
    return 0;
}

// dx_lo[N][Hp][Wp][Cin] = gradient w.r.t. the low-res input (the upsampling's 2x2 block sum folded in);
// v_dy: scratch of cg_conv2d_ups2_wino_v_floats(..., 4*Cout) floats.
int cg_conv2d_ups2_wino_dgrad(void* stream, const float* dy, const float* u_bwd, float* dx_lo, float* v_dy, int N, int Hp,
                              int Wp, int Cin, int Cout) {
    CG_REQUIRE(dy && u_bwd && dx_lo && v_dy, "cg_conv2d_ups2_wino_dgrad: null pointer");
    CG_REQUIRE(wino_dims_ok(N, Hp, Wp, Cin, Cout), "cg_conv2d_ups2_wino_dgrad: unsupported dimensions");
    hipStream_t st = cg::S(stream);
    const int T = N * (Hp / 2) * (Wp / 2);
    hipLaunchKernelGGL(wino_input_transform_kernel<1>, dim3(cg::ew_grid((long)T * Cout)), dim3(256), 0, st, dy, v_dy, N, Hp, Wp,
                       4 * Cout, Cout);
    CG_LAUNCH_CHECK();
    return cg_conv2d_ups2_wino_gemm(stream, v_dy, u_bwd, nullptr, dx_lo, N, Hp, Wp, Cin, Cout, 1);
}

// The same data gradient with its K rows (the four phases' channels) in 2 or 4 slices over blockIdx.z and a fixed-order sum of the partial
// results: the unsplit launch has T/64 x Cin/128 workgroups, which leaves most of the chip idle where Cin is 128 (G32up's first 5x5 layer).  part: cg_conv2d_ups2_wino_dgrad_part_floats() floats (0 = keep the unsplit launch).
static int wino_dgrad_slices(int N, int Hp, int Wp, int Cin, int Cout) {
    const int force = 0;
    const long T = (long)N * (Hp / 2) * (Wp / 2);
    const bool glds = cg::opt(cg::OPT_WINO_GLDS) != 0 && T * 4L * Cout * 4L < 0x7fffffffL &&
                      16L * 4L * Cout * Cin * 4L < 0x7fffffffL;
    if (!glds || !wino_dims_ok(N, Hp, Wp, Cin, Cout)) return 1;
    // workgroups of the unsplit launch: at one per CU or more it stays (G32up-c's layer at batch 128: 256, splitting measured +0.5 % on the
    // step), below that 2 or 4 slices (G32up's 128 -> 256 layer at batch 256: 64 workgroups -> 4 slices, config #3 9.52 -> 9.25 ms)
    const long wgs = (long)cg::cdiv((int)T, 64) * (Cin / 128);
    int sl = force == 1 || force == 2 || force == 4 ? force : (wgs >= cg::kNumCU ? 1 : (wgs * 2 >= cg::kNumCU ? 2 : 4));
    while (sl > 1 && (4 * Cout / sl) % 64 != 0) sl >>= 1;
    return sl;
}
size_t cg_conv2d_ups2_wino_dgrad_part_floats(int N, int Hp, int Wp, int Cin, int Cout) {
    const int sl = wino_dgrad_slices(N, Hp, Wp, Cin, Cout);
    return sl > 1 ? (size_t)sl * N * Hp * Wp * Cin : 0;
}
int cg_conv2d_ups2_wino_dgrad_split(void* stream, const float* dy, const float* u_bwd, float* dx_lo, float* v_dy, float* part, int N, int Hp,
                                    int Wp, int Cin, int Cout) {
    const int sl = wino_dgrad_slices(N, Hp, Wp, Cin, Cout);
    if (sl <= 1 || !part) return cg_conv2d_ups2_wino_dgrad(stream, dy, u_bwd, dx_lo, v_dy, N, Hp, Wp, Cin, Cout);
    CG_REQUIRE(dy && u_bwd && dx_lo && v_dy, "cg_conv2d_ups2_wino_dgrad_split: null pointer");
    const int T = N * (Hp / 2) * (Wp / 2);
    hipLaunchKernelGGL(wino_input_transform_kernel<1>, dim3(cg::ew_grid((long)T * Cout)), dim3(256), 0, cg::S(stream), dy, v_dy, N, Hp, Wp, 4 * Cout, Cout);
    CG_LAUNCH_CHECK();
    if (wino_gemm_launch(stream, v_dy, u_bwd, nullptr, part, N, Hp, Wp, Cin, Cout, 1, nullptr, 16, sl)) return 1;
    const long n = (long)T * 4 * Cin;
    if (sl == 4) hipLaunchKernelGGL(wino22_sum_kernel<4>, dim3(cg::ew_grid(n / 4)), dim3(256), 0, cg::S(stream), part, dx_lo, n / 4, n);
    else hipLaunchKernelGGL(wino22_sum_kernel<2>, dim3(cg::ew_grid(n / 4)), dim3(256), 0, cg::S(stream), part, dx_lo, n / 4, n);
    CG_LAUNCH_CHECK();
    return 0;
}

static size_t wino_align(size_t b) { return (b + 255) / 256 * 256; }

size_t cg_conv2d_ups2_wino_wgrad_workspace_bytes(int N, int Hp, int Wp, int Cin, int Cout) {
    if (!wino_dims_ok(N, Hp, Wp, Cin, Cout)) return 0;
    const size_t T = (size_t)N * (Hp / 2) * (Wp / 2);
    return wino_align(16 * T * 4 * Cout * sizeof(float)) + wino_align((size_t)16 * 4 * Cout * Cin * sizeof(float)) +
           wino_align(std::max(sizeof(double) * Cout, (size_t)cg::kNumCU * 8 * 4 * Cout * sizeof(float))) +   // gradBias: cg_bias_grad's scratch / the dy transform's partials

           cg_conv2d_wgrad_workspace_bytes_strided(16, (int)T, 1, 1, Cin, 4 * Cout, 1, 1, 0, 0, 0);
}

// gw_canonical[Cout][Cin][5][5] += scale * dW, gb += scale * sum dy, from the transformed input v the forward left behind.
int cg_conv2d_ups2_wino_wgrad(void* stream, const float* v, const float* dy, float* gw_canonical, float* gb, int N, int Hp,
                              int Wp, int Cin, int Cout, float scale, void* ws, size_t ws_bytes) {
    CG_REQUIRE(v && dy && gw_canonical, "cg_conv2d_ups2_wino_wgrad: null pointer");
    CG_REQUIRE(wino_dims_ok(N, Hp, Wp, Cin, Cout), "cg_conv2d_ups2_wino_wgrad: unsupported dimensions");
    const size_t need = cg_conv2d_ups2_wino_wgrad_workspace_bytes(N, Hp, Wp, Cin, Cout);
    CG_REQUIRE(ws && ws_bytes >= need && (uintptr_t)ws % 16 == 0, "cg_conv2d_ups2_wino_wgrad: workspace too small (%zu < %zu)",
               ws_bytes, need);
    hipStream_t st = cg::S(stream);
    const int T = N * (Hp / 2) * (Wp / 2);
    const int C4 = 4 * Cout;
    char* base = (char*)ws;
    float* mdy = (float*)base;                    base += wino_align((size_t)16 * T * C4 * sizeof(float));
    float* dut = (float*)base;                    base += wino_align((size_t)16 * C4 * Cin * sizeof(float));
    const size_t bws_bytes = wino_align(std::max(sizeof(double) * Cout, (size_t)cg::kNumCU * 8 * 4 * Cout * sizeof(float)));
    void* bws = base;                             base += bws_bytes;
    void* tws = base;
    const size_t tws_bytes = ws_bytes - (size_t)(base - (char*)ws);
    // (gradBias partial sums from this transform - it reads every dy anyway - were parity-clean but cost the step 0.6 % in round 4: the
    // separate 59 us pass is not on the pass's critical path, the longer transform is; the kernel keeps the argument, nobody passes it)
    const bool fuse = false;
    const int tgrid = (int)cg::ew_grid((long)T * Cout);
    hipLaunchKernelGGL(wino_dy_transform_kernel, dim3(tgrid), dim3(256), 0, st, dy, mdy, N, Hp, Wp, Cout, fuse ? (float*)bws : nullptr);
    CG_LAUNCH_CHECK();
    CG_HIP(hipMemsetAsync(dut, 0, (size_t)16 * C4 * Cin * sizeof(float), st));
    // 16 TN GEMMs dU_xi^T[pco][ci] = sum_tile Mdy_xi[tile][pco] V_xi[tile][ci]: the planes of V, Mdy and dU are equally spaced, so all
    // 16 run as ONE launch + one reduction (rounds 1-3: four launches of four)
    const int groups_per_launch = 16;
    for (int q = 0; q < 16; q += groups_per_launch) {
        if (cg_conv2d_wgrad_strided(stream, groups_per_launch, v + q * (long)T * Cin, (long)T * Cin, mdy + q * (long)T * C4, (long)T * C4,
                                    dut + q * (long)C4 * Cin, (long)C4 * Cin, T, 1, 1, Cin, C4, 1, 1, 0, 0, 0, 1.f, tws, tws_bytes)) return 1;
    }
    hipLaunchKernelGGL(wino_wgrad_finish_kernel, dim3(cg::ew_grid((long)Cout * Cin)), dim3(256), 0, st, dut, gw_canonical, Cout,
                       Cin, scale);
    CG_LAUNCH_CHECK();
    if (fuse) {
        hipLaunchKernelGGL(wino_bias_finish_kernel, dim3(cg::cdiv(Cout, 32)), dim3(256), 0, st, (const float*)bws, tgrid, Cout, gb, scale);
        CG_LAUNCH_CHECK();
        return 0;
    }
    if (gb) return cg_bias_grad(stream, dy, gb, (long)N * 4 * Hp * Wp, Cout, scale, bws, wino_align(sizeof(double) * Cout));
    return 0;
}



}  // extern "C"
