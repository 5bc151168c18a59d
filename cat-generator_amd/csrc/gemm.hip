// Implicit-GEMM convolution / linear kernels on the exact-fp32 MFMA
// (v_mfma_f32_32x32x2_f32) for gfx950.
//
// Replaces the THNN/THCUNN/cudnn kernels the reference reaches through
// cudnn.SpatialConvolution (models.lua:206,212,218,222), nn.SpatialConvolution
// (models.lua:646-685,844-846) and nn.Linear (models.lua:199,697,700,850,853):
//   igemm_nn : updateOutput and updateGradInput  (Y[m][n] = sum_k A(m,k) W[k][n])
//   igemm_tn : accGradParameters                 (dW[k][n] = sum_m A(m,k) dY[m][n])
// A(m,k) is never materialised: m -> output pixel (n,oy,ox), k -> (tap,ci),
// gathered from the NHWC activation with zero padding and, if ups==1, through
// a virtual nearest-neighbour 2x upsampling (models.lua:205,211,217).
//
// Tiling (wave64, 4 waves / workgroup): block tile BM x BN, K step 16, each
// wave owns (BM/WM) x (BN/WN) as MI x NI MFMA 32x32 accumulators.  Both
// operands sit K-major in LDS ([k][m] / [k][n]) so a fragment read is one
// conflict-free ds_read_b32 per operand per MFMA; global->register->LDS
// staging is double buffered (one barrier per K tile) with the next tile's
// global loads issued before the MFMA block of the current one.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 16;

struct ConvGeom {
    int M;            // N*Ho*Wo output pixels
    int HoWo, Wo;     // output spatial
    int Hl, Wl;       // logical input spatial (physical << ups)
    int Hp, Wp;       // physical input spatial
    int Cin, Cout;
    int kW;           // tap -> (ky = tap / kW, kx = tap % kW)
    int padH, padW, ups;
    int Ktot;         // kH*kW*Cin
};

struct NNArgs {
    const float* x;
    const float* w;     // [Ktot][Cout]
    const float* bias;  // [Cout] or null
    float* y;           // [M][Cout] or split partials [S][M][Cout]
    ConvGeom g;
    int kchunk;         // K range per split (multiple of BK)
    long split_stride;  // M*Cout
};

struct TNArgs {
    const float* x;
    const float* dy;    // [M][Cout]
    float* part;        // [S][Ktot][Cout]
    ConvGeom g;
    int pchunk;         // pixels per split (multiple of BK)
};

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// ---------------------------------------------------------------------------
// NN: Y[m][n] = sum_k A(m,k) * W[k][n]
// ---------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN, bool VECA, bool VECB>
__global__ __launch_bounds__(256) void igemm_nn_kernel(NNArgs a) {
    static_assert(WM * WN == 4, "4 waves per workgroup");
    constexpr int MI = BM / WM / 32;
    constexpr int NI = BN / WN / 32;
    static_assert(MI >= 1 && NI >= 1, "wave tile >= 32x32");
    constexpr int LDA = BM + 2;  // 4*LDA % 32 == 8: transposed ds_write_b32 conflict-free
    constexpr int LDB = BN + 4;  // rows stay 16-B aligned for ds_write_b128
    constexpr int A_TILE = BK * LDA, B_TILE = BK * LDB;
    constexpr int AROWS = BM / 64;  // float4 per thread per tile (A)
    static_assert(BM % 64 == 0, "BM multiple of 64");
    constexpr int NVEC = BN / 4;
    constexpr int BRPP = 256 / NVEC;  // B rows per pass
    constexpr int BPASS = (BK + BRPP - 1) / BRPP;

    __shared__ __attribute__((aligned(16))) float smem[2 * A_TILE + 2 * B_TILE];
    float* As = smem;
    float* Bs = smem + 2 * A_TILE;

    const ConvGeom& g = a.g;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wm0 = (wave / WN) * (BM / WM);
    const int wn0 = (wave % WN) * (BN / WN);

    const int ntn = (g.Cout + BN - 1) / BN;
    const int tn = blockIdx.x % ntn, tm = blockIdx.x / ntn;
    const int m0 = tm * BM, n0 = tn * BN;
    const int split = blockIdx.y;
    const int ks = split * a.kchunk;
    const int kend = min(g.Ktot, ks + a.kchunk);
    const int T = (kend - ks + BK - 1) / BK;

    // ---- A staging: thread owns k-vector a_kv (4 consecutive k) of rows a_r + 64p
    const int a_kv = tid & 3, a_r = tid >> 2;
    int a_pix[AROWS], a_oy[AROWS], a_ox[AROWS];
    bool a_ok[AROWS];
#pragma unroll
    for (int p = 0; p < AROWS; ++p) {
        const int m = m0 + a_r + 64 * p;
        a_ok[p] = m < g.M;
        const int mm = a_ok[p] ? m : 0;
        const int n = mm / g.HoWo;
        const int rem = mm - n * g.HoWo;
        a_oy[p] = rem / g.Wo;
        a_ox[p] = rem - a_oy[p] * g.Wo;
        a_pix[p] = n * g.Hp * g.Wp;
    }
    // ---- B staging
    const int b_nv = tid % NVEC, b_kr = tid / NVEC;

    float4 areg[AROWS];
    float4 breg[BPASS];

    auto load_tile = [&](int k0) {
        const int k = k0 + 4 * a_kv;
        if (VECA) {
            const bool kok = k < kend;
            const int tap = k / g.Cin;
            const int ci = k - tap * g.Cin;
            const int ky = tap / g.kW;
            const int kx = tap - ky * g.kW;
#pragma unroll
            for (int p = 0; p < AROWS; ++p) {
                const int iy = a_oy[p] + ky - g.padH;
                const int ix = a_ox[p] + kx - g.padW;
                const bool ok = a_ok[p] && kok && (unsigned)iy < (unsigned)g.Hl && (unsigned)ix < (unsigned)g.Wl;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ok) {
                    const long off = ((long)(a_pix[p] + (iy >> g.ups) * g.Wp + (ix >> g.ups))) * g.Cin + ci;
                    v = ld4(a.x + off);
                }
                areg[p] = v;
            }
        } else {
            float tmp[AROWS][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int kj = k + j;
                const bool kok = kj < kend;
                const int tap = kj / g.Cin;
                const int ci = kj - tap * g.Cin;
                const int ky = tap / g.kW;
                const int kx = tap - ky * g.kW;
#pragma unroll
                for (int p = 0; p < AROWS; ++p) {
                    const int iy = a_oy[p] + ky - g.padH;
                    const int ix = a_ox[p] + kx - g.padW;
                    const bool ok = a_ok[p] && kok && (unsigned)iy < (unsigned)g.Hl && (unsigned)ix < (unsigned)g.Wl;
                    float v = 0.f;
                    if (ok) {
                        const long off = ((long)(a_pix[p] + (iy >> g.ups) * g.Wp + (ix >> g.ups))) * g.Cin + ci;
                        v = a.x[off];
                    }
                    tmp[p][j] = v;
                }
            }
#pragma unroll
            for (int p = 0; p < AROWS; ++p) areg[p] = make_float4(tmp[p][0], tmp[p][1], tmp[p][2], tmp[p][3]);
        }
#pragma unroll
        for (int q = 0; q < BPASS; ++q) {
            const int kr = b_kr + q * BRPP;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (kr < BK) {
                const int kk = k0 + kr;
                const int n = n0 + 4 * b_nv;
                if (kk < kend) {
                    const float* wp = a.w + (long)kk * g.Cout + n;
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
    };

    auto store_tile = [&](int buf) {
        float* A = As + buf * A_TILE;
        float* B = Bs + buf * B_TILE;
#pragma unroll
        for (int p = 0; p < AROWS; ++p) {
            const int r = a_r + 64 * p;
            A[(4 * a_kv + 0) * LDA + r] = areg[p].x;
            A[(4 * a_kv + 1) * LDA + r] = areg[p].y;
            A[(4 * a_kv + 2) * LDA + r] = areg[p].z;
            A[(4 * a_kv + 3) * LDA + r] = areg[p].w;
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

    if (T > 0) {
        load_tile(ks);
        store_tile(0);
    }
    __syncthreads();

    for (int t = 0; t < T; ++t) {
        const int buf = t & 1;
        if (t + 1 < T) load_tile(ks + (t + 1) * BK);
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
        if (t + 1 < T) store_tile(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: D[i][j], i = (r&3) + 8*(r>>2) + 4*h (pixel), j = l31 (channel)
    float* yout = a.y + (long)split * a.split_stride;
    const bool add_bias = (a.bias != nullptr) && (gridDim.y == 1);
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int n = n0 + wn0 + j * 32 + l31;
        if (n >= g.Cout) continue;
        const float bv = add_bias ? a.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (m < g.M) yout[(long)m * g.Cout + n] = acc[i][j][r] + bv;
            }
        }
    }
}

// split-K reduce for NN: y[m][n] = bias[n] + sum_s part[s][m][n]
__global__ void nn_splitk_reduce_kernel(const float* part, const float* bias, float* y, long MN, int Cout, int S) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < MN; i += (long)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int k = 0; k < S; ++k) s += part[(long)k * MN + i];
        if (bias) s += bias[i % Cout];
        y[i] = s;
    }
}

// ---------------------------------------------------------------------------
// TN: dW[k][n] = sum_m A(m,k) * dY[m][n]   (k = (tap,ci) rows, m = pixels reduced)
// ---------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN, bool VECA, bool VECB>
__global__ __launch_bounds__(256) void igemm_tn_kernel(TNArgs a) {
    static_assert(WM * WN == 4, "4 waves per workgroup");
    constexpr int MI = BM / WM / 32;
    constexpr int NI = BN / WN / 32;
    constexpr int LDA = BM + 4;
    constexpr int LDB = BN + 4;
    constexpr int A_TILE = BK * LDA, B_TILE = BK * LDB;
    constexpr int AVEC = BM / 4, ARPP = 256 / AVEC, APASS = (BK + ARPP - 1) / ARPP;
    constexpr int BVEC = BN / 4, BRPP = 256 / BVEC, BPASS = (BK + BRPP - 1) / BRPP;

    __shared__ __attribute__((aligned(16))) float smem[2 * A_TILE + 2 * B_TILE];
    float* As = smem;
    float* Bs = smem + 2 * A_TILE;

    const ConvGeom& g = a.g;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wm0 = (wave / WN) * (BM / WM);
    const int wn0 = (wave % WN) * (BN / WN);

    const int ntn = (g.Cout + BN - 1) / BN;
    const int tn = blockIdx.x % ntn, tm = blockIdx.x / ntn;
    const int m0 = tm * BM, n0 = tn * BN;  // m0: row of dW (tap,ci)
    const int split = blockIdx.y;
    const int ps = split * a.pchunk;
    const int pend = min(g.M, ps + a.pchunk);
    const int T = (pend - ps + BK - 1) / BK;

    // A staging: fixed (tap,ci) columns per thread, pixel rows vary per tile
    const int a_mv = tid % AVEC, a_kr = tid / AVEC;
    int a_ci[VECA ? 1 : 4], a_dy[VECA ? 1 : 4], a_dx[VECA ? 1 : 4];
    bool a_cok[VECA ? 1 : 4];
#pragma unroll
    for (int j = 0; j < (VECA ? 1 : 4); ++j) {
        const int mm = m0 + 4 * a_mv + j;
        a_cok[j] = mm < g.Ktot;
        const int mc = a_cok[j] ? mm : 0;
        const int tap = mc / g.Cin;
        a_ci[j] = mc - tap * g.Cin;
        const int ky = tap / g.kW;
        a_dy[j] = ky - g.padH;
        a_dx[j] = (tap - ky * g.kW) - g.padW;
    }
    const int b_nv = tid % BVEC, b_kr = tid / BVEC;

    float4 areg[APASS];
    float4 breg[BPASS];

    auto load_tile = [&](int p0) {
#pragma unroll
        for (int q = 0; q < APASS; ++q) {
            const int kr = a_kr + q * ARPP;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (kr < BK) {
                const int pix = p0 + kr;
                if (pix < pend) {
                    const int n = pix / g.HoWo;
                    const int rem = pix - n * g.HoWo;
                    const int oy = rem / g.Wo;
                    const int ox = rem - oy * g.Wo;
                    const int base = n * g.Hp * g.Wp;
                    if (VECA) {
                        const int iy = oy + a_dy[0], ix = ox + a_dx[0];
                        if (a_cok[0] && (unsigned)iy < (unsigned)g.Hl && (unsigned)ix < (unsigned)g.Wl) {
                            const long off = ((long)(base + (iy >> g.ups) * g.Wp + (ix >> g.ups))) * g.Cin + a_ci[0];
                            v = ld4(a.x + off);
                        }
                    } else {
                        float t4[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int jj = VECA ? 0 : j;
                            const int iy = oy + a_dy[jj], ix = ox + a_dx[jj];
                            float e = 0.f;
                            if (a_cok[jj] && (unsigned)iy < (unsigned)g.Hl && (unsigned)ix < (unsigned)g.Wl) {
                                const long off = ((long)(base + (iy >> g.ups) * g.Wp + (ix >> g.ups))) * g.Cin + a_ci[jj];
                                e = a.x[off];
                            }
                            t4[j] = e;
                        }
                        v = make_float4(t4[0], t4[1], t4[2], t4[3]);
                    }
                }
            }
            areg[q] = v;
        }
#pragma unroll
        for (int q = 0; q < BPASS; ++q) {
            const int kr = b_kr + q * BRPP;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (kr < BK) {
                const int pix = p0 + kr;
                const int n = n0 + 4 * b_nv;
                if (pix < pend) {
                    const float* dp = a.dy + (long)pix * g.Cout + n;
                    if (VECB) {
                        if (n < g.Cout) v = ld4(dp);
                    } else {
                        if (n + 0 < g.Cout) v.x = dp[0];
                        if (n + 1 < g.Cout) v.y = dp[1];
                        if (n + 2 < g.Cout) v.z = dp[2];
                        if (n + 3 < g.Cout) v.w = dp[3];
                    }
                }
            }
            breg[q] = v;
        }
    };

    auto store_tile = [&](int buf) {
        float* A = As + buf * A_TILE;
        float* B = Bs + buf * B_TILE;
#pragma unroll
        for (int q = 0; q < APASS; ++q) {
            const int kr = a_kr + q * ARPP;
            if (kr < BK) *reinterpret_cast<float4*>(A + kr * LDA + 4 * a_mv) = areg[q];
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

    if (T > 0) {
        load_tile(ps);
        store_tile(0);
    }
    __syncthreads();

    for (int t = 0; t < T; ++t) {
        const int buf = t & 1;
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
        if (t + 1 < T) store_tile(buf ^ 1);
        __syncthreads();
    }

    float* pout = a.part + (long)split * g.Ktot * g.Cout;
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

// gw[co][ci][tap] += scale * sum_s part[s][tap*Cin+ci][co]
__global__ void wgrad_reduce_kernel(const float* part, float* gw, int Ktot, int Cin, int Cout, int KK, int S, float scale) {
    const long total = (long)Ktot * Cout;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int k = 0; k < S; ++k) s += part[(long)k * total + i];
        const int co = (int)(i % Cout);
        const int mm = (int)(i / Cout);
        const int tap = mm / Cin;
        const int ci = mm - tap * Cin;
        float* dst = gw + ((long)co * Cin + ci) * KK + tap;
        *dst += scale * s;
    }
}

// canonical [Cout][Cin][KK] -> wf[tap*Cin+ci][Cout], wb[(KK-1-tap)*Cout+co][Cin]
__global__ void pack_weight_kernel(const float* w, float* wf, float* wb, int Cout, int Cin, int KK) {
    const long total = (long)Cout * Cin * KK;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        // iterate in wf order so the (larger-stride) writes of wf are coalesced
        const int co = (int)(i % Cout);
        const long r = i / Cout;
        const int ci = (int)(r % Cin);
        const int tap = (int)(r / Cin);
        const float v = w[((long)co * Cin + ci) * KK + tap];
        if (wf) wf[i] = v;
        if (wb) wb[((long)(KK - 1 - tap) * Cout + co) * Cin + ci] = v;
    }
}

// ---- host-side dispatch -----------------------------------------------------
struct TileCfg { int bm, bn; };

// pick the block tile: BN covers Cout with least padding, BM chosen so the grid
// fills the 256 CUs at >= ~2 workgroups each when the problem allows it.
static TileCfg pick_tile(long M, int Cout) {
    int bn = Cout > 64 ? 128 : (Cout > 32 ? 64 : 32);
    int bm = 128;
    if (bn == 128) {
        const long blocks128 = ((M + 127) / 128) * ((Cout + 127) / 128);
        if (blocks128 < 2 * cg::kNumCU) bm = 64;
    } else if (bn == 64) {
        const long blocks128 = ((M + 127) / 128) * ((Cout + 63) / 64);
        if (blocks128 < 2 * cg::kNumCU) bm = 64;
    } else {
        bm = 128;  // (128,32) only
    }
    return {bm, bn};
}

static int pick_splits(long tiles, long kiters) {
    // aim for >= 2 workgroups per CU, keep >= 8 K-iterations per split
    int s = 1;
    while (tiles * s < 2 * cg::kNumCU && kiters / (s * 2) >= 8 && s < 64) s *= 2;
    return s;
}

template <int BM, int BN, int WM, int WN>
static void launch_nn(const NNArgs& a, dim3 grid, hipStream_t st, bool veca, bool vecb) {
    if (veca && vecb) hipLaunchKernelGGL((igemm_nn_kernel<BM, BN, WM, WN, true, true>), grid, dim3(256), 0, st, a);
    else if (veca) hipLaunchKernelGGL((igemm_nn_kernel<BM, BN, WM, WN, true, false>), grid, dim3(256), 0, st, a);
    else if (vecb) hipLaunchKernelGGL((igemm_nn_kernel<BM, BN, WM, WN, false, true>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((igemm_nn_kernel<BM, BN, WM, WN, false, false>), grid, dim3(256), 0, st, a);
}

template <int BM, int BN, int WM, int WN>
static void launch_tn(const TNArgs& a, dim3 grid, hipStream_t st, bool veca, bool vecb) {
    if (veca && vecb) hipLaunchKernelGGL((igemm_tn_kernel<BM, BN, WM, WN, true, true>), grid, dim3(256), 0, st, a);
    else if (veca) hipLaunchKernelGGL((igemm_tn_kernel<BM, BN, WM, WN, true, false>), grid, dim3(256), 0, st, a);
    else if (vecb) hipLaunchKernelGGL((igemm_tn_kernel<BM, BN, WM, WN, false, true>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((igemm_tn_kernel<BM, BN, WM, WN, false, false>), grid, dim3(256), 0, st, a);
}

static int make_geom(ConvGeom& g, int N, int Hp, int Wp, int Cin, int Cout, int kH, int kW, int padH, int padW, int ups) {
    if (N <= 0 || Hp <= 0 || Wp <= 0 || Cin <= 0 || Cout <= 0 || kH <= 0 || kW <= 0 || padH < 0 || padW < 0 ||
        (ups != 0 && ups != 1))
        return cg::fail("conv2d: bad geometry N=%d H=%d W=%d Cin=%d Cout=%d k=%dx%d pad=%d,%d ups=%d", N, Hp, Wp, Cin,
                        Cout, kH, kW, padH, padW, ups);
    g.Hp = Hp; g.Wp = Wp;
    g.Hl = Hp << ups; g.Wl = Wp << ups;
    const int Ho = g.Hl + 2 * padH - kH + 1, Wo = g.Wl + 2 * padW - kW + 1;
    if (Ho <= 0 || Wo <= 0) return cg::fail("conv2d: empty output %dx%d", Ho, Wo);
    const long M = (long)N * Ho * Wo;
    if (M > 0x7fffffffL || (long)N * Hp * Wp > 0x7fffffffL) return cg::fail("conv2d: pixel count overflows int32");
    g.M = (int)M; g.HoWo = Ho * Wo; g.Wo = Wo;
    g.Cin = Cin; g.Cout = Cout; g.kW = kW; g.padH = padH; g.padW = padW; g.ups = ups;
    g.Ktot = kH * kW * Cin;
    return 0;
}

struct NNPlan { TileCfg tc; int splits; int kchunk; };
static NNPlan plan_nn(const ConvGeom& g) {
    NNPlan p;
    p.tc = pick_tile(g.M, g.Cout);
    const long tiles = (long)cg::cdiv(g.M, p.tc.bm) * cg::cdiv(g.Cout, p.tc.bn);
    const long kiters = cg::cdiv(g.Ktot, BK);
    p.splits = pick_splits(tiles, kiters);
    p.kchunk = cg::cdiv(kiters, p.splits) * BK;
    p.splits = cg::cdiv(g.Ktot, p.kchunk);
    return p;
}

struct TNPlan { TileCfg tc; int splits; int pchunk; };
static TNPlan plan_tn(const ConvGeom& g) {
    TNPlan p;
    // rows of dW = Ktot (tap,ci); cols = Cout
    int bn = g.Cout > 64 ? 128 : (g.Cout > 32 ? 64 : 32);
    int bm = g.Ktot > 64 ? 128 : 64;
    if (bn == 32) bm = 128;
    p.tc = {bm, bn};
    const long tiles = (long)cg::cdiv(g.Ktot, bm) * cg::cdiv(g.Cout, bn);
    const long piters = cg::cdiv(g.M, BK);
    int s = 1;
    while (tiles * s < 3 * cg::kNumCU && piters / (s * 2) >= 8 && s < 256) s *= 2;
    p.pchunk = cg::cdiv(piters, s) * BK;
    p.splits = cg::cdiv(g.M, p.pchunk);
    return p;
}

}  // namespace

extern "C" {

size_t cg_conv2d_workspace_bytes(int N, int Hp, int Wp, int Cin, int Cout, int kH, int kW, int padH, int padW, int ups) {
    ConvGeom g;
    if (make_geom(g, N, Hp, Wp, Cin, Cout, kH, kW, padH, padW, ups)) return 0;
    NNPlan p = plan_nn(g);
    return p.splits > 1 ? (size_t)p.splits * g.M * g.Cout * sizeof(float) : 0;
}

int cg_conv2d_forward(void* stream, const float* x, const float* wpk, const float* bias, float* y, int N, int Hp, int Wp,
                      int Cin, int Cout, int kH, int kW, int padH, int padW, int ups, void* ws, size_t ws_bytes) {
    CG_REQUIRE(x && wpk && y, "cg_conv2d_forward: null pointer");
    NNArgs a;
    if (make_geom(a.g, N, Hp, Wp, Cin, Cout, kH, kW, padH, padW, ups)) return 1;
    const ConvGeom& g = a.g;
    NNPlan p = plan_nn(g);
    const size_t need = p.splits > 1 ? (size_t)p.splits * g.M * g.Cout * sizeof(float) : 0;
    CG_REQUIRE(need == 0 || (ws && ws_bytes >= need), "cg_conv2d_forward: workspace too small (%zu < %zu)", ws_bytes, need);
    a.x = x; a.w = wpk; a.bias = bias;
    a.y = p.splits > 1 ? (float*)ws : y;
    a.kchunk = p.kchunk;
    a.split_stride = (long)g.M * g.Cout;
    const bool veca = (Cin % 4 == 0) && ((uintptr_t)x % 16 == 0);
    const bool vecb = (Cout % 4 == 0) && ((uintptr_t)wpk % 16 == 0);
    hipStream_t st = cg::S(stream);
    dim3 grid(cg::cdiv(g.M, p.tc.bm) * cg::cdiv(g.Cout, p.tc.bn), p.splits);
    if (p.tc.bm == 128 && p.tc.bn == 128) launch_nn<128, 128, 2, 2>(a, grid, st, veca, vecb);
    else if (p.tc.bm == 64 && p.tc.bn == 128) launch_nn<64, 128, 2, 2>(a, grid, st, veca, vecb);
    else if (p.tc.bm == 128 && p.tc.bn == 64) launch_nn<128, 64, 2, 2>(a, grid, st, veca, vecb);
    else if (p.tc.bm == 64 && p.tc.bn == 64) launch_nn<64, 64, 2, 2>(a, grid, st, veca, vecb);
    else launch_nn<128, 32, 4, 1>(a, grid, st, veca, vecb);
    CG_LAUNCH_CHECK();
    if (p.splits > 1) {
        const long MN = (long)g.M * g.Cout;
        hipLaunchKernelGGL(nn_splitk_reduce_kernel, dim3(cg::ew_grid(MN)), dim3(256), 0, st, (const float*)ws, bias, y, MN,
                           g.Cout, p.splits);
        CG_LAUNCH_CHECK();
    }
    return 0;
}

size_t cg_conv2d_wgrad_workspace_bytes(int N, int Hp, int Wp, int Cin, int Cout, int kH, int kW, int padH, int padW,
                                       int ups) {
    ConvGeom g;
    if (make_geom(g, N, Hp, Wp, Cin, Cout, kH, kW, padH, padW, ups)) return 0;
    TNPlan p = plan_tn(g);
    return (size_t)p.splits * g.Ktot * g.Cout * sizeof(float);
}

int cg_conv2d_wgrad(void* stream, const float* x, const float* dy, float* gw, int N, int Hp, int Wp, int Cin, int Cout,
                    int kH, int kW, int padH, int padW, int ups, float scale, void* ws, size_t ws_bytes) {
    CG_REQUIRE(x && dy && gw, "cg_conv2d_wgrad: null pointer");
    TNArgs a;
    if (make_geom(a.g, N, Hp, Wp, Cin, Cout, kH, kW, padH, padW, ups)) return 1;
    const ConvGeom& g = a.g;
    TNPlan p = plan_tn(g);
    const size_t need = (size_t)p.splits * g.Ktot * g.Cout * sizeof(float);
    CG_REQUIRE(ws && ws_bytes >= need, "cg_conv2d_wgrad: workspace too small (%zu < %zu)", ws_bytes, need);
    a.x = x; a.dy = dy; a.part = (float*)ws; a.pchunk = p.pchunk;
    const bool veca = (Cin % 4 == 0) && ((uintptr_t)x % 16 == 0);
    const bool vecb = (Cout % 4 == 0) && ((uintptr_t)dy % 16 == 0);
    hipStream_t st = cg::S(stream);
    dim3 grid(cg::cdiv(g.Ktot, p.tc.bm) * cg::cdiv(g.Cout, p.tc.bn), p.splits);
    if (p.tc.bm == 128 && p.tc.bn == 128) launch_tn<128, 128, 2, 2>(a, grid, st, veca, vecb);
    else if (p.tc.bm == 64 && p.tc.bn == 128) launch_tn<64, 128, 2, 2>(a, grid, st, veca, vecb);
    else if (p.tc.bm == 128 && p.tc.bn == 64) launch_tn<128, 64, 2, 2>(a, grid, st, veca, vecb);
    else if (p.tc.bm == 64 && p.tc.bn == 64) launch_tn<64, 64, 2, 2>(a, grid, st, veca, vecb);
    else launch_tn<128, 32, 4, 1>(a, grid, st, veca, vecb);
    CG_LAUNCH_CHECK();
    const long total = (long)g.Ktot * g.Cout;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(cg::ew_grid(total)), dim3(256), 0, st, (const float*)ws, gw, g.Ktot,
                       g.Cin, g.Cout, kH * kW, p.splits, scale);
    CG_LAUNCH_CHECK();
    return 0;
}

int cg_pack_conv_weight(void* stream, const float* w, float* wf, float* wb, int Cout, int Cin, int kH, int kW) {
    CG_REQUIRE(w && (wf || wb), "cg_pack_conv_weight: null pointer");
    CG_REQUIRE(Cout > 0 && Cin > 0 && kH > 0 && kW > 0, "cg_pack_conv_weight: bad dims");
    const long total = (long)Cout * Cin * kH * kW;
    hipLaunchKernelGGL(pack_weight_kernel, dim3(cg::ew_grid(total)), dim3(256), 0, cg::S(stream), w, wf, wb, Cout, Cin,
                       kH * kW);
    CG_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
