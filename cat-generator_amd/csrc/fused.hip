// Fused memory-bound chains of the G+D step: what Torch7 runs as separate modules, one pass over HBM each, runs
// here as one kernel per chain.  Arithmetic per element is exactly that of the unfused kernels in ops.hip (same
// operation order), so switching the fusion on or off changes no bit of the forward results.
//
//   act -> pool(2x2) -> spatial dropout        models.lua:649-651 (PReLU, AvgPool, SpatialDropout .2),
//                                              :656-658 / :682-684 (PReLU, MaxPool, SpatialDropout .2),
//                                              :847-848 (LeakyReLU, AvgPool in the localisation nets)
//   batch-norm (training) -> PReLU             models.lua:207-208, 213-214, 219-220 (G32up-c), :145-146,150-151 (G32up)
//
// The batch statistics arrive as per-tile partial sums from the producing GEMM's epilogue (gemm.hip / winograd.hip)
// and are folded by cg_bn_stats_finalize, so the convolution output is read once (normalise + PReLU) instead of
// three times (statistics, normalise, PReLU).
#include "common.h"
#include <stdlib.h>
#include <algorithm>

namespace {
using cg::block_sum_256;

#define V4_LOOP(i, n4) \
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < (n4); i += (long)gridDim.x * blockDim.x)
__device__ __forceinline__ float4 ldv(const float* p, long i) { return reinterpret_cast<const float4*>(p)[i]; }
__device__ __forceinline__ void stv(float* p, long i, float4 v) { reinterpret_cast<float4*>(p)[i] = v; }
__device__ __forceinline__ float4 ldc(const float* p, int c4) { return make_float4(p[4 * c4], p[4 * c4 + 1], p[4 * c4 + 2], p[4 * c4 + 3]); }

struct AlphaTab { const float* a0; const float* a1; const float* a2; const float* a3; };
__device__ __forceinline__ const float* tab_sel(const AlphaTab& t, int g) { return g == 0 ? t.a0 : (g == 1 ? t.a1 : (g == 2 ? t.a2 : t.a3)); }

// act: 0 identity, 1 PReLU (x > 0 ? x : a x), 2 LeakyReLU (x >= 0 ? x : a x)   [ops.hip prelu_fwd / lrelu_fwd]
__device__ __forceinline__ float actf(int act, float v, float a) {
    return act == 0 ? v : (act == 1 ? (v > 0.f ? v : a * v) : (v >= 0.f ? v : a * v));
}
__device__ __forceinline__ float4 act4(int act, float4 v, float a) {
    return make_float4(actf(act, v.x, a), actf(act, v.y, a), actf(act, v.z, a), actf(act, v.w, a));
}
// d(act)/dx applied to an incoming gradient d
__device__ __forceinline__ float dactf(int act, float x, float d, float a) {
    return act == 0 ? d : (act == 1 ? (x > 0.f ? d : a * d) : (x >= 0.f ? d : a * d));
}

// ---------------------------------------------------------------------------------------------------------------
// y[n, oy, ox, c] = mask[n, c] * pool2x2( act(x[n, 2oy.., 2ox.., c]) )       x: [G*NG, H, W, C] NHWC, C % 4 == 0
// Sample n belongs to group n / NG (the stacked batch of D32_st3's identical branches, each with its own PReLU slope).
// ---------------------------------------------------------------------------------------------------------------
template <bool MAX>
__global__ __launch_bounds__(256) void act_pool2_fwd_k(const float* __restrict__ x, float* __restrict__ y,
                                                       const float* __restrict__ mask, AlphaTab at, int act, float slope,
                                                       int NG, long total4, int H, int W, int C4) {
    const int Ho = H >> 1, Wo = W >> 1;
    V4_LOOP(i, total4) {
        const int c4 = (int)(i % C4);
        long r = i / C4;
        const int ox = (int)(r % Wo); r /= Wo;
        const int oy = (int)(r % Ho);
        const long n = r / Ho;
        const float a = act == 1 ? *tab_sel(at, (int)(n / NG)) : slope;
        const long b = ((n * H + 2 * oy) * W + 2 * ox) * C4 + c4;
        const float4 v00 = act4(act, ldv(x, b), a), v01 = act4(act, ldv(x, b + C4), a);
        const float4 v10 = act4(act, ldv(x, b + (long)W * C4), a), v11 = act4(act, ldv(x, b + (long)W * C4 + C4), a);
        float4 o;
#define POOL1(f)                                                        \
        if (MAX) {                                                      \
            float m = v00.f;                                            \
            if (v01.f > m) m = v01.f;                                   \
            if (v10.f > m) m = v10.f;                                   \
            if (v11.f > m) m = v11.f;                                   \
            o.f = m;                                                    \
        } else {                                                        \
            o.f = (v00.f + v01.f + v10.f + v11.f) * 0.25f;              \
        }
        POOL1(x) POOL1(y) POOL1(z) POOL1(w)
#undef POOL1
        if (mask) {
            const float4 mk = ldc(mask, (int)(n * C4 + c4));
            o.x *= mk.x; o.y *= mk.y; o.z *= mk.z; o.w *= mk.w;
        }
        stv(y, i, o);
    }
}

// dx = act'(x) * pool2x2^T( mask * gy );  PReLU: gpart[group][block] = sum_{x <= 0} x * d  (d = gradient w.r.t. act(x)).
// grid (blocks per group, groups): a workgroup never straddles two groups, so its partial belongs to one slope.
template <bool MAX>
__global__ __launch_bounds__(256) void act_pool2_bwd_k(const float* __restrict__ x, const float* __restrict__ gy,
                                                       const float* __restrict__ mask, float* __restrict__ dx, AlphaTab at,
                                                       int act, float slope, int NG, long per_group4, int H, int W, int C4,
                                                       double* __restrict__ gpart) {
    __shared__ double sh[4];
    const int Ho = H >> 1, Wo = W >> 1;
    const int grp = blockIdx.y;
    const float a = act == 1 ? *tab_sel(at, grp) : slope;
    const long base4 = (long)grp * per_group4;      // first pooled float4 of this group
    float s = 0.f;
    V4_LOOP(j, per_group4) {
        const long i = base4 + j;
        const int c4 = (int)(i % C4);
        long r = i / C4;
        const int ox = (int)(r % Wo); r /= Wo;
        const int oy = (int)(r % Ho);
        const long n = r / Ho;
        float4 g = ldv(gy, i);
        if (mask) {
            const float4 mk = ldc(mask, (int)(n * C4 + c4));
            g.x *= mk.x; g.y *= mk.y; g.z *= mk.z; g.w *= mk.w;
        }
        const long b = ((n * H + 2 * oy) * W + 2 * ox) * C4 + c4;
        const long o01 = C4, o10 = (long)W * C4, o11 = (long)W * C4 + C4;
        const float4 x00 = ldv(x, b), x01 = ldv(x, b + o01), x10 = ldv(x, b + o10), x11 = ldv(x, b + o11);
        float4 d00, d01, d10, d11;
#define BWD1(f)                                                                                         \
        {                                                                                               \
            float e00, e01, e10, e11;                                                                   \
            if (MAX) {   /* first maximum of act(x) in scan order takes the gradient (maxpool2_bwd_k) */ \
                const float v00 = actf(act, x00.f, a), v01 = actf(act, x01.f, a);                      \
                const float v10 = actf(act, x10.f, a), v11 = actf(act, x11.f, a);                      \
                int arg = 0; float m = v00;                                                             \
                if (v01 > m) { m = v01; arg = 1; }                                                      \
                if (v10 > m) { m = v10; arg = 2; }                                                      \
                if (v11 > m) { m = v11; arg = 3; }                                                      \
                e00 = arg == 0 ? g.f : 0.f; e01 = arg == 1 ? g.f : 0.f;                                 \
                e10 = arg == 2 ? g.f : 0.f; e11 = arg == 3 ? g.f : 0.f;                                 \
            } else {                                                                                    \
                e00 = e01 = e10 = e11 = g.f * 0.25f;                                                    \
            }                                                                                           \
            d00.f = dactf(act, x00.f, e00, a); d01.f = dactf(act, x01.f, e01, a);                       \
            d10.f = dactf(act, x10.f, e10, a); d11.f = dactf(act, x11.f, e11, a);                       \
            if (act == 1)                                                                               \
                s += (x00.f <= 0.f ? x00.f * e00 : 0.f) + (x01.f <= 0.f ? x01.f * e01 : 0.f) +          \
                     (x10.f <= 0.f ? x10.f * e10 : 0.f) + (x11.f <= 0.f ? x11.f * e11 : 0.f);           \
        }
        BWD1(x) BWD1(y) BWD1(z) BWD1(w)
#undef BWD1
        stv(dx, b, d00); stv(dx, b + o01, d01); stv(dx, b + o10, d10); stv(dx, b + o11, d11);
    }
    if (!gpart) return;
    const double t = block_sum_256((double)s, sh);
    if (threadIdx.x == 0) gpart[(long)grp * gridDim.x + blockIdx.x] = t;
}

// nn.Concat(2) over <= 4 branches in one launch (models.lua:688-692): dst[m][off_b + c] = src_b[m][c]; SPLIT: the reverse
// (the gradient slices handed to the branches); channel counts in float4 units.
struct Cat4 { float* p0; float* p1; float* p2; float* p3; int c0, c1, c2, c3; };
template <bool SPLIT>
__global__ __launch_bounds__(256) void concat4_v4k(Cat4 b, float* __restrict__ whole, long total4, int Ct4) {
    V4_LOOP(i, total4) {
        const int c = (int)(i % Ct4);
        const long m = i / Ct4;
        float* p; int cl, cw;
        if (c < b.c0) { p = b.p0; cl = c; cw = b.c0; }
        else if (c < b.c0 + b.c1) { p = b.p1; cl = c - b.c0; cw = b.c1; }
        else if (c < b.c0 + b.c1 + b.c2) { p = b.p2; cl = c - b.c0 - b.c1; cw = b.c2; }
        else { p = b.p3; cl = c - b.c0 - b.c1 - b.c2; cw = b.c3; }
        if (SPLIT) stv(p, m * cw + cl, ldv(whole, i));
        else stv(whole, i, ldv(p, m * cw + cl));
    }
}
// out = ((a + b) + c) + d: the accumulation order of nn.Concat's gradInput (copy, then one add per further branch)
__global__ __launch_bounds__(256) void sum4_v4k(const float* a, const float* b, const float* c, const float* d, float* out, int n,
                                               long n4) {
    V4_LOOP(i, n4) {
        float4 s = ldv(a, i);
        const float4 u = ldv(b, i);
        s.x += u.x; s.y += u.y; s.z += u.z; s.w += u.w;
        if (n > 2) { const float4 v = ldv(c, i); s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
        if (n > 3) { const float4 v = ldv(d, i); s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
        stv(out, i, s);
    }
}

// nn.Concat(2) -> nn.SpatialDropout (training) in one pass.  DROPFWD: whole = concat * mask with the mask drawn in the kernel -
// element (n, c) of the [N][Ct] mask is draw offset + n*Ct + c of the counter stream, what rng_bernoulli_k would have produced -
// and stored by the thread that owns pixel 0 of the sample; !DROPFWD: the branches' gradient slices = whole * mask (mask read).
template <bool DROPFWD>
__global__ __launch_bounds__(256) void concat4_drop_v4k(Cat4 b, float* __restrict__ whole, float* __restrict__ noise, long total4, int Ct4,
                                                        long HW, float keep, float value, uint64_t seed, uint64_t offset,
                                                        const uint64_t* __restrict__ base) {
    if (DROPFWD && base) offset += *base;
    V4_LOOP(i, total4) {
        const int c = (int)(i % Ct4);
        const long m = i / Ct4;
        const long n = m / HW;
        float* p; int cl, cw;
        if (c < b.c0) { p = b.p0; cl = c; cw = b.c0; }
        else if (c < b.c0 + b.c1) { p = b.p1; cl = c - b.c0; cw = b.c1; }
        else if (c < b.c0 + b.c1 + b.c2) { p = b.p2; cl = c - b.c0 - b.c1; cw = b.c2; }
        else { p = b.p3; cl = c - b.c0 - b.c1 - b.c2; cw = b.c3; }
        const long mi = n * Ct4 + c;
        if (DROPFWD) {
            const uint64_t ctr = offset + (uint64_t)mi * 4u;
            float4 mk;
            mk.x = cg::u01(seed, ctr) < keep ? value : 0.f;
            mk.y = cg::u01(seed, ctr + 1) < keep ? value : 0.f;
            mk.z = cg::u01(seed, ctr + 2) < keep ? value : 0.f;
            mk.w = cg::u01(seed, ctr + 3) < keep ? value : 0.f;
            if (m == n * HW) stv(noise, mi, mk);
            float4 v = ldv(p, m * cw + cl);
            v.x *= mk.x; v.y *= mk.y; v.z *= mk.z; v.w *= mk.w;
            stv(whole, i, v);
        } else {
            const float4 mk = ldv(noise, mi);
            float4 v = ldv(whole, i);
            v.x *= mk.x; v.y *= mk.y; v.z *= mk.z; v.w *= mk.w;
            stv(p, m * cw + cl, v);
        }
    }
}

// The discriminator's head nn.Dropout -> nn.Linear(F, O <= 4) -> nn.Sigmoid (models.lua:699-701), one wave per sample:
// mask drawn in place (element n*F + f of the [N][F] mask = draw offset + n*F + f), xd = x * mask, z = xd . w[o] + b[o], p = sigmoid(z)
template <int O>
__global__ __launch_bounds__(256) void head_fwd_k(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                                  float* __restrict__ noise, float* __restrict__ xd, float* __restrict__ z,
                                                  float* __restrict__ p, int N, int F, float keep, float value, uint64_t seed,
                                                  uint64_t offset, const uint64_t* __restrict__ base) {
    if (base) offset += *base;
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    float acc[O];
#pragma unroll
    for (int o = 0; o < O; ++o) acc[o] = 0.f;
    for (int f = lane; f < F; f += 64) {
        const long i = (long)n * F + f;
        const float mk = cg::u01(seed, offset + (uint64_t)i) < keep ? value : 0.f;
        const float v = x[i] * mk;
        noise[i] = mk; xd[i] = v;
#pragma unroll
        for (int o = 0; o < O; ++o) acc[o] += v * w[(long)o * F + f];
    }
#pragma unroll
    for (int o = 0; o < O; ++o) {
        const float t = cg::wave_sum(acc[o]);
        if (lane == 0) {
            const float zz = t + (b ? b[o] : 0.f);
            z[(long)n * O + o] = zz;
            p[(long)n * O + o] = 1.f / (1.f + expf(-zz));
        }
    }
}

// backward of the head from dp = dLoss/dp:  gz = dp p (1 - p);  gw[o][f] += scale sum_n gz[n][o] xd[n][f];  gb[o] += scale sum_n gz[n][o];
// gx[n][f] = mask[n][f] sum_o gz[n][o] w[o][f].  Workgroup = 64 features x 4 sample lanes; a feature's four partial sums are added
// in lane order, every sum runs in sample order: the same bits on every run.
template <int O>
__global__ __launch_bounds__(256) void head_bwd_k(const float* __restrict__ dp, const float* __restrict__ p, const float* __restrict__ xd,
                                                  const float* __restrict__ noise, const float* __restrict__ w, float* __restrict__ gx,
                                                  float* gw, float* gb, int N, int F, float scale) {
    extern __shared__ float sh[];            // gz[N][O], then part[4][64][O]
    float* gz = sh;
    float* part = sh + (size_t)N * O;
    for (int i = threadIdx.x; i < N * O; i += 256) { const float v = p[i]; gz[i] = dp[i] * (1.f - v) * v; }
    __syncthreads();
    const int fl = threadIdx.x & 63, nl = threadIdx.x >> 6;
    const int f = blockIdx.x * 64 + fl;
    float acc[O], wv[O];
#pragma unroll
    for (int o = 0; o < O; ++o) { acc[o] = 0.f; wv[o] = f < F ? w[(long)o * F + f] : 0.f; }
    if (f < F) {
#pragma unroll 4
        for (int n = nl; n < N; n += 4) {
            const long i = (long)n * F + f;
            const float h = gw ? xd[i] : 0.f, mk = noise[i];
            float g = 0.f;
#pragma unroll
            for (int o = 0; o < O; ++o) { const float t = gz[n * O + o]; acc[o] += t * h; g += t * wv[o]; }
            gx[i] = g * mk;
        }
    }
    if (!gw) return;
#pragma unroll
    for (int o = 0; o < O; ++o) part[(nl * 64 + fl) * O + o] = acc[o];
    __syncthreads();
    if (nl == 0 && f < F) {
#pragma unroll
        for (int o = 0; o < O; ++o) {
            const float t = ((part[fl * O + o] + part[(64 + fl) * O + o]) + part[(128 + fl) * O + o]) + part[(192 + fl) * O + o];
            gw[(long)o * F + f] += scale * t;
        }
    }
    if (gb && blockIdx.x == 0 && threadIdx.x < O) {
        float t = 0.f;
        for (int n = 0; n < N; ++n) t += gz[n * O + threadIdx.x];
        gb[threadIdx.x] += scale * t;
    }
}

struct GalphaTab { float* g0; float* g1; float* g2; float* g3; };
// galpha[group] += scale * sum_b gpart[group][b]: one workgroup per group, fixed summation order
__global__ __launch_bounds__(256) void galpha_groups_reduce_k(const double* gpart, int nparts, GalphaTab gt, float scale) {
    __shared__ double sh[4];
    const int grp = blockIdx.x;
    double s = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 256) s += gpart[(long)grp * nparts + i];
    const double t = block_sum_256(s, sh);
    float* ga = grp == 0 ? gt.g0 : (grp == 1 ? gt.g1 : (grp == 2 ? gt.g2 : gt.g3));
    if (threadIdx.x == 0 && ga) *ga += scale * (float)t;
}

// PReLU backward over a stacked batch of G equal groups, each with its own slope (D32_st3's identical branches):
// dx = x > 0 ? dy : a_g dy;  gpart[g][block] = sum_{x <= 0} x dy     (prelu_bwd_v4k of ops.hip, one launch for G modules)
__global__ __launch_bounds__(256) void prelu_bwd_groups_v4k(const float* __restrict__ x, const float* __restrict__ dy, AlphaTab at,
                                                            float* __restrict__ dx, long per_group4, double* __restrict__ gpart) {
    __shared__ double sh[4];
    const int grp = blockIdx.y;
    const float a = *tab_sel(at, grp);
    const long base4 = (long)grp * per_group4;
    float s = 0.f;
    V4_LOOP(j, per_group4) {
        const long i = base4 + j;
        const float4 v = ldv(x, i), g = ldv(dy, i);
        float4 o;
        o.x = v.x > 0.f ? g.x : a * g.x; o.y = v.y > 0.f ? g.y : a * g.y;
        o.z = v.z > 0.f ? g.z : a * g.z; o.w = v.w > 0.f ? g.w : a * g.w;
        stv(dx, i, o);
        s += (v.x <= 0.f ? v.x * g.x : 0.f) + (v.y <= 0.f ? v.y * g.y : 0.f) + (v.z <= 0.f ? v.z * g.z : 0.f) +
             (v.w <= 0.f ? v.w * g.w : 0.f);
    }
    if (!gpart) return;
    const double t = block_sum_256((double)s, sh);
    if (threadIdx.x == 0) gpart[(long)grp * gridDim.x + blockIdx.x] = t;
}

// ---------------------------------------------------------------------------------------------------------------
// batch statistics from GEMM-epilogue partials: sums[c] = sum_r part[r][0][c], sums[C + c] = sum_r part[r][1][c] (fp64)
// block = 16 columns x 16 row lanes over the 2C columns of the [P][2C] partial matrix; fixed order -> deterministic
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bn_stats_finalize_k(const float* __restrict__ part, int P, int C, double* __restrict__ sums) {
    __shared__ double sh[16][17];
    const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int col = blockIdx.x * 16 + cl;          // 0..2C-1: (stat, channel) = (col / C, col % C)
    double s = 0.0;
    if (col < 2 * C) {
        const int st = col / C, c = col - st * C;
        const float* p = part + (long)st * C + c;
#pragma unroll 4
        for (int r = rl; r < P; r += 16) s += (double)p[(long)r * 2 * C];
    }
    sh[rl][cl] = s;
    __syncthreads();
    if (rl == 0 && col < 2 * C) {
        double t = 0.0;
#pragma unroll
        for (int r = 0; r < 16; ++r) t += sh[r][cl];
        sums[col] = t;
    }
}

// The running-statistics half of bn_act_fwd_v4k / bn_forward alone, from the same fp64 sums and with the same arithmetic: a pass
// that ran with running_mean == NULL (two generator forwards side by side, adversarial.py) applies its update afterwards, in the
// reference's order (nn.SpatialBatchNormalization updateOutput, models.lua:207).
__global__ __launch_bounds__(256) void bn_running_update_k(const double* __restrict__ sums, double count, float momentum, float* running_mean,
                                                           float* running_var, int C) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const double mean = sums[c] / count;
    double var = sums[C + c] / count - mean * mean;
    if (var < 0.0) var = 0.0;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
    const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
}

// training-mode batch-norm + PReLU in one pass; every thread owns one channel quad (the grid stride is a multiple of C/4),
// derives its mean / invstd from the fp64 sums exactly as bn_prepare_k does; workgroup 0 also writes save_mean /
// save_invstd and moves the running statistics.
__global__ __launch_bounds__(256) void bn_act_fwd_v4k(const float* __restrict__ x, float* __restrict__ y,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      const double* __restrict__ sums, double count, float eps, float momentum,
                                                      float* running_mean, float* running_var, float* save_mean,
                                                      float* save_invstd, const float* alpha, long n4, int C4) {
    const int C = C4 * 4;
    if (blockIdx.x == 0) {
        for (int c = threadIdx.x; c < C; c += 256) {
            const double mean = sums[c] / count;
            double var = sums[C + c] / count - mean * mean;
            if (var < 0.0) var = 0.0;
            save_mean[c] = (float)mean;
            save_invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
            if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
            if (running_var) {
                const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
                running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
            }
        }
    }
    const long i0 = blockIdx.x * 256L + threadIdx.x;
    if (i0 >= n4) return;
    const int c4 = (int)(i0 % C4);
    float mu[4], is[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = 4 * c4 + k;
        const double mean = sums[c] / count;
        double var = sums[C + c] / count - mean * mean;
        if (var < 0.0) var = 0.0;
        mu[k] = (float)mean;
        is[k] = (float)(1.0 / sqrt(var + (double)eps));
    }
    const float4 g = ldc(gamma, c4), b = ldc(beta, c4);
    const float a = alpha ? *alpha : 1.f;
    for (long i = i0; i < n4; i += (long)gridDim.x * 256) {
        float4 v = ldv(x, i);
        v.x = (v.x - mu[0]) * is[0] * g.x + b.x; v.y = (v.y - mu[1]) * is[1] * g.y + b.y;
        v.z = (v.z - mu[2]) * is[2] * g.z + b.z; v.w = (v.w - mu[3]) * is[3] * g.w + b.w;
        if (alpha) {
            v.x = v.x > 0.f ? v.x : a * v.x; v.y = v.y > 0.f ? v.y : a * v.y;
            v.z = v.z > 0.f ? v.z : a * v.z; v.w = v.w > 0.f ? v.w : a * v.w;
        }
        stv(y, i, v);
    }
}

// backward statistics of PReLU(BN(x)) from x and the gradient dy w.r.t. the PReLU output:
//   u = (x - mean) * invstd * gamma + beta (recomputed, bit-equal to the forward);  d = u > 0 ? dy : alpha * dy
//   sums[c] = sum d, sums[C + c] = sum d * xhat, sums[2C] = sum_{u <= 0} u * dy   (fp64; block = QB channel quads x RL row
//   lanes as colreduce4_k; per-workgroup partials go to the stream's scratch and are added in the fixed two-level order of
//   cg::ColTree: no floating-point atomics, the same bits on every run)
template <int QB>
__global__ __launch_bounds__(256) void bn_act_bwd_stats_k(const float* __restrict__ x, const float* __restrict__ dy,
                                                          const float* __restrict__ mean, const float* __restrict__ invstd,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          const float* alpha, long M, int C, long rows_per_block,
                                                          double* __restrict__ sums, cg::ColTree tree) {
    constexpr int RL = 256 / QB;
    double* part = tree.part;
    double* part_alpha = tree.extra;
    __shared__ double sh[2][RL][QB * 4 + 2];
    __shared__ double shg[4];
    const int ql = threadIdx.x % QB, rl = threadIdx.x / QB;
    const int q = blockIdx.x * QB + ql;
    const int cq = C >> 2;
    const long r0 = (long)blockIdx.y * rows_per_block;
    const long r1 = min(M, r0 + rows_per_block);
    double a1[4] = {0.0, 0.0, 0.0, 0.0}, a2[4] = {0.0, 0.0, 0.0, 0.0};
    double ga = 0.0;
    const float al = alpha ? *alpha : 1.f;
    if (q < cq) {
        const float4 mu = ldc(mean, q), is = ldc(invstd, q), g = ldc(gamma, q), b = ldc(beta, q);
        for (long rb = r0 + rl; rb < r1; rb += 64L * RL) {
            const long re = min(r1, rb + 64L * RL);
            float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
            float sg = 0.f;
#define ST1(f)                                                          \
                {                                                       \
                    const float xh = (v.f - mu.f) * is.f;               \
                    const float u = xh * g.f + b.f;                     \
                    const float dd = (!alpha || u > 0.f) ? d.f : al * d.f; \
                    s1.f += dd; s2.f += dd * xh;                        \
                    if (alpha && u <= 0.f) sg += u * d.f;               \
                }
            // Batches of UB rows: all 2 UB loads of a batch are issued before the first value is used.  Beside another queue's GEMM this
            // kernel gets ONE workgroup per CU at best, so the bytes in flight per wave decide its rate (round 5; same summation order)
            constexpr int UB = 8;
            long r = rb;
            for (; r + (UB - 1) * RL < re; r += UB * RL) {
                float4 vv[UB], dv[UB];
#pragma unroll
                for (int k = 0; k < UB; ++k) { const long i = (r + (long)k * RL) * cq + q; vv[k] = ldv(x, i); dv[k] = ldv(dy, i); }
#pragma unroll
                for (int k = 0; k < UB; ++k) { const float4 v = vv[k], d = dv[k]; ST1(x) ST1(y) ST1(z) ST1(w) }
            }
            for (; r < re; r += RL) {
                const long i = r * cq + q;
                const float4 v = ldv(x, i), d = ldv(dy, i);
                ST1(x) ST1(y) ST1(z) ST1(w)
            }
#undef ST1
            a1[0] += s1.x; a1[1] += s1.y; a1[2] += s1.z; a1[3] += s1.w;
            a2[0] += s2.x; a2[1] += s2.y; a2[2] += s2.z; a2[3] += s2.w;
            ga += sg;
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) { sh[0][rl][ql * 4 + k] = a1[k]; sh[1][rl][ql * 4 + k] = a2[k]; }
    __syncthreads();
    for (int idx = threadIdx.x; idx < QB * 4 * 2; idx += 256) {
        const int which = idx / (QB * 4), cl = idx % (QB * 4);
        const int c = blockIdx.x * QB * 4 + cl;
        if (c >= C) continue;
        double t = 0.0;
#pragma unroll
        for (int r = 0; r < RL; ++r) t += sh[which][r][cl];
        cg::st_agent(&part[(size_t)blockIdx.y * (2 * C) + which * C + c], t);
    }
    {
        const double t = alpha ? block_sum_256(ga, shg) : 0.0;
        if (threadIdx.x == 0) cg::st_agent(&part_alpha[blockIdx.y * gridDim.x + blockIdx.x], t);
    }
    if (cg::col_tree_finish(tree, (int)blockIdx.y, (int)gridDim.y, gridDim.x, 2 * C, sums)) {
        double a = 0.0;
        for (int i = threadIdx.x; i < (int)(gridDim.x * gridDim.y); i += 256) a += cg::ld_agent(&part_alpha[i]);
        a = block_sum_256(a, shg);
        if (threadIdx.x == 0) sums[2 * C] = a;
    }
}

// dx = gamma * invstd * (d - sum(d)/count - xhat * sum(d xhat)/count) with d as above; workgroup 0 also accumulates
// the parameter gradients: ggamma += scale * local sum(d xhat), gbeta += scale * local sum(d), galpha += scale * local sum
__global__ __launch_bounds__(256) void bn_act_bwd_v4k(const float* __restrict__ x, const float* __restrict__ dy,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      const float* __restrict__ mean, const float* __restrict__ invstd,
                                                      const float* alpha, const double* __restrict__ sums, double count,
                                                      const double* __restrict__ local_sums, long n4, int C4,
                                                      float* __restrict__ dx, float* ggamma, float* gbeta, float* galpha,
                                                      float scale) {
    const int C = C4 * 4;
    if (blockIdx.x == 0) {
        for (int c = threadIdx.x; c < C; c += 256) {
            if (gbeta) gbeta[c] += scale * (float)local_sums[c];
            if (ggamma) ggamma[c] += scale * (float)local_sums[C + c];
        }
        if (galpha && alpha) {
            if (threadIdx.x == 0) *galpha += scale * (float)local_sums[2 * C];
        }
    }
    const long i0 = blockIdx.x * 256L + threadIdx.x;
    if (i0 >= n4) return;
    const int c4 = (int)(i0 % C4);
    const float4 g = ldc(gamma, c4), b = ldc(beta, c4), mu = ldc(mean, c4), is = ldc(invstd, c4);
    float m1[4], m2[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        m1[k] = (float)(sums[4 * c4 + k] / count);
        m2[k] = (float)(sums[C + 4 * c4 + k] / count);
    }
    const float al = alpha ? *alpha : 1.f;
    for (long i = i0; i < n4; i += (long)gridDim.x * 256) {
        const float4 v = ldv(x, i), d = ldv(dy, i);
        float4 o;
#define BW1(f, k)                                                       \
        {                                                               \
            const float xh = (v.f - mu.f) * is.f;                       \
            const float u = xh * g.f + b.f;                             \
            const float dd = (!alpha || u > 0.f) ? d.f : al * d.f;      \
            o.f = g.f * is.f * (dd - m1[k] - xh * m2[k]);               \
        }
        BW1(x, 0) BW1(y, 1) BW1(z, 2) BW1(w, 3)
#undef BW1
        stv(dx, i, o);
    }
}

}  // namespace

static inline bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

// grid whose stride (grid * 256 float4s) is a multiple of C4, so that a thread's channel quad is loop invariant
static int fixed_channel_grid(long n4, int C4) {
    long b = (n4 + 255) / 256;
    const long cap = (long)cg::kNumCU * cg::kEwWgsPerCU;   // as cg::ew_grid
    if (b > cap) b = cap;
    if (256 % C4 != 0) {
        // stride multiple of C4 needs grid % (C4 / gcd(C4, 256)) == 0
        int gcd = C4, t = 256;
        while (t) { const int r = gcd % t; gcd = t; t = r; }
        const int m = C4 / gcd;
        b = std::max<long>(m, b / m * m);
    }
    return (int)std::max<long>(1, b);
}

extern "C" {

int cg_act_pool2_mask_forward(void* stream, const float* x, float* y, const float* mask, int ngroups, int n_per_group, int H,
                              int W, int C, int act, float slope, const float* const* alpha, int pool_max) {
    CG_REQUIRE(x && y, "cg_act_pool2_mask_forward: null pointer");
    CG_REQUIRE(ngroups >= 1 && ngroups <= 4 && n_per_group > 0 && H > 0 && W > 0 && C > 0, "cg_act_pool2_mask_forward: bad dims");
    CG_REQUIRE((H & 1) == 0 && (W & 1) == 0 && C % 4 == 0 && al16(x) && al16(y),
               "cg_act_pool2_mask_forward: needs even H, W, C %% 4 == 0 and 16-byte aligned tensors");
    CG_REQUIRE(act >= 0 && act <= 2, "cg_act_pool2_mask_forward: unknown activation %d", act);
    AlphaTab at{nullptr, nullptr, nullptr, nullptr};
    if (act == 1) {
        CG_REQUIRE(alpha, "cg_act_pool2_mask_forward: PReLU needs the slope pointers");
        const float* t[4] = {nullptr, nullptr, nullptr, nullptr};
        for (int g = 0; g < ngroups; ++g) { CG_REQUIRE(alpha[g], "cg_act_pool2_mask_forward: null slope (group %d)", g); t[g] = alpha[g]; }
        at = AlphaTab{t[0], t[1], t[2], t[3]};
    }
    const long total4 = (long)ngroups * n_per_group * (H / 2) * (W / 2) * (C / 4);
    const dim3 grid(cg::ew_grid(total4));
    if (pool_max)
        hipLaunchKernelGGL(act_pool2_fwd_k<true>, grid, dim3(256), 0, cg::S(stream), x, y, mask, at, act, slope, n_per_group,
                           total4, H, W, C / 4);
    else
        hipLaunchKernelGGL(act_pool2_fwd_k<false>, grid, dim3(256), 0, cg::S(stream), x, y, mask, at, act, slope, n_per_group,
                           total4, H, W, C / 4);
    CG_LAUNCH_CHECK();
    return 0;
}

static int act_pool_bwd_blocks(long per_group4) { return (int)std::max<long>(1, std::min<long>((per_group4 + 255) / 256, cg::kNumCU * 4)); }

size_t cg_act_pool2_mask_backward_workspace_bytes(int ngroups, int n_per_group, int H, int W, int C) {
    if (ngroups < 1 || n_per_group < 1 || H < 2 || W < 2 || C < 4) return 0;
    const long per_group4 = (long)n_per_group * (H / 2) * (W / 2) * (C / 4);
    return sizeof(double) * (size_t)ngroups * act_pool_bwd_blocks(per_group4);
}

int cg_act_pool2_mask_backward(void* stream, const float* x, const float* gy, const float* mask, float* dx, int ngroups,
                               int n_per_group, int H, int W, int C, int act, float slope, const float* const* alpha,
                               float* const* galpha, float scale, int pool_max, void* ws, size_t ws_bytes) {
    CG_REQUIRE(x && gy && dx, "cg_act_pool2_mask_backward: null pointer");
    CG_REQUIRE(ngroups >= 1 && ngroups <= 4 && n_per_group > 0 && H > 0 && W > 0 && C > 0, "cg_act_pool2_mask_backward: bad dims");
    CG_REQUIRE((H & 1) == 0 && (W & 1) == 0 && C % 4 == 0 && al16(x) && al16(gy) && al16(dx),
               "cg_act_pool2_mask_backward: needs even H, W, C %% 4 == 0 and 16-byte aligned tensors");
    CG_REQUIRE(act >= 0 && act <= 2, "cg_act_pool2_mask_backward: unknown activation %d", act);
    AlphaTab at{nullptr, nullptr, nullptr, nullptr};
    GalphaTab gt{nullptr, nullptr, nullptr, nullptr};
    bool want = false;
    if (act == 1) {
        CG_REQUIRE(alpha, "cg_act_pool2_mask_backward: PReLU needs the slope pointers");
        const float* t[4] = {nullptr, nullptr, nullptr, nullptr};
        float* gp[4] = {nullptr, nullptr, nullptr, nullptr};
        for (int g = 0; g < ngroups; ++g) {
            CG_REQUIRE(alpha[g], "cg_act_pool2_mask_backward: null slope (group %d)", g);
            t[g] = alpha[g];
            gp[g] = galpha ? galpha[g] : nullptr;
            want = want || gp[g];
        }
        at = AlphaTab{t[0], t[1], t[2], t[3]};
        gt = GalphaTab{gp[0], gp[1], gp[2], gp[3]};
    }
    const long per_group4 = (long)n_per_group * (H / 2) * (W / 2) * (C / 4);
    const int nblk = act_pool_bwd_blocks(per_group4);
    double* gpart = nullptr;
    if (want) {
        const size_t need = sizeof(double) * (size_t)ngroups * nblk;
        CG_REQUIRE(ws && ws_bytes >= need && ((uintptr_t)ws % 8) == 0, "cg_act_pool2_mask_backward: workspace too small (%zu < %zu)",
                   ws_bytes, need);
        gpart = (double*)ws;
    }
    const dim3 grid(nblk, ngroups);
    if (pool_max)
        hipLaunchKernelGGL(act_pool2_bwd_k<true>, grid, dim3(256), 0, cg::S(stream), x, gy, mask, dx, at, act, slope, n_per_group,
                           per_group4, H, W, C / 4, gpart);
    else
        hipLaunchKernelGGL(act_pool2_bwd_k<false>, grid, dim3(256), 0, cg::S(stream), x, gy, mask, dx, at, act, slope, n_per_group,
                           per_group4, H, W, C / 4, gpart);
    CG_LAUNCH_CHECK();
    if (want) {
        hipLaunchKernelGGL(galpha_groups_reduce_k, dim3(ngroups), dim3(256), 0, cg::S(stream), (const double*)gpart, nblk, gt, scale);
        CG_LAUNCH_CHECK();
    }
    return 0;
}

size_t cg_prelu_backward_grouped_workspace_bytes(int ngroups, long n_per_group) {
    if (ngroups < 1 || n_per_group < 4) return 0;
    return sizeof(double) * (size_t)ngroups * act_pool_bwd_blocks(n_per_group / 4);
}

int cg_prelu_backward_grouped(void* stream, const float* x, const float* dy, const float* const* alpha, float* dx,
                              float* const* galpha, float scale, int ngroups, long n_per_group, void* ws, size_t ws_bytes) {
    CG_REQUIRE(x && dy && dx && alpha, "cg_prelu_backward_grouped: null pointer");
    CG_REQUIRE(ngroups >= 1 && ngroups <= 4 && n_per_group > 0 && n_per_group % 4 == 0 && al16(x) && al16(dy) && al16(dx),
               "cg_prelu_backward_grouped: 1..4 groups of a multiple of 4 elements, 16-byte aligned");
    const float* t[4] = {nullptr, nullptr, nullptr, nullptr};
    float* gp[4] = {nullptr, nullptr, nullptr, nullptr};
    bool want = false;
    for (int g = 0; g < ngroups; ++g) {
        CG_REQUIRE(alpha[g], "cg_prelu_backward_grouped: null slope (group %d)", g);
        t[g] = alpha[g];
        gp[g] = galpha ? galpha[g] : nullptr;
        want = want || gp[g];
    }
    const AlphaTab at{t[0], t[1], t[2], t[3]};
    const GalphaTab gt{gp[0], gp[1], gp[2], gp[3]};
    const long per_group4 = n_per_group / 4;
    const int nblk = act_pool_bwd_blocks(per_group4);
    double* gpart = nullptr;
    if (want) {
        const size_t need = sizeof(double) * (size_t)ngroups * nblk;
        CG_REQUIRE(ws && ws_bytes >= need && ((uintptr_t)ws % 8) == 0, "cg_prelu_backward_grouped: workspace too small (%zu < %zu)",
                   ws_bytes, need);
        gpart = (double*)ws;
    }
    hipLaunchKernelGGL(prelu_bwd_groups_v4k, dim3(nblk, ngroups), dim3(256), 0, cg::S(stream), x, dy, at, dx, per_group4, gpart);
    CG_LAUNCH_CHECK();
    if (want) {
        hipLaunchKernelGGL(galpha_groups_reduce_k, dim3(ngroups), dim3(256), 0, cg::S(stream), (const double*)gpart, nblk, gt, scale);
        CG_LAUNCH_CHECK();
    }
    return 0;
}

static int cat4_launch(void* stream, int n, float* const* parts, const int* C, float* whole, long M, bool split, const char* who) {
    CG_REQUIRE(n >= 1 && n <= 4 && parts && C && whole && M > 0, "%s: 1..4 parts", who);
    float* p[4] = {nullptr, nullptr, nullptr, nullptr};
    int c4[4] = {0, 0, 0, 0};
    long Ct = 0;
    for (int i = 0; i < n; ++i) {
        CG_REQUIRE(parts[i] && C[i] > 0 && C[i] % 4 == 0 && al16(parts[i]), "%s: part %d needs C %% 4 == 0 and 16-byte alignment", who, i);
        p[i] = parts[i]; c4[i] = C[i] / 4; Ct += C[i];
    }
    CG_REQUIRE(al16(whole), "%s: unaligned tensor", who);
    Cat4 b{p[0], p[1], p[2], p[3], c4[0], c4[1], c4[2], c4[3]};
    const long total4 = M * (Ct / 4);
    if (split) hipLaunchKernelGGL(concat4_v4k<true>, dim3(cg::ew_grid(total4)), dim3(256), 0, cg::S(stream), b, whole, total4, (int)(Ct / 4));
    else hipLaunchKernelGGL(concat4_v4k<false>, dim3(cg::ew_grid(total4)), dim3(256), 0, cg::S(stream), b, whole, total4, (int)(Ct / 4));
    CG_LAUNCH_CHECK();
    return 0;
}

int cg_concat_channels(void* stream, int n, const float* const* src, const int* C, float* dst, long M) {
    return cat4_launch(stream, n, (float* const*)src, C, dst, M, false, "cg_concat_channels");
}
int cg_split_channels(void* stream, int n, const float* src, float* const* dst, const int* C, long M) {
    return cat4_launch(stream, n, dst, C, (float*)src, M, true, "cg_split_channels");
}
int cg_sum_n(void* stream, int n, const float* const* src, float* out, long count) {
    CG_REQUIRE(n >= 2 && n <= 4 && src && out && count > 0 && count % 4 == 0 && al16(out), "cg_sum_n: 2..4 aligned tensors of a multiple of 4 elements");
    for (int i = 0; i < n; ++i) CG_REQUIRE(src[i] && al16(src[i]), "cg_sum_n: tensor %d", i);
    hipLaunchKernelGGL(sum4_v4k, dim3(cg::ew_grid(count / 4)), dim3(256), 0, cg::S(stream), src[0], src[1], n > 2 ? src[2] : nullptr,
                       n > 3 ? src[3] : nullptr, out, n, count / 4);
    CG_LAUNCH_CHECK();
    return 0;
}

static int cat4_drop_args(int n, float* const* p, const int* C, const float* whole, const float* noise, Cat4& b, long& Ct, const char* who) {
    CG_REQUIRE(n >= 1 && n <= 4 && p && C && whole && noise, "%s: 1..4 branches", who);
    float* pp[4] = {nullptr, nullptr, nullptr, nullptr}; int c4[4] = {0, 0, 0, 0};
    Ct = 0;
    for (int i = 0; i < n; ++i) {
        CG_REQUIRE(p[i] && C[i] > 0 && C[i] % 4 == 0 && al16(p[i]), "%s: branch %d needs a multiple of 4 channels and an aligned tensor", who, i);
        pp[i] = p[i]; c4[i] = C[i] / 4; Ct += C[i];
    }
    CG_REQUIRE(al16(whole) && al16(noise), "%s: unaligned tensor", who);
    b = Cat4{pp[0], pp[1], pp[2], pp[3], c4[0], c4[1], c4[2], c4[3]};
    return 0;
}

int cg_concat_channels_dropout(void* stream, int n, const float* const* src, const int* C, float* dst, float* noise, int N, long HW,
                               float keep_prob, float value, uint64_t seed, uint64_t offset, const uint64_t* base) {
    Cat4 b; long Ct;
    if (cat4_drop_args(n, (float* const*)src, C, dst, noise, b, Ct, "cg_concat_channels_dropout")) return 1;
    CG_REQUIRE(N > 0 && HW > 0, "cg_concat_channels_dropout: empty tensor");
    const long total4 = (long)N * HW * (Ct / 4);
    hipLaunchKernelGGL(concat4_drop_v4k<true>, dim3(cg::ew_grid(total4)), dim3(256), 0, cg::S(stream), b, dst, noise, total4, (int)(Ct / 4), HW,
                       keep_prob, value, seed, offset, base);
    CG_LAUNCH_CHECK();
    return 0;
}

int cg_split_channels_masked(void* stream, int n, const float* src, const float* noise, float* const* dst, const int* C, int N, long HW) {
    Cat4 b; long Ct;
    if (cat4_drop_args(n, dst, C, src, noise, b, Ct, "cg_split_channels_masked")) return 1;
    CG_REQUIRE(N > 0 && HW > 0, "cg_split_channels_masked: empty tensor");
    const long total4 = (long)N * HW * (Ct / 4);
    hipLaunchKernelGGL(concat4_drop_v4k<false>, dim3(cg::ew_grid(total4)), dim3(256), 0, cg::S(stream), b, (float*)src, (float*)noise, total4,
                       (int)(Ct / 4), HW, 0.f, 0.f, (uint64_t)0, (uint64_t)0, (const uint64_t*)nullptr);
    CG_LAUNCH_CHECK();
    return 0;
}

int cg_drop_linear_sigmoid_supported(int N, int F, int O) { return N > 0 && F > 0 && O >= 1 && O <= 4 && (size_t)N * O * 4 + 4 * 64 * 4 * 4 <= 60000; }

int cg_drop_linear_sigmoid_forward(void* stream, const float* x, const float* w, const float* b, float* noise, float* xd, float* z, float* p,
                                   int N, int F, int O, float keep_prob, float value, uint64_t seed, uint64_t offset, const uint64_t* base) {
    CG_REQUIRE(x && w && noise && xd && z && p, "cg_drop_linear_sigmoid_forward: null pointer");
    CG_REQUIRE(cg_drop_linear_sigmoid_supported(N, F, O), "cg_drop_linear_sigmoid_forward: unsupported shape N=%d F=%d O=%d", N, F, O);
    const dim3 grid(cg::cdiv(N, 4));
#define CG_HEAD_F(OO) hipLaunchKernelGGL(head_fwd_k<OO>, grid, dim3(256), 0, cg::S(stream), x, w, b, noise, xd, z, p, N, F, keep_prob, value, seed, offset, base)
    switch (O) { case 1: CG_HEAD_F(1); break; case 2: CG_HEAD_F(2); break; case 3: CG_HEAD_F(3); break; default: CG_HEAD_F(4); break; }
#undef CG_HEAD_F
    CG_LAUNCH_CHECK();
    return 0;
}

int cg_drop_linear_sigmoid_backward(void* stream, const float* dp, const float* p, const float* xd, const float* noise, const float* w,
                                    float* gx, float* gw, float* gb, int N, int F, int O, float scale) {
    CG_REQUIRE(dp && p && noise && w && gx && (xd || !gw), "cg_drop_linear_sigmoid_backward: null pointer");
    CG_REQUIRE(cg_drop_linear_sigmoid_supported(N, F, O), "cg_drop_linear_sigmoid_backward: unsupported shape N=%d F=%d O=%d", N, F, O);
    const dim3 grid(cg::cdiv(F, 64));
    const size_t lds = ((size_t)N * O + 4 * 64 * O) * sizeof(float);
#define CG_HEAD_B(OO) hipLaunchKernelGGL(head_bwd_k<OO>, grid, dim3(256), lds, cg::S(stream), dp, p, xd, noise, w, gx, gw, gb, N, F, scale)
    switch (O) { case 1: CG_HEAD_B(1); break; case 2: CG_HEAD_B(2); break; case 3: CG_HEAD_B(3); break; default: CG_HEAD_B(4); break; }
#undef CG_HEAD_B
    CG_LAUNCH_CHECK();
    return 0;
}

int cg_bn_stats_finalize(void* stream, const float* partials, long rows, int C, double* sums) {
    CG_REQUIRE(partials && sums && rows > 0 && rows < (1L << 30) && C > 0, "cg_bn_stats_finalize: bad args");
    hipLaunchKernelGGL(bn_stats_finalize_k, dim3(cg::cdiv(2L * C, 16)), dim3(256), 0, cg::S(stream), partials, (int)rows, C, sums);
    CG_LAUNCH_CHECK();
    return 0;
}

int cg_bn_running_update(void* stream, const double* sums, double count, int C, float momentum, float* running_mean, float* running_var) {
    CG_REQUIRE(sums && running_mean && running_var && C > 0 && count > 0, "cg_bn_running_update: bad args");
    hipLaunchKernelGGL(bn_running_update_k, dim3(cg::cdiv((long)C, 256)), dim3(256), 0, cg::S(stream), sums, count, momentum, running_mean, running_var, C);
    CG_LAUNCH_CHECK();
    return 0;
}

int cg_bn_act_forward(void* stream, const float* x, float* y, const float* gamma, const float* beta, const double* sums,
                      double count, long M, int C, float eps, float momentum, float* running_mean, float* running_var,
                      float* save_mean, float* save_invstd, const float* alpha) {
    CG_REQUIRE(x && y && gamma && beta && sums && save_mean && save_invstd && C > 0 && count > 0 && M > 0, "cg_bn_act_forward: bad args");
    if (C % 4 == 0 && al16(x) && al16(y)) {
        const long n4 = M * (C / 4);
        hipLaunchKernelGGL(bn_act_fwd_v4k, dim3(fixed_channel_grid(n4, C / 4)), dim3(256), 0, cg::S(stream), x, y, gamma, beta, sums,
                           count, eps, momentum, running_mean, running_var, save_mean, save_invstd, alpha, n4, C / 4);
        CG_LAUNCH_CHECK();
        return 0;
    }
    // unaligned / odd channel counts: the separate passes
    if (cg_bn_forward(stream, x, y, gamma, beta, sums, count, M, C, eps, momentum, running_mean, running_var, save_mean, save_invstd))
        return 1;
    return alpha ? cg_prelu_forward(stream, y, alpha, y, M * C) : 0;
}

size_t cg_bn_act_backward_sums(int C) { return C > 0 ? (size_t)2 * C + 1 : 0; }

int cg_bn_act_backward_stats(void* stream, const float* x, const float* dy, const float* save_mean, const float* save_invstd,
                             const float* gamma, const float* beta, const float* alpha, long M, int C, double* sums) {
    CG_REQUIRE(x && dy && save_mean && save_invstd && gamma && beta && sums && C > 0 && M > 0, "cg_bn_act_backward_stats: bad args");
    CG_REQUIRE(C % 4 == 0 && al16(x) && al16(dy), "cg_bn_act_backward_stats: needs C %% 4 == 0 and 16-byte aligned tensors");
    const int qb = C >= 128 ? 32 : 16;
    const int cblocks = cg::cdiv(C / 4, qb);
    const int cmul = cg::kColReduceWgsPerCU;
    long chunks = std::max(1L, std::min((M + 63) / 64, (long)cg::kNumCU * cmul / cblocks));
    const long rows_per_block = ((M + chunks - 1) / chunks + 3) / 4 * 4;
    chunks = (M + rows_per_block - 1) / rows_per_block;
    const dim3 grid(cblocks, (unsigned)chunks);
    char* scr = (char*)cg::col_scratch(cg::S(stream));
    if (!scr) return 1;
    cg::ColTree tree;
    CG_REQUIRE(cg::col_tree_layout(scr, chunks, (size_t)2 * C, (size_t)chunks * cblocks, tree), "cg_bn_act_backward_stats: partials exceed the scratch");
    if (qb == 32)
        hipLaunchKernelGGL(bn_act_bwd_stats_k<32>, grid, dim3(256), 0, cg::S(stream), x, dy, save_mean, save_invstd, gamma, beta, alpha,
                           M, C, rows_per_block, sums, tree);
    else
        hipLaunchKernelGGL(bn_act_bwd_stats_k<16>, grid, dim3(256), 0, cg::S(stream), x, dy, save_mean, save_invstd, gamma, beta, alpha,
                           M, C, rows_per_block, sums, tree);
    CG_LAUNCH_CHECK();
    return 0;
}

int cg_bn_act_backward(void* stream, const float* x, const float* dy, const float* gamma, const float* beta,
                       const float* save_mean, const float* save_invstd, const float* alpha, const double* sums, double count,
                       const double* local_sums, long M, int C, float* dx, float* ggamma, float* gbeta, float* galpha,
                       float scale) {
    CG_REQUIRE(x && dy && gamma && beta && save_mean && save_invstd && sums && local_sums && dx && C > 0 && count > 0 && M > 0,
               "cg_bn_act_backward: bad args");
    CG_REQUIRE(C % 4 == 0 && al16(x) && al16(dy) && al16(dx), "cg_bn_act_backward: needs C %% 4 == 0 and 16-byte aligned tensors");
    const long n4 = M * (C / 4);
    hipLaunchKernelGGL(bn_act_bwd_v4k, dim3(fixed_channel_grid(n4, C / 4)), dim3(256), 0, cg::S(stream), x, dy, gamma, beta, save_mean,
                       save_invstd, alpha, sums, count, local_sums, n4, C / 4, dx, ggamma, gbeta, galpha, scale);
    CG_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
