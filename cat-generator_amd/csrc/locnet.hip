// locnet.hip - the localisation network of a spatial transformer as ONE launch each way (cg_locnet_forward / _backward).
//
// models.lua:842-860 builds it as ten modules - AvgPool(2) -> conv3x3 Cin->16 -> LeakyReLU -> conv3x3 16->16 -> LeakyReLU ->
// AvgPool(2) -> View -> Linear(16 (S/2)^2 -> 64) -> LeakyReLU -> Linear(64 -> P) - and :877-878 puts AffineTransformMatrixGenerator and
// AffineGridGeneratorBHWD behind it.  Per SAMPLE that is 1.5 MFLOP on 16 KB of activations: as separate launches (ten forward, a
// dozen backward, each a few microseconds of work behind a launch and a dependent-load round trip) the chain costs D32_st3 about
// 0.3 ms per training step on its critical path - the first transformer sits in front of everything D does (measured with the
// launches dropped: profiles/r03_exp_skip_locnet.txt).  Here one workgroup owns one sample: the pooled input, both convolutions'
// activations and the kernels live in LDS, the two small linear layers stream their weights from L2, and the workgroup ends by
// writing the sampling grid (forward) or the gradient w.r.t. the transformer's input (backward).
//
// Arithmetic per element is that of the separate entry points (cg_avgpool2_*, cg_conv2d_forward, cg_leakyrelu_*, cg_affine_*) up
// to fp32 re-association of the convolution / linear sums (taps outer, input planes inner, one fp32 chain of <= 576 terms).
// Weights are read in their canonical Torch7 layouts straight from the flat parameter vector: no packed copies.
// The weight gradients stay with the GEMM path: the backward launch leaves the per-layer gradients w.r.t. the pre-activations
// (and the forward its activations) as plain tensors, and the planner runs cg_conv2d_wgrad on them off the critical path.
#include "common.h"

namespace {

constexpr int LC = 16;   // planes of both convolutions (models.lua:844,846)
constexpr int LH = 64;   // hidden units of the first linear layer (models.lua:850)

struct LocW { const float *w1, *b1, *w2, *b2, *w3, *b3, *w4, *b4; };

struct LocFwd {
    LocW g[4];
    const float* x; int x_shared;                   // [.., 2S, 2S, Cin] NHWC; x_shared: every group reads samples [0, N)
    int G, N, S, Cin, P, Hg, Wg, ur, us, ut; float slope;
    float *pbuf, *h1buf, *m2buf, *h2buf, *h3buf, *params, *grid;
};
struct LocBwd {
    LocW g[4];
    int G, N, S, Cin, P, Hg, Wg, ur, us, ut; float slope;
    const float *h1buf, *m2buf, *h3buf, *params, *ggrid;
    float *ga1, *ga2, *g3, *g4, *gx;
};

__device__ __forceinline__ float lrelu(float v, float s) { return v >= 0.f ? v : s * v; }

// acc[0..3] += 3x3 / pad 1 convolution at pixel (y, x) for output planes co0..co0+3; in: LDS [S*S][CP], wl: LDS [(tap*C + ci)][16]
__device__ __forceinline__ void conv_px4(const float* in, int CP, int C, const float* wl, int S, int y, int x, int co0, float acc[4]) {
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
        if (yy < 0 || yy >= S || xx < 0 || xx >= S) continue;
        const float* ip = in + (yy * S + xx) * CP;
        const float* wp = wl + (t * C) * LC + co0;
        for (int ci = 0; ci < C; ++ci) {
            const float v = ip[ci];
            const float4 ww = *reinterpret_cast<const float4*>(wp + ci * LC);
            acc[0] += v * ww.x; acc[1] += v * ww.y; acc[2] += v * ww.z; acc[3] += v * ww.w;
        }
    }
}

__device__ __forceinline__ void affine_T(const float* prm, int ur, int us, int ut, float T[6], float* cs = nullptr) {
    int k = 0;
    float th = 0.f, sc = 1.f, tx = 0.f, ty = 0.f;
    if (ur) th = prm[k++];
    if (us) sc = prm[k++];
    if (ut) { tx = prm[k]; ty = prm[k + 1]; }
    const float c = cosf(th), s = sinf(th);
    T[0] = c * sc; T[1] = -s * sc; T[2] = c * sc * tx - s * sc * ty;
    T[3] = s * sc; T[4] = c * sc;  T[5] = s * sc * tx + c * sc * ty;
    if (cs) { cs[0] = c; cs[1] = s; cs[2] = sc; cs[3] = tx; cs[4] = ty; }
}

__global__ __launch_bounds__(256) void locnet_fwd_k(LocFwd a) {
    extern __shared__ float sm[];
    const int S = a.S, Cin = a.Cin, S2 = S * S, CP = Cin + 1, Sh = S / 2, K3 = LC * Sh * Sh, tid = threadIdx.x;
    const int smp = blockIdx.x, g = smp / a.N, n = smp - g * a.N;
    const LocW w = a.g[g];
    float* p = sm;                       // pooled input [S2][CP]
    float* w1l = p + S2 * CP;            // [(tap*Cin + ci)][16]
    float* h1 = w1l + 9 * Cin * LC;      // LeakyReLU(conv1) [S2][17]
    float* w2l = h1 + S2 * (LC + 1);     // [(tap*16 + ci)][16]
    float* h2 = w2l + 9 * LC * LC;       // pooled LeakyReLU(conv2), (c, y, x) order [K3]
    float* h3 = h2 + K3;                 // [64]
    float* prm = h3 + LH;                // [8]
    const float sl = a.slope;

    // AvgPool(2,2,2,2) of the sample (cg_avgpool2_forward's order of additions)
    const float* xs = a.x + (size_t)(a.x_shared ? n : smp) * (4 * S2) * Cin;
    for (int i = tid; i < S2 * Cin; i += 256) {
        const int c = i % Cin, px = (i / Cin) % S, py = i / (Cin * S);
        const size_t b = ((size_t)(2 * py) * (2 * S) + 2 * px) * Cin + c;
        const float v = (xs[b] + xs[b + Cin] + xs[b + (size_t)2 * S * Cin] + xs[b + (size_t)2 * S * Cin + Cin]) * 0.25f;
        p[(py * S + px) * CP + c] = v;
        a.pbuf[(size_t)smp * S2 * Cin + i] = v;
    }
    // kernels: canonical [co][ci][ky][kx] -> [(tap*C + ci)][co]
    for (int i = tid; i < LC * Cin * 9; i += 256) {
        const int t = i % 9, ci = (i / 9) % Cin, co = i / (9 * Cin);
        w1l[(t * Cin + ci) * LC + co] = w.w1[i];
    }
    for (int i = tid; i < LC * LC * 9; i += 256) {
        const int t = i % 9, ci = (i / 9) % LC, co = i / (9 * LC);
        w2l[(t * LC + ci) * LC + co] = w.w2[i];
    }
    __syncthreads();
    // conv3x3 Cin -> 16 + LeakyReLU: items = (pixel, quad of output planes); a wave shares its quad (S2 % 64 == 0)
    for (int it = tid; it < S2 * 4; it += 256) {
        const int pix = it % S2, co0 = (it / S2) * 4;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        conv_px4(p, CP, Cin, w1l, S, pix / S, pix % S, co0, acc);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float v = lrelu(acc[k] + w.b1[co0 + k], sl);
            h1[pix * (LC + 1) + co0 + k] = v;
            a.h1buf[((size_t)smp * S2 + pix) * LC + co0 + k] = v;
        }
    }
    __syncthreads();
    // conv3x3 16 -> 16 + LeakyReLU + AvgPool(2): items = (pooled pixel, quad of output planes), four pixels each
    for (int it = tid; it < Sh * Sh * 4; it += 256) {
        const int pp = it % (Sh * Sh), co0 = (it / (Sh * Sh)) * 4;
        const int py = pp / Sh, px = pp % Sh;
        float v[4][4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int y = 2 * py + (q >> 1), x = 2 * px + (q & 1);
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
            conv_px4(h1, LC + 1, LC, w2l, S, y, x, co0, acc);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                v[q][k] = lrelu(acc[k] + w.b2[co0 + k], sl);
                a.m2buf[((size_t)smp * S2 + y * S + x) * LC + co0 + k] = v[q][k];
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float pv = (v[0][k] + v[1][k] + v[2][k] + v[3][k]) * 0.25f;
            const int idx = (co0 + k) * Sh * Sh + pp;          // nn.View(16*h*h): (c, y, x) order of the NCHW map
            h2[idx] = pv;
            a.h2buf[(size_t)smp * K3 + idx] = pv;
        }
    }
    __syncthreads();
    // Linear(K3 -> 64) + LeakyReLU: four lanes per output
    {
        const int o = tid >> 2, part = tid & 3;
        const float* wr = w.w3 + (size_t)o * K3;
        float s = 0.f;
        for (int k = part; k < K3; k += 4) s += h2[k] * wr[k];
        s += __shfl_xor(s, 1, 64);
        s += __shfl_xor(s, 2, 64);
        if (part == 0) {
            const float v = lrelu(s + w.b3[o], sl);
            h3[o] = v;
            a.h3buf[(size_t)smp * LH + o] = v;
        }
    }
    __syncthreads();
    // Linear(64 -> P)
    if (tid < a.P) {
        const float* wr = w.w4 + tid * LH;
        float s = 0.f;
        for (int k = 0; k < LH; ++k) s += h3[k] * wr[k];
        s += w.b4[tid];
        prm[tid] = s;
        a.params[(size_t)smp * a.P + tid] = s;
    }
    __syncthreads();
    // AffineTransformMatrixGenerator + AffineGridGeneratorBHWD
    float T[6];
    affine_T(prm, a.ur, a.us, a.ut, T);
    const int H = a.Hg, W = a.Wg;
    float* gr = a.grid + (size_t)smp * H * W * 2;
    for (int i = tid; i < H * W; i += 256) {
        const int ii = i / W, j = i - ii * W;
        const float y = H > 1 ? -1.f + 2.f * (float)ii / (float)(H - 1) : -1.f;
        const float x = W > 1 ? -1.f + 2.f * (float)j / (float)(W - 1) : -1.f;
        gr[i * 2 + 0] = T[0] * y + T[1] * x + T[2];
        gr[i * 2 + 1] = T[3] * y + T[4] * x + T[5];
    }
}

// Backward of the same chain for one sample: ggrid -> gT -> gparams -> ... -> gradient w.r.t. the (un-pooled) input.  Leaves
// g4 = dL/d(params), g3 = dL/d(pre-activation of Linear 1), ga2 / ga1 = dL/d(pre-activation of conv 2 / conv 1) for the weight
// gradients.
__global__ __launch_bounds__(256) void locnet_bwd_k(LocBwd a) {
    extern __shared__ float sm[];
    __shared__ double shd[4];
    const int S = a.S, Cin = a.Cin, S2 = S * S, Sh = S / 2, K3 = LC * Sh * Sh, tid = threadIdx.x;
    const int smp = blockIdx.x, g = smp / a.N;
    const LocW w = a.g[g];
    float* ga2 = sm;                         // [S2][17]
    float* ga1 = ga2 + S2 * (LC + 1);        // [S2][17]
    float* w2l = ga1 + S2 * (LC + 1);        // flipped: [(tap*16 + co)][16 ci]
    float* w1l = w2l + 9 * LC * LC;          // flipped: [(tap*16 + co)][CinQ] (Cin rounded up to 4)
    const int CinQ = (Cin + 3) & ~3;
    float* gp = w1l + 9 * LC * CinQ;         // gradient w.r.t. the pooled input [S2][CinQ]
    float* gh2 = gp + S2 * CinQ;             // [K3]
    float* g3s = gh2 + K3;                   // [64]
    float* g4s = g3s + LH;                   // [8]
    const float sl = a.slope;

    // affine_grid_backward: gT[r][:] = sum_{i,j} ggrid[i,j,r] * (y_i, x_j, 1)   (fp64 block sums, fixed order)
    const int H = a.Hg, W = a.Wg;
    const float* gg = a.ggrid + (size_t)smp * H * W * 2;
    double acc[6] = {0, 0, 0, 0, 0, 0};
    for (int i = tid; i < H * W; i += 256) {
        const int ii = i / W, j = i - ii * W;
        const float y = H > 1 ? -1.f + 2.f * (float)ii / (float)(H - 1) : -1.f;
        const float x = W > 1 ? -1.f + 2.f * (float)j / (float)(W - 1) : -1.f;
        const float g0 = gg[i * 2], g1 = gg[i * 2 + 1];
        acc[0] += g0 * y; acc[1] += g0 * x; acc[2] += g0;
        acc[3] += g1 * y; acc[4] += g1 * x; acc[5] += g1;
    }
    float gT[6];
    __shared__ float gTs[6];
    for (int k = 0; k < 6; ++k) {
        const double t = cg::block_sum_256(acc[k], shd);
        if (tid == 0) gTs[k] = (float)t;
    }
    // kernels, flipped for the data gradients: dx[ci] at pixel q = sum_{tap, co} dy[q - off(tap)][co] * w[co][ci][tap]
    for (int i = tid; i < LC * LC * 9; i += 256) {
        const int t = i % 9, ci = (i / 9) % LC, co = i / (9 * LC);
        w2l[((8 - t) * LC + co) * LC + ci] = w.w2[i];
    }
    for (int i = tid; i < 9 * LC * CinQ; i += 256) w1l[i] = 0.f;
    __syncthreads();
    for (int k = 0; k < 6; ++k) gT[k] = gTs[k];
    for (int i = tid; i < LC * Cin * 9; i += 256) {
        const int t = i % 9, ci = (i / 9) % Cin, co = i / (9 * Cin);
        w1l[((8 - t) * LC + co) * CinQ + ci] = w.w1[i];
    }
    // affine_matrix_backward (cg_affine_matrix_backward's formulas)
    if (tid == 0) {
        const float* prm = a.params + (size_t)smp * a.P;
        float T[6], cs[5];
        affine_T(prm, a.ur, a.us, a.ut, T, cs);
        const float c = cs[0], s = cs[1], sc = cs[2], tx = cs[3], ty = cs[4];
        int k = 0;
        if (a.ur) {
            const float d0 = -s * sc, d1 = -c * sc, d2 = -s * sc * tx - c * sc * ty;
            const float d3 = c * sc, d4 = -s * sc, d5 = c * sc * tx - s * sc * ty;
            g4s[k++] = gT[0] * d0 + gT[1] * d1 + gT[2] * d2 + gT[3] * d3 + gT[4] * d4 + gT[5] * d5;
        }
        if (a.us) g4s[k++] = gT[0] * c + gT[1] * (-s) + gT[2] * (c * tx - s * ty) + gT[3] * s + gT[4] * c + gT[5] * (s * tx + c * ty);
        if (a.ut) {
            g4s[k] = gT[2] * (c * sc) + gT[5] * (s * sc);
            g4s[k + 1] = gT[2] * (-s * sc) + gT[5] * (c * sc);
        }
        for (int q = 0; q < a.P; ++q) a.g4[(size_t)smp * a.P + q] = g4s[q];
    }
    __syncthreads();
    // Linear(64 -> P) backward + LeakyReLU backward: g3[k] = lrelu'(a3[k]) * sum_p w4[p][k] g4[p]
    if (tid < LH) {
        float s = 0.f;
        for (int q = 0; q < a.P; ++q) s += w.w4[q * LH + tid] * g4s[q];
        const float h = a.h3buf[(size_t)smp * LH + tid];
        const float v = h >= 0.f ? s : sl * s;
        g3s[tid] = v;
        a.g3[(size_t)smp * LH + tid] = v;
    }
    __syncthreads();
    // Linear(K3 -> 64) backward: gh2[i] = sum_o w3[o][i] g3[o]
    for (int i = tid; i < K3; i += 256) {
        float s = 0.f;
        for (int o = 0; o < LH; ++o) s += w.w3[(size_t)o * K3 + i] * g3s[o];
        gh2[i] = s;
    }
    __syncthreads();
    // AvgPool backward (x 0.25) + LeakyReLU backward at conv 2's output
    for (int i = tid; i < S2 * LC; i += 256) {
        const int c = i % LC, pix = i / LC, y = pix / S, x = pix % S;
        const float gv = gh2[c * Sh * Sh + (y >> 1) * Sh + (x >> 1)] * 0.25f;
        const float m = a.m2buf[((size_t)smp * S2 + pix) * LC + c];
        const float v = m >= 0.f ? gv : sl * gv;
        ga2[pix * (LC + 1) + c] = v;
        a.ga2[((size_t)smp * S2 + pix) * LC + c] = v;
    }
    __syncthreads();
    // conv 2 data gradient + LeakyReLU backward at conv 1's output
    for (int it = tid; it < S2 * 4; it += 256) {
        const int pix = it % S2, c0 = (it / S2) * 4;
        float acc4[4] = {0.f, 0.f, 0.f, 0.f};
        conv_px4(ga2, LC + 1, LC, w2l, S, pix / S, pix % S, c0, acc4);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float h = a.h1buf[((size_t)smp * S2 + pix) * LC + c0 + k];
            const float v = h >= 0.f ? acc4[k] : sl * acc4[k];
            ga1[pix * (LC + 1) + c0 + k] = v;
            a.ga1[((size_t)smp * S2 + pix) * LC + c0 + k] = v;
        }
    }
    __syncthreads();
    // conv 1 data gradient: items = (pixel, quad of INPUT planes); the flipped kernel is [(tap*16 + co)][CinQ]
    const int nq = CinQ / 4;
    for (int it = tid; it < S2 * nq; it += 256) {
        const int pix = it % S2, c0 = (it / S2) * 4;
        const int y = pix / S, x = pix % S;
        float acc4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
            if (yy < 0 || yy >= S || xx < 0 || xx >= S) continue;
            const float* ip = ga1 + (yy * S + xx) * (LC + 1);
            const float* wp = w1l + (t * LC) * CinQ + c0;
            for (int co = 0; co < LC; ++co) {
                const float v = ip[co];
                const float4 ww = *reinterpret_cast<const float4*>(wp + co * CinQ);
                acc4[0] += v * ww.x; acc4[1] += v * ww.y; acc4[2] += v * ww.z; acc4[3] += v * ww.w;
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) gp[pix * CinQ + c0 + k] = acc4[k];
    }
    __syncthreads();
    // AvgPool backward to the transformer's input resolution
    float* gx = a.gx + (size_t)smp * (4 * S2) * Cin;
    for (int i = tid; i < 4 * S2 * Cin; i += 256) {
        const int c = i % Cin, xx = (i / Cin) % (2 * S), yy = i / (Cin * 2 * S);
        gx[i] = gp[((yy >> 1) * S + (xx >> 1)) * CinQ + c] * 0.25f;
    }
}

size_t fwd_lds_floats(int S, int Cin) {
    const int S2 = S * S, K3 = LC * (S / 2) * (S / 2);
    return (size_t)S2 * (Cin + 1) + 9 * Cin * LC + (size_t)S2 * (LC + 1) + 9 * LC * LC + K3 + LH + 8;
}
size_t bwd_lds_floats(int S, int Cin) {
    const int S2 = S * S, K3 = LC * (S / 2) * (S / 2), CinQ = (Cin + 3) & ~3;
    return (size_t)2 * S2 * (LC + 1) + 9 * LC * LC + (size_t)9 * LC * CinQ + (size_t)S2 * CinQ + K3 + LH + 8;
}

}  // namespace

extern "C" {

// 1 if the fused launches cover this localisation net: power-of-two pooled size S >= 8 (so that a wave shares its quad of output
// planes), the activations of one sample within LDS.
int cg_locnet_supported(int S, int Cin, int P) {
    if (S < 8 || (S & (S - 1)) || Cin < 1 || P < 1 || P > 4) return 0;
    return fwd_lds_floats(S, Cin) * 4 <= 160 * 1024 && bwd_lds_floats(S, Cin) * 4 <= 160 * 1024 ? 1 : 0;
}

int cg_locnet_forward(void* stream, int ngroups, int n_per_group, const float* x, int x_shared, const float* const* weights, int S, int Cin,
                      int P, int use_rot, int use_scale, int use_trans, float slope, int Hg, int Wg, float* pooled, float* h1, float* m2,
                      float* h2, float* h3, float* params, float* grid) {
    CG_REQUIRE(x && weights && pooled && h1 && m2 && h2 && h3 && params && grid, "cg_locnet_forward: null pointer");
    CG_REQUIRE(ngroups >= 1 && ngroups <= 4 && n_per_group > 0, "cg_locnet_forward: %d groups of %d samples", ngroups, n_per_group);
    CG_REQUIRE(cg_locnet_supported(S, Cin, P), "cg_locnet_forward: S %d Cin %d P %d not supported", S, Cin, P);
    CG_REQUIRE(P == (use_rot ? 1 : 0) + (use_scale ? 1 : 0) + (use_trans ? 2 : 0), "cg_locnet_forward: P does not match the transform");
    LocFwd a;
    for (int g = 0; g < ngroups; ++g) {
        const float* const* w = weights + 8 * g;
        for (int k = 0; k < 8; ++k) CG_REQUIRE(w[k], "cg_locnet_forward: null weight pointer");
        a.g[g] = LocW{w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7]};
    }
    a.x = x; a.x_shared = x_shared; a.G = ngroups; a.N = n_per_group; a.S = S; a.Cin = Cin; a.P = P; a.Hg = Hg; a.Wg = Wg;
    a.ur = use_rot ? 1 : 0; a.us = use_scale ? 1 : 0; a.ut = use_trans ? 1 : 0; a.slope = slope;
    a.pbuf = pooled; a.h1buf = h1; a.m2buf = m2; a.h2buf = h2; a.h3buf = h3; a.params = params; a.grid = grid;
    const size_t lds = fwd_lds_floats(S, Cin) * 4;
    static bool attr = false;
    if (!attr) { CG_HIP(hipFuncSetAttribute((const void*)locnet_fwd_k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); attr = true; }
    hipLaunchKernelGGL(locnet_fwd_k, dim3(ngroups * n_per_group), dim3(256), lds, cg::S(stream), a);
    CG_LAUNCH_CHECK();
    return 0;
}

int cg_locnet_backward(void* stream, int ngroups, int n_per_group, const float* const* weights, int S, int Cin, int P, int use_rot,
                       int use_scale, int use_trans, float slope, int Hg, int Wg, const float* h1, const float* m2, const float* h3,
                       const float* params, const float* ggrid, float* ga1, float* ga2, float* g3, float* g4, float* gx) {
    CG_REQUIRE(weights && h1 && m2 && h3 && params && ggrid && ga1 && ga2 && g3 && g4 && gx, "cg_locnet_backward: null pointer");
    CG_REQUIRE(ngroups >= 1 && ngroups <= 4 && n_per_group > 0, "cg_locnet_backward: %d groups of %d samples", ngroups, n_per_group);
    CG_REQUIRE(cg_locnet_supported(S, Cin, P), "cg_locnet_backward: S %d Cin %d P %d not supported", S, Cin, P);
    LocBwd a;
    for (int g = 0; g < ngroups; ++g) {
        const float* const* w = weights + 8 * g;
        a.g[g] = LocW{w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7]};
    }
    a.G = ngroups; a.N = n_per_group; a.S = S; a.Cin = Cin; a.P = P; a.Hg = Hg; a.Wg = Wg;
    a.ur = use_rot ? 1 : 0; a.us = use_scale ? 1 : 0; a.ut = use_trans ? 1 : 0; a.slope = slope;
    a.h1buf = h1; a.m2buf = m2; a.h3buf = h3; a.params = params; a.ggrid = ggrid;
    a.ga1 = ga1; a.ga2 = ga2; a.g3 = g3; a.g4 = g4; a.gx = gx;
    const size_t lds = bwd_lds_floats(S, Cin) * 4;
    static bool attr = false;
    if (!attr) { CG_HIP(hipFuncSetAttribute((const void*)locnet_bwd_k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); attr = true; }
    hipLaunchKernelGGL(locnet_bwd_k, dim3(ngroups * n_per_group), dim3(256), lds, cg::S(stream), a);
    CG_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
