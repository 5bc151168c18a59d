// locnet.hip - the localisation network of a spatial transformer as ONE launch each way (cg_locnet_forward / _backward).
//
// models.lua:842-860 builds it as ten modules - AvgPool(2) -> conv3x3 Cin->16 -> LeakyReLU -> conv3x3 16->16 -> LeakyReLU ->
// AvgPool(2) -> View -> Linear(16 (S/2)^2 -> 64) -> LeakyReLU -> Linear(64 -> P) - and :877-878 puts AffineTransformMatrixGenerator and
// AffineGridGeneratorBHWD behind it.  Per SAMPLE that is 1.5 MFLOP on 16 KB of activations: as separate launches (ten forward, a
// dozen backward, each a few microseconds of work behind a launch and a dependent-load round trip) the chain costs D32_st3 about
// 0.3 ms per training step on its critical path - the first transformer sits in front of everything D does (measured with the
// launches dropped: profiles/r03_exp_skip_locnet.txt; profiles/r03_locnet_phases.txt has the per-phase times of these kernels).
// Here one workgroup owns one sample: the pooled input, both convolutions'
// activations and the kernels live in LDS, the two small linear layers stream their weights from L2, and the workgroup ends by
// writing the sampling grid (forward) or the gradient w.r.t. the transformer's input (backward).
//
// Arithmetic per element is that of the separate entry points (cg_avgpool2_*, cg_conv2d_forward, cg_leakyrelu_*, cg_affine_*) up
// to fp32 re-association of the convolution / linear sums (taps outer, input planes inner, one fp32 chain of <= 576 terms,
// or three chains added in a fixed order where the taps are split over threads).
// The linear layers read their canonical weights straight from the flat parameter vector; the two convolutions take
// cg_pack_conv_weight's copies (already [(tap, ci)][co] / flipped [(tap, co)][ci]: straight copies into LDS).
// The weight gradients stay with the GEMM path: the backward launch leaves the per-layer gradients w.r.t. the pre-activations
// (and the forward its activations) as plain tensors, and the planner runs cg_conv2d_wgrad on them off the critical path.
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int NT = 768;  // threads per workgroup (one sample): 3 waves per SIMD hide the LDS / L2 latencies of the short dependent phases
constexpr int LC = 16;   // planes of both convolutions (models.lua:844,846)
constexpr int LH = 64;   // hidden units of the first linear layer (models.lua:850)

struct LocW { const float *w1, *b1, *w2, *b2, *w3, *b3, *w4, *b4, *wf1, *wb1, *wf2, *wb2; };   // wf / wb: cg_pack_conv_weight's copies

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

// 3x3 / pad 1 convolution of one sample held in LDS, register-tiled: an item is 4 pixels x 4 output planes (16 accumulators; per
// (tap, input plane) one broadcast float4 of weights and four activations feed 16 FMAs).  The four pixels of an item lie a quarter
// of the image apart (pixel g + q*S*S/4), so that consecutive lanes own consecutive pixels: with the odd channel stride CP their
// LDS reads fall into distinct banks.  in: [S*S][CP], wl: [(tap*C + ci)][WS] with the output planes contiguous, zero: >= C zeros in
// LDS that out-of-image taps read instead of branching.  raw: [S*S][OP] receives the plain sums for output planes [0, OUT)
// (OUT % 4 == 0).  With few items the nine taps are split over 3 or 9 threads per item whose partial sums meet in `part` and are
// added in a fixed order.  All NT threads must call it; it ends with a barrier.
__host__ __device__ inline int conv_split(int S, int OUT) {   // tap groups: 1 (all nine taps per item), 3 (one kernel row) or 9 (one tap)
    const int items = ((S * S) >> 2) * (OUT >> 2);
    return 2 * items >= NT ? 1 : (6 * items >= NT ? 3 : 9);
}
__device__ void conv3x3_lds(const float* in, int CP, int C, const float* wl, int WS, const float* zero, int S, int OUT, float* raw, int OP,
                            float* part) {
    const int tid = threadIdx.x, nq = (S * S) >> 2, noq = OUT >> 2, items = nq * noq, ls = 31 - __clz(S);
    const int R = conv_split(S, OUT), tpr = 9 / R;
    for (int it = tid; it < items * R; it += NT) {
        const int r = it / items, id = it - r * items;
        const int grp = id % nq, o0 = (id / nq) * 4;
        float acc[4][4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { acc[q][0] = acc[q][1] = acc[q][2] = acc[q][3] = 0.f; }
        int py[4], px[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int pix = grp + q * nq; py[q] = pix >> ls; px[q] = pix & (S - 1); }
        for (int t = r * tpr; t < r * tpr + tpr; ++t) {
            const int dy = t / 3 - 1, dx = t - (t / 3) * 3 - 1;
            const float* ip[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int yy = py[q] + dy, xx = px[q] + dx;
                ip[q] = ((unsigned)yy < (unsigned)S && (unsigned)xx < (unsigned)S) ? in + ((yy << ls) + xx) * CP : zero;
            }
            const float* wp = wl + (size_t)(t * C) * WS + o0;
#pragma unroll 8
            for (int ci = 0; ci < C; ++ci) {
                const float4 ww = *reinterpret_cast<const float4*>(wp + ci * WS);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float v = ip[q][ci];
                    acc[q][0] += v * ww.x; acc[q][1] += v * ww.y; acc[q][2] += v * ww.z; acc[q][3] += v * ww.w;
                }
            }
        }
        if (R == 1) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float* o = raw + (grp + q * nq) * OP + o0;
                o[0] = acc[q][0]; o[1] = acc[q][1]; o[2] = acc[q][2]; o[3] = acc[q][3];
            }
        } else {
            // part[r][k][item]: consecutive lanes (items) store to consecutive banks (an item-major [item][16] layout put 32 lanes on two banks)
            float* o = part + (size_t)r * items * 16 + id;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                o[(q * 4) * items] = acc[q][0]; o[(q * 4 + 1) * items] = acc[q][1]; o[(q * 4 + 2) * items] = acc[q][2]; o[(q * 4 + 3) * items] = acc[q][3];
            }
        }
    }
    __syncthreads();
    if (R > 1) {
        for (int e = tid; e < items * 16; e += NT) {
            const int k = e / items, id = e - k * items, q = k >> 2, c = k & 3;
            const int grp = id % nq, o0 = (id / nq) * 4;
            float v = part[e];
            for (int r = 1; r < R; ++r) v += part[(size_t)r * items * 16 + e];      // fixed order
            raw[(grp + q * nq) * OP + o0 + c] = v;
        }
        __syncthreads();
    }
}
__host__ __device__ inline int conv_part_floats(int S, int OUT) {
    const int items = ((S * S) >> 2) * (OUT >> 2), R = conv_split(S, OUT);
    return R == 1 ? 0 : R * items * 16;
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

// LDS plan shared by the kernels and the host-side size check (floats)
struct FwdLds { int p, w1l, h1, w2l, m2, h2, h3, prm, zero, part, total; };
__host__ __device__ inline FwdLds fwd_lds(int S, int Cin) {
    FwdLds L; const int S2 = S * S, K3 = LC * (S / 2) * (S / 2);
    int o = 0;
    L.p = o; o += S2 * (Cin | 1);            // odd stride
    L.w1l = o; o += 9 * Cin * LC;
    L.h1 = o; o += S2 * (LC + 1);
    L.w2l = o; o += 9 * LC * LC;
    L.m2 = o; o += S2 * (LC + 1);
    L.h2 = o; o += K3;
    L.h3 = o; o += LH;
    L.prm = o; o += 8;
    L.zero = o; o += (Cin > LC ? Cin : LC) + 4;
    L.part = o; o += conv_part_floats(S, LC);
    L.total = o;
    return L;
}
struct BwdLds { int ga2, ga1, w2l, w1l, gp, gh2, g3, g4, zero, part, total; };
__host__ __device__ inline BwdLds bwd_lds(int S, int Cin) {
    BwdLds L; const int S2 = S * S, K3 = LC * (S / 2) * (S / 2), CinQ = (Cin + 3) & ~3;
    int o = 0;
    L.ga2 = o; o += S2 * (LC + 1);
    L.ga1 = o; o += S2 * (LC + 1);
    L.w2l = o; o += 9 * LC * LC;
    L.w1l = o; o += 9 * LC * CinQ;
    L.gp = o; o += S2 * (CinQ + 1);
    L.gh2 = o; o += K3;
    L.g3 = o; o += LH;
    L.g4 = o; o += 8;
    L.zero = o; o += LC + 4;
    const int p2 = conv_part_floats(S, LC), p1 = conv_part_floats(S, CinQ);
    L.part = o; o += p2 > p1 ? p2 : p1;
    L.total = o;
    return L;
}

__global__ __launch_bounds__(NT) void locnet_fwd_k(LocFwd a) {
    extern __shared__ float sm[];
    const int S = a.S, Cin = a.Cin, S2 = S * S, CP = Cin | 1, Sh = S / 2, K3 = LC * Sh * Sh, tid = threadIdx.x;
    const int smp = blockIdx.x, g = smp / a.N, n = smp - g * a.N;
    const LocW w = a.g[g];
    const FwdLds L = fwd_lds(S, Cin);
    float *p = sm + L.p, *w1l = sm + L.w1l, *h1 = sm + L.h1, *w2l = sm + L.w2l, *m2 = sm + L.m2, *h2 = sm + L.h2, *h3 = sm + L.h3,
          *prm = sm + L.prm, *zero = sm + L.zero, *part = sm + L.part;
    const float sl = a.slope;

    // AvgPool(2,2,2,2) of the sample (cg_avgpool2_forward's order of additions)
    const float* xs = a.x + (size_t)(a.x_shared ? n : smp) * (4 * S2) * Cin;
    for (int i = tid; i < S2 * Cin; i += NT) {
        const int c = i % Cin, px = (i / Cin) % S, py = i / (Cin * S);
        const size_t b = ((size_t)(2 * py) * (2 * S) + 2 * px) * Cin + c;
        const float v = (xs[b] + xs[b + Cin] + xs[b + (size_t)2 * S * Cin] + xs[b + (size_t)2 * S * Cin + Cin]) * 0.25f;
        p[(py * S + px) * CP + c] = v;
        a.pbuf[(size_t)smp * S2 * Cin + i] = v;
    }
    // kernels as [(tap*C + ci)][co]
    // (cg_pack_conv_weight's forward copies already have this layout: straight, coalesced copies)
    for (int d = tid; d < 9 * Cin * LC; d += NT) w1l[d] = w.wf1[d];
    for (int d = tid; d < 9 * LC * LC; d += NT) w2l[d] = w.wf2[d];
    for (int i = tid; i < (Cin > LC ? Cin : LC) + 4; i += NT) zero[i] = 0.f;
    __syncthreads();
    // conv3x3 Cin -> 16, then bias + LeakyReLU in place
    conv3x3_lds(p, CP, Cin, w1l, LC, zero, S, LC, h1, LC + 1, part);
    for (int i = tid; i < S2 * LC; i += NT) {
        const int c = i % LC, pix = i / LC;
        const float v = lrelu(h1[pix * (LC + 1) + c] + w.b1[c], sl);
        h1[pix * (LC + 1) + c] = v;
        a.h1buf[((size_t)smp * S2 + pix) * LC + c] = v;
    }
    __syncthreads();
    // conv3x3 16 -> 16, bias + LeakyReLU, AvgPool(2)
    conv3x3_lds(h1, LC + 1, LC, w2l, LC, zero, S, LC, m2, LC + 1, part);
    for (int i = tid; i < S2 * LC; i += NT) {
        const int c = i % LC, pix = i / LC;
        const float v = lrelu(m2[pix * (LC + 1) + c] + w.b2[c], sl);
        m2[pix * (LC + 1) + c] = v;
        a.m2buf[((size_t)smp * S2 + pix) * LC + c] = v;
    }
    __syncthreads();
    for (int i = tid; i < K3; i += NT) {          // i = (c, py, px): nn.View(16*h*h) flattens the NCHW map
        const int px = i % Sh, py = (i / Sh) % Sh, c = i / (Sh * Sh);
        const float* q = m2 + ((2 * py) * S + 2 * px) * (LC + 1) + c;
        const float pv = (q[0] + q[LC + 1] + q[S * (LC + 1)] + q[(S + 1) * (LC + 1)]) * 0.25f;
        h2[i] = pv;
        a.h2buf[(size_t)smp * K3 + i] = pv;
    }
    __syncthreads();
    // Linear(K3 -> 64) + LeakyReLU: eight waves take 8 outputs each; a wave's lanes stride over k and keep all 8 running sums, so every
    // step has 8 (x2 unrolled) independent coalesced row loads in flight (the weight matrix streams from L2: latency, not arithmetic,
    // is the cost); fixed shuffle tree per output at the end
    {
        const int wv = tid >> 6, lane = tid & 63;
        if (wv < 8) {
            const float* wr = w.w3 + (size_t)(wv * 8) * K3;
            float s8[8];
#pragma unroll
            for (int oo = 0; oo < 8; ++oo) s8[oo] = 0.f;
#pragma unroll 2
            for (int k = lane; k < K3; k += 64) {
                const float hv = h2[k];
#pragma unroll
                for (int oo = 0; oo < 8; ++oo) s8[oo] += hv * wr[(size_t)oo * K3 + k];
            }
#pragma unroll
            for (int oo = 0; oo < 8; ++oo) {
                float sv = s8[oo];
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) sv += __shfl_down(sv, off, 64);
                if (lane == 0) {
                    const int o = wv * 8 + oo;
                    const float v = lrelu(sv + w.b3[o], sl);
                    h3[o] = v;
                    a.h3buf[(size_t)smp * LH + o] = v;
                }
            }
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
    for (int i = tid; i < H * W; i += NT) {
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
__global__ __launch_bounds__(NT) void locnet_bwd_k(LocBwd a) {
    extern __shared__ float sm[];
    __shared__ double shd[6][NT / 64];
    __shared__ float gTs[6];
    const int S = a.S, Cin = a.Cin, S2 = S * S, Sh = S / 2, K3 = LC * Sh * Sh, tid = threadIdx.x, CinQ = (Cin + 3) & ~3;
    const int smp = blockIdx.x, g = smp / a.N;
    const LocW w = a.g[g];
    const BwdLds L = bwd_lds(S, Cin);
    float *ga2 = sm + L.ga2, *ga1 = sm + L.ga1, *w2l = sm + L.w2l, *w1l = sm + L.w1l, *gp = sm + L.gp, *gh2 = sm + L.gh2, *g3s = sm + L.g3,
          *g4s = sm + L.g4, *zero = sm + L.zero, *part = sm + L.part;
    const float sl = a.slope;

    // affine_grid_backward: gT[r][:] = sum_{i,j} ggrid[i,j,r] * (y_i, x_j, 1)   (fp64 block sums, fixed order)
    const int H = a.Hg, W = a.Wg;
    const float* gg = a.ggrid + (size_t)smp * H * W * 2;
    double acc[6] = {0, 0, 0, 0, 0, 0};
    for (int i = tid; i < H * W; i += NT) {
        const int ii = i / W, j = i - ii * W;
        const float y = H > 1 ? -1.f + 2.f * (float)ii / (float)(H - 1) : -1.f;
        const float x = W > 1 ? -1.f + 2.f * (float)j / (float)(W - 1) : -1.f;
        const float g0 = gg[i * 2], g1 = gg[i * 2 + 1];
        acc[0] += g0 * y; acc[1] += g0 * x; acc[2] += g0;
        acc[3] += g1 * y; acc[4] += g1 * x; acc[5] += g1;
    }
    for (int k = 0; k < 6; ++k) {             // wave sums (fixed shuffle tree), then the waves in order
        const double t = cg::wave_sum(acc[k]);
        if ((tid & 63) == 0) shd[k][tid >> 6] = t;
    }
    __syncthreads();
    if (tid < 6) {
        double t = 0.0;
        for (int wv = 0; wv < NT / 64; ++wv) t += shd[tid][wv];
        gTs[tid] = (float)t;
    }
    // kernels, flipped for the data gradients: dx[ci] at pixel q = sum_{tap, co} dy[q - off(tap)][co] * w[co][ci][tap]
    // (cg_pack_conv_weight's backward copies: [((8 - t)*16 + co)][ci], rows of Cin floats)
    for (int d = tid; d < 9 * LC * LC; d += NT) w2l[d] = w.wb2[d];
    for (int d = tid; d < 9 * LC * CinQ; d += NT) {
        const int ci = d % CinQ, row = d / CinQ;
        w1l[d] = ci < Cin ? w.wb1[row * Cin + ci] : 0.f;
    }
    for (int i = tid; i < LC + 4; i += NT) zero[i] = 0.f;
    __syncthreads();
    // affine_matrix_backward (cg_affine_matrix_backward's formulas)
    if (tid == 0) {
        float gT[6];
        for (int k = 0; k < 6; ++k) gT[k] = gTs[k];
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
    // Linear(K3 -> 64) backward: gh2[i] = sum_o w3[o][i] g3[o]   (consecutive lanes, consecutive i: coalesced rows)
    for (int i0 = tid; i0 < K3; i0 += 4 * NT) {            // up to four columns per thread, 16 rows at a time: 64 loads in flight
        float s4[4] = {0.f, 0.f, 0.f, 0.f};
        for (int o0 = 0; o0 < LH; o0 += 16) {
            float wv[4][16];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = i0 + NT * j;
#pragma unroll
                for (int o = 0; o < 16; ++o) wv[j][o] = i < K3 ? w.w3[(size_t)(o0 + o) * K3 + i] : 0.f;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int o = 0; o < 16; ++o) s4[j] += wv[j][o] * g3s[o0 + o];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) if (i0 + NT * j < K3) gh2[i0 + NT * j] = s4[j];
    }
    __syncthreads();
    // AvgPool backward (x 0.25) + LeakyReLU backward at conv 2's output
    for (int i = tid; i < S2 * LC; i += NT) {
        const int c = i % LC, pix = i / LC, y = pix / S, x = pix % S;
        const float gv = gh2[c * Sh * Sh + (y >> 1) * Sh + (x >> 1)] * 0.25f;
        const float m = a.m2buf[((size_t)smp * S2 + pix) * LC + c];
        const float v = m >= 0.f ? gv : sl * gv;
        ga2[pix * (LC + 1) + c] = v;
        a.ga2[((size_t)smp * S2 + pix) * LC + c] = v;
    }
    __syncthreads();
    // conv 2 data gradient, then LeakyReLU backward at conv 1's output (in place)
    conv3x3_lds(ga2, LC + 1, LC, w2l, LC, zero, S, LC, ga1, LC + 1, part);
    for (int i = tid; i < S2 * LC; i += NT) {
        const int c = i % LC, pix = i / LC;
        const float h = a.h1buf[((size_t)smp * S2 + pix) * LC + c];
        const float r = ga1[pix * (LC + 1) + c];
        const float v = h >= 0.f ? r : sl * r;
        ga1[pix * (LC + 1) + c] = v;
        a.ga1[((size_t)smp * S2 + pix) * LC + c] = v;
    }
    __syncthreads();
    // conv 1 data gradient (output planes = the layer's input planes, padded to a multiple of 4)
    conv3x3_lds(ga1, LC + 1, LC, w1l, CinQ, zero, S, CinQ, gp, CinQ + 1, part);
    // AvgPool backward to the transformer's input resolution
    float* gx = a.gx + (size_t)smp * (4 * S2) * Cin;
    for (int i = tid; i < 4 * S2 * Cin; i += NT) {
        const int c = i % Cin, xx = (i / Cin) % (2 * S), yy = i / (Cin * 2 * S);
        gx[i] = gp[((yy >> 1) * S + (xx >> 1)) * (CinQ + 1) + c] * 0.25f;
    }
}


// ================================================================================================ round 4: the same chain on the MFMA
// The kernels above keep one sample's maps AND kernels in ~100 KB of LDS (one workgroup of 768 threads per CU, the 384 samples of
// the three branch transformers in 1.5 rounds) and run the two convolutions at VALU + LDS-read rates with the nine taps split over
// threads whose partial sums meet in LDS: 114 us forward / 116-166 us backward for the three-branch launch at batch 128, ~0.01 of any
// roofline (profiles/r03_locnet_phases.txt).  Here, for the shapes D32_st3 uses at 32x32 (S 16 / 3 planes; S 8 / 64 planes):
//   * both convolutions are implicit GEMMs on v_mfma_f32_16x16x4_f32 (M = 16 pixels, N = the 16 output planes, K = 4 input planes of
//     one tap per instruction): the A operand is ONE ds_read_b32 per lane out of a zero-haloed [(S+2)^2][C + 2] image - no bounds
//     test, the tap a compile-time address offset, pixel stride = 2 (mod 32) so that 16 pixels x 2 K-lanes fill 32 distinct banks -
//     and the B operand (the packed kernel) sits in REGISTERS: each wave loads its 36 fragments straight from L2 at the top of the
//     kernel, before the input is even pooled.  No kernel copy in LDS, no tap split, no partial-sum round trip (the 64-plane first
//     convolution splits its INPUT PLANES over the four waves instead: 36 K-steps each, one fixed-order 4-term sum at the end);
//   * a workgroup is 4 waves and 30-50 KB of LDS: 3-5 samples share a CU, so the launch is one round and the dependent phases of one
//     sample (global load -> pool -> conv -> conv -> pool -> 2 linear layers -> grid) hide under the others';
//   * the 16 K3 x 64 linear layer streams its rows as 16-byte loads, 4-8 rows in flight per wave, against register-held input.
// Buffers, layouts and entry points are unchanged (the weight gradients stay on the GEMM path); sums re-associate within fp32
// (K-steps of 4 input planes accumulate in tap order).  The round-3 VALU kernels remain the fallback for the geometries v2::has() does not list.
namespace v2 {

constexpr int NW = 4, NT2 = 64 * NW;
typedef float f4 __attribute__((ext_vector_type(4)));

template <int S_, int CIN_> struct Cfg {
    static constexpr int S = S_, CIN = CIN_, S2 = S * S, SP = S + 2, NP = SP * SP, MT = S2 / 16;
    static constexpr int CK1 = (CIN + 3) & ~3;          // input planes per tap as GEMM K (zero-padded to the K-step)
    static constexpr int CP1 = CK1 + 2, CP2 = LC + 2;   // LDS pixel strides: = 2 (mod 4), 6 / 18 / 66 fill the 32 banks of a ds_read_b32 group
    static constexpr bool KSPLIT = CIN >= 4 * NW * 4;   // conv 1: input planes over the waves (else: M tiles over the waves)
    static constexpr int NCH1 = KSPLIT ? CK1 / (4 * NW) : CK1 / 4;   // K-steps per tap a wave runs in conv 1
    static constexpr int NMT1 = KSPLIT ? MT : MT / NW;  // M tiles per wave in conv 1
    static constexpr int NMT2 = MT / NW;                // ... in conv 2 (M split)
    static constexpr int Sh = S / 2, K3 = LC * Sh * Sh;
    static constexpr int RS = 20;                        // row stride of the K-split partials
    static_assert(MT % NW == 0 && (S & (S - 1)) == 0, "shape");
    // forward LDS (floats)
    static constexpr int F_P = 0, F_H1 = F_P + NP * CP1, F_M2 = F_H1 + NP * CP2, F_H2 = F_M2 + S2 * (LC + 1), F_H3 = F_H2 + K3, F_PRM = F_H3 + LH,
                         F_TOT = F_PRM + 8;
    static constexpr int F_RED = F_P;                    // the K-split partials reuse the pooled input's image once every wave is through with it
    static_assert(!KSPLIT || NW * S2 * RS <= NP * CP1, "partials fit the input image");
    // backward
    static constexpr bool NSPLIT = CIN >= 16 * NW;      // conv 1 data gradient: output (= the layer's input) planes over the waves
    static constexpr int CQ = (CIN + 3) & ~3;
    static constexpr int B_GA2 = 0, B_GA1 = B_GA2 + NP * CP2, B_GP = B_GA1 + NP * CP2, B_GH2 = B_GP + S2 * (CQ + 1), B_G3 = B_GH2 + K3,
                         B_G4 = B_G3 + LH, B_TOT = B_G4 + 8;
    static_assert((NP * CP1) % 4 == 0 && (NP * CP2) % 4 == 0 && (S2 * (LC + 1)) % 4 == 0, "16-byte sections");
};

__device__ __forceinline__ int pp(int pix, int S, int SP) { return ((pix / S) + 1) * SP + (pix % S) + 1; }   // S: compile-time power of two

// acc[m] += A_m (16 pixels x 4 planes) * B (4 planes x 16) over 9 taps x NCH K-steps.  in: zero-haloed LDS image, pixel stride CP;
// ab[m]: this lane's element of tile m at tap (0,0), plane kq (+ the wave's plane offset); b: this lane's B elements, [tap][K-step].
template <int NMT, int NCH, int CP, int SP>
__device__ __forceinline__ void conv_mfma(const float* in, const int (&ab)[NMT], const float (&b)[9 * NCH], f4 (&acc)[NMT]) {
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int toff = ((t / 3 - 1) * SP + (t % 3 - 1)) * CP;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
#pragma unroll
            for (int m = 0; m < NMT; ++m)
                acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(in[ab[m] + toff + 4 * j], b[t * NCH + j], acc[m], 0, 0, 0);
        }
    }
}

// B fragments of a 16 -> 16 convolution from a packed copy w[(tap*16 + k)][16]: lane (n = l & 15, kq = l >> 4) takes w[(t*16 + 4j + kq)][n]
__device__ __forceinline__ void load_b16(const float* __restrict__ w, int lane, float (&b)[36]) {
    const int n = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) b[t * 4 + j] = w[((t * LC + 4 * j + kq) * LC) + n];
}

template <int S, int CIN>
__global__ __launch_bounds__(NT2) void locnet_fwd2_k(LocFwd a) {
    // Highest wave priority: this is a latency-bound kernel at the head of D's chain that in the D-step becomes ready beside the second
    // generator pass's Winograd GEMM (which keeps every SIMD's MFMA pipe and LDS busy): 249 -> 190 us in the traced step, step 5.62 -> 5.59 ms
    // same box (profiles/r06_sweeps.txt, call r06o).  What remains is contention a priority cannot remove (19 us alone).
    __builtin_amdgcn_s_setprio(3);
    using C = Cfg<S, CIN>;
    extern __shared__ float sm[];
    float *pP = sm + C::F_P, *h1P = sm + C::F_H1, *m2 = sm + C::F_M2, *red = sm + C::F_RED, *h2 = sm + C::F_H2, *h3 = sm + C::F_H3, *prm = sm + C::F_PRM;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, n = lane & 15, kq = lane >> 4;
    const int smp = blockIdx.x, g = smp / a.N, ns = smp - g * a.N;
    const LocW w = a.g[g];
    const float sl = a.slope;
    constexpr int S2 = C::S2, SP = C::SP;

    // (0) B fragments of both convolutions: the first loads of the kernel, in flight while the input is pooled
    float b1[9 * C::NCH1], b2[36];
    if (C::KSPLIT) {
        const int cb = (C::CK1 / NW) * wv;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int j = 0; j < C::NCH1; ++j) b1[t * C::NCH1 + j] = w.wf1[((t * CIN + cb + 4 * j + kq) * LC) + n];
    } else {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int j = 0; j < C::NCH1; ++j) { const int ci = 4 * j + kq; b1[t * C::NCH1 + j] = ci < CIN ? w.wf1[((t * CIN + ci) * LC) + n] : 0.f; }
    }
    load_b16(w.wf2, lane, b2);
    // (1) zero the haloed images (the interiors are overwritten below)
    {
        float4* z = reinterpret_cast<float4*>(sm);
        for (int i = tid; i < C::F_M2 / 4; i += NT2) z[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    // (2) AvgPool(2,2,2,2) of the sample (cg_avgpool2_forward's order of additions)
    {
        const float* xs = a.x + (size_t)(a.x_shared ? ns : smp) * (4 * S2) * CIN;
        float* pb = a.pbuf + (size_t)smp * S2 * CIN;
        if (CIN % 4 == 0) {
            constexpr int C4 = CIN / 4 > 0 ? CIN / 4 : 1;
            for (int i = tid; i < S2 * C4; i += NT2) {
                const int c4 = i % C4, pix = i / C4, px = pix % S, py = pix / S;
                const float4* q = reinterpret_cast<const float4*>(xs + ((size_t)(2 * py) * (2 * S) + 2 * px) * CIN) + c4;
                const float4 v0 = q[0], v1 = q[C4], v2 = q[2 * S * C4], v3 = q[2 * S * C4 + C4];
                float4 v;
                v.x = (v0.x + v1.x + v2.x + v3.x) * 0.25f; v.y = (v0.y + v1.y + v2.y + v3.y) * 0.25f;
                v.z = (v0.z + v1.z + v2.z + v3.z) * 0.25f; v.w = (v0.w + v1.w + v2.w + v3.w) * 0.25f;
                float2* d = reinterpret_cast<float2*>(pP + pp(pix, S, SP) * C::CP1 + 4 * c4);   // 8-byte aligned: CP1 is even
                d[0] = make_float2(v.x, v.y); d[1] = make_float2(v.z, v.w);
                reinterpret_cast<float4*>(pb)[i] = v;
            }
        } else {
            for (int i = tid; i < S2 * CIN; i += NT2) {
                const int c = i % CIN, pix = i / CIN, px = pix % S, py = pix / S;
                const size_t b = ((size_t)(2 * py) * (2 * S) + 2 * px) * CIN + c;
                const float v = (xs[b] + xs[b + CIN] + xs[b + (size_t)2 * S * CIN] + xs[b + (size_t)2 * S * CIN + CIN]) * 0.25f;
                pP[pp(pix, S, SP) * C::CP1 + c] = v;
                pb[i] = v;
            }
        }
    }
    __syncthreads();
    // (3) conv3x3 CIN -> 16, bias, LeakyReLU
    {
        f4 acc[C::NMT1];
        int ab[C::NMT1];
#pragma unroll
        for (int m = 0; m < C::NMT1; ++m) {
            acc[m] = (f4){0.f, 0.f, 0.f, 0.f};
            const int mt = C::KSPLIT ? m : wv * C::NMT1 + m;
            ab[m] = pp(mt * 16 + n, S, SP) * C::CP1 + kq + (C::KSPLIT ? (C::CK1 / NW) * wv : 0);
        }
        conv_mfma<C::NMT1, C::NCH1, C::CP1, SP>(pP, ab, b1, acc);
        if (C::KSPLIT) {
            __syncthreads();   // every wave has read its planes of the input image: the partials go on top of it
#pragma unroll
            for (int m = 0; m < C::NMT1; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[(wv * S2 + m * 16 + kq * 4 + r) * C::RS + n] = acc[m][r];
            __syncthreads();
            for (int e = tid; e < S2 * LC; e += NT2) {
                const int c = e & (LC - 1), pix = e >> 4;
                float v = red[pix * C::RS + c];
#pragma unroll
                for (int q = 1; q < NW; ++q) v += red[(q * S2 + pix) * C::RS + c];     // fixed order
                v = lrelu(v + w.b1[c], sl);
                h1P[pp(pix, S, SP) * C::CP2 + c] = v;
                a.h1buf[((size_t)smp * S2 + pix) * LC + c] = v;
            }
        } else {
            const float bias = w.b1[n];
#pragma unroll
            for (int m = 0; m < C::NMT1; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int pix = (wv * C::NMT1 + m) * 16 + kq * 4 + r;
                    const float v = lrelu(acc[m][r] + bias, sl);
                    h1P[pp(pix, S, SP) * C::CP2 + n] = v;
                    a.h1buf[((size_t)smp * S2 + pix) * LC + n] = v;
                }
        }
    }
    __syncthreads();
    // (4) conv3x3 16 -> 16, bias, LeakyReLU
    {
        f4 acc[C::NMT2];
        int ab[C::NMT2];
#pragma unroll
        for (int m = 0; m < C::NMT2; ++m) {
            acc[m] = (f4){0.f, 0.f, 0.f, 0.f};
            ab[m] = pp((wv * C::NMT2 + m) * 16 + n, S, SP) * C::CP2 + kq;
        }
        conv_mfma<C::NMT2, 4, C::CP2, SP>(h1P, ab, b2, acc);
        const float bias = w.b2[n];
#pragma unroll
        for (int m = 0; m < C::NMT2; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int pix = (wv * C::NMT2 + m) * 16 + kq * 4 + r;
                const float v = lrelu(acc[m][r] + bias, sl);
                m2[pix * (LC + 1) + n] = v;
                a.m2buf[((size_t)smp * S2 + pix) * LC + n] = v;
            }
    }
    __syncthreads();
    // (5) AvgPool(2); i = (c, py, px): nn.View(16*h*h) flattens the NCHW map
    for (int i = tid; i < C::K3; i += NT2) {
        const int px = i % C::Sh, py = (i / C::Sh) % C::Sh, c = i / (C::Sh * C::Sh);
        const float* q = m2 + ((2 * py) * S + 2 * px) * (LC + 1) + c;
        const float pv = (q[0] + q[LC + 1] + q[S * (LC + 1)] + q[(S + 1) * (LC + 1)]) * 0.25f;
        h2[i] = pv;
        a.h2buf[(size_t)smp * C::K3 + i] = pv;
    }
    __syncthreads();
    // (6) Linear(K3 -> 64) + LeakyReLU: a wave takes 16 outputs, RB rows at a time; lane l holds input elements 4l + 256j .. +3 and
    // reads the same columns of every row as 16-byte loads (a row = K3 / 256 coalesced 1 KB requests per wave)
    {
        constexpr int KJ = C::K3 / 256, RB = KJ >= 4 ? 4 : 8;
        static_assert(C::K3 % 256 == 0, "K3");
        float4 hv[KJ];
#pragma unroll
        for (int j = 0; j < KJ; ++j) hv[j] = *reinterpret_cast<const float4*>(h2 + 4 * lane + 256 * j);
        const float* wr = w.w3 + (size_t)(16 * wv) * C::K3 + 4 * lane;
#pragma unroll 1
        for (int ob = 0; ob < 16; ob += RB) {
            float4 wq[RB][KJ];
#pragma unroll
            for (int r = 0; r < RB; ++r)
#pragma unroll
                for (int j = 0; j < KJ; ++j) wq[r][j] = *reinterpret_cast<const float4*>(wr + (size_t)(ob + r) * C::K3 + 256 * j);
#pragma unroll
            for (int r = 0; r < RB; ++r) {
                float sv = 0.f;
#pragma unroll
                for (int j = 0; j < KJ; ++j) sv += hv[j].x * wq[r][j].x + hv[j].y * wq[r][j].y + hv[j].z * wq[r][j].z + hv[j].w * wq[r][j].w;
                sv = cg::wave_sum(sv);
                if (lane == 0) {
                    const int o = 16 * wv + ob + r;
                    const float v = lrelu(sv + w.b3[o], sl);
                    h3[o] = v;
                    a.h3buf[(size_t)smp * LH + o] = v;
                }
            }
        }
    }
    __syncthreads();
    // (7) Linear(64 -> P): one wave, lane k holds h3[k]
    if (wv == 0) {
        const float hk = h3[lane];
        for (int p_ = 0; p_ < a.P; ++p_) {
            const float sv = cg::wave_sum(hk * w.w4[p_ * LH + lane]);
            if (lane == 0) { const float v = sv + w.b4[p_]; prm[p_] = v; a.params[(size_t)smp * a.P + p_] = v; }
        }
    }
    __syncthreads();
    // (8) AffineTransformMatrixGenerator + AffineGridGeneratorBHWD
    float T[6];
    affine_T(prm, a.ur, a.us, a.ut, T);
    const int H = a.Hg, W = a.Wg;
    float2* gr = reinterpret_cast<float2*>(a.grid + (size_t)smp * H * W * 2);
    for (int i = tid; i < H * W; i += NT2) {
        const int ii = i / W, j = i - ii * W;
        const float y = H > 1 ? -1.f + 2.f * (float)ii / (float)(H - 1) : -1.f;
        const float x = W > 1 ? -1.f + 2.f * (float)j / (float)(W - 1) : -1.f;
        gr[i] = make_float2(T[0] * y + T[1] * x + T[2], T[3] * y + T[4] * x + T[5]);
    }
}

template <int S, int CIN>
__global__ __launch_bounds__(NT2) void locnet_bwd2_k(LocBwd a) {
    __builtin_amdgcn_s_setprio(3);   // as locnet_fwd2_k
    using C = Cfg<S, CIN>;
    extern __shared__ float sm[];
    __shared__ double shd[6][NW];
    __shared__ float gTs[6];
    float *ga2P = sm + C::B_GA2, *ga1P = sm + C::B_GA1, *gp = sm + C::B_GP, *gh2 = sm + C::B_GH2, *g3s = sm + C::B_G3, *g4s = sm + C::B_G4;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, n = lane & 15, kq = lane >> 4;
    const int smp = blockIdx.x, g = smp / a.N;
    const LocW w = a.g[g];
    const float sl = a.slope;
    constexpr int S2 = C::S2, SP = C::SP, K3 = C::K3, Sh = C::Sh, CQ = C::CQ;

    // (0) B fragments of both data-gradient convolutions (cg_pack_conv_weight's flipped copies [((8 - t)*16 + co)][ci])
    float b2[36], b1[36];
    load_b16(w.wb2, lane, b2);
    {
        const int nb = C::NSPLIT ? 16 * wv + n : n;       // output plane (= input plane of the layer) this lane's B column is
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) b1[t * 4 + j] = nb < CIN ? w.wb1[((t * LC + 4 * j + kq) * CIN) + nb] : 0.f;
    }
    {
        float4* z = reinterpret_cast<float4*>(sm);
        for (int i = tid; i < C::B_GP / 4; i += NT2) z[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // (1) affine_grid_backward: gT[r][:] = sum_{i,j} ggrid[i,j,r] * (y_i, x_j, 1)   (fp64 block sums, fixed order)
    const int H = a.Hg, W = a.Wg;
    {
        const float2* gg = reinterpret_cast<const float2*>(a.ggrid + (size_t)smp * H * W * 2);
        double acc[6] = {0, 0, 0, 0, 0, 0};
        for (int i = tid; i < H * W; i += NT2) {
            const int ii = i / W, j = i - ii * W;
            const float y = H > 1 ? -1.f + 2.f * (float)ii / (float)(H - 1) : -1.f;
            const float x = W > 1 ? -1.f + 2.f * (float)j / (float)(W - 1) : -1.f;
            const float2 gv = gg[i];
            acc[0] += gv.x * y; acc[1] += gv.x * x; acc[2] += gv.x;
            acc[3] += gv.y * y; acc[4] += gv.y * x; acc[5] += gv.y;
        }
        for (int k = 0; k < 6; ++k) {
            const double t = cg::wave_sum(acc[k]);
            if (lane == 0) shd[k][wv] = t;
        }
    }
    __syncthreads();
    // (2) affine_matrix_backward (cg_affine_matrix_backward's formulas), one lane
    if (tid == 0) {
        float gT[6];
        for (int k = 0; k < 6; ++k) { double t = 0.0; for (int q = 0; q < NW; ++q) t += shd[k][q]; gT[k] = (float)t; gTs[k] = gT[k]; }
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
    // (3) Linear(64 -> P) backward + LeakyReLU backward
    if (tid < LH) {
        float s = 0.f;
        for (int q = 0; q < a.P; ++q) s += w.w4[q * LH + tid] * g4s[q];
        const float h = a.h3buf[(size_t)smp * LH + tid];
        const float v = h >= 0.f ? s : sl * s;
        g3s[tid] = v;
        a.g3[(size_t)smp * LH + tid] = v;
    }
    __syncthreads();
    // (4) Linear(K3 -> 64) backward: gh2[i] = sum_o w3[o][i] g3[o]; a thread owns K3 / 256 consecutive columns (one 16-byte or 4-byte
    // load per row), 16 rows in flight, rows added in order
    {
        constexpr int CW = K3 / NT2;      // 1 or 4
        static_assert(CW == 1 || CW == 4, "K3");
        float s4[4] = {0.f, 0.f, 0.f, 0.f};
        const float* wc = w.w3 + CW * tid;
#pragma unroll 1
        for (int o0 = 0; o0 < LH; o0 += 16) {
            float wq[16][CW];
#pragma unroll
            for (int o = 0; o < 16; ++o) {
                if (CW == 4) { const float4 t = *reinterpret_cast<const float4*>(wc + (size_t)(o0 + o) * K3); wq[o][0] = t.x; wq[o][1 % CW] = t.y; wq[o][2 % CW] = t.z; wq[o][3 % CW] = t.w; }
                else wq[o][0] = wc[(size_t)(o0 + o) * K3];
            }
#pragma unroll
            for (int o = 0; o < 16; ++o)
#pragma unroll
                for (int j = 0; j < CW; ++j) s4[j] += wq[o][j] * g3s[o0 + o];
        }
#pragma unroll
        for (int j = 0; j < CW; ++j) gh2[CW * tid + j] = s4[j];
    }
    __syncthreads();
    // (5) AvgPool backward (x 0.25) + LeakyReLU backward at conv 2's output -> ga2 (haloed image + global, for the weight gradient)
    for (int i = tid; i < S2 * LC; i += NT2) {
        const int c = i & (LC - 1), pix = i >> 4, y = pix / S, x = pix % S;
        const float gv = gh2[c * Sh * Sh + (y >> 1) * Sh + (x >> 1)] * 0.25f;
        const float m = a.m2buf[((size_t)smp * S2 + pix) * LC + c];
        const float v = m >= 0.f ? gv : sl * gv;
        ga2P[pp(pix, S, SP) * C::CP2 + c] = v;
        a.ga2[((size_t)smp * S2 + pix) * LC + c] = v;
    }
    __syncthreads();
    // (6) conv 2 data gradient, LeakyReLU backward at conv 1's output -> ga1
    {
        f4 acc[C::NMT2];
        int ab[C::NMT2];
#pragma unroll
        for (int m = 0; m < C::NMT2; ++m) {
            acc[m] = (f4){0.f, 0.f, 0.f, 0.f};
            ab[m] = pp((wv * C::NMT2 + m) * 16 + n, S, SP) * C::CP2 + kq;
        }
        conv_mfma<C::NMT2, 4, C::CP2, SP>(ga2P, ab, b2, acc);
#pragma unroll
        for (int m = 0; m < C::NMT2; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int pix = (wv * C::NMT2 + m) * 16 + kq * 4 + r;
                const float h = a.h1buf[((size_t)smp * S2 + pix) * LC + n];
                const float v = h >= 0.f ? acc[m][r] : sl * acc[m][r];
                ga1P[pp(pix, S, SP) * C::CP2 + n] = v;
                a.ga1[((size_t)smp * S2 + pix) * LC + n] = v;
            }
    }
    __syncthreads();
    // (7) conv 1 data gradient: N = the layer's input planes (64: one 16-plane tile per wave over all pixels; 3: zero-padded tile, pixels over the waves)
    {
        constexpr int NMT = C::NSPLIT ? C::MT : C::MT / NW;
        f4 acc[NMT];
        int ab[NMT];
#pragma unroll
        for (int m = 0; m < NMT; ++m) {
            acc[m] = (f4){0.f, 0.f, 0.f, 0.f};
            const int mt = C::NSPLIT ? m : wv * NMT + m;
            ab[m] = pp(mt * 16 + n, S, SP) * C::CP2 + kq;
        }
        conv_mfma<NMT, 4, C::CP2, SP>(ga1P, ab, b1, acc);
        const int nb = C::NSPLIT ? 16 * wv + n : n;
#pragma unroll
        for (int m = 0; m < NMT; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int pix = (C::NSPLIT ? m : wv * NMT + m) * 16 + kq * 4 + r;
                if (nb < CQ) gp[pix * (CQ + 1) + nb] = acc[m][r];
            }
    }
    __syncthreads();
    // (8) AvgPool backward to the transformer's input resolution
    float* gx = a.gx + (size_t)smp * (4 * S2) * CIN;
    for (int i = tid; i < 4 * S2 * CIN; i += NT2) {
        const int c = i % CIN, xx = (i / CIN) % (2 * S), yy = i / (CIN * 2 * S);
        gx[i] = gp[((yy >> 1) * S + (xx >> 1)) * (CQ + 1) + c] * 0.25f;
    }
}

template <int S, int CIN> int launch_fwd(hipStream_t st, const LocFwd& a, int nblk) {
    using C = Cfg<S, CIN>;
    constexpr size_t lds = (size_t)C::F_TOT * 4;
    static bool granted = false;
    if (lds > 64 * 1024 && !granted) { CG_HIP(hipFuncSetAttribute((const void*)locnet_fwd2_k<S, CIN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); granted = true; }
    hipLaunchKernelGGL((locnet_fwd2_k<S, CIN>), dim3(nblk), dim3(NT2), lds, st, a);
    return 0;
}
template <int S, int CIN> int launch_bwd(hipStream_t st, const LocBwd& a, int nblk) {
    using C = Cfg<S, CIN>;
    constexpr size_t lds = (size_t)C::B_TOT * 4;
    static bool granted = false;
    if (lds > 64 * 1024 && !granted) { CG_HIP(hipFuncSetAttribute((const void*)locnet_bwd2_k<S, CIN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); granted = true; }
    hipLaunchKernelGGL((locnet_bwd2_k<S, CIN>), dim3(nblk), dim3(NT2), lds, st, a);
    return 0;
}
// (16, 64) - the branch transformers of D32_st3 at 64x64 (config #5) - was parity-clean in round 4 but is NOT instantiated: it needs
// 131 / 118 KB of LDS and 260 VGPRs, i.e. one workgroup per CU for 192 samples, and the ten separate (grouped) launches it would
// replace are faster: config #5 12.67 ms per step without, 12.81 with (profiles/r04_sweeps.txt).
inline bool has(int S, int Cin) { return (S == 16 && Cin == 3) || (S == 8 && Cin == 64); }
inline bool enabled() { return true; }

}  // namespace v2

size_t fwd_lds_floats(int S, int Cin) { return (size_t)fwd_lds(S, Cin).total; }
size_t bwd_lds_floats(int S, int Cin) { return (size_t)bwd_lds(S, Cin).total; }

}  // namespace

extern "C" {

// 1 if the fused launches cover this localisation net: power-of-two pooled size S >= 8 (so that a wave shares its quad of output
// planes), the activations of one sample within LDS.
int cg_locnet_supported(int S, int Cin, int P) {
    if (S < 8 || (S & (S - 1)) || Cin < 1 || P < 1 || P > 4) return 0;
    if (v2::enabled() && v2::has(S, Cin)) return 1;     // the MFMA kernels bring their own (smaller) LDS plan
    const size_t lim = 160 * 1024 - 1024;   // the kernels' few static __shared__ words come out of the same 160 KB
    return fwd_lds_floats(S, Cin) * 4 <= lim && bwd_lds_floats(S, Cin) * 4 <= lim ? 1 : 0;
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
        const float* const* w = weights + 12 * g;
        for (int k = 0; k < 12; ++k) CG_REQUIRE(w[k], "cg_locnet_forward: null weight pointer");
        a.g[g] = LocW{w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7], w[8], w[9], w[10], w[11]};
    }
    a.x = x; a.x_shared = x_shared; a.G = ngroups; a.N = n_per_group; a.S = S; a.Cin = Cin; a.P = P; a.Hg = Hg; a.Wg = Wg;
    a.ur = use_rot ? 1 : 0; a.us = use_scale ? 1 : 0; a.ut = use_trans ? 1 : 0; a.slope = slope;
    a.pbuf = pooled; a.h1buf = h1; a.m2buf = m2; a.h2buf = h2; a.h3buf = h3; a.params = params; a.grid = grid;
    if (v2::enabled() && v2::has(S, Cin)) {   // the MFMA kernels (round 4)
        const int nb = ngroups * n_per_group;
        const int rc = S == 8 ? v2::launch_fwd<8, 64>(cg::S(stream), a, nb) : v2::launch_fwd<16, 3>(cg::S(stream), a, nb);
        if (rc) return rc;
        CG_LAUNCH_CHECK();
        return 0;
    }
    const size_t lds = fwd_lds_floats(S, Cin) * 4;
    static size_t granted = 64 * 1024;
    if (lds > granted) { CG_HIP(hipFuncSetAttribute((const void*)locnet_fwd_k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); granted = lds; }
    hipLaunchKernelGGL(locnet_fwd_k, dim3(ngroups * n_per_group), dim3(NT), lds, cg::S(stream), a);
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
        const float* const* w = weights + 12 * g;
        for (int k = 0; k < 12; ++k) CG_REQUIRE(w[k], "cg_locnet_backward: null weight pointer");
        a.g[g] = LocW{w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7], w[8], w[9], w[10], w[11]};
    }
    a.G = ngroups; a.N = n_per_group; a.S = S; a.Cin = Cin; a.P = P; a.Hg = Hg; a.Wg = Wg;
    a.ur = use_rot ? 1 : 0; a.us = use_scale ? 1 : 0; a.ut = use_trans ? 1 : 0; a.slope = slope;
    a.h1buf = h1; a.m2buf = m2; a.h3buf = h3; a.params = params; a.ggrid = ggrid;
    a.ga1 = ga1; a.ga2 = ga2; a.g3 = g3; a.g4 = g4; a.gx = gx;
    if (v2::enabled() && v2::has(S, Cin)) {
        const int nb = ngroups * n_per_group;
        const int rc = S == 8 ? v2::launch_bwd<8, 64>(cg::S(stream), a, nb) : v2::launch_bwd<16, 3>(cg::S(stream), a, nb);
        if (rc) return rc;
        CG_LAUNCH_CHECK();
        return 0;
    }
    const size_t lds = bwd_lds_floats(S, Cin) * 4;
    static size_t granted = 64 * 1024;
    if (lds > granted) { CG_HIP(hipFuncSetAttribute((const void*)locnet_bwd_k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); granted = lds; }
    hipLaunchKernelGGL(locnet_bwd_k, dim3(ngroups * n_per_group), dim3(NT), lds, cg::S(stream), a);
    CG_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
