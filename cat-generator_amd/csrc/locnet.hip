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

size_t fwd_lds_floats(int S, int Cin) { return (size_t)fwd_lds(S, Cin).total; }
size_t bwd_lds_floats(int S, int Cin) { return (size_t)bwd_lds(S, Cin).total; }

}  // namespace

extern "C" {

// 1 if the fused launches cover this localisation net: power-of-two pooled size S >= 8 (so that a wave shares its quad of output
// planes), the activations of one sample within LDS.
int cg_locnet_supported(int S, int Cin, int P) {
    if (S < 8 || (S & (S - 1)) || Cin < 1 || P < 1 || P > 4) return 0;
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
    const size_t lds = bwd_lds_floats(S, Cin) * 4;
    static size_t granted = 64 * 1024;
    if (lds > granted) { CG_HIP(hipFuncSetAttribute((const void*)locnet_bwd_k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); granted = lds; }
    hipLaunchKernelGGL(locnet_bwd_k, dim3(ngroups * n_per_group), dim3(NT), lds, cg::S(stream), a);
    CG_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
