/*
 * oracle/ops.c — CPU restatement (TEST INFRASTRUCTURE, not product code) of the
 * heavy operators on the hot path of aleju/cat-generator's adversarial.lua step.
 *
 * PARITY UNPINNED: the reference ships no tests, fixtures or golden vectors
 *  and cannot run in this container (no Lua/Torch7); its
 * arithmetic lives in un-vendored, un-pinned Torch7 rocks (torch/nn THNN,
 * soumith/cudnn.torch "cudnn3", qassemoquab/stnbhwd, torch/optim; era late
 * 2015 / early 2016, README.md:89-106).  This file restates the published
 * algorithms of those rocks at the reference's call sites; it is cross-checked
 * against PyTorch-CPU (tests/test_oracle_vs_torch.py), an independent
 * descendant of THNN.
 *
 * Layout: Torch7's — NCHW feature maps, weight [Cout][Cin][kH][kW],
 * linear weight [out][in]; stn tensors are BHWD.  fp32 throughout (sgemm
 * accumulates in float like the BLAS THNN calls).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef float v8sf __attribute__((vector_size(32), aligned(4)));

static inline v8sf ld8(const float* p) { return *(const v8sf*)p; }
static inline void st8(float* p, v8sf v) { *(v8sf*)p = v; }
static inline v8sf bc8(float a) { return (v8sf){a, a, a, a, a, a, a, a}; }

/* C[M][N] (+)= A[M][K] * B[K][N], row-major, single thread.
 * 4x16 register tile over k; plain float accumulation (sgemm class). */
static void sgemm_nn(int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                     int accumulate) {
    const int KB = 256;
    if (!accumulate)
        for (int i = 0; i < M; ++i) memset(C + (size_t)i * ldc, 0, sizeof(float) * N);
    for (int k0 = 0; k0 < K; k0 += KB) {
        const int k1 = k0 + KB < K ? k0 + KB : K;
        int i = 0;
        for (; i + 4 <= M; i += 4) {
            int j = 0;
            for (; j + 16 <= N; j += 16) {
                v8sf c00 = ld8(C + (size_t)(i + 0) * ldc + j), c01 = ld8(C + (size_t)(i + 0) * ldc + j + 8);
                v8sf c10 = ld8(C + (size_t)(i + 1) * ldc + j), c11 = ld8(C + (size_t)(i + 1) * ldc + j + 8);
                v8sf c20 = ld8(C + (size_t)(i + 2) * ldc + j), c21 = ld8(C + (size_t)(i + 2) * ldc + j + 8);
                v8sf c30 = ld8(C + (size_t)(i + 3) * ldc + j), c31 = ld8(C + (size_t)(i + 3) * ldc + j + 8);
                for (int k = k0; k < k1; ++k) {
                    const v8sf b0 = ld8(B + (size_t)k * ldb + j), b1 = ld8(B + (size_t)k * ldb + j + 8);
                    const v8sf a0 = bc8(A[(size_t)(i + 0) * lda + k]);
                    const v8sf a1 = bc8(A[(size_t)(i + 1) * lda + k]);
                    const v8sf a2 = bc8(A[(size_t)(i + 2) * lda + k]);
                    const v8sf a3 = bc8(A[(size_t)(i + 3) * lda + k]);
                    c00 += a0 * b0; c01 += a0 * b1;
                    c10 += a1 * b0; c11 += a1 * b1;
                    c20 += a2 * b0; c21 += a2 * b1;
                    c30 += a3 * b0; c31 += a3 * b1;
                }
                st8(C + (size_t)(i + 0) * ldc + j, c00); st8(C + (size_t)(i + 0) * ldc + j + 8, c01);
                st8(C + (size_t)(i + 1) * ldc + j, c10); st8(C + (size_t)(i + 1) * ldc + j + 8, c11);
                st8(C + (size_t)(i + 2) * ldc + j, c20); st8(C + (size_t)(i + 2) * ldc + j + 8, c21);
                st8(C + (size_t)(i + 3) * ldc + j, c30); st8(C + (size_t)(i + 3) * ldc + j + 8, c31);
            }
            for (; j < N; ++j)
                for (int ii = i; ii < i + 4; ++ii) {
                    float s = C[(size_t)ii * ldc + j];
                    for (int k = k0; k < k1; ++k) s += A[(size_t)ii * lda + k] * B[(size_t)k * ldb + j];
                    C[(size_t)ii * ldc + j] = s;
                }
        }
        for (; i < M; ++i) {
            for (int k = k0; k < k1; ++k) {
                const float a = A[(size_t)i * lda + k];
                const float* b = B + (size_t)k * ldb;
                float* c = C + (size_t)i * ldc;
                for (int j = 0; j < N; ++j) c[j] += a * b[j];
            }
        }
    }
}

static void transpose(const float* A, int rows, int cols, float* At) { /* At[cols][rows] */
    for (int r = 0; r < rows; ++r)
        for (int c = 0; c < cols; ++c) At[(size_t)c * rows + r] = A[(size_t)r * cols + c];
}

/* THNN unfolded-copy, columns [p0, p0 + pc) of it: col[(ci*kH+ky)*kW+kx][p - p0] = x[ci][oy+ky-padH][ox+kx-padW], p = oy*Wo+ox.
 * The convolutions below walk the output pixels in blocks of pc columns so that the unfolded block (K x pc floats) stays in a
 * core's L2 (THNN unfolds the whole image: 26 MB per thread for the 5x5 256->128 layer at 32x32, which is what kept this port from
 * scaling past ~16 threads).  Every output element still accumulates its K products in the same order. */
static void im2col_cols(const float* x, int C, int H, int W, int kH, int kW, int padH, int padW, int Wo, int p0, int pc,
                        float* col) {
    for (int ci = 0; ci < C; ++ci)
        for (int ky = 0; ky < kH; ++ky)
            for (int kx = 0; kx < kW; ++kx) {
                float* dst = col + (size_t)((ci * kH + ky) * kW + kx) * pc;
                int oy = p0 / Wo, ox = p0 % Wo;
                for (int q = 0; q < pc; ++q) {
                    const int iy = oy + ky - padH, ix = ox + kx - padW;
                    dst[q] = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? x[((size_t)ci * H + iy) * W + ix] : 0.f;
                    if (++ox == Wo) { ox = 0; ++oy; }
                }
            }
}
static void col2im_add_cols(const float* col, int C, int H, int W, int kH, int kW, int padH, int padW, int Wo, int p0, int pc,
                            float* x) {
    for (int ci = 0; ci < C; ++ci)
        for (int ky = 0; ky < kH; ++ky)
            for (int kx = 0; kx < kW; ++kx) {
                const float* src = col + (size_t)((ci * kH + ky) * kW + kx) * pc;
                int oy = p0 / Wo, ox = p0 % Wo;
                for (int q = 0; q < pc; ++q) {
                    const int iy = oy + ky - padH, ix = ox + kx - padW;
                    if (iy >= 0 && iy < H && ix >= 0 && ix < W) x[((size_t)ci * H + iy) * W + ix] += src[q];
                    if (++ox == Wo) { ox = 0; ++oy; }
                }
            }
}
/* output pixels per block: K x pc floats <= 1 MB, a multiple of 16 (the sgemm kernel's vector width) */
static int col_block(int K, int P) {
    int pc = (256 * 1024) / (K > 0 ? K : 1);
    pc &= ~15;
    if (pc < 16) pc = 16;
    return pc < P ? pc : P;
}

/* nn.SpatialConvolution / cudnn.SpatialConvolution updateOutput, stride 1
 * (models.lua:206,212,218,222,646-685): THNN SpatialConvolutionMM —
 * out[n] = bias (broadcast) + W[Cout][Cin*kH*kW] * im2col(x[n]). */
void orc_conv2d_forward(const float* x, const float* w, const float* b, float* y, int N, int Cin, int H, int W,
                        int Cout, int kH, int kW, int padH, int padW) {
    const int Ho = H + 2 * padH - kH + 1, Wo = W + 2 * padW - kW + 1, K = Cin * kH * kW, P = Ho * Wo, PC = col_block(K, P);
#pragma omp parallel
    {
        float* col = (float*)malloc(sizeof(float) * (size_t)K * PC);
#pragma omp for schedule(dynamic, 1)
        for (int n = 0; n < N; ++n) {
            float* yn = y + (size_t)n * Cout * P;
            for (int co = 0; co < Cout; ++co) {
                const float bv = b ? b[co] : 0.f;
                for (int p = 0; p < P; ++p) yn[(size_t)co * P + p] = bv;
            }
            for (int p0 = 0; p0 < P; p0 += PC) {
                const int pc = p0 + PC <= P ? PC : P - p0;
                im2col_cols(x + (size_t)n * Cin * H * W, Cin, H, W, kH, kW, padH, padW, Wo, p0, pc, col);
                sgemm_nn(Cout, pc, K, w, K, col, pc, yn + p0, P, 1);
            }
        }
        free(col);
    }
}

/* updateGradInput: col = W^T * dy[n]; dx[n] = col2im(col). */
void orc_conv2d_backward_data(const float* dy, const float* w, float* dx, int N, int Cin, int H, int W, int Cout,
                              int kH, int kW, int padH, int padW) {
    const int Ho = H + 2 * padH - kH + 1, Wo = W + 2 * padW - kW + 1, K = Cin * kH * kW, P = Ho * Wo, PC = col_block(K, P);
    float* wt = (float*)malloc(sizeof(float) * (size_t)K * Cout);
    transpose(w, Cout, K, wt);
#pragma omp parallel
    {
        float* col = (float*)malloc(sizeof(float) * (size_t)K * PC);
#pragma omp for schedule(dynamic, 1)
        for (int n = 0; n < N; ++n) {
            float* dxn = dx + (size_t)n * Cin * H * W;
            memset(dxn, 0, sizeof(float) * (size_t)Cin * H * W);
            for (int p0 = 0; p0 < P; p0 += PC) {
                const int pc = p0 + PC <= P ? PC : P - p0;
                sgemm_nn(K, pc, Cout, wt, Cout, dy + (size_t)n * Cout * P + p0, P, col, pc, 0);
                col2im_add_cols(col, Cin, H, W, kH, kW, padH, padW, Wo, p0, pc, dxn);
            }
        }
        free(col);
    }
    free(wt);
}

/* accGradParameters: gw += scale * sum_n dy[n] * im2col(x[n])^T ; gb += scale * sum dy. */
void orc_conv2d_backward_weight(const float* x, const float* dy, float* gw, float* gb, int N, int Cin, int H, int W,
                                int Cout, int kH, int kW, int padH, int padW, float scale) {
    const int Ho = H + 2 * padH - kH + 1, Wo = W + 2 * padW - kW + 1, K = Cin * kH * kW, P = Ho * Wo, PC = col_block(K, P);
    int nthreads = 1;
#ifdef _OPENMP
    nthreads = omp_get_max_threads();
#endif
    if (nthreads > N) nthreads = N > 0 ? N : 1;
    float* acc = (float*)calloc((size_t)nthreads * Cout * K, sizeof(float));
#pragma omp parallel num_threads(nthreads)
    {
        int tid = 0;
#ifdef _OPENMP
        tid = omp_get_thread_num();
#endif
        float* col = (float*)malloc(sizeof(float) * (size_t)K * PC);
        float* colt = (float*)malloc(sizeof(float) * (size_t)K * PC);
        float* a = acc + (size_t)tid * Cout * K;
#pragma omp for schedule(static)
        for (int n = 0; n < N; ++n) {
            for (int p0 = 0; p0 < P; p0 += PC) {
                const int pc = p0 + PC <= P ? PC : P - p0;
                im2col_cols(x + (size_t)n * Cin * H * W, Cin, H, W, kH, kW, padH, padW, Wo, p0, pc, col);
                transpose(col, K, pc, colt); /* colt[pc][K] */
                sgemm_nn(Cout, K, pc, dy + (size_t)n * Cout * P + p0, P, colt, K, a, K, 1);
            }
        }
        free(col);
        free(colt);
    }
    for (size_t i = 0; i < (size_t)Cout * K; ++i) {
        float s = 0.f;
        for (int t = 0; t < nthreads; ++t) s += acc[(size_t)t * Cout * K + i];
        gw[i] += scale * s;
    }
    free(acc);
    if (gb)
        for (int co = 0; co < Cout; ++co) {
            double s = 0.0;
            for (int n = 0; n < N; ++n) {
                const float* d = dy + ((size_t)n * Cout + co) * P;
                for (int p = 0; p < P; ++p) s += d[p];
            }
            gb[co] += scale * (float)s;
        }
}

/* nn.Linear (models.lua:199,697,700,850,853): y = x W^T + b */
void orc_linear_forward(const float* x, const float* w, const float* b, float* y, int N, int in, int out) {
    float* wt = (float*)malloc(sizeof(float) * (size_t)in * out);
    transpose(w, out, in, wt); /* wt[in][out] */
    const int chunk = 4;
#pragma omp parallel for schedule(static)
    for (int n0 = 0; n0 < N; n0 += chunk) {
        const int m = n0 + chunk <= N ? chunk : N - n0;
        for (int i = 0; i < m; ++i)
            for (int o = 0; o < out; ++o) y[(size_t)(n0 + i) * out + o] = b ? b[o] : 0.f;
        sgemm_nn(m, out, in, x + (size_t)n0 * in, in, wt, out, y + (size_t)n0 * out, out, 1);
    }
    free(wt);
}
/* dx = dy W */
void orc_linear_backward_data(const float* dy, const float* w, float* dx, int N, int in, int out) {
    const int chunk = 4;
#pragma omp parallel for schedule(static)
    for (int n0 = 0; n0 < N; n0 += chunk) {
        const int m = n0 + chunk <= N ? chunk : N - n0;
        sgemm_nn(m, in, out, dy + (size_t)n0 * out, out, w, in, dx + (size_t)n0 * in, in, 0);
    }
}
/* gw[out][in] += scale * dy^T x ; gb += scale * colsum(dy) */
void orc_linear_backward_weight(const float* x, const float* dy, float* gw, float* gb, int N, int in, int out,
                                float scale) {
    float* dyt = (float*)malloc(sizeof(float) * (size_t)N * out);
    transpose(dy, N, out, dyt); /* [out][N] */
    float* tmp = (float*)malloc(sizeof(float) * (size_t)out * in);
    const int chunk = 4;
#pragma omp parallel for schedule(static)
    for (int o0 = 0; o0 < out; o0 += chunk) {
        const int m = o0 + chunk <= out ? chunk : out - o0;
        sgemm_nn(m, in, N, dyt + (size_t)o0 * N, N, x, in, tmp + (size_t)o0 * in, in, 0);
    }
    for (size_t i = 0; i < (size_t)out * in; ++i) gw[i] += scale * tmp[i];
    if (gb)
        for (int o = 0; o < out; ++o) {
            double s = 0.0;
            for (int n = 0; n < N; ++n) s += dyt[(size_t)o * N + n];
            gb[o] += scale * (float)s;
        }
    free(tmp);
    free(dyt);
}

/* nn.BilinearSamplerBHWD (stnbhwd generic/BilinearSamplerBHWD.c; models.lua:888):
 * img [N,Hi,Wi,C], grid [N,Ho,Wo,2] = (y,x) in [-1,1]; src = (coord+1)*(size-1)/2;
 * taps outside the image contribute 0. */
void orc_bilinear_forward(const float* img, const float* grid, float* out, int N, int Hi, int Wi, int C, int Ho,
                          int Wo) {
#pragma omp parallel for schedule(static)
    for (int n = 0; n < N; ++n)
        for (int oy = 0; oy < Ho; ++oy)
            for (int ox = 0; ox < Wo; ++ox) {
                const size_t pix = ((size_t)n * Ho + oy) * Wo + ox;
                const float yf = grid[pix * 2], xf = grid[pix * 2 + 1];
                const float xc = (xf + 1.f) * (float)(Wi - 1) / 2.f;
                const float yc = (yf + 1.f) * (float)(Hi - 1) / 2.f;
                const int x0 = (int)floorf(xc), y0 = (int)floorf(yc);
                const float wx = 1.f - (xc - (float)x0), wy = 1.f - (yc - (float)y0);
                const int in00 = x0 >= 0 && x0 <= Wi - 1 && y0 >= 0 && y0 <= Hi - 1;
                const int in01 = x0 + 1 >= 0 && x0 + 1 <= Wi - 1 && y0 >= 0 && y0 <= Hi - 1;
                const int in10 = x0 >= 0 && x0 <= Wi - 1 && y0 + 1 >= 0 && y0 + 1 <= Hi - 1;
                const int in11 = x0 + 1 >= 0 && x0 + 1 <= Wi - 1 && y0 + 1 >= 0 && y0 + 1 <= Hi - 1;
                for (int c = 0; c < C; ++c) {
                    float v = 0.f;
                    if (in00) v += wx * wy * img[(((size_t)n * Hi + y0) * Wi + x0) * C + c];
                    if (in01) v += (1.f - wx) * wy * img[(((size_t)n * Hi + y0) * Wi + x0 + 1) * C + c];
                    if (in10) v += wx * (1.f - wy) * img[(((size_t)n * Hi + y0 + 1) * Wi + x0) * C + c];
                    if (in11) v += (1.f - wx) * (1.f - wy) * img[(((size_t)n * Hi + y0 + 1) * Wi + x0 + 1) * C + c];
                    out[pix * C + c] = v;
                }
            }
}
void orc_bilinear_backward(const float* img, const float* grid, const float* gout, float* gimg, float* ggrid, int N,
                           int Hi, int Wi, int C, int Ho, int Wo) {
    memset(gimg, 0, sizeof(float) * (size_t)N * Hi * Wi * C);
#pragma omp parallel for schedule(static)
    for (int n = 0; n < N; ++n)
        for (int oy = 0; oy < Ho; ++oy)
            for (int ox = 0; ox < Wo; ++ox) {
                const size_t pix = ((size_t)n * Ho + oy) * Wo + ox;
                const float yf = grid[pix * 2], xf = grid[pix * 2 + 1];
                const float xc = (xf + 1.f) * (float)(Wi - 1) / 2.f;
                const float yc = (yf + 1.f) * (float)(Hi - 1) / 2.f;
                const int x0 = (int)floorf(xc), y0 = (int)floorf(yc);
                const float wx = 1.f - (xc - (float)x0), wy = 1.f - (yc - (float)y0);
                const int in00 = x0 >= 0 && x0 <= Wi - 1 && y0 >= 0 && y0 <= Hi - 1;
                const int in01 = x0 + 1 >= 0 && x0 + 1 <= Wi - 1 && y0 >= 0 && y0 <= Hi - 1;
                const int in10 = x0 >= 0 && x0 <= Wi - 1 && y0 + 1 >= 0 && y0 + 1 <= Hi - 1;
                const int in11 = x0 + 1 >= 0 && x0 + 1 <= Wi - 1 && y0 + 1 >= 0 && y0 + 1 <= Hi - 1;
                float d00 = 0.f, d01 = 0.f, d10 = 0.f, d11 = 0.f;
                for (int c = 0; c < C; ++c) {
                    const float g = gout[pix * C + c];
                    if (in00) {
                        const size_t a = (((size_t)n * Hi + y0) * Wi + x0) * C + c;
                        d00 += img[a] * g; gimg[a] += wx * wy * g;
                    }
                    if (in01) {
                        const size_t a = (((size_t)n * Hi + y0) * Wi + x0 + 1) * C + c;
                        d01 += img[a] * g; gimg[a] += (1.f - wx) * wy * g;
                    }
                    if (in10) {
                        const size_t a = (((size_t)n * Hi + y0 + 1) * Wi + x0) * C + c;
                        d10 += img[a] * g; gimg[a] += wx * (1.f - wy) * g;
                    }
                    if (in11) {
                        const size_t a = (((size_t)n * Hi + y0 + 1) * Wi + x0 + 1) * C + c;
                        d11 += img[a] * g; gimg[a] += (1.f - wx) * (1.f - wy) * g;
                    }
                }
                const float gy = -wx * d00 + wx * d10 - (1.f - wx) * d01 + (1.f - wx) * d11;
                const float gx = -wy * d00 + wy * d01 - (1.f - wy) * d10 + (1.f - wy) * d11;
                ggrid[pix * 2 + 0] = gy * (float)(Hi - 1) / 2.f;
                ggrid[pix * 2 + 1] = gx * (float)(Wi - 1) / 2.f;
            }
}

/* counter-based uniform in [0,1): splitmix64(seed, ctr) — the engine's own
 * generator (TH's MT19937 stream is not reproducible; SURVEY.md §7 "RNG"). */
void orc_rng_u01(float* out, long n, uint64_t seed, uint64_t offset) {
    for (long i = 0; i < n; ++i) {
        uint64_t z = seed + (offset + (uint64_t)i + 1) * 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z = z ^ (z >> 31);
        out[i] = (float)(z >> 40) * (1.0f / 16777216.0f);
    }
}

int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
void orc_set_num_threads(int n) {
#ifdef _OPENMP
    omp_set_num_threads(n);
#else
    (void)n;
#endif
}
