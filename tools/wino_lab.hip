// Laboratory for a FUSED-transform Winograd F(2x2,3x3) convolution with 64 input and 64 output planes (round 6):
// D32_st3's 3x3 layers (models.lua:648,655,659,664,668,673,677) run at direct count on igemm_nn_kernel<128,64> (97 us for the
// 32 x 32 layer at batch 128 = 0.63 of the fp32 MFMA peak).  With K = 64 an unfused Winograd pipeline would write and re-read a V
// four times the size of the input (134 MB per pass): memory-bound at the time the direct kernel takes.  Here nothing but x, y and the
// transformed filters U (256 KB, L2-resident) moves:
//   workgroup = 32 output tiles (8 x 16 pixels) of one image, the 10 x 18 pixel input patch (zero outside the image) in LDS once;
//   wave      = (32 output planes) x (two of the four position rows xi): per position the A operand B^T d B is formed IN REGISTERS
//               from four 16-byte LDS reads (3 VALU per element), the B operand U[pos] is a coalesced 16-byte global load (L2), 32
//               v_mfma_f32_32x32x2_f32 contract the 64 input planes, and A^T M A is folded into the accumulation: the position's
//               accumulator is added with its 0 / +-1 coefficients into the four output accumulators of the tile;
//   the two xi halves of a workgroup meet in LDS at the end, one of them stores y (+ bias).
// Executed MFMA work: 16 / 36 of the direct count.  Checked against an fp64 direct convolution on the host.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/wino_lab.hip -o tools/wino_lab       Run: tools/wino_lab [N] [H] [iters]
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int C = 64;          // input planes = output planes
constexpr int PLD = 68;        // floats per patch pixel (64 + 4: the 16-byte reads of 8 tiles along x hit 8 distinct 4-bank slots)
constexpr int PW = 18, PH = 10;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// U[pos][ci / 4][co][4] = (G g G^T)[xi][nu] for filter g = w[co][ci][:][:];  pos = xi * 4 + nu.  flip = the data gradient's filters
__global__ void pack_u(const float* __restrict__ w, float* __restrict__ U, int flip) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= C * C) return;
    const int co = idx / C, ci = idx % C;
    float g[3][3];
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) g[a][b] = flip ? w[((long)ci * C + co) * 9 + (2 - a) * 3 + (2 - b)] : w[((long)co * C + ci) * 9 + a * 3 + b];
    const float G[4][3] = {{1.f, 0.f, 0.f}, {.5f, .5f, .5f}, {.5f, -.5f, .5f}, {0.f, 0.f, 1.f}};
    float t[4][3];
    for (int i = 0; i < 4; ++i)
        for (int b = 0; b < 3; ++b) t[i][b] = G[i][0] * g[0][b] + G[i][1] * g[1][b] + G[i][2] * g[2][b];
    for (int i = 0; i < 4; ++i)
        for (int k = 0; k < 4; ++k) {
            const float u = t[i][0] * G[k][0] + t[i][1] * G[k][1] + t[i][2] * G[k][2];
            U[(((long)(i * 4 + k) * (C / 4) + ci / 4) * C + co) * 4 + (ci & 3)] = u;
        }
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// VARIANT bits: 1 = XOR the plane quad with the tile row's parity (LDS bank spread between the two tile rows of a 16-lane group);
// TIMING-ONLY (wrong results, reported as such): 2 = no U loads (one quad reused), 4 = no output stores, 8 = no patch load
template <int VARIANT>
__global__ __launch_bounds__(256, 2) void wino_fused_k(const float* __restrict__ x, const float* __restrict__ U, const float* __restrict__ bias,
                                                       float* __restrict__ y, int N, int H, int W) {
    __shared__ __attribute__((aligned(16))) float P[PH * PW * PLD];     // 48 960 bytes
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int nh = wave & 1, ph = wave >> 1;
    const int bxn = W / 16, byn = H / 8;
    int b = blockIdx.x;
    const int bx = b % bxn; b /= bxn;
    const int by = b % byn;
    const int n = b / byn;
    const int y0 = by * 8, x0 = bx * 16;

    // ---- patch -> LDS (zero outside the image)
    for (int idx = tid; idx < ((VARIANT & 8) ? 0 : PH * PW * (C / 4)); idx += 256) {
        const int q = idx & 15, pix = idx >> 4;
        const int pr = pix / PW, pc = pix - pr * PW;
        const int iy = y0 - 1 + pr, ix = x0 - 1 + pc;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = ld4(x + (((long)n * H + iy) * W + ix) * C + q * 4);
        const int qs = (VARIANT & 1) ? (q ^ ((pr >> 1) & 1)) : q;
        *reinterpret_cast<float4*>(P + pix * PLD + qs * 4) = v;
    }
    __syncthreads();

    const int tyl = j >> 3, txl = j & 7;
    const float* pt = P + ((2 * tyl) * PW + 2 * txl) * PLD + h * 4;     // d[0][0] of this lane's tile, plane quad h (+ 2 s per step)
    const float* ub = U + ((long)h * C + nh * 32 + j) * 4;              // + (pos * 16 + 2 s) * C * 4

    float4 u_const = ld4(ub);
    f32x16 Y[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int bb = 0; bb < 2; ++bb)
#pragma unroll
            for (int r = 0; r < 16; ++r) Y[a][bb][r] = 0.f;

    for (int xl = 0; xl < 2; ++xl) {
        const int xi = 2 * ph + xl;
        // B^T rows: xi 0: d0 - d2, 1: d1 + d2, 2: d2 - d1, 3: d1 - d3  ->  first row with +1, second with s2
        const int i1 = xi == 0 ? 0 : (xi == 2 ? 2 : 1), i2 = xi == 0 ? 2 : (xi == 1 ? 2 : (xi == 2 ? 1 : 3));
        const float s2 = xi == 1 ? 1.f : -1.f;
        const float ca0 = xi == 3 ? 0.f : 1.f, ca1 = xi == 0 ? 0.f : (xi == 1 ? 1.f : -1.f);      // A^T column xi
        const float* r1 = pt + i1 * PW * PLD;
        const float* r2 = pt + i2 * PW * PLD;
#pragma unroll
        for (int nu = 0; nu < 4; ++nu) {
            constexpr int J1[4] = {0, 1, 2, 1}, J2[4] = {2, 2, 1, 3};
            constexpr float T2[4] = {-1.f, 1.f, -1.f, -1.f};
            const int j1 = J1[nu], j2 = J2[nu];
            const float t2 = T2[nu];
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            const float* up = ub + (long)((xi * 4 + nu) * 16) * C * 4;
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                int qo = 2 * s * 4;
                const float4 ua = (VARIANT & 2) ? u_const : ld4(up + (long)(2 * s) * C * 4);
                float4 da, db, dc, dd;
                if (VARIANT & 1) {
                    // quad index XORed with the parity of the patch row pair: rows 2 tyl + i -> ((2 tyl + i) >> 1) & 1
                    const int p1 = ((2 * tyl + i1) >> 1) & 1, p2 = ((2 * tyl + i2) >> 1) & 1;
                    const int q1 = ((2 * s + h) ^ p1) * 4 - h * 4, q2 = ((2 * s + h) ^ p2) * 4 - h * 4;
                    da = ld4(r1 + j1 * PLD + q1); db = ld4(r1 + j2 * PLD + q1);
                    dc = ld4(r2 + j1 * PLD + q2); dd = ld4(r2 + j2 * PLD + q2);
                } else {
                    da = ld4(r1 + j1 * PLD + qo); db = ld4(r1 + j2 * PLD + qo);
                    dc = ld4(r2 + j1 * PLD + qo); dd = ld4(r2 + j2 * PLD + qo);
                }
                float4 v;
                v.x = fmaf(s2, fmaf(t2, dd.x, dc.x), fmaf(t2, db.x, da.x));
                v.y = fmaf(s2, fmaf(t2, dd.y, dc.y), fmaf(t2, db.y, da.y));
                v.z = fmaf(s2, fmaf(t2, dd.z, dc.z), fmaf(t2, db.z, da.z));
                v.w = fmaf(s2, fmaf(t2, dd.w, dc.w), fmaf(t2, db.w, da.w));
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v.x, ua.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v.y, ua.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v.z, ua.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v.w, ua.w, acc, 0, 0, 0);
            }
            // A^T M A folded in: column nu of A: nu 0 -> b 0; 1 -> b 0, 1; 2 -> b 0 (+), b 1 (-); 3 -> b 1 (-)
            constexpr float CB0[4] = {1.f, 1.f, 1.f, 0.f}, CB1[4] = {0.f, 1.f, -1.f, -1.f};
            if (CB0[nu] != 0.f) {
#pragma unroll
                for (int r = 0; r < 16; ++r) { Y[0][0][r] = fmaf(ca0 * CB0[nu], acc[r], Y[0][0][r]); Y[1][0][r] = fmaf(ca1 * CB0[nu], acc[r], Y[1][0][r]); }
            }
            if (CB1[nu] != 0.f) {
#pragma unroll
                for (int r = 0; r < 16; ++r) { Y[0][1][r] = fmaf(ca0 * CB1[nu], acc[r], Y[0][1][r]); Y[1][1][r] = fmaf(ca1 * CB1[nu], acc[r], Y[1][1][r]); }
            }
        }
    }

    // ---- the two xi halves meet in LDS (the patch is dead), the lower half stores
    __syncthreads();
    float* red = P + nh * (4 * 16 * 64);
    if (ph == 1) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int bb = 0; bb < 2; ++bb)
#pragma unroll
                for (int r = 0; r < 16; ++r) red[((a * 2 + bb) * 16 + r) * 64 + lane] = Y[a][bb][r];
    }
    __syncthreads();
    if (VARIANT & 4) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) t += Y[0][0][r] + Y[0][1][r] + Y[1][0][r] + Y[1][1][r];
        if (t == 12345.678f) y[tid] = t;
    } else if (ph == 0) {
        const float bco = bias ? bias[nh * 32 + j] : 0.f;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int bb = 0; bb < 2; ++bb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int t = (r & 3) + 8 * (r >> 2) + 4 * h;
                    const int oy = y0 + 2 * (t >> 3) + a, ox = x0 + 2 * (t & 7) + bb;
                    y[(((long)n * H + oy) * W + ox) * C + nh * 32 + j] = (Y[a][bb][r] + red[((a * 2 + bb) * 16 + r) * 64 + lane]) + bco;
                }
    }
}

// ---- version 2: position row xi at compile time (a wave-uniform branch on the wave's half picks rows {0,1} or {2,3}), so the +-1 / 0
// coefficients fold into adds and skipped terms; the four positions of a row are combined first (T_b = sum_nu A[nu][b] M[xi][nu]: 4 adds
// per value) and then added into the output accumulators (1-2 adds per value; 7-10 VALU per value and row before); both halves store.
// VARIANT bits as above (2 / 4 / 8 timing-only).
template <int XI, int VARIANT, int UNR>
__device__ __forceinline__ void wino_row(const float* pt, const float* ub, const float4& u_const, f32x16 (&Y)[2][2]) {
    constexpr int I1 = XI == 0 ? 0 : (XI == 2 ? 2 : 1), I2 = XI == 0 ? 2 : (XI == 1 ? 2 : (XI == 2 ? 1 : 3));
    constexpr bool PLUS = XI == 1;            // second row added (xi = 1) or subtracted
    const float* r1 = pt + I1 * PW * PLD;
    const float* r2 = pt + I2 * PW * PLD;
    f32x16 acc[4];
#pragma unroll
    for (int nu = 0; nu < 4; ++nu)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nu][r] = 0.f;
#pragma unroll UNR
    for (int s = 0; s < 8; ++s) {
        float4 d1[4], d2[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) { d1[c] = ld4(r1 + c * PLD + 8 * s); d2[c] = ld4(r2 + c * PLD + 8 * s); }
        float4 e[4];                           // the row transform, shared by the four columns
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            e[c].x = PLUS ? d1[c].x + d2[c].x : d1[c].x - d2[c].x; e[c].y = PLUS ? d1[c].y + d2[c].y : d1[c].y - d2[c].y;
            e[c].z = PLUS ? d1[c].z + d2[c].z : d1[c].z - d2[c].z; e[c].w = PLUS ? d1[c].w + d2[c].w : d1[c].w - d2[c].w;
        }
        // all four positions' A values first, then the MFMAs rotate over the four accumulators (a VALU instruction between two MFMAs on the
        // SAME accumulator costs ~43 cycles, between different ones ~6: MI355X_MICROARCH.md)
        float4 v[4], ua[4];
#pragma unroll
        for (int nu = 0; nu < 4; ++nu) {
            constexpr int J1[4] = {0, 1, 2, 1}, J2[4] = {2, 2, 1, 3};
            ua[nu] = (VARIANT & 2) ? u_const : ld4(ub + (long)((XI * 4 + nu) * 16 + 2 * s) * C * 4);
            const float4 a = e[J1[nu]], b = e[J2[nu]];
            if (nu == 1) { v[nu].x = a.x + b.x; v[nu].y = a.y + b.y; v[nu].z = a.z + b.z; v[nu].w = a.w + b.w; }
            else { v[nu].x = a.x - b.x; v[nu].y = a.y - b.y; v[nu].z = a.z - b.z; v[nu].w = a.w - b.w; }
        }
#pragma unroll
        for (int nu = 0; nu < 4; ++nu) acc[nu] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[nu].x, ua[nu].x, acc[nu], 0, 0, 0);
#pragma unroll
        for (int nu = 0; nu < 4; ++nu) acc[nu] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[nu].y, ua[nu].y, acc[nu], 0, 0, 0);
#pragma unroll
        for (int nu = 0; nu < 4; ++nu) acc[nu] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[nu].z, ua[nu].z, acc[nu], 0, 0, 0);
#pragma unroll
        for (int nu = 0; nu < 4; ++nu) acc[nu] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[nu].w, ua[nu].w, acc[nu], 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float t0 = (acc[0][r] + acc[1][r]) + acc[2][r], t1 = (acc[1][r] - acc[2][r]) - acc[3][r];
        if (XI != 3) { Y[0][0][r] += t0; Y[0][1][r] += t1; }
        if (XI == 1) { Y[1][0][r] += t0; Y[1][1][r] += t1; }
        if (XI >= 2) { Y[1][0][r] -= t0; Y[1][1][r] -= t1; }
    }
}

template <int VARIANT, int UNR>
__global__ __launch_bounds__(256, 2) void wino_fused2_k(const float* __restrict__ x, const float* __restrict__ U, const float* __restrict__ bias,
                                                        float* __restrict__ y, int N, int H, int W) {
    __shared__ __attribute__((aligned(16))) float P[PH * PW * PLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int nh = wave & 1, ph = wave >> 1;
    const int bxn = W / 16, byn = H / 8;
    int b = blockIdx.x;
    const int bx = b % bxn; b /= bxn;
    const int by = b % byn;
    const int n = b / byn;
    const int y0 = by * 8, x0 = bx * 16;
    for (int idx = tid; idx < ((VARIANT & 8) ? 0 : PH * PW * (C / 4)); idx += 256) {
        const int q = idx & 15, pix = idx >> 4;
        const int pr = pix / PW, pc = pix - pr * PW;
        const int iy = y0 - 1 + pr, ix = x0 - 1 + pc;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = ld4(x + (((long)n * H + iy) * W + ix) * C + q * 4);
        *reinterpret_cast<float4*>(P + pix * PLD + q * 4) = v;
    }
    __syncthreads();
    const int tyl = j >> 3, txl = j & 7;
    const float* pt = P + ((2 * tyl) * PW + 2 * txl) * PLD + h * 4;
    const float* ub = U + ((long)h * C + nh * 32 + j) * 4;
    const float4 u_const = ld4(ub);
    f32x16 Y[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int bb = 0; bb < 2; ++bb)
#pragma unroll
            for (int r = 0; r < 16; ++r) Y[a][bb][r] = 0.f;
    if (ph == 0) { wino_row<0, VARIANT, UNR>(pt, ub, u_const, Y); wino_row<1, VARIANT, UNR>(pt, ub, u_const, Y); }
    else { wino_row<2, VARIANT, UNR>(pt, ub, u_const, Y); wino_row<3, VARIANT, UNR>(pt, ub, u_const, Y); }

    // the halves swap: half 0 finishes and stores output row a = 0 of every tile, half 1 row a = 1
    __syncthreads();
    float* red = P + (ph * 2 + nh) * (2 * 16 * 64);
    {
        // the row the OTHER half finishes (a select, not an index: a run-time index would put Y into scratch memory)
#pragma unroll
        for (int bb = 0; bb < 2; ++bb)
#pragma unroll
            for (int r = 0; r < 16; ++r) red[(bb * 16 + r) * 64 + lane] = ph == 0 ? Y[1][bb][r] : Y[0][bb][r];
    }
    __syncthreads();
    if (VARIANT & 4) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) t += Y[0][0][r] + Y[0][1][r] + Y[1][0][r] + Y[1][1][r];
        if (t == 12345.678f) y[tid] = t;
        return;
    }
    const float* oth = P + ((1 - ph) * 2 + nh) * (2 * 16 * 64);
    const float bco = bias ? bias[nh * 32 + j] : 0.f;
#pragma unroll
    for (int bb = 0; bb < 2; ++bb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int t = (r & 3) + 8 * (r >> 2) + 4 * h;
            const int oy = y0 + 2 * (t >> 3) + ph, ox = x0 + 2 * (t & 7) + bb;
            const float v0 = ph == 0 ? Y[0][bb][r] : oth[(bb * 16 + r) * 64 + lane];     // xi rows 0,1 first, then 2,3: a fixed order
            const float v1 = ph == 0 ? oth[(bb * 16 + r) * 64 + lane] : Y[1][bb][r];
            y[(((long)n * H + oy) * W + ox) * C + nh * 32 + j] = (v0 + v1) + bco;
        }
}

// ---- version 3: wave = ONE position row xi x all 64 output planes (two 32-column MFMA tiles share every A value: 1.5 instead of 3
// transform VALU per MFMA), positions in two pairs (4 accumulators), the row's two output-transform sums T_b kept in 64 registers, the
// four rows meet in LDS (the dead patch) and every wave stores a quarter of the outputs; U through a buffer descriptor with SGPR offsets
// (no address VALU); lanes -> tiles permuted so that every 16-lane group of ds_read_b128 covers two whole tile rows, whose plane quads are
// XORed by the row-pair parity: 16 distinct 16-byte slots per group.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 bufld4(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    const f32x4 f = __builtin_bit_cast(f32x4, v);        // cast the WHOLE vector (see gemm.hip)
    return make_float4(f.x, f.y, f.z, f.w);
}
// row i of an MFMA tile (= lane & 31 of the A operand) -> output tile (ty, tx) of the 4 x 8 block
__device__ __forceinline__ int tile_of_row(int l) {
    const bool ga = l < 4 || (l >= 12 && l < 16) || (l >= 20 && l < 28);
    const int rank = ga ? (l < 4 ? l : (l < 16 ? l - 8 : l - 12)) : (l < 12 ? l - 4 : (l < 20 ? l - 8 : l - 16));
    return (ga ? 0 : 16) + rank;          // ty = result >> 3, tx = result & 7
}

template <int XI, int VARIANT>
__device__ __forceinline__ void wino_row3(const float* P, int tyl, int txl, int h, __amdgpu_buffer_rsrc_t rsU, unsigned uvoff, const float4& u_const,
                                          f32x16 (&T)[2][2]) {
    constexpr int I1 = XI == 0 ? 0 : (XI == 2 ? 2 : 1), I2 = XI == 0 ? 2 : (XI == 1 ? 2 : (XI == 2 ? 1 : 3));
    constexpr bool PLUS = XI == 1;
    const float* pt = P + ((2 * tyl) * PW + 2 * txl) * PLD;
    const float* r1 = pt + I1 * PW * PLD + ((h ^ ((tyl + (I1 >> 1)) & 1)) << 2);
    const float* r2 = pt + I2 * PW * PLD + ((h ^ ((tyl + (I2 >> 1)) & 1)) << 2);
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
        f32x16 acc[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
#pragma unroll 2
        for (int s = 0; s < 8; ++s) {
            float4 e[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float4 d1 = ld4(r1 + c * PLD + 8 * s), d2 = ld4(r2 + c * PLD + 8 * s);
                e[c].x = PLUS ? d1.x + d2.x : d1.x - d2.x; e[c].y = PLUS ? d1.y + d2.y : d1.y - d2.y;
                e[c].z = PLUS ? d1.z + d2.z : d1.z - d2.z; e[c].w = PLUS ? d1.w + d2.w : d1.w - d2.w;
            }
            float4 v[2], ua[2][2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                constexpr int J1[4] = {0, 1, 2, 1}, J2[4] = {2, 2, 1, 3};
                const int nu = 2 * pr + k;
                const float4 a = e[J1[nu]], b = e[J2[nu]];
                if (nu == 1) { v[k].x = a.x + b.x; v[k].y = a.y + b.y; v[k].z = a.z + b.z; v[k].w = a.w + b.w; }
                else { v[k].x = a.x - b.x; v[k].y = a.y - b.y; v[k].z = a.z - b.z; v[k].w = a.w - b.w; }
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    ua[k][nt] = (VARIANT & 2) ? u_const : bufld4(rsU, uvoff, (((XI * 4 + nu) * 16 + 2 * s) * C + nt * 32) * 16);
            }
#define MF(comp)                                                                                                          \
    _Pragma("unroll") for (int k = 0; k < 2; ++k) _Pragma("unroll") for (int nt = 0; nt < 2; ++nt)                          \
        acc[k][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[k].comp, ua[k][nt].comp, acc[k][nt], 0, 0, 0);
            MF(x) MF(y) MF(z) MF(w)
#undef MF
        }
        // T_0 = M_0 + M_1 + M_2, T_1 = M_1 - M_2 - M_3 (columns of A)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (pr == 0) { T[0][nt][r] = acc[0][nt][r] + acc[1][nt][r]; T[1][nt][r] = acc[1][nt][r]; }
                else { T[0][nt][r] += acc[0][nt][r]; T[1][nt][r] = (T[1][nt][r] - acc[0][nt][r]) - acc[1][nt][r]; }
            }
    }
}

template <int VARIANT>
__global__ __launch_bounds__(256, 2) void wino_fused3_k(const float* __restrict__ x, const float* __restrict__ U, const float* __restrict__ bias,
                                                        float* __restrict__ y, int N, int H, int W) {
    __shared__ __attribute__((aligned(16))) float P[PH * PW * PLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int bxn = W / 16, byn = H / 8;
    int b = blockIdx.x;
    const int bx = b % bxn; b /= bxn;
    const int by = b % byn;
    const int n = b / byn;
    const int y0 = by * 8, x0 = bx * 16;
    for (int idx = tid; idx < ((VARIANT & 8) ? 0 : PH * PW * (C / 4)); idx += 256) {
        const int q = idx & 15, pix = idx >> 4;
        const int pr = pix / PW, pc = pix - pr * PW;
        const int iy = y0 - 1 + pr, ix = x0 - 1 + pc;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = ld4(x + (((long)n * H + iy) * W + ix) * C + q * 4);
        *reinterpret_cast<float4*>(P + pix * PLD + ((q ^ ((pr >> 1) & 1)) << 2)) = v;
    }
    __syncthreads();
    const int tl = tile_of_row(j), tyl = tl >> 3, txl = tl & 7;
    const __amdgpu_buffer_rsrc_t rsU = __builtin_amdgcn_make_buffer_rsrc((void*)U, 0, 16 * C * C * 4, 0x00020000);
    const unsigned uvoff = (unsigned)(h * C + j) * 16u;
    const float4 u_const = ld4(U + (h * C + j) * 4);
    f32x16 T[2][2];       // [b][column tile]
    if (wave == 0) wino_row3<0, VARIANT>(P, tyl, txl, h, rsU, uvoff, u_const, T);
    else if (wave == 1) wino_row3<1, VARIANT>(P, tyl, txl, h, rsU, uvoff, u_const, T);
    else if (wave == 2) wino_row3<2, VARIANT>(P, tyl, txl, h, rsU, uvoff, u_const, T);
    else wino_row3<3, VARIANT>(P, tyl, txl, h, rsU, uvoff, u_const, T);

    // Y[0][b] = T0 + T1 + T2, Y[1][b] = T1 - T2 - T3 (subscript = the wave's row).  Wave w finishes (a = w >> 1, column tile w & 1): it
    // keeps its own sums of that tile and reads the others' from LDS.  R1 / R2 = all of rows 1 / 2, R0 / R3 = the tile rows 0 / 3 give away.
    __syncthreads();
    float* R1 = P; float* R2 = P + 4096; float* R0 = P + 8192; float* R3 = P + 8192 + 2048;      // 16 + 16 + 8 + 8 KB
#pragma unroll
    for (int bb = 0; bb < 2; ++bb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (wave == 1) { R1[((bb * 2 + 0) * 16 + r) * 64 + lane] = T[bb][0][r]; R1[((bb * 2 + 1) * 16 + r) * 64 + lane] = T[bb][1][r]; }
            if (wave == 2) { R2[((bb * 2 + 0) * 16 + r) * 64 + lane] = T[bb][0][r]; R2[((bb * 2 + 1) * 16 + r) * 64 + lane] = T[bb][1][r]; }
            if (wave == 0) R0[(bb * 16 + r) * 64 + lane] = T[bb][1][r];
            if (wave == 3) R3[(bb * 16 + r) * 64 + lane] = T[bb][0][r];
        }
    __syncthreads();
    if (VARIANT & 4) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) t += T[0][0][r] + T[0][1][r] + T[1][0][r] + T[1][1][r];
        if (t == 12345.678f) y[tid] = t;
        return;
    }
    const int a = wave >> 1, nt = wave & 1;
    const float bco = bias ? bias[nt * 32 + j] : 0.f;
    float fin[2][16];
#pragma unroll
    for (int bb = 0; bb < 2; ++bb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int k1 = ((bb * 2 + nt) * 16 + r) * 64 + lane, k0 = (bb * 16 + r) * 64 + lane;
            const float t1 = wave == 1 ? T[bb][1][r] : R1[k1];                 // wave 1 finishes tile 1, wave 2 tile 0: their own registers
            const float t2 = wave == 2 ? T[bb][0][r] : R2[k1];
            float v;
            if (a == 0) v = ((wave == 0 ? T[bb][0][r] : R0[k0]) + t1) + t2;
            else v = (t1 - t2) - (wave == 3 ? T[bb][1][r] : R3[k0]);
            fin[bb][r] = v + bco;
        }
    if (VARIANT & 16) {
        // wide stores: the block's 8 x 16 x 64 outputs pass through LDS once more ([pixel][plane]) and leave as 16-byte stores, a wave
        // covering four pixels = 1 KB of one output row (32 four-byte stores per lane, two 128-byte segments each, before)
        __syncthreads();
#pragma unroll
        for (int bb = 0; bb < 2; ++bb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int t = tile_of_row((r & 3) + 8 * (r >> 2) + 4 * h);
                P[((2 * (t >> 3) + a) * 16 + 2 * (t & 7) + bb) * C + nt * 32 + j] = fin[bb][r];
            }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int pix = it * 16 + (tid >> 4), q = tid & 15;
            const float4 v = *reinterpret_cast<const float4*>(P + pix * C + q * 4);
            *reinterpret_cast<float4*>(y + (((long)n * H + y0 + (pix >> 4)) * W + x0 + (pix & 15)) * C + q * 4) = v;
        }
        return;
    }
#pragma unroll
    for (int bb = 0; bb < 2; ++bb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int t = tile_of_row((r & 3) + 8 * (r >> 2) + 4 * h);
            const int oy = y0 + 2 * (t >> 3) + a, ox = x0 + 2 * (t & 7) + bb;
            y[(((long)n * H + oy) * W + ox) * C + nt * 32 + j] = fin[bb][r];
        }
}

// ---- version 4: EIGHT waves per workgroup = (position row xi) x (32-column tile): half the accumulators per wave (T 32 + acc 32
// registers), so four waves per SIMD fit and two workgroups per CU give every SIMD four waves to draw MFMAs from; the A values are
// formed twice (once per column tile).  Lane permutation, swizzle and buffer loads as version 3.
template <int XI, int VARIANT>
__device__ __forceinline__ void wino_row4(const float* P, int tyl, int txl, int h, __amdgpu_buffer_rsrc_t rsU, unsigned uvoff, const float4& u_const,
                                          f32x16 (&T)[2]) {
    constexpr int I1 = XI == 0 ? 0 : (XI == 2 ? 2 : 1), I2 = XI == 0 ? 2 : (XI == 1 ? 2 : (XI == 2 ? 1 : 3));
    constexpr bool PLUS = XI == 1;
    const float* pt = P + ((2 * tyl) * PW + 2 * txl) * PLD;
    const float* r1 = pt + I1 * PW * PLD + ((h ^ ((tyl + (I1 >> 1)) & 1)) << 2);
    const float* r2 = pt + I2 * PW * PLD + ((h ^ ((tyl + (I2 >> 1)) & 1)) << 2);
    f32x16 acc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
#pragma unroll 2
    for (int s = 0; s < 8; ++s) {
        float4 e[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float4 d1 = ld4(r1 + c * PLD + 8 * s), d2 = ld4(r2 + c * PLD + 8 * s);
            e[c].x = PLUS ? d1.x + d2.x : d1.x - d2.x; e[c].y = PLUS ? d1.y + d2.y : d1.y - d2.y;
            e[c].z = PLUS ? d1.z + d2.z : d1.z - d2.z; e[c].w = PLUS ? d1.w + d2.w : d1.w - d2.w;
        }
        float4 v[4], ua[4];
#pragma unroll
        for (int nu = 0; nu < 4; ++nu) {
            constexpr int J1[4] = {0, 1, 2, 1}, J2[4] = {2, 2, 1, 3};
            const float4 a = e[J1[nu]], b = e[J2[nu]];
            if (VARIANT & 64) v[nu] = u_const;
            else if (VARIANT & 32) v[nu] = ld4(r1 + nu * PLD + 8 * s);
            else if (nu == 1) { v[nu].x = a.x + b.x; v[nu].y = a.y + b.y; v[nu].z = a.z + b.z; v[nu].w = a.w + b.w; }
            else { v[nu].x = a.x - b.x; v[nu].y = a.y - b.y; v[nu].z = a.z - b.z; v[nu].w = a.w - b.w; }
            ua[nu] = (VARIANT & 2) ? u_const : bufld4(rsU, uvoff, (((XI * 4 + nu) * 16 + 2 * s) * C) * 16);
        }
#define MF4(comp) _Pragma("unroll") for (int nu = 0; nu < 4; ++nu) acc[nu] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[nu].comp, ua[nu].comp, acc[nu], 0, 0, 0);
        MF4(x) MF4(y) MF4(z) MF4(w)
#undef MF4
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) { T[0][r] = (acc[0][r] + acc[1][r]) + acc[2][r]; T[1][r] = (acc[1][r] - acc[2][r]) - acc[3][r]; }
}

template <int VARIANT>
__global__ __launch_bounds__(512, 2) void wino_fused4_k(const float* __restrict__ x, const float* __restrict__ U, const float* __restrict__ bias,
                                                        float* __restrict__ y, int N, int H, int W) {
    __shared__ __attribute__((aligned(16))) float P[PH * PW * PLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int xi = wave & 3, nt = wave >> 2;
    const int bxn = W / 16, byn = H / 8;
    int b = blockIdx.x;
    const int bx = b % bxn; b /= bxn;
    const int by = b % byn;
    const int n = b / byn;
    const int y0 = by * 8, x0 = bx * 16;
    for (int idx = tid; idx < ((VARIANT & 8) ? 0 : PH * PW * (C / 4)); idx += 512) {
        const int q = idx & 15, pix = idx >> 4;
        const int pr = pix / PW, pc = pix - pr * PW;
        const int iy = y0 - 1 + pr, ix = x0 - 1 + pc;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = ld4(x + (((long)n * H + iy) * W + ix) * C + q * 4);
        *reinterpret_cast<float4*>(P + pix * PLD + ((q ^ ((pr >> 1) & 1)) << 2)) = v;
    }
    __syncthreads();
    const int tl = tile_of_row(j), tyl = tl >> 3, txl = tl & 7;
    const __amdgpu_buffer_rsrc_t rsU = __builtin_amdgcn_make_buffer_rsrc((void*)U, 0, 16 * C * C * 4, 0x00020000);
    const unsigned uvoff = (unsigned)(h * C + nt * 32 + j) * 16u;
    const float4 u_const = ld4(U + (h * C + j) * 4);
    f32x16 T[2];
    if (xi == 0) wino_row4<0, VARIANT>(P, tyl, txl, h, rsU, uvoff, u_const, T);
    else if (xi == 1) wino_row4<1, VARIANT>(P, tyl, txl, h, rsU, uvoff, u_const, T);
    else if (xi == 2) wino_row4<2, VARIANT>(P, tyl, txl, h, rsU, uvoff, u_const, T);
    else wino_row4<3, VARIANT>(P, tyl, txl, h, rsU, uvoff, u_const, T);

    // wave (xi, nt) finishes output (a = xi >> 1, b = xi & 1) of its column tile: Y[0][b] = T0[b] + T1[b] + T2[b], Y[1][b] = T1[b] - T2[b] - T3[b].
    // Published per column tile (1024 floats each): slot 0 T1[0], 1 T2[0], 2 T0[1], 3 T2[1], 4 T3[0], 5 T1[1]
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
    if (VARIANT & 4) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) t += T[0][r] + T[1][r];
        if (t == 12345.678f) y[tid] = t;
        return;
    }
    const int a = xi >> 1, bb = xi & 1;
    const float bco = bias ? bias[nt * 32 + j] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int k = r * 64 + lane;
        float v;
        if (xi == 0) v = (T[0][r] + X[0 * 1024 + k]) + X[1 * 1024 + k];            // T0[0] + T1[0] + T2[0]
        else if (xi == 1) v = (X[2 * 1024 + k] + T[1][r]) + X[3 * 1024 + k];       // T0[1] + T1[1] + T2[1]
        else if (xi == 2) v = (X[0 * 1024 + k] - T[0][r]) - X[4 * 1024 + k];       // T1[0] - T2[0] - T3[0]
        else v = (X[5 * 1024 + k] - X[3 * 1024 + k]) - T[1][r];                    // T1[1] - T2[1] - T3[1]
        const int t = tile_of_row((r & 3) + 8 * (r >> 2) + 4 * h);
        const int oy = y0 + 2 * (t >> 3) + a, ox = x0 + 2 * (t & 7) + bb;
        y[(((long)n * H + oy) * W + ox) * C + nt * 32 + j] = v + bco;
    }
}

static const char* g_only = nullptr;   // argv[4]: run only the variants whose name contains this
static double ref_at(const std::vector<float>& x, const std::vector<float>& w, const std::vector<float>& bias, int H, int W, int n, int oy, int ox, int co) {
    double s = bias[co];
    for (int ty = 0; ty < 3; ++ty)
        for (int tx = 0; tx < 3; ++tx) {
            const int iy = oy + ty - 1, ix = ox + tx - 1;
            if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
            for (int ci = 0; ci < C; ++ci) s += (double)x[(((long)n * H + iy) * W + ix) * C + ci] * (double)w[((long)co * C + ci) * 9 + ty * 3 + tx];
        }
    return s;
}

template <int V, int KV = 1>
static void run(const char* name, const float* dx, const float* dU, const float* db, float* dy, int N, int H, int W, int iters,
                const std::vector<float>& x, const std::vector<float>& w, const std::vector<float>& bias) {
    if (g_only && !strstr(name, g_only)) return;
    const dim3 grid((unsigned)(N * (H / 8) * (W / 16)));
    CK(hipMemset(dy, 0xff, (size_t)N * H * W * C * 4));
    if (KV == 1) hipLaunchKernelGGL((wino_fused_k<V>), grid, dim3(256), 0, 0, dx, dU, db, dy, N, H, W); else if (KV == 5) hipLaunchKernelGGL((wino_fused3_k<V>), grid, dim3(256), 0, 0, dx, dU, db, dy, N, H, W); else if (KV == 6) hipLaunchKernelGGL((wino_fused4_k<V>), grid, dim3(512), 0, 0, dx, dU, db, dy, N, H, W); else hipLaunchKernelGGL((wino_fused2_k<V, (KV == 3 ? 1 : (KV > 1 ? KV : 2))>), grid, dim3(256), 0, 0, dx, dU, db, dy, N, H, W);
    CK(hipDeviceSynchronize());
    std::vector<float> y((size_t)N * H * W * C);
    CK(hipMemcpy(y.data(), dy, y.size() * 4, hipMemcpyDeviceToHost));
    double emax = 0.0, rmax = 0.0;
    unsigned long long seed = 12345;
    for (int k = 0; k < 4000; ++k) {
        seed = seed * 6364136223846793005ull + 1442695040888963407ull;
        const int n = (int)((seed >> 33) % N), oy = k < 64 ? (k & 1 ? H - 1 : 0) : (int)((seed >> 20) % H), ox = k < 128 ? ((k >> 1) & 1 ? W - 1 : 0) : (int)((seed >> 10) % W);
        const int co = (int)((seed >> 3) % C);
        const double r = ref_at(x, w, bias, H, W, n, oy, ox, co);
        const double d = fabs((double)y[(((long)n * H + oy) * W + ox) * C + co] - r);
        if (d > emax) emax = d;
        if (fabs(r) > rmax) rmax = fabs(r);
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) if (KV == 1) hipLaunchKernelGGL((wino_fused_k<V>), grid, dim3(256), 0, 0, dx, dU, db, dy, N, H, W); else if (KV == 5) hipLaunchKernelGGL((wino_fused3_k<V>), grid, dim3(256), 0, 0, dx, dU, db, dy, N, H, W); else if (KV == 6) hipLaunchKernelGGL((wino_fused4_k<V>), grid, dim3(512), 0, 0, dx, dU, db, dy, N, H, W); else hipLaunchKernelGGL((wino_fused2_k<V, (KV == 3 ? 1 : (KV > 1 ? KV : 2))>), grid, dim3(256), 0, 0, dx, dU, db, dy, N, H, W);
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) if (KV == 1) hipLaunchKernelGGL((wino_fused_k<V>), grid, dim3(256), 0, 0, dx, dU, db, dy, N, H, W); else if (KV == 5) hipLaunchKernelGGL((wino_fused3_k<V>), grid, dim3(256), 0, 0, dx, dU, db, dy, N, H, W); else if (KV == 6) hipLaunchKernelGGL((wino_fused4_k<V>), grid, dim3(512), 0, 0, dx, dU, db, dy, N, H, W); else hipLaunchKernelGGL((wino_fused2_k<V, (KV == 3 ? 1 : (KV > 1 ? KV : 2))>), grid, dim3(256), 0, 0, dx, dU, db, dy, N, H, W);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = 1e3 * ms / iters, direct = 2.0 * N * H * W * C * C * 9;
    printf("%-44s %8.1f us  executed %6.1f TF (%.3f of 157.3)  direct-count %6.1f TF   max|d| vs fp64 %.2e (max|ref| %.2f)\n", name, us,
           direct * 16 / 36 / us / 1e6, direct * 16 / 36 / us / 1e6 / 157.3, direct / us / 1e6, emax, rmax);
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 128, H = argc > 2 ? atoi(argv[2]) : 32, iters = argc > 3 ? atoi(argv[3]) : 20;
    const int W = H;
    if (argc > 4) g_only = argv[4];
    if (H % 8 || W % 16) { printf("H %% 8 == 0 and W %% 16 == 0\n"); return 1; }
    std::vector<float> x((size_t)N * H * W * C), w((size_t)C * C * 9), bias(C);
    unsigned long long s = 1;
    auto rnd = [&]() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (float)((s >> 40) & 0xffff) / 65536.f - 0.5f; };
    for (auto& v : x) v = rnd();
    for (auto& v : w) v = rnd() * 0.1f;
    for (auto& v : bias) v = rnd();
    float *dx, *dw, *dU, *db, *dy;
    CK(hipMalloc(&dx, x.size() * 4)); CK(hipMalloc(&dw, w.size() * 4)); CK(hipMalloc(&dU, 16 * C * C * 4)); CK(hipMalloc(&db, C * 4));
    CK(hipMalloc(&dy, x.size() * 4));
    CK(hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dw, w.data(), w.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, bias.data(), C * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(pack_u, dim3(C * C / 256), dim3(256), 0, 0, dw, dU, 0);
    CK(hipDeviceSynchronize());
    printf("fused F(2x2,3x3), 64 -> 64 planes, %d x %d x %d: %d workgroups of 32 tiles\n", N, H, W, N * (H / 8) * (W / 16));
    run<0>("patch in LDS, U from L2, linear quads", dx, dU, db, dy, N, H, W, iters, x, w, bias);
    run<1>("... quads XORed per tile-row parity", dx, dU, db, dy, N, H, W, iters, x, w, bias);
    run<0, 2>("v2: xi at compile time, T-form update, both halves store; K loop unrolled 2", dx, dU, db, dy, N, H, W, iters, x, w, bias);
    run<0, 4>("v2, K loop unrolled 4", dx, dU, db, dy, N, H, W, iters, x, w, bias);
    run<0, 3>("v2, K loop not unrolled (3 = unroll 1)", dx, dU, db, dy, N, H, W, iters, x, w, bias);
    run<0, 5>("v3: wave = one xi row x 64 planes, permuted lanes, buffer U", dx, dU, db, dy, N, H, W, iters, x, w, bias);
    run<0, 6>("v4: 8 waves = (xi row) x (32-column tile), 4 waves per SIMD", dx, dU, db, dy, N, H, W, iters, x, w, bias);
    run<16, 5>("v3 + wide stores (outputs through LDS, 16-byte stores)", dx, dU, db, dy, N, H, W, iters, x, w, bias);
    printf("timing-only variants (results wrong by construction):\n");
    run<2, 5>("v3, no U loads", dx, dU, db, dy, N, H, W, iters, x, w, bias);
    run<4, 5>("v3, no output stores", dx, dU, db, dy, N, H, W, iters, x, w, bias);
    run<8, 5>("v3, no patch load", dx, dU, db, dy, N, H, W, iters, x, w, bias);
    run<14, 5>("v3, no U loads, no stores, no patch load", dx, dU, db, dy, N, H, W, iters, x, w, bias);
    run<24, 5>("v3 + wide stores, no patch load", dx, dU, db, dy, N, H, W, iters, x, w, bias);
    run<2, 6>("v4, no U loads", dx, dU, db, dy, N, H, W, iters, x, w, bias);
    run<4, 6>("v4, no output stores", dx, dU, db, dy, N, H, W, iters, x, w, bias);
    run<8, 6>("v4, no patch load", dx, dU, db, dy, N, H, W, iters, x, w, bias);
    run<14, 6>("v4, no U loads, no stores, no patch load", dx, dU, db, dy, N, H, W, iters, x, w, bias);
    run<14 + 32, 6>("v4, ... and no transform VALU (A = raw patch value)", dx, dU, db, dy, N, H, W, iters, x, w, bias);
    run<14 + 64, 6>("v4, ... and no LDS reads (A constant): MFMA + T sums only", dx, dU, db, dy, N, H, W, iters, x, w, bias);
    run<2, 2>("v2, no U loads", dx, dU, db, dy, N, H, W, iters, x, w, bias);
    run<4, 2>("v2, no output stores", dx, dU, db, dy, N, H, W, iters, x, w, bias);
    run<8, 2>("v2, no patch load", dx, dU, db, dy, N, H, W, iters, x, w, bias);
    run<14, 2>("v2, no U loads, no stores, no patch load", dx, dU, db, dy, N, H, W, iters, x, w, bias);
    run<2>("no U loads", dx, dU, db, dy, N, H, W, iters, x, w, bias);
    run<4>("no output stores", dx, dU, db, dy, N, H, W, iters, x, w, bias);
    run<8>("no patch load", dx, dU, db, dy, N, H, W, iters, x, w, bias);
    run<14>("no U loads, no stores, no patch load", dx, dU, db, dy, N, H, W, iters, x, w, bias);
    return 0;
}
