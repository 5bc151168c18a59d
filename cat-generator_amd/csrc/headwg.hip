// Weight gradient of nn.View(C*H*W) -> nn.Linear(C*H*W, Co) on an NHWC map, written STRAIGHT into the canonical gradWeight (round 6).
//
// Stands in for accGradParameters of D32_st3's head (models.lua:696-697: View(320*8*8) -> Linear(20480, 256)).  The layer is a
// [Co x N] . [N x C*H*W] product with the batch as its K: 1.3 GFLOP for 10.5 MB of x and 21 MB of gradWeight - memory-bound, 8 us of
// MFMA time.  As a convolution with an H x W kernel (how the planned executor runs the layer: cg_pack_conv_weight_map) the generic path
// wrote the [(tap, c)][co] partial plane (21 MB) and then transposed it into gradWeight[co][c][tap] with a second kernel (21 MB read, 42 MB
// read-modify-write): 32 + 27 us in the step.  Here ONE kernel does gradWeight[co][c][tap] += scale * sum_n dy[n][co] x[n][tap][c]:
//
//   * a workgroup owns 128 output rows (co) x 128 columns = 8 input planes x 16 taps; its K loop runs over the batch in tiles of 16 images,
//     both operands dropped into LDS by buffer_load ... lds (dy rows of 128 floats; x rows gathered as 32 granules of 4 planes, one
//     per (plane quad, tap): which granule a lane fetches is free, so the LDS row is laid out [plane quad][tap][4]),
//   * the MFMA's B columns are read in (plane, tap) order - lane = tap, fragment = plane pair - which the [quad][tap][4] row serves without
//     bank conflicts (16 taps x 4 floats = 64 banks), so that an accumulator register holds 16 consecutive taps of one (co, plane):
//     the read-modify-write of gradWeight is 64-byte runs, no transposition pass, no partial plane,
//   * x is read in 32-byte granules; the two workgroups that share the other half of the 64-byte sector (and the two row tiles) are
//     given block numbers congruent mod 8, i.e. the same XCD and L2,
//   * gradBias (column sums of dy) rides on the workgroups of the first column tile.
#include "common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void glds16(__amdgpu_buffer_rsrc_t r, float* lds, unsigned voff, int soff) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 16, voff, soff, 0, 0);
#endif
}

constexpr int BK = 16;            // images per K tile (32 images per tile, 64 KB of LDS: 42 us against 32)
constexpr int TM = 128, TN = 128; // output rows (co) x columns (8 planes x 16 taps) per workgroup
constexpr int CC = 8, TT = 16;

struct HeadArgs {
    const float* x;    // [N][T][C]  (the NHWC map: T = H*W positions = the taps of the H x W kernel)
    const float* dy;   // [N][Co]
    float* gw;         // [Co][C][T] canonical nn.Linear weight gradient (accumulated into)
    float* gb;         // [Co] or null
    int N, T, C, Co;
    float scale;
};

__global__ __launch_bounds__(256, 2) void head_wgrad_k(HeadArgs a) {
    __shared__ __attribute__((aligned(16))) float As0[BK * TM];
    __shared__ __attribute__((aligned(16))) float As1[BK * TM];
    __shared__ __attribute__((aligned(16))) float Bs0[BK * TN];
    __shared__ __attribute__((aligned(16))) float Bs1[BK * TN];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 64;

    // block -> (row tile, plane octet, tap group): the 2 * nco workgroups that read the same 64-byte sectors of x share an XCD
    const int nco = a.Co / TM, nci = a.C / CC, ntp = a.T / TT;
    const int share = 2 * nco, units = (nci / 2) * ntp;
    int L = blockIdx.x, r, u;
    if ((nci & 1) == 0 && (units & 7) == 0) { const int slot = L >> 3; r = slot % share; u = (slot / share) * 8 + (L & 7); }
    else { r = L % share; u = L / share; }
    int tco, ci_oct, tpg;
    if ((nci & 1) == 0) { tco = r % nco; ci_oct = 2 * (u % (nci / 2)) + r / nco; tpg = u / (nci / 2); }
    else { tco = L % nco; const int v = L / nco; ci_oct = v % nci; tpg = v / nci; }
    const int co0 = tco * TM, ci0 = ci_oct * CC, tap0 = tpg * TT;

    __amdgpu_buffer_rsrc_t rsd = __builtin_amdgcn_make_buffer_rsrc((void*)(a.dy + co0), 0, 0x7fffffff, 0x00020000);
    __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x + (long)tap0 * a.C + ci0), 0, 0x7fffffff, 0x00020000);
    // a wave instruction fills two K rows of 32 granules: dy row = 128 consecutive floats; x row = granule g -> (quad g >> 4, tap g & 15)
    unsigned avoff[BK / 8], bvoff[BK / 8];
#pragma unroll
    for (int q = 0; q < BK / 8; ++q) {
        const int kr = q * 8 + wave * 2 + h;
        avoff[q] = (unsigned)(kr * a.Co + 4 * l31) * 4u;
        bvoff[q] = (unsigned)((kr * a.T + (l31 & 15)) * a.C + 4 * (l31 >> 4)) * 4u;
    }
    auto dma_tile = [&](int t, auto bufc) {
        constexpr int buf = decltype(bufc)::value;
        float* A = buf ? As1 : As0;
        float* B = buf ? Bs1 : Bs0;
        const int soa = t * BK * a.Co * 4, sob = t * BK * a.T * a.C * 4;
#pragma unroll
        for (int q = 0; q < BK / 8; ++q) {
            const int row0 = q * 8 + wave * 2;
            glds16(rsd, A + row0 * TM, avoff[q], soa);
            glds16(rsx, B + row0 * TN, bvoff[q], sob);
        }
    };
    // B fragment j of this wave: column f = wn0 + 32 j + l31 -> plane f >> 4, tap f & 15 -> LDS column (plane >> 2) * 64 + 4 tap + (plane & 3)
    int bcol[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int f = wn0 + 32 * j + l31, pl = f >> 4, tp = f & 15;
        bcol[j] = (pl >> 2) * 64 + 4 * tp + (pl & 3);
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r2 = 0; r2 < 16; ++r2) acc[i][j][r2] = 0.f;
    const bool do_bias = a.gb != nullptr && ci_oct == 0 && tpg == 0;
    float bsum = 0.f;

    const int T = a.N / BK;
    dma_tile(0, std::integral_constant<int, 0>{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    auto k_tile = [&](auto bufc, int t) {
        constexpr int buf = decltype(bufc)::value;
        if (t + 1 < T) dma_tile(t + 1, std::integral_constant<int, buf ^ 1>{});
        const float* A = (buf ? As1 : As0) + wm0 + l31;
        const float* B = buf ? Bs1 : Bs0;
        if (do_bias && tid < TM) {
#pragma unroll
            for (int kk = 0; kk < BK; ++kk) bsum += (buf ? As1 : As0)[kk * TM + tid];
        }
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float av[2], bv[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) av[i] = A[(kk + h) * TM + i * 32];
#pragma unroll
            for (int j = 0; j < 2; ++j) bv[j] = B[(kk + h) * TN + bcol[j]];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };
    for (int t = 0; t < T; t += 2) {
        k_tile(std::integral_constant<int, 0>{}, t);
        if (t + 1 < T) k_tile(std::integral_constant<int, 1>{}, t + 1);
    }

    if (do_bias && tid < TM) a.gb[co0 + tid] += a.scale * bsum;
    // gradWeight[co][ci][tap] += scale * acc: per register 2 rows (h) x 2 planes x 16 taps = four 64-byte runs
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void*)a.gw, 0, 0x7fffffff, 0x00020000);
    const int rowb = a.C * a.T * 4;   // bytes per output row
    unsigned vo[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int f = wn0 + 32 * j + l31;
        vo[j] = (unsigned)(((co0 + wm0 + 4 * h) * a.C + ci0 + (f >> 4)) * a.T + tap0 + (f & 15)) * 4u;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        float old[2][16];
#pragma unroll
        for (int r2 = 0; r2 < 16; ++r2)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                old[j][r2] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rg, (int)vo[j], (i * 32 + (r2 & 3) + 8 * (r2 >> 2)) * rowb, 0));
#pragma unroll
        for (int r2 = 0; r2 < 16; ++r2)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(old[j][r2] + a.scale * acc[i][j][r2]), rg, (int)vo[j],
                                                      (i * 32 + (r2 & 3) + 8 * (r2 >> 2)) * rowb, 0);
    }
}

}  // namespace

// nn.View -> nn.Linear as an H x W convolution with a 1 x 1 output grid: T = H*W taps of C planes
bool cg::head_wgrad_ok(int N, int T, int C, int Co) {
    return N > 0 && N % BK == 0 && T % TT == 0 && C % CC == 0 && Co % TM == 0 && (long)N * T * C * 4L < 0x7fffffffL &&
           (long)Co * C * T * 4L < 0x7fffffffL;
}

int cg::head_wgrad(hipStream_t st, const float* x, const float* dy, float* gw, float* gb, int N, int T, int C, int Co, float scale) {
    if (!head_wgrad_ok(N, T, C, Co)) return -1;
    HeadArgs a{x, dy, gw, gb, N, T, C, Co, scale};
    hipLaunchKernelGGL(head_wgrad_k, dim3((Co / TM) * (C / CC) * (T / TT)), dim3(256), 0, st, a);
    CG_LAUNCH_CHECK();
    return 0;
}
