// What does the LDS -> MFMA inner loop of the implicit-GEMM kernels cost by itself?  No global loads, no barriers: a
// workgroup of 4 waves (wave tile 32 x 64: MI = 1, NI = 2, K tile 32 like igemm_nn_kernel<64,128,...,32>) re-reads one
// resident LDS tile `iters` times.  Variants: the ds_read_b32 fragment pattern, the k-quad ds_read_b128 pattern, the same
// with every fragment of the tile requested up front, and a register double buffer that requests the NEXT tile's
// fragments before the current tile's MFMAs.  Build: hipcc --offload-arch=gfx950 -O3 tools/lds_mfma.hip -o tools/lds_mfma
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int BM = 64, BN = 128, BKT = 32;

__device__ __forceinline__ float f4c(const float4& v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : (i == 2 ? v.z : v.w)); }

// MODE 0: b32 fragments (A[k][m], B[k][n]);  1: quads, reads placed by the compiler;  2: quads, all 12 reads of a tile first;
//      3: quads, register double buffer across tiles (reads of tile t+1 before the MFMAs of tile t)
// -DACC_AGPR: plain __launch_bounds__(256) - the compiler then keeps the accumulators in AGPRs (a[0:31]); with the (256, 2)
// bounds of the real kernels they live in arch VGPRs
#ifdef ACC_AGPR
#define LOOP_BOUNDS __launch_bounds__(256)
#else
#define LOOP_BOUNDS __launch_bounds__(256, 2)
#endif
template <int MODE>
__global__ LOOP_BOUNDS void loop_k(float* out, int iters, int dummy) {
    __shared__ __attribute__((aligned(16))) float smem[2 * BKT * BM + 2 * BKT * BN];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
    for (int i = tid; i < 2 * BKT * BM + 2 * BKT * BN; i += 256) smem[i] = (float)((i * 7 + 3) & 15) * 0.125f;
    __syncthreads();
    const int wm0 = (wave >> 1) * 32, wn0 = (wave & 1) * 64;
    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    float* As = smem;
    float* Bs = smem + 2 * BKT * BM;
    if (MODE == 0) {
        for (int it = 0; it < iters; it += 2) {
#pragma unroll
            for (int buf = 0; buf < 2; ++buf) {   // compile-time buffer index: immediates, as in the real kernel
                const float* A = As + buf * BKT * BM + wm0;
                const float* B = Bs + buf * BKT * BN + wn0 + l31;
#pragma unroll
                for (int kk = 0; kk < BKT; kk += 2) {
                    const float av = A[(kk + h) * BM + (l31 ^ (((kk >> 2) & 7) << 2))];
                    const float b0 = B[(kk + h) * BN], b1 = B[(kk + h) * BN + 32];
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b0, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b1, acc[1], 0, 0, 0);
                }
                asm volatile("" ::: "memory");
            }
        }
    } else {
        const float4* A4 = reinterpret_cast<const float4*>(As);
        const float4* B4 = reinterpret_cast<const float4*>(Bs);
        int qa[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) qa[g] = (2 * g + h) * BM + wm0 + (l31 ^ ((2 * g + h) & 7));
        const int qb = h * BN + wn0 + (l31 ^ ((l31 >> 3) & 3));
        float4 af[2][4], bf[2][4][2];
        auto frag = [&](int s, int g, int buf) {
            af[s][g] = A4[buf * (BKT * BM / 4) + qa[g]];
            bf[s][g][0] = B4[buf * (BKT * BN / 4) + qb + g * 2 * BN];
            bf[s][g][1] = B4[buf * (BKT * BN / 4) + qb + g * 2 * BN + 32];
        };
        auto mm = [&](int s, int g) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(f4c(af[s][g], c), f4c(bf[s][g][0], c), acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(f4c(af[s][g], c), f4c(bf[s][g][1], c), acc[1], 0, 0, 0);
            }
        };
        if (MODE == 3) {
#pragma unroll
            for (int g = 0; g < 4; ++g) frag(0, g, 0);
        }
        for (int it = 0; it < iters; it += 2) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int buf = half;
                if (MODE == 1) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) { frag(0, g, buf); mm(0, g); }
                } else if (MODE == 2) {
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int g = 0; g < 4; ++g) frag(0, g, buf);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int g = 0; g < 4; ++g) mm(0, g);
                    __builtin_amdgcn_sched_barrier(0);
                } else {
                    // set `half` is complete (requested one tile ago): request the other set, then multiply this one
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int g = 0; g < 4; ++g) frag(half ^ 1, g, buf);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int g = 0; g < 4; ++g) mm(half, g);
                    __builtin_amdgcn_sched_barrier(0);
                }
                asm volatile("" ::: "memory");
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[j][r];
    if (s == 12345.678f) out[0] = s;
}

template <typename F>
static double run(F launch, double flop) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    launch(); launch();
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0, 0);
    for (int r = 0; r < 5; ++r) launch();
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return flop * 5 / (ms * 1e-3) * 1e-12;
}

int main() {
    float* out;
    if (hipMalloc(&out, 4) != hipSuccess) return 1;
    const int iters = 4096;
#ifdef ACC_AGPR
    printf("accumulators in AGPRs\n");
#else
    printf("accumulators in arch VGPRs\n");
#endif
    printf("%-64s %8s\n", "LDS -> MFMA loop, 64x128x32 tile, 4 waves (32x64 each)", "TFLOP/s");
#define R(MODE, WPC, label)                                                                                             \
    {                                                                                                                   \
        const double flop = 256.0 * WPC * 4 * iters * 32 * 4096.0;                                                      \
        double tf = run([&] { hipLaunchKernelGGL((loop_k<MODE>), dim3(256 * WPC), dim3(256), 0, 0, out, iters, 0); }, flop); \
        printf("%-64s %8.1f  (%.3f of 157.3)\n", label, tf, tf / 157.3);                                                \
    }
    R(0, 1, "b32 fragments, 1 workgroup / CU");
    R(0, 2, "b32 fragments, 2 workgroups / CU");
    R(0, 3, "b32 fragments, 3 workgroups / CU");
    R(1, 2, "quad fragments (compiler-placed), 2 workgroups / CU");
    R(1, 3, "quad fragments (compiler-placed), 3 workgroups / CU");
    R(2, 1, "quad fragments, tile's 12 reads first, 1 workgroup / CU");
    R(2, 2, "quad fragments, tile's 12 reads first, 2 workgroups / CU");
    R(2, 3, "quad fragments, tile's 12 reads first, 3 workgroups / CU");
    R(3, 1, "quad fragments, register double buffer, 1 workgroup / CU");
    R(3, 2, "quad fragments, register double buffer, 2 workgroups / CU");
    R(3, 3, "quad fragments, register double buffer, 3 workgroups / CU");
    return 0;
}
