// Practical fp32-MFMA ceiling on one MI355X: register-only loops of v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32 with
// 1..4 independent accumulator tiles per wave and 1..4 waves per SIMD, optionally with VALU / LDS instructions in the loop
// (the instruction mixes of the implicit-GEMM kernels).  Prints achieved TFLOP/s; 157.3 TF = 256 CUs x 4 SIMDs x 64 FLOP/clk
// x 2.4 GHz.  Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o tools/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, int VALU, int LDS>
__global__ __launch_bounds__(256) void k32(float* out, int iters, float a, float b) {
    __shared__ float4 sm[1024];
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float av = a + threadIdx.x, bv = b;
    float tv[4] = {a, b, a + 1.f, b + 1.f};
    if (LDS) sm[threadIdx.x] = make_float4(a, b, a, b), sm[threadIdx.x + 256] = make_float4(b, a, b, a);
    __syncthreads();
    const float4* sp = sm + (threadIdx.x & 255);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (LDS && (u & 3) == 0) {
                float4 q = sp[(u & 4) ? 256 : 0];
                asm volatile("" : "+v"(q.x), "+v"(q.y), "+v"(q.z), "+v"(q.w));
                av += q.x + q.z; bv += q.y + q.w;
            }
#pragma unroll
            for (int i = 0; i < NACC; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i], 0, 0, 0);
#pragma unroll
                for (int v = 0; v < VALU; ++v) { float& tr = tv[(i * VALU + v + u) & 3]; tr = tr * 1.0001f + b; asm volatile("" : "+v"(tr)); }
            }
        }
    }
    float s = tv[0] + tv[1] + tv[2] + tv[3];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 12345.678f) out[0] = s;
}

template <int NACC>
__global__ __launch_bounds__(256) void k16(float* out, int iters, float a, float b) {
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    float av = a + threadIdx.x, bv = b;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) s += acc[i][r];
    if (s == 12345.678f) out[0] = s;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <typename F>
static double run(F launch, double flop) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); launch();
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int r = 0; r < 5; ++r) launch();
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    return flop * 5 / (ms * 1e-3) * 1e-12;
}

int main() {
    float* out;
    CK(hipMalloc(&out, 4));
    const int iters = 4096;
    const int CUS = 256;
    printf("%-44s %8s\n", "loop (per wave), workgroups of 4 waves", "TFLOP/s");
#define R32(NACC, VALU, LDS, WPC, label)                                                                                  \
    {                                                                                                                      \
        const double flop = (double)CUS * WPC * 4 * iters * 8 * NACC * 4096.0;                                              \
        double tf = run([&] { hipLaunchKernelGGL((k32<NACC, VALU, LDS>), dim3(CUS * WPC), dim3(256), 0, 0, out, iters, 1.f, 2.f); }, flop); \
        printf("%-44s %8.1f  (%.3f of 157.3)\n", label, tf, tf / 157.3);                                                    \
    }
    R32(1, 0, 0, 1, "32x32x2, 1 acc (dependent), 1 wave/SIMD");
    R32(2, 0, 0, 1, "32x32x2, 2 acc, 1 wave/SIMD");
    R32(4, 0, 0, 1, "32x32x2, 4 acc, 1 wave/SIMD");
    R32(4, 0, 0, 2, "32x32x2, 4 acc, 2 waves/SIMD");
    R32(4, 0, 0, 3, "32x32x2, 4 acc, 3 waves/SIMD");
    R32(4, 0, 0, 4, "32x32x2, 4 acc, 4 waves/SIMD");
    R32(2, 0, 0, 2, "32x32x2, 2 acc, 2 waves/SIMD");
    R32(2, 0, 0, 3, "32x32x2, 2 acc, 3 waves/SIMD");
    R32(4, 1, 0, 2, "32x32x2, 4 acc, 2 waves, +1 VALU / MFMA");
    R32(4, 2, 0, 2, "32x32x2, 4 acc, 2 waves, +2 VALU / MFMA");
    R32(4, 4, 0, 2, "32x32x2, 4 acc, 2 waves, +4 VALU / MFMA");
    R32(4, 0, 1, 2, "32x32x2, 4 acc, 2 waves, ds_read_b128 / 16 MFMA");
    R32(2, 0, 1, 3, "32x32x2, 2 acc, 3 waves, ds_read_b128 / 8 MFMA");
#define R16(NACC, WPC, label)                                                                                              \
    {                                                                                                                      \
        const double flop = (double)CUS * WPC * 4 * iters * 8 * NACC * 2048.0;                                              \
        double tf = run([&] { hipLaunchKernelGGL((k16<NACC>), dim3(CUS * WPC), dim3(256), 0, 0, out, iters, 1.f, 2.f); }, flop); \
        printf("%-44s %8.1f  (%.3f of 157.3)\n", label, tf, tf / 157.3);                                                    \
    }
    R16(4, 1, "16x16x4, 4 acc, 1 wave/SIMD");
    R16(4, 2, "16x16x4, 4 acc, 2 waves/SIMD");
    R16(8, 2, "16x16x4, 8 acc, 2 waves/SIMD");
    return 0;
}
