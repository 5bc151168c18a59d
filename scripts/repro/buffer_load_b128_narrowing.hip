// Reproducer for a ROCm 7.2 (-O3) miscompile: extracting the four dwords of a
// __builtin_amdgcn_raw_buffer_load_b128 result one by one makes InstCombine narrow the v4i32 buffer load to a single
// i32 and splat it (output "0 0 0 0 4 4 4 4 ..." instead of "0 1 2 3 4 5 6 7 ...").  Casting the whole vector
// (csrc/gemm.hip bufld4) keeps buffer_load_dwordx4.  Build: hipcc --offload-arch=gfx950 -O3 this.hip -o t && ./t
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* x, float* y, int soff, int bias) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)(x + bias), 0, 0x7fffffff, 0x00020000);
    unsigned voff = threadIdx.x * 16;
    if (threadIdx.x == 3) voff = 0x80000000u;
    u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    y[threadIdx.x * 4 + 0] = __builtin_bit_cast(float, v.x);
    y[threadIdx.x * 4 + 1] = __builtin_bit_cast(float, v.y);
    y[threadIdx.x * 4 + 2] = __builtin_bit_cast(float, v.z);
    y[threadIdx.x * 4 + 3] = __builtin_bit_cast(float, v.w);
}
int main() {
    float *x, *y; float hx[1024], hy[64];
    for (int i = 0; i < 1024; ++i) hx[i] = (float)i;
    hipMalloc(&x, 4096); hipMalloc(&y, 256);
    hipMemcpy(x, hx, 4096, hipMemcpyHostToDevice);
    for (int t = 0; t < 3; ++t) {
        int soff = t == 0 ? 0 : 64, bias = t == 2 ? -8 : 0;
        const float* base = t == 2 ? x + 100 : x;
        hipLaunchKernelGGL(k, dim3(1), dim3(8), 0, 0, base, y, soff, bias);
        hipMemcpy(hy, y, 128, hipMemcpyDeviceToHost);
        printf("soff=%d bias=%d:", soff, bias);
        for (int i = 0; i < 32; ++i) printf(" %g", hy[i]);
        printf("\n");
    }
    return 0;
}
