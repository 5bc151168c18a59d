// Semantics check for buffer_load_dwordx4 ... lds on gfx950: lane l's 16 bytes land at M0 base + 16 l; what is written for a
// lane whose offset is out of the descriptor's range?  (the implicit-GEMM gather relies on "0".)
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(const float* x, float* y, int mode) {
    __shared__ __attribute__((aligned(16))) float sm[512];
    for (int i = threadIdx.x; i < 512; i += 64) sm[i] = -7.f;
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, 0x7fffffff, 0x00020000);
    const unsigned OOB = 0x80000000u;
    // even lanes load their quad (permuted: lane l loads quad l ^ 1), odd lanes are out of range when mode == 1
    unsigned voff = ((threadIdx.x ^ 1) * 16);
    if (mode == 1 && (threadIdx.x & 1)) voff = OOB;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(sm + 256), 16, voff, 0, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);   // vmcnt(0) expcnt(0) lgkmcnt(0)
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += 64) y[i] = sm[i];
}
int main() {
    float *x, *y, hx[512], hy[512];
    for (int i = 0; i < 512; ++i) hx[i] = (float)i;
    if (hipMalloc(&x, 2048) != hipSuccess || hipMalloc(&y, 2048) != hipSuccess) return 1;
    (void)hipMemcpy(x, hx, 2048, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 2; ++mode) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, x, y, mode);
        (void)hipMemcpy(hy, y, 2048, hipMemcpyDeviceToHost);
        printf("mode %d: untouched half [0]=%g [255]=%g; lanes 0..3 ->", mode, hy[0], hy[255]);
        for (int i = 256; i < 256 + 16; ++i) printf(" %g", hy[i]);
        printf("\n");
    }
    return 0;
}
