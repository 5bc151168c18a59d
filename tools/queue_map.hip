// Which HIP streams share a hardware queue?  GPU_MAX_HW_QUEUES (default 4) queues serve all streams of a process; two kernels on streams
// of one queue run back to back whatever the dependency graph says.  Creates NS non-blocking streams in order and times a pair of
// 1-workgroup spin kernels on every pair (and against the null stream): ~1x = different queues, ~2x = the same queue.
// Build: hipcc --offload-arch=gfx950 -O2 tools/queue_map.hip -o tools/queue_map
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void spin(long cycles, int* out) {
    const long t0 = clock64();
    while (clock64() - t0 < cycles) { }
    if (cycles < 0) *out = 1;
}
int main(int argc, char** argv) {
    const int NS = argc > 1 ? atoi(argv[1]) : 12;
    hipStream_t s[64];
    s[0] = nullptr;
    for (int i = 1; i <= NS; ++i) CK(hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking));
    int* d; CK(hipMalloc(&d, 4));
    const long cyc = 200000;   // ~2 ms of the 100 MHz clock64 counter
    auto run = [&](int a, int b) {
        CK(hipDeviceSynchronize());
        auto t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s[a], cyc, d);
        if (b >= 0) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s[b], cyc, d);
        CK(hipDeviceSynchronize());
        return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    };
    run(1, 2);
    const double one = run(1, -1);
    printf("one kernel: %.2f ms; streams 0 (null) .. %d; '#' = the pair serialises (same hardware queue)\n    ", one, NS);
    for (int j = 0; j <= NS; ++j) printf("%2d ", j);
    printf("\n");
    int cls[64]; for (int i = 0; i <= NS; ++i) cls[i] = -1;
    int ncls = 0;
    for (int i = 0; i <= NS; ++i) {
        printf("%2d  ", i);
        for (int j = 0; j <= NS; ++j) {
            if (i == j) { printf(" . "); continue; }
            const double t = run(i, j);
            printf(" %c ", t > 1.6 * one ? '#' : ' ');
            if (t > 1.6 * one && j < i && cls[i] < 0) cls[i] = cls[j];
        }
        if (cls[i] < 0) cls[i] = ncls++;
        printf("  queue class %d\n", cls[i]);
    }
    return 0;
}
