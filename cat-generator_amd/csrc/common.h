// Shared helpers for the gfx950 kernels behind include/catgan.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../include/catgan.h"

namespace cg {

// thread-local error string returned by cg_last_error()
char* err_buf();
int fail(const char* fmt, ...);

inline hipStream_t S(void* s) { return reinterpret_cast<hipStream_t>(s); }

#define CG_HIP(call)                                                              \
    do {                                                                          \
        hipError_t e__ = (call);                                                  \
        if (e__ != hipSuccess)                                                    \
            return cg::fail("%s:%d %s -> %s", __FILE__, __LINE__, #call,          \
                            hipGetErrorString(e__));                              \
    } while (0)

#define CG_LAUNCH_CHECK()                                                         \
    do {                                                                          \
        hipError_t e__ = hipGetLastError();                                       \
        if (e__ != hipSuccess)                                                    \
            return cg::fail("%s:%d kernel launch -> %s", __FILE__, __LINE__,      \
                            hipGetErrorString(e__));                              \
    } while (0)

#define CG_REQUIRE(cond, ...)                                                     \
    do {                                                                          \
        if (!(cond)) return cg::fail(__VA_ARGS__);                                \
    } while (0)

constexpr int kNumCU = 256;  // MI355X: 8 XCDs x 32 CUs

// Tunables (cg_set_option / cg_get_option, catgan.h).  Default = the environment variable of the same name if set,
// else the built-in value; cg_set_option overrides both at run time (tests force every kernel variant this way).
enum Opt {
    OPT_SPLIT_TARGET, OPT_SPLIT_MINK, OPT_TN_SMAX, OPT_TN_TARGET, OPT_SKINNY, OPT_GEMM_SLOW, OPT_GEMM_BK32,
    OPT_COLREDUCE_WGS_PER_CU, OPT_WINO_WAVES, OPT_WINO_BK, OPT_NN_TILE, OPT_TN_TILE, OPT_NN_SPLITS, OPT_TN_SPLITS,
    OPT_EPILOGUE_STATS, OPT_SAMPLER_ATOMICS, OPT_XCD_SWIZZLE, OPT_NN_QUAD, OPT_TN_QUAD, OPT_WINO_QUAD, OPT_NN_PF, OPT_NN_GLDS, OPT_TN_GLDS, OPT_WINO_GLDS, OPT_COUNT
};
long opt(Opt o);

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// Per-stream scratch of the deterministic column reductions (batch-norm statistics / backward sums, bias gradients): every
// workgroup writes its per-channel partials here, the LAST one to arrive (a ticket counter, kept at the front of the block and
// left at zero again) adds them up in a fixed order - no floating-point atomics, so a batch-norm layer gives the same bits on
// every run (the property the reference protects by pinning stn's sampler to the CPU, models.lua:889-899).  One block per
// stream, allocated on the stream's first use (which therefore must not happen inside a graph capture: warm up first).
constexpr size_t kColScratchBytes = 8u << 20;
void* col_scratch(hipStream_t stream);   // nullptr + cg::fail() on error
// Drop every weight-gradient reduction still queued by cg_conv2d_wgrad_*_deferred (any stream): a pass that failed half way must not
// leave partial sums behind for the next flush to add to a gradient.
void wgrad_discard_all();

// grid for memory-bound grid-stride kernels: enough blocks to fill 256 CUs x 8
static inline int ew_grid(long n, int per_block = 256) {
    long b = (n + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > kNumCU * 8) b = kNumCU * 8;
    return (int)b;
}

// ---- device-side helpers -------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// true in exactly one workgroup of the grid: the last one to arrive at this point (everything the others wrote before their
// arrival is visible to it).  counter must be zero before the launch; the last workgroup leaves it at zero again.
__device__ __forceinline__ bool last_block_arrives(unsigned* counter, unsigned total) {
    __shared__ int last_flag;
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned t = atomicAdd(counter, 1u);
        last_flag = (t == total - 1) ? 1 : 0;
        if (last_flag) *counter = 0u;
    }
    __syncthreads();
    if (last_flag) __threadfence();
    return last_flag != 0;
}

// block-wide sum for 256-thread blocks; result valid in thread 0
__device__ __forceinline__ double block_sum_256(double v, double* sh /*[4]*/) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0) r = sh[0] + sh[1] + sh[2] + sh[3];
    __syncthreads();
    return r;
}

}  // namespace cg
