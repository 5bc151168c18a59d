// Shared helpers for the gfx950 kernels behind include/catgan.h.
#pragma once
#include <hip/hip_runtime.h>
#include <algorithm>
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

// Timing-only builds (VERDICT r05 #6e: the sensitivity numbers that steer the work must be reproducible from the repo).
//   CG_BUILD_DEFINES="-DCG_TIMING_PROBE=<mask>" python -c "import __graft_entry__ as g; g.build()"
// makes every K loop of the masked GEMM families run HALF its tiles - the same launches, the same memory-bound kernels around them,
// wrong results by construction - so that `python bench.py` shows how much of a family's duration the step's time follows:
//   1 direct implicit GEMMs (igemm_nn / igemm_nng)   2 weight gradients (igemm_tn / igemm_tng)   4 Winograd GEMMs (wino_gemm / wino_gemm_g)
//   8 the fused Winograd kernel's position loops (wino3_fused_k)
// Never set in a build whose results are used; cg_abi_version() of such a build returns -1 so that every host refuses it.
#ifndef CG_TIMING_PROBE
#define CG_TIMING_PROBE 0
#endif
#define CG_PROBE_HALF(fam, T) (((CG_TIMING_PROBE) & (fam)) ? ((T) + 1) / 2 : (T))
// grid caps of the memory-bound kernels, swept in round 4 and constants since round 6 (profiles/r04_sweeps.txt): 4 workgroups per CU for the
// grid-stride kernels (8 until round 4: beside another queue's GEMM fewer, longer-lived workgroups get through sooner), 1 per CU for the
// column reductions
constexpr int kEwWgsPerCU = 4, kColReduceWgsPerCU = 1;

// Tunables (cg_set_option / cg_get_option, catgan.h).  Default = the environment variable of the same name if set,
// else the built-in value; cg_set_option overrides both at run time (tests force every kernel variant this way).
enum Opt {
    OPT_SKINNY, OPT_GEMM_BK32, OPT_WINO_BK, OPT_NN_TILE, OPT_TN_TILE, OPT_NN_SPLITS, OPT_TN_SPLITS,
    OPT_XCD_SWIZZLE, OPT_NN_GLDS, OPT_TN_GLDS, OPT_WINO_GLDS, OPT_PAD_SKIP, OPT_WINO3, OPT_COUNT
};
long opt(Opt o);

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// Per-stream scratch of the deterministic column reductions (batch-norm statistics / backward sums, bias gradients): every
// workgroup writes its per-channel partials here, the LAST one to arrive (a ticket counter, kept at the front of the block and
// left at zero again) adds them up in a fixed order - no floating-point atomics, so a batch-norm layer gives the same bits on
// every run (the property the reference protects by pinning stn's sampler to the CPU, models.lua:889-899).  One block per
// stream out of a pool of eight allocated on the FIRST use of any stream (that one must be outside a graph capture; a stream that
// first appears inside a capture takes a pooled block without allocating).
constexpr size_t kColScratchBytes = 8u << 20;
void* col_scratch(hipStream_t stream);   // nullptr + cg::fail() on error
// ---- hardware queues (round 5).  The HIP runtime serves ALL streams of a process from GPU_MAX_HW_QUEUES = 4 hardware queues,
// assigned in creation order; two streams of one queue run back to back whatever the dependency graph says, and WHICH streams share a
// queue decides how much of a plan's overlap exists (measured: +-10 % per step between creation orders; one extra stream created by
// the library once moved the host's side stream onto the main stream's queue, +2.7 %; more than 4 queues is far worse - 8: 7.4 ms,
// 5: 11 ms per step against 5.9, the queues then time-slice the chip).  So side streams are not created where they are needed: they
// come out of one pool whose streams are CLASSIFIED, once, by a timing probe against the caller's stream (two spinning kernels on a
// pair of streams: together they take one kernel's time on different queues, two on the same), and a plan asks for "the k-th stream of
// queue class c": class 0 = the caller's own queue, 1..3 = the other three, numbered in the order the probe meets them.
hipStream_t queue_stream(hipStream_t ref, int cls, int slot);   // nullptr + cg::fail() on error
int queue_class(hipStream_t ref, hipStream_t s);                // class of a pool stream (or of ref itself: 0); -1 unknown
// Drop every weight-gradient reduction still queued by cg_conv2d_wgrad_*_deferred (any stream): a pass that failed half way must not
// leave partial sums behind for the next flush to add to a gradient.
void wgrad_discard_all();

// 3x3 convolutions with <= 3 output planes as a scatter-form GEMM on the MFMA, input read once (csrc/skinny.hip, round 6): widths that
// are a multiple of 32 up to 128, 64 or 128 input planes.  CG_SKINNY: 1 = these where they apply (else round 1's VALU kernels of
// gemm.hip), 2 = the VALU kernels always, 0 = the generic implicit GEMM.
bool skinny_mfma_ok(int Cin, int Cout, int H, int W);
int skinny_mfma_forward(hipStream_t st, const float* x, const float* w, const float* bias, float* y, int N, int H, int W, int Cin, int Cout);
int skinny_mfma_wgrad(hipStream_t st, const float* x, const float* dy, float* part, float* bias_part, int N, int H, int W, int Cin, int Cout,
                      int max_blocks);   // returns the number of partial planes written, -1 on error

// Fused-transform Winograd F(2x2,3x3) for plain 64 -> 64 plane 3x3 convolutions (csrc/wino3.hip, round 6).  CG_WINO3: 1 = where the
// geometry fits and the launch has >= 2 workgroups per CU, 2 = wherever the geometry fits (tests), 0 = never (the direct implicit GEMM).
// headwg.hip: weight gradient of nn.View -> nn.Linear on an NHWC map (T = H*W taps of C planes) straight into the canonical gradWeight
bool head_wgrad_ok(int N, int T, int C, int Co);
int head_wgrad(hipStream_t st, const float* x, const float* dy, float* gw, float* gb, int N, int T, int C, int Co, float scale);
bool wino3_geom_ok(int ngroups, int N, int H, int W, int Cin, int Cout, int kH, int kW, int padH, int padW, int ups);
int wino3_note_pack(hipStream_t st, int n, const float* const* w, float* const* wf, float* const* wb, const int* Cout, const int* Cin,
                    const int* kH, const int* kW, const int* wb_map);
int wino3_forward(hipStream_t st, int ngroups, const float* const* x, const float* const* wpk, const float* const* bias, float* const* y,
                  int N, int H, int W);   // 1 launched, 0 not this path, -1 error

// grid for memory-bound grid-stride kernels: enough blocks to fill 256 CUs x 8
static inline int ew_grid(long n, int per_block = 256) {
    long b = (n + per_block - 1) / per_block;
    if (b < 1) b = 1;
    // kEwWgsPerCU = 4 (8 until round 4): beside a GEMM from another queue a memory-bound kernel waits for a slot per workgroup,
    // so fewer, longer-lived workgroups get through sooner - same-box step 6.36 -> 6.27 ms, alone no difference (profiles/r04_sweeps.txt)
    const long cap = (long)kNumCU * kEwWgsPerCU;
    if (b > cap) b = cap;
    return (int)b;
}

// ---- device-side helpers -------------------------------------------------
// the engine's counter-based generator: splitmix64 of (seed, counter), 24-bit mantissa uniform in [0,1)
// (host twin tensor.SplitMix, oracle twin orc_rng_u01)
__device__ __forceinline__ float u01(uint64_t seed, uint64_t ctr) {
    uint64_t z = seed + (ctr + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (float)(z >> 40) * (1.0f / 16777216.0f);
}
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

// Cross-workgroup hand-over of the partial sums.  The eight XCDs have private L2s, so an agent-scope fence costs a write-back
// walk plus an invalidate of the XCD's whole L2; issued by every wave of every workgroup (__threadfence in all threads) that is
// 128 walks per XCD, serialised: +30 us on an 18 us reduction (measured).  CG_COL_HANDOVER picks the protocol:
//   0  __threadfence() in every thread on both sides (the textbook form)
//   3  one release fence per WORKGROUP (thread 0, after the barrier that orders the other waves' stores before it - fences are
//      cumulative), one acquire fence per wave of the finishing workgroups only
//   4  no fences: partials are published and read with returning agent-scope atomics (swap / or-with-zero), which execute at the
//      device's coherence point like the double atomicAdds this replaces
// Same box, ms per step: 0 = 7.23, 3 = 7.13 - 7.16, 4 = 7.12 - 7.15 (profiles/r03_colreduce_handover.txt).  A fence-free form with
// write-through (sc1) stores / sc1 loads and only s_waitcnt in front of the ticket measured 7.10 - 7.13 but is NOT covered by the
// documented memory model (is a store's vmcnt acknowledgement its visibility to the other XCDs?) and failed parity runs that a
// later-found host-side race (the scratch's memset against non-blocking streams, see col_scratch) may or may not explain: not shipped.
#ifndef CG_COL_HANDOVER
#define CG_COL_HANDOVER 3
#endif
__device__ __forceinline__ void st_agent(double* p, double v) {
#if CG_COL_HANDOVER == 4
    const unsigned long long old = __hip_atomic_exchange((unsigned long long*)p, (unsigned long long)__double_as_longlong(v),
                                                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("" ::"v"(old));   // keep the returning form: its s_waitcnt means the swap has been performed
#else
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
__device__ __forceinline__ double ld_agent(const double* p) {
#if CG_COL_HANDOVER == 4
    unsigned long long zero = 0ull;
    asm volatile("" : "+v"(zero));   // opaque, or the compiler turns the idempotent read-modify-write into a plain load
    return __longlong_as_double((long long)__hip_atomic_fetch_or((unsigned long long*)const_cast<double*>(p), zero, __ATOMIC_RELAXED,
                                                                 __HIP_MEMORY_SCOPE_AGENT));
#else
    return __hip_atomic_load(const_cast<double*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}

// true in exactly one workgroup of the grid: the last one to arrive at this point.  What the others published with st_agent before
// their arrival is visible to its ld_agent loads.  counter must be zero before the launch; the last workgroup leaves it at zero again.
__device__ __forceinline__ bool last_block_arrives(unsigned* counter, unsigned total) {
    __shared__ int last_flag;
#if CG_COL_HANDOVER == 0
    __threadfence();
#else
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    __syncthreads();
    if (threadIdx.x == 0) {
#if CG_COL_HANDOVER == 3
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
#endif
        const unsigned t = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last_flag = (t == total - 1) ? 1 : 0;
        if (last_flag) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
#if CG_COL_HANDOVER == 0
    if (last_flag) __threadfence();
#elif CG_COL_HANDOVER == 3
    if (last_flag) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
    return last_flag != 0;
}

// ---- second stage of the deterministic column reductions ------------------------------------------------------------
// A launch of `chunks` row chunks leaves chunks x nk fp64 partials.  Adding them in ONE workgroup costs chunks dependent-latency
// loads per column (measured: 256 chunks = 90 us on the tail of an 18 us kernel), so the sum is a fixed two-level tree: chunks are
// grouped G at a time, the last workgroup to arrive of a GROUP adds its G rows (16 loads in flight per column) into the group's row,
// the last group to finish adds the group rows.  Which workgroup does a sum depends on timing, the ORDER of every addition does not.
struct ColTree {
    int G, ngroups;
    unsigned* counters;   // [0] groups finished, [1 + g] arrivals of group g; all zero between launches
    double* part;         // [chunks][nk]
    double* grp;          // [ngroups][nk]
    double* extra;        // caller-defined tail (e.g. the PReLU slope partials)
};
constexpr int kColTreeMaxGroups = 32;
static inline bool col_tree_layout(char* scr, long chunks, size_t nk, size_t nextra, ColTree& t) {
    t.G = (int)std::max<long>(16, (chunks + kColTreeMaxGroups - 1) / kColTreeMaxGroups);
    t.ngroups = (int)((chunks + t.G - 1) / t.G);
    t.counters = (unsigned*)scr;
    t.part = (double*)(scr + 256);
    t.grp = t.part + (size_t)chunks * nk;
    t.extra = t.grp + (size_t)t.ngroups * nk;
    return 256 + sizeof(double) * ((size_t)chunks * nk + (size_t)t.ngroups * nk + nextra) <= kColScratchBytes;
}

// out[k] = rows[0][k] + rows[1][k] + ... in row order, loads 16 rows ahead of the additions (rows published with st_agent)
__device__ __forceinline__ void col_sum_rows(const double* rows, int nrows, int nk, double* out) {
    for (int k = threadIdx.x; k < nk; k += blockDim.x) {
        double t = 0.0;
        int y = 0;
        for (; y + 16 <= nrows; y += 16) {
            double v[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = ld_agent(&rows[(size_t)(y + j) * nk + k]);
#pragma unroll
            for (int j = 0; j < 16; ++j) t += v[j];
        }
        for (; y < nrows; ++y) t += ld_agent(&rows[(size_t)y * nk + k]);
        st_agent(&out[k], t);
    }
}

// called by every workgroup after it stored its partial row (chunk = its row chunk, blocks_per_chunk = workgroups sharing a chunk);
// true in the one workgroup that wrote the final sums
__device__ __forceinline__ bool col_tree_finish(const ColTree& t, int chunk, int chunks, unsigned blocks_per_chunk, int nk, double* sums) {
    const int g = chunk / t.G;
    const int in_group = min(t.G, chunks - g * t.G);
    if (!last_block_arrives(t.counters + 1 + g, (unsigned)in_group * blocks_per_chunk)) return false;
    col_sum_rows(t.part + (size_t)g * t.G * nk, in_group, nk, t.grp + (size_t)g * nk);
    if (!last_block_arrives(t.counters, (unsigned)t.ngroups)) return false;
    col_sum_rows(t.grp, t.ngroups, nk, sums);
    return true;
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
