// Memory-bound kernels of the G+D step (everything that is not a GEMM):
// activations, BCE, batch-norm, pooling, dropout, layout changes, spatial
// transformer pieces, fused penalty+clamp+Adam, small reductions.  NHWC fp32.
// Each kernel cites the reference module it stands in for (see catgan.h).
#include "common.h"
#include <stdarg.h>
#include <stdlib.h>
#include <algorithm>

namespace cg {
char* err_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}
int fail(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return 1;
}

namespace {
struct OptDef { const char* name; long dflt; };
// order = enum Opt (common.h)
const OptDef kOptDefs[OPT_COUNT] = {
    {"CG_SKINNY", 1}, {"CG_GEMM_BK32", 1}, {"CG_WINO_BK", 0}, {"CG_NN_TILE", 0}, {"CG_TN_TILE", 0}, {"CG_NN_SPLITS", 0}, {"CG_TN_SPLITS", 0},
    {"CG_XCD_SWIZZLE", 7}, {"CG_NN_GLDS", 1}, {"CG_TN_GLDS", 1}, {"CG_WINO_GLDS", 1}, {"CG_PAD_SKIP", 20}, {"CG_WINO3", 1},
};
long g_opt_val[OPT_COUNT];
int g_opt_state[OPT_COUNT];   // 0 = not read yet, 1 = default / environment, 2 = set through the ABI
}  // namespace

}  // namespace cg
#include <chrono>
#include <mutex>
#include <unordered_map>
#include <vector>
namespace cg {
void* col_scratch(hipStream_t stream) {
    // a pool of blocks allocated (and their ticket counters zeroed) in chunks of kPool, handed to streams as they appear.  A stream
    // that shows up inside a graph capture (torch.cuda.graph captures on a stream of its own) must find a free block - nothing can
    // be allocated then - so every hand-over outside a capture leaves at least kReserve blocks behind (a plan's side and
    // weight-gradient streams, several nets per process: streams are not scarce any more).
    constexpr int kPool = 8, kReserve = 4;
    static std::unordered_map<hipStream_t, void*> blocks;
    static std::vector<void*> pool;
    static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    auto it = blocks.find(stream);
    if (it != blocks.end()) return it->second;
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    const bool capturing = stream && hipStreamIsCapturing(stream, &st) == hipSuccess && st != hipStreamCaptureStatusNone;
    if (capturing && pool.empty()) {
        cg::fail("column reduce: no scratch block left for a stream first used inside a graph capture (run one pass on it before cg_graph_begin)");
        return nullptr;
    }
    if (!capturing && (int)pool.size() <= kReserve) {
        char* p = nullptr;
        // ANOTHER stream of the process may be inside a graph capture right now (a plan's side / weight-gradient streams make
        // refills frequent): allocation, a memset and a synchronisation are "potentially unsafe" calls that a capture in global mode
        // forbids to every thread - so the refill runs with this thread's capture mode relaxed, and the memset is not on the null stream
        // (which would drag a capturing blocking stream in).  Nothing here touches the capturing stream's work; the blocks are handed
        // out only to streams that are not capturing.
        // No stream of its own for the memset: HIP maps streams onto a handful of hardware queues in creation order, and one more stream
        // moved the training host's side stream onto the main stream's queue (round 5: both generator forwards ran back to back, +2.7 %
        // per step).  The memset goes on the REQUESTING stream - not capturing, we checked - and only that stream is waited for.
        hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
        (void)hipThreadExchangeStreamCaptureMode(&mode);
        const bool ok = hipMalloc((void**)&p, kColScratchBytes * kPool) == hipSuccess &&
                        hipMemsetAsync(p, 0, kColScratchBytes * kPool, stream) == hipSuccess &&   // finished before any kernel draws a ticket
                        hipStreamSynchronize(stream) == hipSuccess;
        (void)hipThreadExchangeStreamCaptureMode(&mode);
        if (!ok) {
            (void)hipGetLastError();
            if (pool.empty()) { cg::fail("column reduce: cannot allocate %zu bytes of scratch", kColScratchBytes * kPool); return nullptr; }
        } else
            for (int i = kPool - 1; i >= 0; --i) pool.insert(pool.begin(), p + (size_t)i * kColScratchBytes);   // older blocks go first
    }
    void* blk = pool.back();
    pool.pop_back();
    blocks[stream] = blk;
    return blk;
}
namespace {
__global__ void queue_probe_spin_k(unsigned long long ticks, int* sink) {   // wall_clock64: 100 MHz, independent of the shader clock
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) { }
    if (ticks == ~0ull) *sink = 1;
}
struct QueuePool {
    std::vector<hipStream_t> s;
    std::unordered_map<hipStream_t, std::vector<int>> cls;   // per reference stream: class of every pool stream
    std::mutex mu;
};
QueuePool& qpool() { static QueuePool p; return p; }
constexpr int kQueuePool = 24;

// milliseconds for one spinning kernel on a (and one on b, if b != a's sentinel) to finish
double probe_pair(hipStream_t a, hipStream_t b, bool both, unsigned long long ticks) {
    (void)hipStreamSynchronize(a);
    if (both) (void)hipStreamSynchronize(b);
    const auto t0 = std::chrono::steady_clock::now();
    hipLaunchKernelGGL(queue_probe_spin_k, dim3(1), dim3(64), 0, a, ticks, (int*)nullptr);
    if (both) hipLaunchKernelGGL(queue_probe_spin_k, dim3(1), dim3(64), 0, b, ticks, (int*)nullptr);
    (void)hipStreamSynchronize(a);
    if (both) (void)hipStreamSynchronize(b);
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}
bool same_queue(hipStream_t a, hipStream_t b) {
    const unsigned long long ticks = 30000;   // 0.3 ms
    int votes = 0;
    for (int r = 0; r < 3; ++r) {
        const double one = probe_pair(a, a, false, ticks), two = probe_pair(a, b, true, ticks);
        if (two > 1.6 * one) ++votes;
    }
    return votes >= 2;
}
}  // namespace

int queue_classify(hipStream_t ref) {     // fills qpool().cls[ref]; caller holds the lock
    QueuePool& p = qpool();
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (ref && hipStreamIsCapturing(ref, &st) == hipSuccess && st != hipStreamCaptureStatusNone)
        return cg::fail("queue_stream: a stream's hardware queue cannot be probed inside a graph capture (run one pass on it before cg_graph_begin)");
    while ((int)p.s.size() < kQueuePool) {
        hipStream_t q;
        if (hipStreamCreateWithFlags(&q, hipStreamNonBlocking) != hipSuccess) return cg::fail("queue_stream: hipStreamCreate failed");
        p.s.push_back(q);
    }
    std::vector<int> c(p.s.size(), -1);
    std::vector<hipStream_t> rep;           // representative of class 1, 2, 3
    for (size_t i = 0; i < p.s.size(); ++i) {
        if (same_queue(ref, p.s[i])) { c[i] = 0; continue; }
        for (size_t k = 0; k < rep.size() && c[i] < 0; ++k)
            if (same_queue(rep[k], p.s[i])) c[i] = (int)k + 1;
        if (c[i] < 0) { rep.push_back(p.s[i]); c[i] = (int)rep.size(); }
    }
    (void)hipGetLastError();
    p.cls[ref] = c;
    return 0;
}

hipStream_t queue_stream(hipStream_t ref, int cls, int slot) {
    QueuePool& p = qpool();
    std::lock_guard<std::mutex> lk(p.mu);
    if (!p.cls.count(ref) && queue_classify(ref)) return nullptr;
    const std::vector<int>& c = p.cls[ref];
    int nclass = 0;
    for (int v : c) nclass = std::max(nclass, v + 1);
    if (nclass < 2) {
        // GPU_MAX_HW_QUEUES=1, a serialising profiler / HIP_LAUNCH_BLOCKING, or a busy shared GPU that skewed the probe: results never
        // depend on the queue class (only the overlap does), so hand out distinct pool streams and say so once
        static bool warned = false;
        if (!warned) { warned = true; fprintf(stderr, "cg: the hardware-queue probe found one queue only - side streams are plain pool streams (overlap not guaranteed)\n"); }
        return p.s[(size_t)((cls & 3) * (kQueuePool / 4) + slot % (kQueuePool / 4)) % p.s.size()];
    }
    cls = cls % nclass;                     // fewer than four queues (GPU_MAX_HW_QUEUES < 4): wrap
    int seen = 0;
    hipStream_t last = nullptr;
    for (size_t i = 0; i < c.size(); ++i)
        if (c[i] == cls) { last = p.s[i]; if (seen++ == slot) return p.s[i]; }
    if (last) return last;                  // more roles than pool streams of this class: share the last one
    cg::fail("queue_stream: no pool stream on hardware queue class %d", cls);
    return nullptr;
}

int queue_class(hipStream_t ref, hipStream_t s) {
    if (s == ref) return 0;
    QueuePool& p = qpool();
    std::lock_guard<std::mutex> lk(p.mu);
    auto it = p.cls.find(ref);
    if (it == p.cls.end()) return -1;
    for (size_t i = 0; i < p.s.size(); ++i) if (p.s[i] == s) return it->second[i];
    return -1;
}

unsigned long g_opt_epoch = 1;   // bumped by cg_set_option: compiled plans (net.hip) re-derive workspace sizes / dispatch-dependent rows
long opt(Opt o) {
    if (g_opt_state[o] == 0) {
        const char* e = getenv(kOptDefs[o].name);
        g_opt_val[o] = e ? atol(e) : kOptDefs[o].dflt;
        g_opt_state[o] = 1;
    }
    return g_opt_val[o];
}
}  // namespace cg

namespace {
using cg::block_sum_256;
using cg::last_block_arrives;
using cg::wave_sum;

#define GRID_STRIDE(i, n) \
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < (n); i += (long)gridDim.x * blockDim.x)

// ---------------------------------------------------------------- input pipeline
// one lane per pixel; every operation correctly rounded on its own, like the numpy loader's
__global__ void images_u8_to_f32_k(const unsigned char* __restrict__ src, float* __restrict__ dst, long npix, int cs) {
#pragma clang fp contract(off)   // no fma: the same three products and two sums the numpy loader rounds one by one
    GRID_STRIDE(i, npix) {
        const float r = __fdiv_rn((float)src[3 * i], 255.f), g = __fdiv_rn((float)src[3 * i + 1], 255.f), b = __fdiv_rn((float)src[3 * i + 2], 255.f);
        if (cs == 0) { dst[3 * i] = r; dst[3 * i + 1] = g; dst[3 * i + 2] = b; }
        else {
            const float t0 = 0.21f * r, t1 = 0.72f * g, t2 = 0.07f * b;   // plain operators: the pragma above governs them
            dst[i] = (t0 + t1) + t2;
        }
    }
}

// image.load -> image.scale -> NN_UTILS.rgbToColorSpace (dataset.lua:123-131,166) on the device: 8-bit RGB as decoded, [N][Hs][Ws][3]
// -> fp32 NHWC [N][Hd][Wd][C].  image.scale's default mode is the torch `image` rock's separable 'bilinear' [upstream, recalled:
// generic/image.c scaleLinear_rowcol]: first every row to the target width, then every column to the target height, each pass in
// fp32.  One axis, source length Ls, target length Ld, scale = (float)Ls / Ld (down) or (float)(Ls-1)/(Ld-1) (up):
//   Ld <  Ls: box average with fractional ends - out[d] = (w0 s[i0] + s[i0+1] + ... + s[i1-1] + f1 s[i1]) / (w0 + ... + f1), where
//             [i0 + (1-w0), i1 + f1) = [d scale, (d+1) scale) and the last term is dropped when i1 == Ls;
//   Ld >  Ls: out[d] = (1-f) s[i] + f s[i+1] with i + f = d scale, the last sample copied;          Ld == Ls: a copy.
// Every operation is a single correctly rounded fp32 one in the order of that loop (no fma), so the host loader's numpy restatement
// and this kernel agree bit for bit.  One lane per output element.
// s[k - base] holds source sample k for the few k this target sample touches
__device__ __forceinline__ float scale_axis(const float* w, int base, int Ls, int Ld, int d) {
#pragma clang fp contract(off)
#define s(k) w[(k) - base]
    if (Ld == Ls) return s(d);
    if (Ld > Ls) {
        if (d == Ld - 1 || Ls == 1) return s(Ls == 1 ? 0 : Ls - 1);
        const float scale = __fdiv_rn((float)(Ls - 1), (float)(Ld - 1));
        float sf = (float)d * scale;
        const int si = (int)sf;
        sf -= (float)si;
        const float a = (1.f - sf) * s(si), b = sf * s(si + 1);
        return a + b;
    }
    const float scale = __fdiv_rn((float)Ls, (float)Ld);
    float f0 = (float)d * scale;
    int i0 = (int)f0;
    f0 -= (float)i0;
    if (d == 0) { i0 = 0; f0 = 0.f; }
    float f1 = (float)(d + 1) * scale;
    const int i1 = (int)f1;
    f1 -= (float)i1;
    float acc = (1.f - f0) * s(i0), n = 1.f - f0;
    for (int si = i0 + 1; si < i1; ++si) { acc += s(si); n += 1.f; }
    if (i1 < Ls) { const float t = f1 * s(i1); acc += t; n += f1; }
    return __fdiv_rn(acc, n);
#undef s
}
__global__ void images_u8_scale_k(const unsigned char* __restrict__ src, float* __restrict__ dst, int N, int Hs, int Ws, int Hd, int Wd, int cs) {
#pragma clang fp contract(off)
    const long total = (long)N * Hd * Wd;
    GRID_STRIDE(i, total) {
        const int ox = (int)(i % Wd), oy = (int)((i / Wd) % Hd);
        const long n = i / ((long)Wd * Hd);
        const unsigned char* im = src + n * (long)Hs * Ws * 3;
        // vertical footprint of this output row (the second pass reads the first pass's fp32 rows)
        int y0 = oy, y1 = oy;
        if (Hd < Hs) {
            const float scale = __fdiv_rn((float)Hs, (float)Hd);
            y0 = oy == 0 ? 0 : (int)((float)oy * scale);
            y1 = min(Hs - 1, (int)((float)(oy + 1) * scale));
        } else if (Hd > Hs) {
            const float scale = __fdiv_rn((float)(Hs - 1), (float)(Hd - 1));
            y0 = (oy == Hd - 1 || Hs == 1) ? Hs - 1 : (int)((float)oy * scale);
            y1 = min(Hs - 1, y0 + 1);
        }
        float rgb[3];
        for (int c = 0; c < 3; ++c) {
            float col[8];   // first pass at column ox for the source rows y0..y1 (<= 8 rows: down-scaling factors up to 6)
            for (int y = y0; y <= y1 && y - y0 < 8; ++y) {
                // horizontal footprint of ox in row y
                int x0 = ox, x1 = ox;
                if (Wd < Ws) {
                    const float sc = __fdiv_rn((float)Ws, (float)Wd);
                    x0 = ox == 0 ? 0 : (int)((float)ox * sc);
                    x1 = min(Ws - 1, (int)((float)(ox + 1) * sc));
                } else if (Wd > Ws) {
                    const float sc = __fdiv_rn((float)(Ws - 1), (float)(Wd - 1));
                    x0 = (ox == Wd - 1 || Ws == 1) ? Ws - 1 : (int)((float)ox * sc);
                    x1 = min(Ws - 1, x0 + 1);
                }
                float row[8];
                for (int x = x0; x <= x1 && x - x0 < 8; ++x) row[x - x0] = __fdiv_rn((float)im[((long)y * Ws + x) * 3 + c], 255.f);
                col[y - y0] = scale_axis(row, x0, Ws, Wd, ox);
            }
            rgb[c] = scale_axis(col, y0, Hs, Hd, oy);
        }
        if (cs == 0) { dst[3 * i] = rgb[0]; dst[3 * i + 1] = rgb[1]; dst[3 * i + 2] = rgb[2]; }
        else {
            const float t0 = 0.21f * rgb[0], t1 = 0.72f * rgb[1], t2 = 0.07f * rgb[2];   // z = ((0 + .21 r) + .72 g) + .07 b (nn_utils.lua:268-270)
            dst[i] = (t0 + t1) + t2;
        }
    }
}

// ---------------------------------------------------------------- activations
// Memory-bound elementwise kernels move 16 B per lane (float4) when the buffers are 16-B aligned; `n4` counts
// whole float4s, the (<4)-element tail is handled by the last lanes with scalar accesses.
#define V4_LOOP(i, n4) \
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < (n4); i += (long)gridDim.x * blockDim.x)
__device__ __forceinline__ float4 ldv(const float* p, long i) { return reinterpret_cast<const float4*>(p)[i]; }
__device__ __forceinline__ void stv(float* p, long i, float4 v) { reinterpret_cast<float4*>(p)[i] = v; }
// per-channel parameters are views into the flat vector (any 4-B alignment): four scalar loads
__device__ __forceinline__ float4 ldc(const float* p, int c4) { return make_float4(p[4 * c4], p[4 * c4 + 1], p[4 * c4 + 2], p[4 * c4 + 3]); }

__global__ void prelu_fwd_v4k(const float* x, const float* alpha, float* y, long n4) {
    const float a = *alpha;
    V4_LOOP(i, n4) {
        float4 v = ldv(x, i);
        v.x = v.x > 0.f ? v.x : a * v.x; v.y = v.y > 0.f ? v.y : a * v.y;
        v.z = v.z > 0.f ? v.z : a * v.z; v.w = v.w > 0.f ? v.w : a * v.w;
        stv(y, i, v);
    }
}
__global__ void prelu_bwd_v4k(const float* x, const float* dy, const float* alpha, float* dx, double* gpart, long n4) {
    __shared__ double sh[4];
    const float a = *alpha;
    float s = 0.f;
    V4_LOOP(i, n4) {
        const float4 v = ldv(x, i), g = ldv(dy, i);
        float4 o;
        o.x = v.x > 0.f ? g.x : a * g.x; o.y = v.y > 0.f ? g.y : a * g.y;
        o.z = v.z > 0.f ? g.z : a * g.z; o.w = v.w > 0.f ? g.w : a * g.w;
        stv(dx, i, o);
        s += (v.x <= 0.f ? v.x * g.x : 0.f) + (v.y <= 0.f ? v.y * g.y : 0.f) + (v.z <= 0.f ? v.z * g.z : 0.f) +
             (v.w <= 0.f ? v.w * g.w : 0.f);
    }
    if (!gpart) return;
    const double t = block_sum_256((double)s, sh);
    if (threadIdx.x == 0) gpart[blockIdx.x] = t;
}
__global__ void lrelu_fwd_v4k(const float* x, float* y, float s, long n4) {
    V4_LOOP(i, n4) {
        float4 v = ldv(x, i);
        v.x = v.x >= 0.f ? v.x : s * v.x; v.y = v.y >= 0.f ? v.y : s * v.y;
        v.z = v.z >= 0.f ? v.z : s * v.z; v.w = v.w >= 0.f ? v.w : s * v.w;
        stv(y, i, v);
    }
}
__global__ void lrelu_bwd_v4k(const float* x, const float* dy, float* dx, float s, long n4) {
    V4_LOOP(i, n4) {
        const float4 v = ldv(x, i);
        float4 g = ldv(dy, i);
        g.x = v.x >= 0.f ? g.x : s * g.x; g.y = v.y >= 0.f ? g.y : s * g.y;
        g.z = v.z >= 0.f ? g.z : s * g.z; g.w = v.w >= 0.f ? g.w : s * g.w;
        stv(dx, i, g);
    }
}
__global__ void add_v4k(const float* a, const float* b, float* o, long n4) {
    V4_LOOP(i, n4) {
        const float4 u = ldv(a, i), v = ldv(b, i);
        stv(o, i, make_float4(u.x + v.x, u.y + v.y, u.z + v.z, u.w + v.w));
    }
}
__global__ void axpy_v4k(float al, const float* x, float* y, long n4) {
    V4_LOOP(i, n4) {
        const float4 u = ldv(x, i);
        float4 v = ldv(y, i);
        v.x += al * u.x; v.y += al * u.y; v.z += al * u.z; v.w += al * u.w;
        stv(y, i, v);
    }
}
// spatial dropout with C % 4 == 0: mask[n][c..c+3] is one float4
__global__ void mask_mul_sp_v4k(const float* x, const float* mask, float* y, long n4, long HWC4, int C4) {
    V4_LOOP(i, n4) {
        const long n = i / HWC4;
        const float4 m = ldc(mask, (int)(n * C4 + (i % C4)));
        float4 v = ldv(x, i);
        v.x *= m.x; v.y *= m.y; v.z *= m.z; v.w *= m.w;
        stv(y, i, v);
    }
}
__global__ void mask_mul_v4k(const float* x, const float* mask, float* y, long n4) {
    V4_LOOP(i, n4) {
        const float4 m = ldv(mask, i);
        float4 v = ldv(x, i);
        v.x *= m.x; v.y *= m.y; v.z *= m.z; v.w *= m.w;
        stv(y, i, v);
    }
}
// batch-norm apply / backward with C % 4 == 0
__global__ void bn_apply_v4k(const float* x, float* y, const float* gamma, const float* beta, const float* mean,
                             const float* invstd, long n4, int C4) {
    V4_LOOP(i, n4) {
        const int c = (int)(i % C4);
        const float4 g = ldc(gamma, c), b = ldc(beta, c), m = ldc(mean, c), s = ldc(invstd, c);
        float4 v = ldv(x, i);
        v.x = (v.x - m.x) * s.x * g.x + b.x; v.y = (v.y - m.y) * s.y * g.y + b.y;
        v.z = (v.z - m.z) * s.z * g.z + b.z; v.w = (v.w - m.w) * s.w * g.w + b.w;
        stv(y, i, v);
    }
}
__global__ void bn_bwd_v4k(const float* x, const float* dy, const float* gamma, const float* mean, const float* invstd,
                           const double* sums, double count, long n4, int C4, float* dx) {
    V4_LOOP(i, n4) {
        const int c = (int)(i % C4);
        const float4 g = ldc(gamma, c), m = ldc(mean, c), s = ldc(invstd, c);
        const float4 v = ldv(x, i), d = ldv(dy, i);
        const int C = C4 * 4;
        float4 o;
#define BNB(f, k)                                                                      \
        {                                                                              \
            const float xh = (v.f - m.f) * s.f;                                        \
            const float m1 = (float)(sums[4 * c + k] / count), m2 = (float)(sums[C + 4 * c + k] / count); \
            o.f = g.f * s.f * (d.f - m1 - xh * m2);                                    \
        }
        BNB(x, 0) BNB(y, 1) BNB(z, 2) BNB(w, 3)
#undef BNB
        stv(dx, i, o);
    }
}

__global__ void prelu_fwd_k(const float* x, const float* alpha, float* y, long n) {
    const float a = *alpha;
    GRID_STRIDE(i, n) {
        const float v = x[i];
        y[i] = v > 0.f ? v : a * v;
    }
}
__global__ void prelu_bwd_k(const float* x, const float* dy, const float* alpha, float* dx, double* gpart, long n) {
    __shared__ double sh[4];
    const float a = *alpha;
    float s = 0.f;
    GRID_STRIDE(i, n) {
        const float v = x[i], g = dy[i];
        dx[i] = v > 0.f ? g : a * g;
        if (v <= 0.f) s += v * g;
    }
    if (!gpart) return;
    const double t = block_sum_256((double)s, sh);
    if (threadIdx.x == 0) gpart[blockIdx.x] = t;
}
// *galpha += scale * sum of the per-workgroup partials, in a fixed order (one same-address float atomic per
// workgroup costs ~12 ns each on this part: 25 us for a 2048-workgroup launch, and is order-dependent)
__global__ void prelu_galpha_reduce_k(const double* gpart, int nparts, float* galpha, float scale) {
    __shared__ double sh[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 256) s += gpart[i];
    const double t = block_sum_256(s, sh);
    if (threadIdx.x == 0) *galpha += (float)(t * scale);
}
__global__ void lrelu_fwd_k(const float* x, float* y, float s, long n) {
    GRID_STRIDE(i, n) {
        const float v = x[i];
        y[i] = v >= 0.f ? v : s * v;
    }
}
__global__ void lrelu_bwd_k(const float* x, const float* dy, float* dx, float s, long n) {
    GRID_STRIDE(i, n) { dx[i] = x[i] >= 0.f ? dy[i] : s * dy[i]; }
}
__global__ void sigmoid_fwd_k(const float* x, float* y, long n) {
    GRID_STRIDE(i, n) { y[i] = 1.f / (1.f + expf(-x[i])); }
}
__global__ void sigmoid_bwd_k(const float* y, const float* dy, float* dx, long n) {
    GRID_STRIDE(i, n) {
        const float v = y[i];
        dx[i] = dy[i] * (1.f - v) * v;
    }
}

// ------------------------------------------------------------------------ BCE
__global__ void bce_fwd_k(const float* p, const float* t, float* loss, long n) {
    // single block: n is the batch size
    __shared__ double sh[4];
    double s = 0.0;
    for (long i = threadIdx.x; i < n; i += blockDim.x) {
        const float x = p[i], y = t[i];
        s -= (double)(logf(x + 1e-12f) * y + logf(1.f - x + 1e-12f) * (1.f - y));
    }
    const double tot = block_sum_256(s, sh);
    if (threadIdx.x == 0) *loss = (float)(tot / (double)n);
}
__global__ void bce_bwd_k(const float* p, const float* t, float* dp, long n) {
    const float norm = 1.f / (float)n;
    GRID_STRIDE(i, n) {
        const float x = p[i], y = t[i];
        dp[i] = -norm * (y - x) / ((1.f - x + 1e-12f) * (x + 1e-12f));
    }
}

// ------------------------------------------------------------- column reduces
// x: [M][C].  grid = (ceil(C/64), row chunks); block = 64 channels x 4 row lanes.
// MODE 0: (x, x^2)   MODE 1: (dy, dy*xhat)   MODE 2: (dy, 0)
template <int MODE>
__global__ __launch_bounds__(256) void colreduce_k(const float* x, const float* dy, const float* mean,
                                                   const float* invstd, long M, int C, long rows_per_block,
                                                   double* sums, cg::ColTree tree) {
    double* part = tree.part;
    __shared__ double sh1[4][64], sh2[4][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const long r0 = (long)blockIdx.y * rows_per_block;
    const long r1 = min(M, r0 + rows_per_block);
    double a1 = 0.0, a2 = 0.0;
    if (c < C) {
        float s1 = 0.f, s2 = 0.f;
        float mu = 0.f, is = 0.f;
        if (MODE == 1) { mu = mean[c]; is = invstd[c]; }
        int cnt = 0;
        for (long r = r0 + rl; r < r1; r += 4) {
            const long i = r * C + c;
            if (MODE == 0) {
                const float v = x[i];
                s1 += v; s2 += v * v;
            } else if (MODE == 1) {
                const float g = dy[i];
                s1 += g; s2 += g * ((x[i] - mu) * is);
            } else {
                s1 += dy[i];
            }
            if (++cnt == 64) {  // bound fp32 partial length
                a1 += s1; a2 += s2; s1 = 0.f; s2 = 0.f; cnt = 0;
            }
        }
        a1 += s1; a2 += s2;
    }
    sh1[rl][cl] = a1; sh2[rl][cl] = a2;
    __syncthreads();
    if (rl == 0 && c < C) {
        const double t1 = sh1[0][cl] + sh1[1][cl] + sh1[2][cl] + sh1[3][cl];
        const double t2 = sh2[0][cl] + sh2[1][cl] + sh2[2][cl] + sh2[3][cl];
        const int nk = (MODE == 2 ? 1 : 2) * C;
        cg::st_agent(&part[(size_t)blockIdx.y * nk + c], t1);
        if (MODE != 2) cg::st_agent(&part[(size_t)blockIdx.y * nk + C + c], t2);
    }
    cg::col_tree_finish(tree, (int)blockIdx.y, (int)gridDim.y, gridDim.x, (MODE == 2 ? 1 : 2) * C, sums);
}

// float4 form (C % 4 == 0, 16-byte aligned rows): block = QB channel quads x RL row lanes (QB*RL = 256), four
// independent row loads in flight per thread, fp32 partials of <= 64 rows folded into doubles.
template <int MODE, int QB>
__global__ __launch_bounds__(256) void colreduce4_k(const float* x, const float* dy, const float* mean,
                                                    const float* invstd, long M, int C, long rows_per_block,
                                                    double* sums, cg::ColTree tree) {
    constexpr int RL = 256 / QB;
    double* part = tree.part;
    __shared__ double sh[2][RL][QB * 4 + 2];
    const int ql = threadIdx.x % QB, rl = threadIdx.x / QB;
    const int q = blockIdx.x * QB + ql;          // channel quad
    const int cq = C >> 2;
    const long r0 = (long)blockIdx.y * rows_per_block;
    const long r1 = min(M, r0 + rows_per_block);
    double a1[4] = {0.0, 0.0, 0.0, 0.0}, a2[4] = {0.0, 0.0, 0.0, 0.0};
    if (q < cq) {
        float4 mu = make_float4(0.f, 0.f, 0.f, 0.f), is = mu;
        if (MODE == 1) { mu = ldc(mean, q); is = ldc(invstd, q); }
        for (long rb = r0 + rl; rb < r1; rb += 64L * RL) {
            const long re = min(r1, rb + 64L * RL);
            float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
#pragma unroll 4
            for (long r = rb; r < re; r += RL) {
                const long i = r * cq + q;
                if (MODE == 0) {
                    const float4 v = ldv(x, i);
                    s1.x += v.x; s1.y += v.y; s1.z += v.z; s1.w += v.w;
                    s2.x += v.x * v.x; s2.y += v.y * v.y; s2.z += v.z * v.z; s2.w += v.w * v.w;
                } else if (MODE == 1) {
                    const float4 g = ldv(dy, i), v = ldv(x, i);
                    s1.x += g.x; s1.y += g.y; s1.z += g.z; s1.w += g.w;
                    s2.x += g.x * ((v.x - mu.x) * is.x); s2.y += g.y * ((v.y - mu.y) * is.y);
                    s2.z += g.z * ((v.z - mu.z) * is.z); s2.w += g.w * ((v.w - mu.w) * is.w);
                } else {
                    const float4 g = ldv(dy, i);
                    s1.x += g.x; s1.y += g.y; s1.z += g.z; s1.w += g.w;
                }
            }
            a1[0] += s1.x; a1[1] += s1.y; a1[2] += s1.z; a1[3] += s1.w;
            a2[0] += s2.x; a2[1] += s2.y; a2[2] += s2.z; a2[3] += s2.w;
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) { sh[0][rl][ql * 4 + k] = a1[k]; sh[1][rl][ql * 4 + k] = a2[k]; }
    __syncthreads();
    // 256 threads finish QB*4 channels x 2 sums
    for (int idx = threadIdx.x; idx < QB * 4 * (MODE == 2 ? 1 : 2); idx += 256) {
        const int which = idx / (QB * 4), cl = idx % (QB * 4);
        const int c = blockIdx.x * QB * 4 + cl;
        if (c >= C) continue;
        double t = 0.0;
#pragma unroll
        for (int r = 0; r < RL; ++r) t += sh[which][r][cl];
        cg::st_agent(&part[(size_t)blockIdx.y * ((MODE == 2 ? 1 : 2) * C) + which * C + c], t);
    }
    cg::col_tree_finish(tree, (int)blockIdx.y, (int)gridDim.y, gridDim.x, (MODE == 2 ? 1 : 2) * C, sums);
}

__global__ void bias_grad_finish_k(const double* sums, float* gb, int C, float scale) {
    GRID_STRIDE(c, C) { gb[c] += scale * (float)sums[c]; }
}

// --------------------------------------------------------------- batch-norm
__global__ void bn_prepare_k(const double* sums, double count, int C, float eps, float momentum, float* running_mean,
                             float* running_var, float* save_mean, float* save_invstd) {
    GRID_STRIDE(c, C) {
        const double mean = sums[c] / count;
        double var = sums[C + c] / count - mean * mean;
        if (var < 0.0) var = 0.0;
        const float invstd = (float)(1.0 / sqrt(var + (double)eps));
        save_mean[c] = (float)mean;
        save_invstd[c] = invstd;
        if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
        if (running_var) {
            const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
        }
    }
}
__global__ void bn_apply_k(const float* x, float* y, const float* gamma, const float* beta, const float* mean,
                           const float* invstd, long total, int C) {
    GRID_STRIDE(i, total) {
        const int c = (int)(i % C);
        y[i] = (x[i] - mean[c]) * invstd[c] * gamma[c] + beta[c];
    }
}
__global__ void bn_eval_k(const float* x, float* y, const float* gamma, const float* beta, const float* rm,
                          const float* rv, long total, int C, float eps) {
    GRID_STRIDE(i, total) {
        const int c = (int)(i % C);
        y[i] = (x[i] - rm[c]) / sqrtf(rv[c] + eps) * gamma[c] + beta[c];
    }
}
__global__ void bn_bwd_k(const float* x, const float* dy, const float* gamma, const float* mean, const float* invstd,
                         const double* sums, double count, long total, int C, float* dx) {
    GRID_STRIDE(i, total) {
        const int c = (int)(i % C);
        const float is = invstd[c];
        const float xh = (x[i] - mean[c]) * is;
        const float m1 = (float)(sums[c] / count);
        const float m2 = (float)(sums[C + c] / count);
        dx[i] = gamma[c] * is * (dy[i] - m1 - xh * m2);
    }
}
__global__ void bn_bwd_param_k(const double* local_sums, int C, float* ggamma, float* gbeta, float scale) {
    GRID_STRIDE(c, C) {
        if (gbeta) gbeta[c] += scale * (float)local_sums[c];
        if (ggamma) ggamma[c] += scale * (float)local_sums[C + c];
    }
}

// ------------------------------------------------------- resampling / pooling
__global__ void ups2_fwd_k(const float* x, float* y, int N, int H, int W, int C) {
    const long total = (long)N * 2 * H * 2 * W * C;
    GRID_STRIDE(i, total) {
        const int c = (int)(i % C);
        long r = i / C;
        const int ox = (int)(r % (2 * W)); r /= (2 * W);
        const int oy = (int)(r % (2 * H));
        const long n = r / (2 * H);
        y[i] = x[((n * H + (oy >> 1)) * W + (ox >> 1)) * C + c];
    }
}
__global__ void ups2_bwd_k(const float* dy, float* dx, int N, int H, int W, int C) {
    const long total = (long)N * H * W * C;
    GRID_STRIDE(i, total) {
        const int c = (int)(i % C);
        long r = i / C;
        const int xx = (int)(r % W); r /= W;
        const int yy = (int)(r % H);
        const long n = r / H;
        const long W2 = 2 * W;
        const long b = ((n * 2 * H + 2 * yy) * W2 + 2 * xx) * C + c;
        dx[i] = (dy[b] + dy[b + C]) + (dy[b + W2 * C] + dy[b + W2 * C + C]);
    }
}
template <bool MAX>
__global__ void pool2_fwd_k(const float* x, float* y, int N, int H, int W, int C) {
    const int Ho = H / 2, Wo = W / 2;
    const long total = (long)N * Ho * Wo * C;
    GRID_STRIDE(i, total) {
        const int c = (int)(i % C);
        long r = i / C;
        const int ox = (int)(r % Wo); r /= Wo;
        const int oy = (int)(r % Ho);
        const long n = r / Ho;
        const long b = ((n * H + 2 * oy) * W + 2 * ox) * C + c;
        const float v00 = x[b], v01 = x[b + C], v10 = x[b + (long)W * C], v11 = x[b + (long)W * C + C];
        if (MAX) {
            float m = v00;
            if (v01 > m) m = v01;
            if (v10 > m) m = v10;
            if (v11 > m) m = v11;
            y[i] = m;
        } else {
            y[i] = (v00 + v01 + v10 + v11) * 0.25f;
        }
    }
}
__global__ void avgpool2_bwd_k(const float* dy, float* dx, int N, int H, int W, int C) {
    const int Ho = H / 2, Wo = W / 2;
    const long total = (long)N * H * W * C;
    GRID_STRIDE(i, total) {
        const int c = (int)(i % C);
        long r = i / C;
        const int xx = (int)(r % W); r /= W;
        const int yy = (int)(r % H);
        const long n = r / H;
        float v = 0.f;
        if ((yy >> 1) < Ho && (xx >> 1) < Wo) v = dy[((n * Ho + (yy >> 1)) * Wo + (xx >> 1)) * C + c] * 0.25f;
        dx[i] = v;
    }
}
__global__ void maxpool2_bwd_k(const float* x, const float* dy, float* dx, int N, int H, int W, int C) {
    // one thread per pooled output: route the gradient to the first max in scan order
    const int Ho = H / 2, Wo = W / 2;
    const long total = (long)N * Ho * Wo * C;
    GRID_STRIDE(i, total) {
        const int c = (int)(i % C);
        long r = i / C;
        const int ox = (int)(r % Wo); r /= Wo;
        const int oy = (int)(r % Ho);
        const long n = r / Ho;
        const long b = ((n * H + 2 * oy) * W + 2 * ox) * C + c;
        const long o01 = C, o10 = (long)W * C, o11 = (long)W * C + C;
        const float v00 = x[b], v01 = x[b + o01], v10 = x[b + o10], v11 = x[b + o11];
        int arg = 0; float m = v00;
        if (v01 > m) { m = v01; arg = 1; }
        if (v10 > m) { m = v10; arg = 2; }
        if (v11 > m) { m = v11; arg = 3; }
        const float g = dy[i];
        dx[b] = arg == 0 ? g : 0.f;
        dx[b + o01] = arg == 1 ? g : 0.f;
        dx[b + o10] = arg == 2 ? g : 0.f;
        dx[b + o11] = arg == 3 ? g : 0.f;
    }
}

// -------------------------------------------------------------------- dropout
__global__ void mask_mul_k(const float* x, const float* mask, float* y, long total, long HWC, int C, int spatial) {
    GRID_STRIDE(i, total) {
        float m;
        if (spatial) {
            const long n = i / HWC;
            m = mask[n * C + (i % C)];
        } else {
            m = mask[i];
        }
        y[i] = x[i] * m;
    }
}
using cg::u01;
__global__ void rng_bernoulli_k(float* out, long n, float keep, float value, uint64_t seed, uint64_t offset,
                                const uint64_t* base) {
    if (base) offset += *base;
    GRID_STRIDE(i, n) { out[i] = u01(seed, offset + (uint64_t)i) < keep ? value : 0.f; }
}
__global__ void rng_uniform_k(float* out, long n, float lo, float hi, uint64_t seed, uint64_t offset,
                              const uint64_t* base) {
    if (base) offset += *base;
    GRID_STRIDE(i, n) { out[i] = lo + (hi - lo) * u01(seed, offset + (uint64_t)i); }
}
__global__ void rng_randint_k(int32_t* out, long n, int32_t range, uint64_t seed, uint64_t offset, const uint64_t* base) {
    if (base) offset += *base;
    GRID_STRIDE(i, n) {
        int32_t v = (int32_t)(u01(seed, offset + (uint64_t)i) * (float)range);
        out[i] = v < range ? v : range - 1;
    }
}
// G consecutive blocks of n floats, block g drawn at its own position off<g> of the counter stream (the dropout masks of
// D32_st3's identical branches, which a branch-after-branch walk would draw at different positions)
__global__ void rng_bernoulli_groups_k(float* out, long n, int G, float keep, float value, uint64_t seed, uint64_t off0,
                                       uint64_t off1, uint64_t off2, uint64_t off3, const uint64_t* base) {
    const uint64_t b = base ? *base : 0ull;
    GRID_STRIDE(i, n * G) {
        const int g = (int)(i / n);
        const uint64_t off = g == 0 ? off0 : (g == 1 ? off1 : (g == 2 ? off2 : off3));
        out[i] = u01(seed, b + off + (uint64_t)(i - (long)g * n)) < keep ? value : 0.f;
    }
}
__global__ void counter_add_k(uint64_t* c, uint64_t d) { if (threadIdx.x == 0 && blockIdx.x == 0) *c += d; }

// --------------------------------------------------------------------- layout
__global__ void nchw_to_nhwc_k(const float* in, float* out, int N, int C, int H, int W) {
    const long total = (long)N * C * H * W;
    GRID_STRIDE(i, total) {  // i indexes the NHWC output
        const int c = (int)(i % C);
        long r = i / C;
        const int x = (int)(r % W); r /= W;
        const int y = (int)(r % H);
        const long n = r / H;
        out[i] = in[((n * C + c) * H + y) * W + x];
    }
}
__global__ void nhwc_to_nchw_k(const float* in, float* out, int N, int C, int H, int W) {
    const long total = (long)N * C * H * W;
    GRID_STRIDE(i, total) {  // i indexes the NCHW output
        const int x = (int)(i % W);
        long r = i / W;
        const int y = (int)(r % H); r /= H;
        const int c = (int)(r % C);
        const long n = r / C;
        out[i] = in[((n * H + y) * W + x) * C + c];
    }
}
__global__ void copy_channels_k(const float* src, float* dst, long M, int Csrc, int so, int Cdst, int dof, int Cc) {
    const long total = M * Cc;
    GRID_STRIDE(i, total) {
        const int c = (int)(i % Cc);
        const long m = i / Cc;
        dst[m * Cdst + dof + c] = src[m * Csrc + so + c];
    }
}
__global__ void gather_rows_k(const float* src, const int32_t* idx, float* dst, long nrows, long rowlen) {
    const long total = nrows * rowlen;
    GRID_STRIDE(i, total) {
        const long r = i / rowlen, j = i - r * rowlen;
        dst[i] = src[(long)idx[r] * rowlen + j];
    }
}
__global__ void fill_k(float* x, float v, long n) { GRID_STRIDE(i, n) x[i] = v; }
__global__ void add_k(const float* a, const float* b, float* o, long n) { GRID_STRIDE(i, n) o[i] = a[i] + b[i]; }
__global__ void axpy_k(float al, const float* x, float* y, long n) { GRID_STRIDE(i, n) y[i] += al * x[i]; }
__global__ void axpy_sign_k(float al, const float* x, float* y, long n) {
    GRID_STRIDE(i, n) {
        const float v = x[i];
        y[i] += al * (v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f));
    }
}
__global__ void scale_k(float* x, float al, long n) { GRID_STRIDE(i, n) x[i] *= al; }
__global__ void clamp_k(float* x, float lo, float hi, long n) {
    GRID_STRIDE(i, n) { x[i] = fminf(fmaxf(x[i], lo), hi); }
}
template <bool SQ>
__global__ void sumred_k(const float* x, long n, double* out) {
    __shared__ double sh[4];
    double s = 0.0;
    GRID_STRIDE(i, n) {
        const float v = x[i];
        s += SQ ? (double)v * (double)v : (double)fabsf(v);
    }
    const double t = block_sum_256(s, sh);
    if (threadIdx.x == 0) atomicAdd(out, t);
}

// -------------------------------------------------------- spatial transformer
// T = R(theta) * S(s) * Tr(tx,ty); rows: [c*s, -sn*s, c*s*tx - sn*s*ty], [sn*s, c*s, sn*s*tx + c*s*ty]
__global__ void affine_matrix_fwd_k(const float* params, float* T, int N, int ur, int us, int ut) {
    const int P = ur + us + 2 * ut;
    GRID_STRIDE(n, N) {
        const float* p = params + (long)n * P;
        int k = 0;
        float th = 0.f, sc = 1.f, tx = 0.f, ty = 0.f;
        if (ur) th = p[k++];
        if (us) sc = p[k++];
        if (ut) { tx = p[k]; ty = p[k + 1]; }
        const float c = cosf(th), s = sinf(th);
        float* t = T + (long)n * 6;
        t[0] = c * sc;  t[1] = -s * sc; t[2] = c * sc * tx - s * sc * ty;
        t[3] = s * sc;  t[4] = c * sc;  t[5] = s * sc * tx + c * sc * ty;
    }
}
__global__ void affine_matrix_bwd_k(const float* params, const float* gT, float* gparams, int N, int ur, int us, int ut) {
    const int P = ur + us + 2 * ut;
    GRID_STRIDE(n, N) {
        const float* p = params + (long)n * P;
        int k = 0;
        float th = 0.f, sc = 1.f, tx = 0.f, ty = 0.f;
        if (ur) th = p[k++];
        if (us) sc = p[k++];
        if (ut) { tx = p[k]; ty = p[k + 1]; }
        const float c = cosf(th), s = sinf(th);
        const float* g = gT + (long)n * 6;
        float* gp = gparams + (long)n * P;
        k = 0;
        if (ur) {
            // d/dtheta: c -> -s, s -> c
            const float d0 = -s * sc, d1 = -c * sc, d2 = -s * sc * tx - c * sc * ty;
            const float d3 = c * sc, d4 = -s * sc, d5 = c * sc * tx - s * sc * ty;
            gp[k++] = g[0] * d0 + g[1] * d1 + g[2] * d2 + g[3] * d3 + g[4] * d4 + g[5] * d5;
        }
        if (us) {
            gp[k++] = g[0] * c + g[1] * (-s) + g[2] * (c * tx - s * ty) + g[3] * s + g[4] * c + g[5] * (s * tx + c * ty);
        }
        if (ut) {
            gp[k] = g[2] * (c * sc) + g[5] * (s * sc);
            gp[k + 1] = g[2] * (-s * sc) + g[5] * (c * sc);
        }
    }
}
__global__ void affine_grid_fwd_k(const float* T, float* grid, int N, int H, int W) {
    const long total = (long)N * H * W;
    GRID_STRIDE(i, total) {
        const int j = (int)(i % W);
        long r = i / W;
        const int ii = (int)(r % H);
        const long n = r / H;
        const float y = H > 1 ? -1.f + 2.f * (float)ii / (float)(H - 1) : -1.f;
        const float x = W > 1 ? -1.f + 2.f * (float)j / (float)(W - 1) : -1.f;
        const float* t = T + n * 6;
        grid[i * 2 + 0] = t[0] * y + t[1] * x + t[2];
        grid[i * 2 + 1] = t[3] * y + t[4] * x + t[5];
    }
}
// one block per sample: gT[n][r][:] = sum_{i,j} ggrid[n,i,j,r] * (y_i, x_j, 1)
__global__ __launch_bounds__(256) void affine_grid_bwd_k(const float* ggrid, float* gT, int N, int H, int W) {
    __shared__ double sh[4];
    const int n = blockIdx.x;
    double acc[6] = {0, 0, 0, 0, 0, 0};
    const int HW = H * W;
    for (int p = threadIdx.x; p < HW; p += blockDim.x) {
        const int ii = p / W, j = p - ii * W;
        const float y = H > 1 ? -1.f + 2.f * (float)ii / (float)(H - 1) : -1.f;
        const float x = W > 1 ? -1.f + 2.f * (float)j / (float)(W - 1) : -1.f;
        const float g0 = ggrid[((long)n * HW + p) * 2 + 0];
        const float g1 = ggrid[((long)n * HW + p) * 2 + 1];
        acc[0] += g0 * y; acc[1] += g0 * x; acc[2] += g0;
        acc[3] += g1 * y; acc[4] += g1 * x; acc[5] += g1;
    }
    for (int k = 0; k < 6; ++k) {
        const double t = block_sum_256(acc[k], sh);
        if (threadIdx.x == 0) gT[(long)n * 6 + k] = (float)t;
    }
}

struct BilinTaps {
    int y0, x0;
    float wy0, wx0;  // weight of the top / left tap
    bool in00, in01, in10, in11;
};
__device__ __forceinline__ BilinTaps bilin_taps(float yf, float xf, int Hi, int Wi) {
    BilinTaps t;
    const float xc = (xf + 1.f) * (float)(Wi - 1) * 0.5f;
    const float yc = (yf + 1.f) * (float)(Hi - 1) * 0.5f;
    const float x0f = floorf(xc), y0f = floorf(yc);
    t.x0 = (int)x0f; t.y0 = (int)y0f;
    t.wx0 = 1.f - (xc - x0f);
    t.wy0 = 1.f - (yc - y0f);
    const bool xin0 = t.x0 >= 0 && t.x0 <= Wi - 1, xin1 = t.x0 + 1 >= 0 && t.x0 + 1 <= Wi - 1;
    const bool yin0 = t.y0 >= 0 && t.y0 <= Hi - 1, yin1 = t.y0 + 1 >= 0 && t.y0 + 1 <= Hi - 1;
    t.in00 = yin0 && xin0; t.in01 = yin0 && xin1; t.in10 = yin1 && xin0; t.in11 = yin1 && xin1;
    return t;
}
// Nimg: the image batch; sample n of the grid / output batch reads image n % Nimg (cg_bilinear_sampler_*_shared: G sibling
// transformers sampling the SAME images with their own grids, as one launch over G * Nimg samples)
__global__ void bilinear_fwd_k(const float* img, const float* grid, float* out, int N, int Nimg, int Hi, int Wi, int C, int Ho,
                               int Wo) {
    const long total = (long)N * Ho * Wo * C;
    GRID_STRIDE(i, total) {
        const int c = (int)(i % C);
        const long pix = i / C;
        const long n = (pix / ((long)Ho * Wo)) % Nimg;
        const BilinTaps t = bilin_taps(grid[pix * 2], grid[pix * 2 + 1], Hi, Wi);
        const float* b = img + (((n * Hi + t.y0) * (long)Wi + t.x0)) * C + c;
        float v = 0.f;
        if (t.in00) v += t.wx0 * t.wy0 * b[0];
        if (t.in01) v += (1.f - t.wx0) * t.wy0 * b[C];
        if (t.in10) v += t.wx0 * (1.f - t.wy0) * b[(long)Wi * C];
        if (t.in11) v += (1.f - t.wx0) * (1.f - t.wy0) * b[(long)Wi * C + C];
        out[i] = v;
    }
}
// the same with one lane per channel QUAD (C % 4 == 0, 16-B aligned tensors): the taps are derived once per four outputs and
// every access moves 16 B; per channel the operations and their order are those of the scalar kernel (bit-identical)
__global__ __launch_bounds__(256) void bilinear_fwd_v4k(const float* __restrict__ img, const float* __restrict__ grid,
                                                        float* __restrict__ out, int N, int Nimg, int Hi, int Wi, int C4, int Ho, int Wo) {
    const long total = (long)N * Ho * Wo * C4;
    GRID_STRIDE(i, total) {
        const int c = (int)(i % C4);
        const long pix = i / C4;
        const long n = (pix / ((long)Ho * Wo)) % Nimg;
        const float2 gyx = *reinterpret_cast<const float2*>(grid + pix * 2);
        const BilinTaps t = bilin_taps(gyx.x, gyx.y, Hi, Wi);
        const long b = (((n * Hi + t.y0) * (long)Wi + t.x0)) * C4 + c;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        auto tap = [&](bool in, float w, long off) {
            if (in) { const float4 q = ldv(img, b + off); v.x += w * q.x; v.y += w * q.y; v.z += w * q.z; v.w += w * q.w; }
        };
        tap(t.in00, t.wx0 * t.wy0, 0);
        tap(t.in01, (1.f - t.wx0) * t.wy0, C4);
        tap(t.in10, t.wx0 * (1.f - t.wy0), (long)Wi * C4);
        tap(t.in11, (1.f - t.wx0) * (1.f - t.wy0), (long)Wi * C4 + C4);
        stv(out, i, v);
    }
}
// G lanes per output pixel (64/G pixels per wave), lanes stride the channels; G = smallest power of two >= min(C, 64)
template <int G>
__global__ __launch_bounds__(256) void bilinear_bwd_k(const float* img, const float* grid, const float* gout,
                                                      float* gimg, float* ggrid, int N, int Nimg, int Hi, int Wi, int C, int Ho,
                                                      int Wo) {
    const long npix = (long)N * Ho * Wo;
    const int sub = threadIdx.x & (G - 1);
    const long grp0 = (blockIdx.x * (long)blockDim.x + threadIdx.x) / G;
    const long ngrp = ((long)gridDim.x * blockDim.x) / G;
    const long iters = (npix + ngrp - 1) / ngrp;   // same trip count for every lane of a wave (shuffles below)
    for (long it = 0; it < iters; ++it) {
        const long pix = grp0 + it * ngrp;
        const bool live = pix < npix;
        BilinTaps t;
        t.in00 = t.in01 = t.in10 = t.in11 = false; t.x0 = t.y0 = 0; t.wx0 = t.wy0 = 0.f;
        long n = 0;
        if (live) {
            n = pix / ((long)Ho * Wo);
            t = bilin_taps(grid[pix * 2], grid[pix * 2 + 1], Hi, Wi);
        }
        const long b = ((n * Hi + t.y0) * (long)Wi + t.x0) * C;
        const float* im = img + ((n % Nimg) - n) * (long)Hi * Wi * C;   // the image this sample reads (its own unless shared)
        const long o01 = C, o10 = (long)Wi * C, o11 = (long)Wi * C + C;
        float d00 = 0.f, d01 = 0.f, d10 = 0.f, d11 = 0.f;
        if (live)
            for (int c = sub; c < C; c += G) {
                const float g = gout[pix * C + c];
                if (t.in00) { d00 += im[b + c] * g;       atomicAdd(&gimg[b + c], t.wx0 * t.wy0 * g); }
                if (t.in01) { d01 += im[b + o01 + c] * g; atomicAdd(&gimg[b + o01 + c], (1.f - t.wx0) * t.wy0 * g); }
                if (t.in10) { d10 += im[b + o10 + c] * g; atomicAdd(&gimg[b + o10 + c], t.wx0 * (1.f - t.wy0) * g); }
                if (t.in11) { d11 += im[b + o11 + c] * g; atomicAdd(&gimg[b + o11 + c], (1.f - t.wx0) * (1.f - t.wy0) * g); }
            }
#pragma unroll
        for (int off = G >> 1; off > 0; off >>= 1) {
            d00 += __shfl_down(d00, off, G); d01 += __shfl_down(d01, off, G);
            d10 += __shfl_down(d10, off, G); d11 += __shfl_down(d11, off, G);
        }
        if (live && sub == 0) {
            const float gy = -t.wx0 * d00 + t.wx0 * d10 - (1.f - t.wx0) * d01 + (1.f - t.wx0) * d11;
            const float gx = -t.wy0 * d00 + t.wy0 * d01 - (1.f - t.wy0) * d10 + (1.f - t.wy0) * d11;
            ggrid[pix * 2 + 0] = gy * (float)(Hi - 1) * 0.5f;
            ggrid[pix * 2 + 1] = gx * (float)(Wi - 1) * 0.5f;
        }
    }
}

// Deterministic backward (no float atomics): the reference pins this module to the CPU because stn's GPU scatter was
// "non-reproducible" (models.lua:889-899).  One workgroup = one sample x S pixel slots.  It stages the taps of ALL the
// sample's output pixels in LDS and buckets them by their top-left source cell (y0, x0) - counting sort with integer LDS
// atomics, every bucket then sorted by output-pixel index, so the bucket contents AND their order are reproducible.  Then
//   (A) for its S OUTPUT pixels: ggrid from the channel dot products (G lanes per pixel, sub-wave reduction), and
//   (B) for its S SOURCE pixels: gimg[q] = sum over the output pixels in the four buckets (y-1..y, x-1..x) that cover q,
//       in bucket order - a gather: every gimg element is written exactly once (no memset), in a fixed summation order.
// G = lanes per pixel (smallest power of two >= min(C, 64)); CACC = channels per lane.
// VW = floats per lane access: 4 when C % 4 == 0 and the tensors are 16-B aligned (a wave then covers 64 / G pixels with G = C / 4
// lanes each instead of one pixel per 64 channels), else 1.  A lane owns channels (sub + G * k) * VW .. + VW - 1.
template <int VW> struct BVec;
template <> struct BVec<1> {
    typedef float T;
    static __device__ __forceinline__ T zero() { return 0.f; }
    static __device__ __forceinline__ T ld(const float* p) { return *p; }
    static __device__ __forceinline__ void st(float* p, T v) { *p = v; }
    static __device__ __forceinline__ float dot(T a, T b) { return a * b; }
    static __device__ __forceinline__ void fma(T& acc, float w, T g) { acc += w * g; }
};
template <> struct BVec<4> {
    typedef float4 T;
    static __device__ __forceinline__ T zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
    static __device__ __forceinline__ T ld(const float* p) { return *reinterpret_cast<const float4*>(p); }
    static __device__ __forceinline__ void st(float* p, T v) { *reinterpret_cast<float4*>(p) = v; }
    static __device__ __forceinline__ float dot(T a, T b) { return ((a.x * b.x + a.y * b.y) + a.z * b.z) + a.w * b.w; }
    static __device__ __forceinline__ void fma(T& acc, float w, T g) { acc.x += w * g.x; acc.y += w * g.y; acc.z += w * g.z; acc.w += w * g.w; }
};

template <int G, int CACC, int VW>
__global__ __launch_bounds__(256) void bilinear_bwd_det_k(const float* __restrict__ img, const float* __restrict__ grid,
                                                          const float* __restrict__ gout, float* __restrict__ gimg,
                                                          float* __restrict__ ggrid, int Nimg, int Hi, int Wi, int C, int Ho, int Wo,
                                                          int nchunks, int S) {
    extern __shared__ float bl_sh[];
    const int P = Ho * Wo, Q = Hi * Wi;
    const int CW = Wi + 1, NC = (Hi + 1) * CW;        // buckets: (y0 + 1, x0 + 1), y0 in [-1, Hi-1], x0 in [-1, Wi-1]
    int* ty0 = reinterpret_cast<int*>(bl_sh);
    int* tx0 = ty0 + P;
    float* twy = bl_sh + 2 * P;
    float* twx = bl_sh + 3 * P;
    int* list = reinterpret_cast<int*>(bl_sh) + 4 * P;  // [P] output pixels grouped by bucket
    int* cstart = list + P;                              // [NC + 1]
    int* ccnt = cstart + NC + 1;                         // [NC]
    __shared__ int part[256];
    const long n = blockIdx.x / nchunks;
    const int chunk = blockIdx.x % nchunks;
    const int tid = threadIdx.x;
    for (int c = tid; c < NC; c += 256) ccnt[c] = 0;
    __syncthreads();
    for (int p = tid; p < P; p += 256) {
        const BilinTaps t = bilin_taps(grid[(n * P + p) * 2], grid[(n * P + p) * 2 + 1], Hi, Wi);
        ty0[p] = t.y0; tx0[p] = t.x0; twy[p] = t.wy0; twx[p] = t.wx0;
        if (t.y0 >= -1 && t.y0 <= Hi - 1 && t.x0 >= -1 && t.x0 <= Wi - 1) atomicAdd(&ccnt[(t.y0 + 1) * CW + t.x0 + 1], 1);
    }
    __syncthreads();
    // exclusive scan of the bucket counts: per-thread chunk sums, wave-level prefix sums of the 256 partials (shuffles) plus
    // the three wave totals, chunk-local offsets
    const int per = (NC + 255) / 256;
    int mine = 0;
    for (int c = tid * per; c < min(NC, (tid + 1) * per); ++c) mine += ccnt[c];
    {
        const int lane = tid & 63, wv = tid >> 6;
        int incl = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int up = __shfl_up(incl, off, 64);
            if (lane >= off) incl += up;
        }
        __shared__ int wtot[4];
        if (lane == 63) wtot[wv] = incl;
        __syncthreads();
        int base = 0;
        for (int w = 0; w < wv; ++w) base += wtot[w];
        part[tid] = base + incl - mine;
        if (tid == 255) cstart[NC] = base + incl;
    }
    __syncthreads();
    {
        int run = part[tid];
        for (int c = tid * per; c < min(NC, (tid + 1) * per); ++c) { cstart[c] = run; run += ccnt[c]; ccnt[c] = 0; }
    }
    __syncthreads();
    for (int p = tid; p < P; p += 256) {
        const int y0 = ty0[p], x0 = tx0[p];
        if (y0 >= -1 && y0 <= Hi - 1 && x0 >= -1 && x0 <= Wi - 1) {
            const int c = (y0 + 1) * CW + x0 + 1;
            list[cstart[c] + atomicAdd(&ccnt[c], 1)] = p;    // arrival order is arbitrary ...
        }
    }
    __syncthreads();
    // ... so order every bucket by output-pixel index: an element's place is the number of smaller indices in its bucket (all threads, one
    // element each - a bucket is short unless the transformer collapses, and then P^2 / 256 broadcast reads per thread still bound it;
    // round 6: the insertion sort a single thread ran per bucket took 2-8 ms when every output pixel fell into one cell)
    int* sorted = ccnt + NC;                                 // [P]
    {
        const int placed = cstart[NC];
        for (int i = tid; i < placed; i += 256) {
            const int v = list[i];
            const int c = (ty0[v] + 1) * CW + tx0[v] + 1;
            const int b = cstart[c], e = b + ccnt[c];
            int rank = 0;
            for (int j = b; j < e; ++j) rank += list[j] < v ? 1 : 0;
            sorted[b + rank] = v;
        }
    }
    __syncthreads();
    list = sorted;
    const float* gsample = gout + n * (long)P * C;
    const float* isample = img + (n % Nimg) * (long)Q * C;
    const int sub = tid & (G - 1), grp = tid / G, ngrp = 256 / G;
    // ---- (A) grid gradient of this workgroup's output pixels (every lane of a wave runs the same trip count)
    for (int j0 = 0; j0 < S; j0 += ngrp) {
        const int p = chunk * S + j0 + grp;
        const bool live = j0 + grp < S && p < P;
        float d00 = 0.f, d01 = 0.f, d10 = 0.f, d11 = 0.f;
        float wy0 = 0.f, wx0 = 0.f;
        if (live) {
            const int y0 = ty0[p], x0 = tx0[p];
            wy0 = twy[p]; wx0 = twx[p];
            const bool xin0 = x0 >= 0 && x0 <= Wi - 1, xin1 = x0 + 1 >= 0 && x0 + 1 <= Wi - 1;
            const bool yin0 = y0 >= 0 && y0 <= Hi - 1, yin1 = y0 + 1 >= 0 && y0 + 1 <= Hi - 1;
            const long b = ((long)y0 * Wi + x0) * C;
            const long o01 = C, o10 = (long)Wi * C, o11 = (long)Wi * C + C;
#pragma unroll
            for (int k = 0; k < CACC; ++k) {
                const int c = (sub + G * k) * VW;
                if (c < C) {
                    typedef BVec<VW> V;
                    const typename V::T g = V::ld(gsample + (long)p * C + c);
                    const typename V::T i00 = (yin0 && xin0) ? V::ld(isample + b + c) : V::zero(), i01 = (yin0 && xin1) ? V::ld(isample + b + o01 + c) : V::zero();
                    const typename V::T i10 = (yin1 && xin0) ? V::ld(isample + b + o10 + c) : V::zero(), i11 = (yin1 && xin1) ? V::ld(isample + b + o11 + c) : V::zero();
                    d00 += V::dot(i00, g); d01 += V::dot(i01, g); d10 += V::dot(i10, g); d11 += V::dot(i11, g);
                }
            }
        }
#pragma unroll
        for (int off = G >> 1; off > 0; off >>= 1) {
            d00 += __shfl_down(d00, off, G); d01 += __shfl_down(d01, off, G);
            d10 += __shfl_down(d10, off, G); d11 += __shfl_down(d11, off, G);
        }
        if (live && sub == 0) {
            const float gy = -wx0 * d00 + wx0 * d10 - (1.f - wx0) * d01 + (1.f - wx0) * d11;
            const float gx = -wy0 * d00 + wy0 * d01 - (1.f - wy0) * d10 + (1.f - wy0) * d11;
            ggrid[(n * P + p) * 2 + 0] = gy * (float)(Hi - 1) * 0.5f;
            ggrid[(n * P + p) * 2 + 1] = gx * (float)(Wi - 1) * 0.5f;
        }
    }
    // ---- (B) image gradient of this workgroup's source pixels
    constexpr int kHeavy = 64;                               // more taps than this on one source pixel: the whole workgroup sums them (below)
    __shared__ float coop[256 * CACC * VW];
    bool any_heavy = false;
    for (int j0 = 0; j0 < S; j0 += ngrp) {
        const int q = chunk * S + j0 + grp;
        if (!(j0 + grp < S && q < Q)) continue;
        const int y = q / Wi, x = q - y * Wi;
        // the four buckets whose 2x2 footprint contains (y, x): (y0, x0) = (y-1, x-1), (y-1, x), (y, x-1), (y, x)
        const int c00 = y * CW + x;                      // bucket index of (y0 = y-1, x0 = x-1)
        const int s0 = cstart[c00], n0 = ccnt[c00];
        const int s1 = cstart[c00 + 1], n1 = ccnt[c00 + 1];
        const int s2 = cstart[c00 + CW], n2 = ccnt[c00 + CW];
        const int s3 = cstart[c00 + CW + 1], n3 = ccnt[c00 + CW + 1];
        const int e1 = n0, e2 = n0 + n1, e3 = n0 + n1 + n2, total = e3 + n3;
        if (total > kHeavy) { any_heavy = true; continue; }
        typedef BVec<VW> V;
        typename V::T acc[CACC];
#pragma unroll
        for (int k = 0; k < CACC; ++k) acc[k] = V::zero();
        for (int i0 = 0; i0 < total; i0 += 4) {          // four gradient loads in flight, consumed in bucket order
            int ph[4];
            float w[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u;
                if (i < total) {
                    const int e = i < e1 ? s0 + i : (i < e2 ? s1 + i - e1 : (i < e3 ? s2 + i - e2 : s3 + i - e3));
                    ph[u] = list[e];
                    const float wy = (y == ty0[ph[u]]) ? twy[ph[u]] : 1.f - twy[ph[u]];
                    const float wx = (x == tx0[ph[u]]) ? twx[ph[u]] : 1.f - twx[ph[u]];
                    w[u] = wx * wy;                       // same product order as the forward: (x weight) * (y weight)
                } else {
                    ph[u] = ph[0]; w[u] = 0.f;            // padding: adds an exact +0
                }
            }
            typename V::T g[4][CACC];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int k = 0; k < CACC; ++k) {
                    const int c = (sub + G * k) * VW;
                    g[u][k] = c < C ? V::ld(gsample + (long)ph[u] * C + c) : V::zero();
                }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int k = 0; k < CACC; ++k) V::fma(acc[k], w[u], g[u][k]);
        }
#pragma unroll
        for (int k = 0; k < CACC; ++k) {
            const int c = (sub + G * k) * VW;
            if (c < C) V::st(gimg + (n * Q + q) * C + c, acc[k]);
        }
    }
    // A source pixel under a collapsed transformer (scale -> 0: every output pixel samples the same cell) collects up to P taps; one lane
    // group summing them in a row is what made this kernel 75-750x slower there.  Such pixels are summed by the WHOLE workgroup: lane group r
    // takes the taps r, r + ngrp, ... in bucket order, the ngrp partial sums are added in group order - a fixed order again, so the result
    // is reproducible (it differs from the one-group order by fp32 re-association only, and only where more than kHeavy taps meet).
    if (__syncthreads_or(any_heavy ? 1 : 0)) {
        typedef BVec<VW> V;
        for (int j = 0; j < S; ++j) {
            const int q = chunk * S + j;
            if (q >= Q) break;
            const int y = q / Wi, x = q - y * Wi;
            const int c00 = y * CW + x;
            const int s0 = cstart[c00], n0 = ccnt[c00];
            const int s1 = cstart[c00 + 1], n1 = ccnt[c00 + 1];
            const int s2 = cstart[c00 + CW], n2 = ccnt[c00 + CW];
            const int s3 = cstart[c00 + CW + 1], n3 = ccnt[c00 + CW + 1];
            const int e1 = n0, e2 = n0 + n1, e3 = n0 + n1 + n2, total = e3 + n3;
            if (total <= kHeavy) continue;                // uniform over the workgroup
            typename V::T acc[CACC];
#pragma unroll
            for (int k = 0; k < CACC; ++k) acc[k] = V::zero();
            for (int i0 = grp; i0 < total; i0 += 2 * ngrp) {     // two loads in flight per lane group
                int ph[2];
                float w[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int i = i0 + u * ngrp;
                    if (i < total) {
                        const int e = i < e1 ? s0 + i : (i < e2 ? s1 + i - e1 : (i < e3 ? s2 + i - e2 : s3 + i - e3));
                        ph[u] = list[e];
                        const float wy = (y == ty0[ph[u]]) ? twy[ph[u]] : 1.f - twy[ph[u]];
                        const float wx = (x == tx0[ph[u]]) ? twx[ph[u]] : 1.f - twx[ph[u]];
                        w[u] = wx * wy;
                    } else {
                        ph[u] = ph[0]; w[u] = 0.f;
                    }
                }
                typename V::T g[2][CACC];
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int k = 0; k < CACC; ++k) {
                        const int c = (sub + G * k) * VW;
                        g[u][k] = c < C ? V::ld(gsample + (long)ph[u] * C + c) : V::zero();
                    }
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int k = 0; k < CACC; ++k) V::fma(acc[k], w[u], g[u][k]);
            }
            __syncthreads();                               // the previous heavy pixel's partial sums have been read
#pragma unroll
            for (int k = 0; k < CACC; ++k) V::st(coop + (tid * CACC + k) * VW, acc[k]);
            __syncthreads();
            if (grp == 0) {
                typename V::T tot[CACC];
#pragma unroll
                for (int k = 0; k < CACC; ++k) tot[k] = V::zero();
                for (int r = 0; r < ngrp; ++r)
#pragma unroll
                    for (int k = 0; k < CACC; ++k) V::fma(tot[k], 1.f, V::ld(coop + ((r * G + sub) * CACC + k) * VW));
#pragma unroll
                for (int k = 0; k < CACC; ++k) {
                    const int c = (sub + G * k) * VW;
                    if (c < C) V::st(gimg + (n * Q + q) * C + c, tot[k]);
                }
            }
        }
    }
}

// ------------------------------------------------------------------ optimiser
__global__ void adam_k(float* p, float* g, float* m, float* v, long n, float step, float b1, float b2, float eps,
                       float l1, float l2, float clampv, int write_back, float lr, const uint64_t* t_dev) {
    if (t_dev) {  // replay-safe: bias correction from the device-side step counter
        const double t = (double)*t_dev;
        step = (float)((double)lr * sqrt(1.0 - pow((double)b2, t)) / (1.0 - pow((double)b1, t)));
    }
    GRID_STRIDE(i, n) {
        const float pi = p[i];
        float gi = g[i];
        if (l1 != 0.f || l2 != 0.f) {
            const float sg = pi > 0.f ? 1.f : (pi < 0.f ? -1.f : 0.f);
            gi += sg * l1 + pi * l2;
        }
        if (clampv > 0.f) gi = fminf(fmaxf(gi, -clampv), clampv);
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        p[i] = pi - step * mi / (sqrtf(vi) + eps);
        if (write_back) g[i] = gi;
    }
}
__device__ __forceinline__ float pen_clamp(float pi, float gi, float l1, float l2, float clampv) {
    if (l1 != 0.f || l2 != 0.f) {
        const float sg = pi > 0.f ? 1.f : (pi < 0.f ? -1.f : 0.f);
        gi += sg * l1 + pi * l2;
    }
    if (clampv > 0.f) gi = fminf(fmaxf(gi, -clampv), clampv);
    return gi;
}
__global__ void sgd_k(float* p, float* g, float* v, long n, float lr, float mom, float damp, float l1, float l2,
                      float clampv, int write_back) {
    GRID_STRIDE(i, n) {
        const float pi = p[i];
        const float gi = pen_clamp(pi, g[i], l1, l2, clampv);
        float d = gi;
        if (mom != 0.f) { d = mom * v[i] + (1.f - damp) * gi; v[i] = d; }
        p[i] = pi - lr * d;
        if (write_back) g[i] = gi;
    }
}
__global__ void adagrad_k(float* p, float* g, float* var, long n, float lr, float l1, float l2, float clampv,
                          int write_back) {
    GRID_STRIDE(i, n) {
        const float pi = p[i];
        const float gi = pen_clamp(pi, g[i], l1, l2, clampv);
        const float vi = var[i] + gi * gi;
        var[i] = vi;
        p[i] = pi - lr * gi / (sqrtf(vi) + 1e-10f);
        if (write_back) g[i] = gi;
    }
}
__global__ void confusion_k(const float* out, const float* tgt, int32_t* counts, long n) {
    GRID_STRIDE(i, n) {
        const int pred = out[i] > 0.5f ? 1 : 0;
        const int t = tgt[i] > 0.5f ? 1 : 0;
        atomicAdd(&counts[pred * 2 + t], 1);
    }
}

}  // namespace

static inline bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }
#define EW_LAUNCH(kern, n, ...)                                                                  \
    do {                                                                                         \
        if ((n) > 0) {                                                                           \
            hipLaunchKernelGGL(kern, dim3(cg::ew_grid(n)), dim3(256), 0, cg::S(stream), __VA_ARGS__); \
            CG_LAUNCH_CHECK();                                                                   \
        }                                                                                        \
    } while (0)

extern "C" {

int cg_abi_version(void) { return CG_TIMING_PROBE ? -1 : CG_ABI_VERSION; }   // a timing-only build (common.h) is refused by every host
int cg_set_option(const char* name, long value) {
    CG_REQUIRE(name, "cg_set_option: null name");
    for (int i = 0; i < cg::OPT_COUNT; ++i)
        if (strcmp(name, cg::kOptDefs[i].name) == 0) {
            ++cg::g_opt_epoch;
            if (value == -1) { cg::g_opt_state[i] = 0; return 0; }   // back to the environment / built-in default
            cg::g_opt_val[i] = value; cg::g_opt_state[i] = 2;
            return 0;
        }
    return cg::fail("cg_set_option: unknown option %s", name);
}
int cg_get_option(const char* name, long* value) {
    CG_REQUIRE(name && value, "cg_get_option: null pointer");
    for (int i = 0; i < cg::OPT_COUNT; ++i)
        if (strcmp(name, cg::kOptDefs[i].name) == 0) { *value = cg::opt((cg::Opt)i); return 0; }
    return cg::fail("cg_get_option: unknown option %s", name);
}
const char* cg_last_error(void) { return cg::err_buf(); }
int cg_device_count(int* count) { CG_REQUIRE(count, "null"); CG_HIP(hipGetDeviceCount(count)); return 0; }
int cg_set_device(int device) { CG_HIP(hipSetDevice(device)); return 0; }
int cg_malloc(void** dptr, size_t bytes) { CG_REQUIRE(dptr, "null"); CG_HIP(hipMalloc(dptr, bytes)); return 0; }
int cg_free(void* dptr) { CG_HIP(hipFree(dptr)); return 0; }
int cg_memcpy_h2d(void* stream, void* dst, const void* src, size_t bytes) {
    CG_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, cg::S(stream))); return 0;
}
int cg_memcpy_d2h(void* stream, void* dst, const void* src, size_t bytes) {
    CG_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, cg::S(stream)));
    CG_HIP(hipStreamSynchronize(cg::S(stream)));
    return 0;
}
int cg_memcpy_d2d(void* stream, void* dst, const void* src, size_t bytes) {
    CG_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, cg::S(stream))); return 0;
}
// 16 bytes per lane, grid-stride: the runtime's fill kernel (hipMemsetAsync -> __amd_rocclr_fillBufferAligned) takes 34 us for the
// 26.7 MB of D's flat gradient (0.8 TB/s) at the head of every fevalD (adversarial.lua:80 zeroes the gradients before the forward pass)
__global__ __launch_bounds__(256) void zero16_k(float4* __restrict__ p, long n16) {
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    GRID_STRIDE(i, n16) p[i] = z;
}
int cg_memset_zero(void* stream, void* dst, size_t bytes) {
    if (bytes >= (1u << 20) && (uintptr_t)dst % 16 == 0) {     // large, aligned: our own kernel; the tail (< 16 bytes) and small buffers: the runtime
        const long n16 = (long)(bytes / 16);
        EW_LAUNCH(zero16_k, n16, (float4*)dst, n16);
        const size_t done = (size_t)n16 * 16;
        if (done < bytes) CG_HIP(hipMemsetAsync((char*)dst + done, 0, bytes - done, cg::S(stream)));
        return 0;
    }
    CG_HIP(hipMemsetAsync(dst, 0, bytes, cg::S(stream))); return 0;
}
int cg_stream_create(void** stream) {
    CG_REQUIRE(stream, "null");
    hipStream_t s; CG_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); *stream = (void*)s; return 0;
}
int cg_stream_on_queue(void* ref_stream, int queue_class, int slot, void** stream) {
    CG_REQUIRE(stream && queue_class >= 0 && slot >= 0, "cg_stream_on_queue: bad arguments");
    hipStream_t s = cg::queue_stream(cg::S(ref_stream), queue_class, slot);
    if (!s) return 1;
    *stream = (void*)s;
    return 0;
}
int cg_stream_destroy(void* stream) { CG_HIP(hipStreamDestroy(cg::S(stream))); return 0; }
int cg_stream_sync(void* stream) { CG_HIP(hipStreamSynchronize(cg::S(stream))); return 0; }

int cg_host_alloc(void** hptr, size_t bytes) {
    CG_REQUIRE(hptr && bytes > 0, "cg_host_alloc: bad arguments");
    CG_HIP(hipHostMalloc(hptr, bytes, hipHostMallocDefault));
    return 0;
}
int cg_host_free(void* hptr) { if (hptr) CG_HIP(hipHostFree(hptr)); return 0; }
int cg_event_create(void** event) {
    CG_REQUIRE(event, "cg_event_create: null pointer");
    hipEvent_t e; CG_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming)); *event = (void*)e; return 0;
}
int cg_event_destroy(void* event) { if (event) CG_HIP(hipEventDestroy((hipEvent_t)event)); return 0; }
int cg_event_record(void* event, void* stream) {
    CG_REQUIRE(event, "cg_event_record: null event");
    CG_HIP(hipEventRecord((hipEvent_t)event, cg::S(stream))); return 0;
}
int cg_event_sync(void* event) {
    CG_REQUIRE(event, "cg_event_sync: null event");
    CG_HIP(hipEventSynchronize((hipEvent_t)event)); return 0;
}
int cg_stream_wait_event(void* stream, void* event) {
    CG_REQUIRE(event, "cg_stream_wait_event: null event");
    CG_HIP(hipStreamWaitEvent(cg::S(stream), (hipEvent_t)event, 0)); return 0;
}
int cg_images_u8_to_f32(void* stream, const unsigned char* src, float* dst, long npixels, int colorspace) {
    CG_REQUIRE(src && dst && npixels > 0 && (colorspace == 0 || colorspace == 1), "cg_images_u8_to_f32: bad arguments");
    EW_LAUNCH(images_u8_to_f32_k, npixels, src, dst, npixels, colorspace); return 0;
}

int cg_images_u8_scale_to_f32(void* stream, const unsigned char* src, float* dst, int N, int Hs, int Ws, int Hd, int Wd, int colorspace) {
    CG_REQUIRE(src && dst && N > 0 && Hs > 0 && Ws > 0 && Hd > 0 && Wd > 0 && (colorspace == 0 || colorspace == 1), "cg_images_u8_scale_to_f32: bad arguments");
    CG_REQUIRE(Hs <= 6 * Hd && Ws <= 6 * Wd, "cg_images_u8_scale_to_f32: down-scaling by more than 6 is not supported");
    const long total = (long)N * Hd * Wd;
    EW_LAUNCH(images_u8_scale_k, total, src, dst, N, Hs, Ws, Hd, Wd, colorspace); return 0;
}

int cg_prelu_forward(void* stream, const float* x, const float* alpha, float* y, long n) {
    CG_REQUIRE(x && alpha && y, "cg_prelu_forward: null pointer");
    if (n % 4 == 0 && al16(x) && al16(y)) { EW_LAUNCH(prelu_fwd_v4k, n / 4, x, alpha, y, n / 4); return 0; }
    EW_LAUNCH(prelu_fwd_k, n, x, alpha, y, n); return 0;
}
size_t cg_prelu_backward_workspace_bytes(long n) { return n > 0 ? sizeof(double) * (size_t)cg::ew_grid(n) : 0; }
int cg_prelu_backward(void* stream, const float* x, const float* dy, const float* alpha, float* dx, float* galpha,
                      float scale, long n, void* ws, size_t ws_bytes) {
    CG_REQUIRE(x && dy && alpha && dx, "cg_prelu_backward: null pointer");
    if (n <= 0) return 0;
    const bool v4 = n % 4 == 0 && al16(x) && al16(dy) && al16(dx);
    const int nblk = cg::ew_grid(v4 ? n / 4 : n);
    double* gpart = nullptr;
    if (galpha) {
        CG_REQUIRE(ws && ws_bytes >= sizeof(double) * nblk && ((uintptr_t)ws % 8) == 0,
                   "cg_prelu_backward: workspace too small (%zu < %zu)", ws_bytes, sizeof(double) * nblk);
        gpart = (double*)ws;
    }
    if (v4) EW_LAUNCH(prelu_bwd_v4k, n / 4, x, dy, alpha, dx, gpart, n / 4);
    else EW_LAUNCH(prelu_bwd_k, n, x, dy, alpha, dx, gpart, n);
    if (galpha) {
        hipLaunchKernelGGL(prelu_galpha_reduce_k, dim3(1), dim3(256), 0, cg::S(stream), gpart, nblk, galpha, scale);
        CG_LAUNCH_CHECK();
    }
    return 0;
}
int cg_leakyrelu_forward(void* stream, const float* x, float* y, float slope, long n) {
    CG_REQUIRE(x && y, "cg_leakyrelu_forward: null pointer");
    if (n % 4 == 0 && al16(x) && al16(y)) { EW_LAUNCH(lrelu_fwd_v4k, n / 4, x, y, slope, n / 4); return 0; }
    EW_LAUNCH(lrelu_fwd_k, n, x, y, slope, n); return 0;
}
int cg_leakyrelu_backward(void* stream, const float* x, const float* dy, float* dx, float slope, long n) {
    CG_REQUIRE(x && dy && dx, "cg_leakyrelu_backward: null pointer");
    if (n % 4 == 0 && al16(x) && al16(dy) && al16(dx)) { EW_LAUNCH(lrelu_bwd_v4k, n / 4, x, dy, dx, slope, n / 4); return 0; }
    EW_LAUNCH(lrelu_bwd_k, n, x, dy, dx, slope, n); return 0;
}
int cg_sigmoid_forward(void* stream, const float* x, float* y, long n) {
    CG_REQUIRE(x && y, "cg_sigmoid_forward: null pointer");
    EW_LAUNCH(sigmoid_fwd_k, n, x, y, n); return 0;
}
int cg_sigmoid_backward(void* stream, const float* y, const float* dy, float* dx, long n) {
    CG_REQUIRE(y && dy && dx, "cg_sigmoid_backward: null pointer");
    EW_LAUNCH(sigmoid_bwd_k, n, y, dy, dx, n); return 0;
}
int cg_bce_forward(void* stream, const float* p, const float* t, float* loss, long n) {
    CG_REQUIRE(p && t && loss && n > 0, "cg_bce_forward: bad args");
    hipLaunchKernelGGL(bce_fwd_k, dim3(1), dim3(256), 0, cg::S(stream), p, t, loss, n);
    CG_LAUNCH_CHECK(); return 0;
}
int cg_bce_backward(void* stream, const float* p, const float* t, float* dp, long n) {
    CG_REQUIRE(p && t && dp, "cg_bce_backward: null pointer");
    EW_LAUNCH(bce_bwd_k, n, p, t, dp, n); return 0;
}

static int colreduce_launch(void* stream, int mode, const float* x, const float* dy, const float* mean,
                            const float* invstd, long M, int C, double* sums, int nsums) {
    if (M <= 0) { CG_HIP(hipMemsetAsync(sums, 0, sizeof(double) * nsums * C, cg::S(stream))); return 0; }
    const bool v4 = C % 4 == 0 && (!x || al16(x)) && (!dy || al16(dy));
    const int qb = C >= 128 ? 32 : 16;   // channel quads per block in the float4 form
    const int cblocks = v4 ? cg::cdiv(C / 4, qb) : cg::cdiv(C, 64);
    // row chunks: every chunk ends in one double atomic per channel, and same-address atomics serialise (~10 ns each),
    // so aim at one workgroup per CU in total (CG_COLREDUCE_WGS_PER_CU) rather than at maximum occupancy
    const int cmul = cg::kColReduceWgsPerCU;
    long chunks = std::max(1L, std::min((M + 63) / 64, (long)cg::kNumCU * cmul / cblocks));
    const long rows_per_block = ((M + chunks - 1) / chunks + 3) / 4 * 4;
    chunks = (M + rows_per_block - 1) / rows_per_block;
    dim3 grid(cblocks, (unsigned)chunks);
    // deterministic sum inside one launch: partials in the stream's scratch, added in a fixed two-level order (cg::ColTree)
    char* scr = (char*)cg::col_scratch(cg::S(stream));
    if (!scr) return 1;
    cg::ColTree tree;
    CG_REQUIRE(cg::col_tree_layout(scr, chunks, (size_t)nsums * C, 0, tree), "column reduce: %ld chunks x %d sums exceed the scratch", chunks, nsums * C);
#define CG_COLRED(K) hipLaunchKernelGGL(K, grid, dim3(256), 0, cg::S(stream), x, dy, mean, invstd, M, C, rows_per_block, sums, tree)
    if (v4 && qb == 32) {
        if (mode == 0) CG_COLRED((colreduce4_k<0, 32>)); else if (mode == 1) CG_COLRED((colreduce4_k<1, 32>)); else CG_COLRED((colreduce4_k<2, 32>));
    } else if (v4) {
        if (mode == 0) CG_COLRED((colreduce4_k<0, 16>)); else if (mode == 1) CG_COLRED((colreduce4_k<1, 16>)); else CG_COLRED((colreduce4_k<2, 16>));
    } else {
        if (mode == 0) CG_COLRED(colreduce_k<0>); else if (mode == 1) CG_COLRED(colreduce_k<1>); else CG_COLRED(colreduce_k<2>);
    }
#undef CG_COLRED
    CG_LAUNCH_CHECK();
    return 0;
}

int cg_bias_grad(void* stream, const float* dy, float* gb, long M, int C, float scale, void* ws, size_t ws_bytes) {
    CG_REQUIRE(dy && gb && C > 0, "cg_bias_grad: bad args");
    CG_REQUIRE(ws && ws_bytes >= sizeof(double) * (size_t)C, "cg_bias_grad: workspace too small");
    double* scratch = (double*)ws;
    if (colreduce_launch(stream, 2, nullptr, dy, nullptr, nullptr, M, C, scratch, 1)) return 1;
    EW_LAUNCH(bias_grad_finish_k, (long)C, (const double*)scratch, gb, C, scale);
    return 0;
}

int cg_bn_stats(void* stream, const float* x, long M, int C, double* sums) {
    CG_REQUIRE(x && sums && C > 0, "cg_bn_stats: bad args");
    return colreduce_launch(stream, 0, x, nullptr, nullptr, nullptr, M, C, sums, 2);
}
int cg_bn_forward(void* stream, const float* x, float* y, const float* gamma, const float* beta, const double* sums,
                  double count, long M, int C, float eps, float momentum, float* running_mean, float* running_var,
                  float* save_mean, float* save_invstd) {
    CG_REQUIRE(x && y && gamma && beta && sums && save_mean && save_invstd && C > 0 && count > 0, "cg_bn_forward: bad args");
    EW_LAUNCH(bn_prepare_k, (long)C, sums, count, C, eps, momentum, running_mean, running_var, save_mean, save_invstd);
    const long total = M * C;
    if (C % 4 == 0 && al16(x) && al16(y)) {
        EW_LAUNCH(bn_apply_v4k, total / 4, x, y, gamma, beta, (const float*)save_mean, (const float*)save_invstd, total / 4,
                  C / 4);
        return 0;
    }
    EW_LAUNCH(bn_apply_k, total, x, y, gamma, beta, (const float*)save_mean, (const float*)save_invstd, total, C);
    return 0;
}
int cg_bn_forward_eval(void* stream, const float* x, float* y, const float* gamma, const float* beta,
                       const float* running_mean, const float* running_var, long M, int C, float eps) {
    CG_REQUIRE(x && y && gamma && beta && running_mean && running_var && C > 0, "cg_bn_forward_eval: bad args");
    const long total = M * C;
    EW_LAUNCH(bn_eval_k, total, x, y, gamma, beta, running_mean, running_var, total, C, eps);
    return 0;
}
int cg_bn_backward_stats(void* stream, const float* x, const float* dy, const float* save_mean, const float* save_invstd,
                         long M, int C, double* sums) {
    CG_REQUIRE(x && dy && save_mean && save_invstd && sums && C > 0, "cg_bn_backward_stats: bad args");
    return colreduce_launch(stream, 1, x, dy, save_mean, save_invstd, M, C, sums, 2);
}
int cg_bn_backward(void* stream, const float* x, const float* dy, const float* gamma, const float* save_mean,
                   const float* save_invstd, const double* sums, double count, const double* local_sums, long M, int C,
                   float* dx, float* ggamma, float* gbeta, float scale) {
    CG_REQUIRE(x && dy && gamma && save_mean && save_invstd && sums && local_sums && dx && C > 0 && count > 0,
               "cg_bn_backward: bad args");
    const long total = M * C;
    if (C % 4 == 0 && al16(x) && al16(dy) && al16(dx)) {
        EW_LAUNCH(bn_bwd_v4k, total / 4, x, dy, gamma, save_mean, save_invstd, sums, count, total / 4, C / 4, dx);
    } else {
        EW_LAUNCH(bn_bwd_k, total, x, dy, gamma, save_mean, save_invstd, sums, count, total, C, dx);
    }
    EW_LAUNCH(bn_bwd_param_k, (long)C, local_sums, C, ggamma, gbeta, scale);
    return 0;
}

int cg_upsample2x_forward(void* stream, const float* x, float* y, int N, int H, int W, int C) {
    CG_REQUIRE(x && y, "cg_upsample2x_forward: null pointer");
    const long total = (long)N * 4 * H * W * C;
    EW_LAUNCH(ups2_fwd_k, total, x, y, N, H, W, C); return 0;
}
int cg_upsample2x_backward(void* stream, const float* dy, float* dx, int N, int H, int W, int C) {
    CG_REQUIRE(dy && dx, "cg_upsample2x_backward: null pointer");
    const long total = (long)N * H * W * C;
    EW_LAUNCH(ups2_bwd_k, total, dy, dx, N, H, W, C); return 0;
}
int cg_avgpool2_forward(void* stream, const float* x, float* y, int N, int H, int W, int C) {
    CG_REQUIRE(x && y, "cg_avgpool2_forward: null pointer");
    const long total = (long)N * (H / 2) * (W / 2) * C;
    EW_LAUNCH(pool2_fwd_k<false>, total, x, y, N, H, W, C); return 0;
}
int cg_avgpool2_backward(void* stream, const float* dy, float* dx, int N, int H, int W, int C) {
    CG_REQUIRE(dy && dx, "cg_avgpool2_backward: null pointer");
    const long total = (long)N * H * W * C;
    EW_LAUNCH(avgpool2_bwd_k, total, dy, dx, N, H, W, C); return 0;
}
int cg_maxpool2_forward(void* stream, const float* x, float* y, int N, int H, int W, int C) {
    CG_REQUIRE(x && y, "cg_maxpool2_forward: null pointer");
    const long total = (long)N * (H / 2) * (W / 2) * C;
    EW_LAUNCH(pool2_fwd_k<true>, total, x, y, N, H, W, C); return 0;
}
int cg_maxpool2_backward(void* stream, const float* x, const float* dy, float* dx, int N, int H, int W, int C) {
    CG_REQUIRE(x && dy && dx, "cg_maxpool2_backward: null pointer");
    CG_REQUIRE(H % 2 == 0 && W % 2 == 0, "cg_maxpool2_backward: odd spatial size");
    const long total = (long)N * (H / 2) * (W / 2) * C;
    EW_LAUNCH(maxpool2_bwd_k, total, x, dy, dx, N, H, W, C); return 0;
}
int cg_mask_mul(void* stream, const float* x, const float* mask, float* y, int N, long HW, int C, int spatial) {
    CG_REQUIRE(x && mask && y, "cg_mask_mul: null pointer");
    const long total = (long)N * HW * C;
    if (al16(x) && al16(y)) {
        if (spatial && C % 4 == 0) {
            EW_LAUNCH(mask_mul_sp_v4k, total / 4, x, mask, y, total / 4, HW * C / 4, C / 4); return 0;
        }
        if (!spatial && total % 4 == 0 && al16(mask)) { EW_LAUNCH(mask_mul_v4k, total / 4, x, mask, y, total / 4); return 0; }
    }
    EW_LAUNCH(mask_mul_k, total, x, mask, y, total, HW * C, C, spatial); return 0;
}
int cg_rng_bernoulli_dev(void* stream, float* out, long n, float keep_prob, float value, uint64_t seed, uint64_t offset,
                         const uint64_t* base) {
    CG_REQUIRE(out, "cg_rng_bernoulli: null pointer");
    EW_LAUNCH(rng_bernoulli_k, n, out, n, keep_prob, value, seed, offset, base); return 0;
}
int cg_rng_bernoulli_dev_grouped(void* stream, float* out, long n_per_group, int ngroups, float keep_prob, float value,
                                 uint64_t seed, uint64_t off0, uint64_t off1, uint64_t off2, uint64_t off3,
                                 const uint64_t* base) {
    CG_REQUIRE(out && ngroups >= 1 && ngroups <= 4, "cg_rng_bernoulli_dev_grouped: bad args");
    EW_LAUNCH(rng_bernoulli_groups_k, n_per_group * ngroups, out, n_per_group, ngroups, keep_prob, value, seed, off0, off1, off2,
              off3, base);
    return 0;
}
int cg_rng_uniform_dev(void* stream, float* out, long n, float lo, float hi, uint64_t seed, uint64_t offset,
                       const uint64_t* base) {
    CG_REQUIRE(out, "cg_rng_uniform: null pointer");
    EW_LAUNCH(rng_uniform_k, n, out, n, lo, hi, seed, offset, base); return 0;
}
int cg_rng_randint_dev(void* stream, int32_t* out, long n, int32_t range, uint64_t seed, uint64_t offset,
                       const uint64_t* base) {
    CG_REQUIRE(out && range > 0, "cg_rng_randint: bad args");
    EW_LAUNCH(rng_randint_k, n, out, n, range, seed, offset, base); return 0;
}
int cg_rng_bernoulli(void* stream, float* out, long n, float keep_prob, float value, uint64_t seed, uint64_t offset) {
    return cg_rng_bernoulli_dev(stream, out, n, keep_prob, value, seed, offset, nullptr);
}
int cg_rng_uniform(void* stream, float* out, long n, float lo, float hi, uint64_t seed, uint64_t offset) {
    return cg_rng_uniform_dev(stream, out, n, lo, hi, seed, offset, nullptr);
}
int cg_rng_randint(void* stream, int32_t* out, long n, int32_t range, uint64_t seed, uint64_t offset) {
    return cg_rng_randint_dev(stream, out, n, range, seed, offset, nullptr);
}
int cg_counter_add(void* stream, uint64_t* counter, uint64_t delta) {
    CG_REQUIRE(counter, "cg_counter_add: null pointer");
    hipLaunchKernelGGL(counter_add_k, dim3(1), dim3(64), 0, cg::S(stream), counter, delta);
    CG_LAUNCH_CHECK(); return 0;
}
int cg_nchw_to_nhwc(void* stream, const float* in, float* out, int N, int C, int H, int W) {
    CG_REQUIRE(in && out, "cg_nchw_to_nhwc: null pointer");
    const long total = (long)N * C * H * W;
    EW_LAUNCH(nchw_to_nhwc_k, total, in, out, N, C, H, W); return 0;
}
int cg_nhwc_to_nchw(void* stream, const float* in, float* out, int N, int C, int H, int W) {
    CG_REQUIRE(in && out, "cg_nhwc_to_nchw: null pointer");
    const long total = (long)N * C * H * W;
    EW_LAUNCH(nhwc_to_nchw_k, total, in, out, N, C, H, W); return 0;
}
int cg_copy_channels(void* stream, const float* src, float* dst, long M, int Csrc, int src_off, int Cdst, int dst_off,
                     int Ccopy) {
    CG_REQUIRE(src && dst, "cg_copy_channels: null pointer");
    CG_REQUIRE(src_off >= 0 && dst_off >= 0 && src_off + Ccopy <= Csrc && dst_off + Ccopy <= Cdst,
               "cg_copy_channels: slice out of range");
    const long total = M * Ccopy;
    EW_LAUNCH(copy_channels_k, total, src, dst, M, Csrc, src_off, Cdst, dst_off, Ccopy); return 0;
}
int cg_gather_rows(void* stream, const float* src, const int32_t* idx, float* dst, long nrows, long rowlen) {
    CG_REQUIRE(src && idx && dst, "cg_gather_rows: null pointer");
    const long total = nrows * rowlen;
    EW_LAUNCH(gather_rows_k, total, src, idx, dst, nrows, rowlen); return 0;
}
int cg_fill(void* stream, float* x, float value, long n) {
    CG_REQUIRE(x, "cg_fill: null pointer");
    EW_LAUNCH(fill_k, n, x, value, n); return 0;
}
int cg_add(void* stream, const float* a, const float* b, float* out, long n) {
    CG_REQUIRE(a && b && out, "cg_add: null pointer");
    if (n % 4 == 0 && al16(a) && al16(b) && al16(out)) { EW_LAUNCH(add_v4k, n / 4, a, b, out, n / 4); return 0; }
    EW_LAUNCH(add_k, n, a, b, out, n); return 0;
}
int cg_axpy(void* stream, float alpha, const float* x, float* y, long n) {
    CG_REQUIRE(x && y, "cg_axpy: null pointer");
    if (n % 4 == 0 && al16(x) && al16(y)) { EW_LAUNCH(axpy_v4k, n / 4, alpha, x, y, n / 4); return 0; }
    EW_LAUNCH(axpy_k, n, alpha, x, y, n); return 0;
}
int cg_axpy_sign(void* stream, float alpha, const float* x, float* y, long n) {
    CG_REQUIRE(x && y, "cg_axpy_sign: null pointer");
    EW_LAUNCH(axpy_sign_k, n, alpha, x, y, n); return 0;
}
int cg_scale(void* stream, float* x, float alpha, long n) {
    CG_REQUIRE(x, "cg_scale: null pointer");
    EW_LAUNCH(scale_k, n, x, alpha, n); return 0;
}
int cg_clamp(void* stream, float* x, float lo, float hi, long n) {
    CG_REQUIRE(x, "cg_clamp: null pointer");
    EW_LAUNCH(clamp_k, n, x, lo, hi, n); return 0;
}
int cg_sumsq(void* stream, const float* x, long n, double* out) {
    CG_REQUIRE(x && out, "cg_sumsq: null pointer");
    CG_HIP(hipMemsetAsync(out, 0, sizeof(double), cg::S(stream)));
    EW_LAUNCH(sumred_k<true>, n, x, n, out); return 0;
}
int cg_sumabs(void* stream, const float* x, long n, double* out) {
    CG_REQUIRE(x && out, "cg_sumabs: null pointer");
    CG_HIP(hipMemsetAsync(out, 0, sizeof(double), cg::S(stream)));
    EW_LAUNCH(sumred_k<false>, n, x, n, out); return 0;
}

int cg_affine_matrix_forward(void* stream, const float* params, float* T, int N, int use_rot, int use_scale,
                             int use_trans) {
    CG_REQUIRE(params && T, "cg_affine_matrix_forward: null pointer");
    CG_REQUIRE(use_rot || use_scale || use_trans, "cg_affine_matrix_forward: fully parametrised form not supported");
    EW_LAUNCH(affine_matrix_fwd_k, (long)N, params, T, N, use_rot ? 1 : 0, use_scale ? 1 : 0, use_trans ? 1 : 0);
    return 0;
}
int cg_affine_matrix_backward(void* stream, const float* params, const float* gT, float* gparams, int N, int use_rot,
                              int use_scale, int use_trans) {
    CG_REQUIRE(params && gT && gparams, "cg_affine_matrix_backward: null pointer");
    EW_LAUNCH(affine_matrix_bwd_k, (long)N, params, gT, gparams, N, use_rot ? 1 : 0, use_scale ? 1 : 0, use_trans ? 1 : 0);
    return 0;
}
int cg_affine_grid_forward(void* stream, const float* T, float* grid, int N, int H, int W) {
    CG_REQUIRE(T && grid, "cg_affine_grid_forward: null pointer");
    const long total = (long)N * H * W;
    EW_LAUNCH(affine_grid_fwd_k, total, T, grid, N, H, W); return 0;
}
int cg_affine_grid_backward(void* stream, const float* ggrid, float* gT, int N, int H, int W) {
    CG_REQUIRE(ggrid && gT && N > 0, "cg_affine_grid_backward: bad args");
    hipLaunchKernelGGL(affine_grid_bwd_k, dim3(N), dim3(256), 0, cg::S(stream), ggrid, gT, N, H, W);
    CG_LAUNCH_CHECK(); return 0;
}
namespace {
int sampler_forward(void* stream, const float* img, const float* grid, float* out, int N, int Nimg, int Hi, int Wi, int C, int Ho,
                    int Wo) {
    const long total = (long)N * Ho * Wo * C;
    if (C % 4 == 0 && al16(img) && al16(out) && (uintptr_t)grid % 8 == 0) {
        EW_LAUNCH(bilinear_fwd_v4k, total / 4, img, grid, out, N, Nimg, Hi, Wi, C / 4, Ho, Wo);
        return 0;
    }
    EW_LAUNCH(bilinear_fwd_k, total, img, grid, out, N, Nimg, Hi, Wi, C, Ho, Wo); return 0;
}

int sampler_backward(void* stream, const float* img, const float* grid, const float* gout, float* gimg, float* ggrid, int N, int Nimg,
                     int Hi, int Wi, int C, int Ho, int Wo) {
    const long P = (long)Ho * Wo, Q = (long)Hi * Wi;
    const size_t shb = ((size_t)6 * P + 2 * ((size_t)(Hi + 1) * (Wi + 1)) + 1) * 4;
    if (shb <= 140 * 1024 && C <= 256 && N > 0) {   // deterministic gather form
        const bool v4 = C % 4 == 0 && al16(img) && al16(gout) && al16(gimg);
        const int units = v4 ? C / 4 : C;                  // lane-sized channel units per pixel
        int G = 4;
        while (G < 64 && G < units) G <<= 1;
        // pixel slots per workgroup: at most 16 workgroups per sample (every workgroup stages and buckets all P taps again), and
        // no more workgroups than keep the chip busy (~4 per CU) when the batch is large
        const int want = std::max(1, std::min(16, (int)(cg::kNumCU * 4 / std::max(1, N))));
        const int S = std::max(std::max(32, 256 / G), cg::cdiv(std::max(P, Q), want));
        const int nchunks = cg::cdiv(std::max(P, Q), S);
        const dim3 grd((unsigned)((long)N * nchunks)), blk(256);
#define CG_BILIN_DET(GG, K, VV) hipLaunchKernelGGL((bilinear_bwd_det_k<GG, K, VV>), grd, blk, shb, cg::S(stream), img, grid, gout, gimg, ggrid, Nimg, Hi, Wi, C, Ho, Wo, nchunks, S)
        if (v4) {
            if (G == 4) CG_BILIN_DET(4, 1, 4);
            else if (G == 8) CG_BILIN_DET(8, 1, 4);
            else if (G == 16) CG_BILIN_DET(16, 1, 4);
            else if (G == 32) CG_BILIN_DET(32, 1, 4);
            else CG_BILIN_DET(64, 1, 4);                   // C <= 256
        } else if (G == 4) CG_BILIN_DET(4, 1, 1);
        else if (G == 8) CG_BILIN_DET(8, 1, 1);
        else if (G == 16) CG_BILIN_DET(16, 1, 1);
        else if (G == 32) CG_BILIN_DET(32, 1, 1);
        else if (C <= 64) CG_BILIN_DET(64, 1, 1);
        else if (C <= 128) CG_BILIN_DET(64, 2, 1);
        else CG_BILIN_DET(64, 4, 1);
#undef CG_BILIN_DET
        CG_LAUNCH_CHECK();
        return 0;
    }
    CG_HIP(hipMemsetAsync(gimg, 0, sizeof(float) * (size_t)N * Hi * Wi * C, cg::S(stream)));
    const long npix = (long)N * Ho * Wo;
    int G = 4;
    while (G < 64 && G < C) G <<= 1;
    long blocks = (npix * G + 255) / 256;
    if (blocks > cg::kNumCU * 8) blocks = cg::kNumCU * 8;
    if (blocks < 1) blocks = 1;
    const dim3 grd((unsigned)blocks), blk(256);
#define CG_BILIN_BWD(GG) hipLaunchKernelGGL(bilinear_bwd_k<GG>, grd, blk, 0, cg::S(stream), img, grid, gout, gimg, ggrid, N, Nimg, Hi, Wi, C, Ho, Wo)
    switch (G) {
        case 4: CG_BILIN_BWD(4); break;
        case 8: CG_BILIN_BWD(8); break;
        case 16: CG_BILIN_BWD(16); break;
        case 32: CG_BILIN_BWD(32); break;
        default: CG_BILIN_BWD(64); break;
    }
#undef CG_BILIN_BWD
    CG_LAUNCH_CHECK(); return 0;
}
}  // namespace

int cg_bilinear_sampler_forward(void* stream, const float* img, const float* grid, float* out, int N, int Hi, int Wi,
                                int C, int Ho, int Wo) {
    CG_REQUIRE(img && grid && out && N > 0, "cg_bilinear_sampler_forward: null pointer");
    return sampler_forward(stream, img, grid, out, N, N, Hi, Wi, C, Ho, Wo);
}
int cg_bilinear_sampler_backward(void* stream, const float* img, const float* grid, const float* gout, float* gimg,
                                 float* ggrid, int N, int Hi, int Wi, int C, int Ho, int Wo) {
    CG_REQUIRE(img && grid && gout && gimg && ggrid && N > 0, "cg_bilinear_sampler_backward: null pointer");
    return sampler_backward(stream, img, grid, gout, gimg, ggrid, N, N, Hi, Wi, C, Ho, Wo);
}
int cg_bilinear_sampler_forward_shared(void* stream, int ngroups, const float* img, const float* grid, float* out, int N, int Hi,
                                       int Wi, int C, int Ho, int Wo) {
    CG_REQUIRE(img && grid && out && N > 0 && ngroups >= 1, "cg_bilinear_sampler_forward_shared: bad arguments");
    return sampler_forward(stream, img, grid, out, ngroups * N, N, Hi, Wi, C, Ho, Wo);
}
int cg_bilinear_sampler_backward_shared(void* stream, int ngroups, const float* img, const float* grid, const float* gout,
                                        float* gimg, float* ggrid, int N, int Hi, int Wi, int C, int Ho, int Wo) {
    CG_REQUIRE(img && grid && gout && gimg && ggrid && N > 0 && ngroups >= 1, "cg_bilinear_sampler_backward_shared: bad arguments");
    return sampler_backward(stream, img, grid, gout, gimg, ggrid, ngroups * N, N, Hi, Wi, C, Ho, Wo);
}

int cg_adam_step(void* stream, float* p, float* g, float* m, float* v, long n, float lr, float beta1, float beta2,
                 float eps, int t, float l1, float l2, float clamp, int write_back_grad) {
    CG_REQUIRE(p && g && m && v && t >= 1, "cg_adam_step: bad args");
    const double bc1 = 1.0 - pow((double)beta1, (double)t);
    const double bc2 = 1.0 - pow((double)beta2, (double)t);
    const float step = (float)((double)lr * sqrt(bc2) / bc1);
    EW_LAUNCH(adam_k, n, p, g, m, v, n, step, beta1, beta2, eps, l1, l2, clamp, write_back_grad, lr, (const uint64_t*)nullptr);
    return 0;
}
int cg_adam_step_dev(void* stream, float* p, float* g, float* m, float* v, long n, float lr, float beta1, float beta2,
                     float eps, const uint64_t* t_dev, float l1, float l2, float clamp, int write_back_grad) {
    CG_REQUIRE(p && g && m && v && t_dev, "cg_adam_step_dev: bad args");
    EW_LAUNCH(adam_k, n, p, g, m, v, n, 0.f, beta1, beta2, eps, l1, l2, clamp, write_back_grad, lr, t_dev);
    return 0;
}
int cg_sgd_step(void* stream, float* p, float* g, float* v, long n, float lr, float momentum, float dampening, float l1,
                float l2, float clamp, int write_back_grad) {
    CG_REQUIRE(p && g && (v || momentum == 0.f), "cg_sgd_step: bad args");
    EW_LAUNCH(sgd_k, n, p, g, v, n, lr, momentum, dampening, l1, l2, clamp, write_back_grad);
    return 0;
}
int cg_adagrad_step(void* stream, float* p, float* g, float* var, long n, float lr, float l1, float l2, float clamp,
                    int write_back_grad) {
    CG_REQUIRE(p && g && var, "cg_adagrad_step: bad args");
    EW_LAUNCH(adagrad_k, n, p, g, var, n, lr, l1, l2, clamp, write_back_grad);
    return 0;
}
int cg_confusion_update(void* stream, const float* outputs, const float* targets, int32_t* counts, long n) {
    CG_REQUIRE(outputs && targets && counts, "cg_confusion_update: null pointer");
    EW_LAUNCH(confusion_k, n, outputs, targets, counts, n); return 0;
}

}  // extern "C"
