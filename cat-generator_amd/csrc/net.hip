// net.hip - the planned executor behind the C ABI (cg_net_*, cg_graph_*; include/catgan.h).
//
// A host (the LuaJIT layer, the Python twin, a C++ driver) describes a network module by module - the constructor calls of
// models.lua:138-160,196-228,640-711,814-906 map one to one onto cg_net_add - binds the parameter tensors its
// getParameters() produced (train.lua:184-185) and then calls cg_net_forward / cg_net_backward where the reference calls
// MODEL_X:forward / :backward / :updateGradInput (adversarial.lua:84-89,182-197).  Everything between those calls lives here:
//
//   * the plan: nn.Sequential's module list is cut into segments that run as fused launches (conv|linear -> activation in the GEMM
//     epilogue; activation -> 2x2 pooling -> spatial dropout in one pass; conv -> batch-norm (training) -> PReLU with the statistics
//     in the GEMM epilogue; nn.View -> nn.Linear reading the NHWC map directly);
//   * lockstep execution of structurally identical nn.Concat branches (D32_st3's three transformer branches, models.lua:653-678):
//     grouped GEMM launches, parameter-free layers as ONE launch over the stacked batch, shared pooling / shared-image sampling;
//   * the second group of branches on a side HIP stream (event fork / join), deferred + batched weight-gradient reductions,
//     one batched weight re-pack per parameter update, sync-BN / gradient-bucket collectives at their places in the sequence;
//   * buffers: every activation, gradient, mask and workspace is allocated when a (network, input shape) pair is first compiled,
//     so a compiled plan launches without allocating - which is what lets cg_graph_begin / _end capture a whole training
//     iteration of host code into one hipGraph and cg_graph_launch replay it.
//
// A plan is compiled by symbolic execution: the module tree is walked once per (input shape, training flags) with tensor
// descriptors instead of data; every launch becomes a closure over resolved pointers and geometry, appended to a flat op list.
// Running a pass is a loop over that list.  Random draws (dropout masks) sit at fixed offsets from the counter-stream position
// the caller passes in, in the order a module-after-module walk draws them, so engine and oracle masks stay bit-identical.
//
// With option "trace" the launches go to recording stubs (net_ktable.inc) instead of the GPU: tests/test_net_plan.py checks the
// planner's launch sequence, data flow and draw order on a machine without a GPU.
#include <stdarg.h>

#include <algorithm>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "common.h"

namespace {

using std::vector;

struct Net;
thread_local Net* g_cur_net = nullptr;   // the net whose ops are running (trace stubs resolve pointers through it)

struct TraceLine {
    std::string s;
    explicit TraceLine(const char* name) { s = "call|"; s += name; }
    void stream(void* st);
    void ptr(const void* p);
    void parr(const void* const* a, int n) {
        if (!a) { s += "|n"; return; }
        s += "|a:" + std::to_string(n) + ":";
        for (int i = 0; i < n; ++i) { if (i) s += ","; s += pname(a[i]); }
    }
    void iarr(const int* a, int n) {
        s += "|I:";
        for (int i = 0; a && i < n; ++i) { if (i) s += ","; s += std::to_string(a[i]); }
    }
    void i(long long v) { s += "|i:" + std::to_string(v); }
    void u(unsigned long long v) { s += "|u:" + std::to_string(v); }
    void f(double v) { char b[64]; snprintf(b, sizeof b, "|f:%a", v); s += b; }
    static std::string pname(const void* p);
    int done();
};

#include "net_ktable.inc"

// ------------------------------------------------------------------------------------------------ descriptors
enum Kind {
    K_SEQ = 0, K_CONCAT = 1, K_CONCATTABLE = 2, K_LINEAR = 3, K_CONV = 4, K_PRELU = 5, K_LRELU = 6, K_SIGMOID = 7, K_BN = 8,
    K_VIEW = 9, K_COPY = 10, K_TRANSPOSE = 11, K_UPS = 12, K_AVGPOOL = 13, K_MAXPOOL = 14, K_SDROP = 15, K_DROP = 16,
    K_AFFMAT = 17, K_AFFGRID = 18, K_SAMPLER = 19, K_COUNT
};
enum { PLAIN = 0, NHWC = 1 };
enum { EXT_NONE = 0, EXT_X = 1, EXT_GY = 2 };

// A tensor as the planner sees it: where it lives, its logical (Torch7) shape, its physical layout.
struct Val {
    float* p = nullptr;        // device address (ext == EXT_NONE)
    int ext = EXT_NONE;        // else: byte offset `off` from the caller's input (EXT_X) / gradOutput (EXT_GY) pointer
    size_t off = 0;
    int nd = 0;
    long d[4] = {0, 0, 0, 0};  // logical shape; fmt NHWC: [N,C,H,W] stored as [N,H>>ups,W>>ups,C]
    int fmt = PLAIN;
    int ups = 0;               // logical H,W are twice the physical ones (virtual nn.SpatialUpSamplingNearest(2))
    uint64_t blk = 0;          // this tensor is slice gi of gc equal slices of the block whose key is blk (0: not a slice)
    int gi = 0, gc = 0;
    bool is_tab = false;       // nn.ConcatTable output: a table of tensors
    vector<Val> tab;
    bool none = true;

    long numel() const { long n = 1; for (int i = 0; i < nd; ++i) n *= d[i]; return n; }
    long phys() const { return numel() >> (2 * ups); }
    uint64_t key() const { return ext ? (((uint64_t)ext << 60) | (uint64_t)off) : (uint64_t)(uintptr_t)p; }
    bool same_shape(const Val& o) const {
        if (nd != o.nd) return false;
        for (int i = 0; i < nd; ++i) if (d[i] != o.d[i]) return false;
        return true;
    }
    Val at(size_t byte_off) const {   // same descriptor, address moved
        Val v = *this;
        if (ext) v.off += byte_off; else v.p = (float*)((char*)p + byte_off);
        return v;
    }
};

Val mkval(float* p, std::initializer_list<long> dims, int fmt = PLAIN, int ups = 0) {
    Val v; v.p = p; v.nd = (int)dims.size(); int i = 0; for (long x : dims) v.d[i++] = x; v.fmt = fmt; v.ups = ups; v.none = false;
    return v;
}
Val reshape(const Val& x, std::initializer_list<long> dims, int fmt, int ups = 0, bool keep_grp = false) {
    Val v = x; v.nd = (int)dims.size(); int i = 0; for (long q : dims) v.d[i++] = q; for (; i < 4; ++i) v.d[i] = 0;
    v.fmt = fmt; v.ups = ups;
    if (!keep_grp) { v.blk = 0; v.gi = v.gc = 0; }
    return v;
}

struct Mod {
    int id = 0, kind = 0, parent = -1;
    vector<int> kids;
    long ia[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    float fa[4] = {0, 0, 0, 0};
    int train = 1;
    float *w = nullptr, *gw = nullptr, *b = nullptr, *gb = nullptr, *rmean = nullptr, *rvar = nullptr;
    // kernel-side copies of a conv / linear layer's weight (shared by every compiled plan of the net)
    float *wf = nullptr, *wb = nullptr;          // plain: wf[(tap*Cin+ci)][Cout], wb (flipped taps) or NULL for 1x1
    int pk_map = -1;                             // 0 plain / 1 map (cg_pack_conv_weight_map) layout currently allocated; -1 none
    long map_c = 0, map_h = 0, map_w = 0;
    bool dirty_plain = true;
    float *wf_ph = nullptr, *wb_ph = nullptr;    // behind a folded 2x upsampling: phase-summed
    float *u_fwd = nullptr, *u_bwd = nullptr;    // Winograd-domain phase kernels
    float *u22 = nullptr, *u22b = nullptr;       // F(2x2,2x2) forward / data-gradient kernels of a 3x3 layer behind a folded upsampling
    bool wino = false, dirty_ups = true;
    bool is_gemm() const { return kind == K_LINEAR || kind == K_CONV; }
    bool is_act() const { return kind == K_PRELU || kind == K_LRELU; }
    bool is_pool() const { return kind == K_AVGPOOL || kind == K_MAXPOOL; }
    bool is_container() const { return kind <= K_CONCATTABLE; }
    long Cin() const { return kind == K_LINEAR ? ia[0] : ia[0]; }
    long Cout() const { return ia[1]; }
    long kW() const { return kind == K_CONV ? ia[2] : 1; }
    long kH() const { return kind == K_CONV ? ia[3] : 1; }
    long padW() const { return kind == K_CONV ? ia[4] : 0; }
    long padH() const { return kind == K_CONV ? ia[5] : 0; }
};

struct Geo {   // the 10 geometry arguments of the convolution entry points
    int N, Hp, Wp, Cin, Cout, kH, kW, padH, padW, ups;
    bool operator==(const Geo& o) const {
        return N == o.N && Hp == o.Hp && Wp == o.Wp && Cin == o.Cin && Cout == o.Cout && kH == o.kH && kW == o.kW && padH == o.padH &&
               padW == o.padW && ups == o.ups;
    }
};
#define GEO(g) (g).N, (g).Hp, (g).Wp, (g).Cin, (g).Cout, (g).kH, (g).kW, (g).padH, (g).padW, (g).ups

struct Seg { int kind; int i, j; };   // kinds below; modules [i, j) of a Sequential
enum { S_ONE = 0, S_GEMM_ACT, S_ACT_POOL, S_VIEW_GEMM, S_VIEW_GEMM_ACT, S_GEMM_BN_ACT, S_CAT_DROP, S_HEAD };

// What a module left behind in one compiled plan (the attributes the Python twin kept on the module objects).
struct MS {
    Val x, out, gin;             // input as consumed, output, gradInput
    Val p;                       // AffineTransformMatrixGenerator: its parameter input
    Val noise, noise_block;      // dropout masks
    Val in_img, in_grid;         // sampler inputs
    uint64_t stk = 0;            // key of the block a stacked forward produced (first sibling only)
    bool shared_in = false;      // pooling: siblings shared one launch
    bool has_shared = false; Val shared_out, shared_grids;   // sampler: shared-image launch
    bool skip = false;           // nn.View fused into the linear layer behind it
    long in_shape[4] = {0, 0, 0, 0}; int in_nd = 0;
    bool map_in = false; long mc = 0, mh = 0, mw = 0;        // nn.Linear consuming an NHWC map
    bool use_wino = false, use_wino22 = false;
    bool fused = false; int fG = 0; Val fX, fmask; long fN = 0, fC = 0, fH = 0, fW = 0;   // act_pool segment state (on the activation)
    bool bn_fused = false; long bnM = 0, bnC = 0;            // gemm_bn_act segment state (on the batch-norm)
    double count = 0;            // BN: samples behind the statistics (x world under sync-BN)
    vector<long> sizes;          // nn.Concat: channels per branch
    vector<Seg> ran;             // nn.Sequential: the plan its forward ran
    bool ran_set = false;
    void* wg_ws = nullptr; size_t wg_ws_bytes = 0; bool wg_pending = false;   // deferred weight-gradient workspace
    bool loc_fused = false; int locG = 0; long locN = 0;                      // [locnet, AffMat, AffGrid] ran as cg_locnet_forward
    bool cat_drop = false;                                                   // nn.Concat: the SpatialDropout behind it ran inside its launch
    bool head_fused = false; Val hz;                                         // nn.Dropout: [Dropout, Linear, Sigmoid] ran as one launch (hz = the linear output)
    void* loc_ws[4] = {nullptr, nullptr, nullptr, nullptr}; size_t loc_ws_bytes[4] = {0, 0, 0, 0};
    std::map<std::string, Val> bufs;
};

struct Run;
struct Op { int sidx; std::function<int(Run&)> fn; };

struct Prog {
    Net* net = nullptr;
    Val in;                                    // input descriptor the plan was compiled for (ext = EXT_X)
    std::map<int, MS> ms;
    vector<Op> fwd;
    vector<Op> bwd[2];                         // [0] updateGradInput only, [1] backward (gradInput + accGradParameters)
    bool have_bwd[2] = {false, false};
    Val out, gin[2];
    long draws = 0;                            // counter-stream draws of one forward
    int nstreams = 1;                          // 1 + side streams this plan uses
    vector<void*> ws; vector<size_t> ws_bytes; // split-K scratch per stream index (grow-only)
    unsigned long opt_epoch = 0;
    vector<std::pair<float*, long>> buckets;   // gradient buckets of the root Sequential (first module index order)
    vector<int> bucket_first;
    std::map<int, int> ups_first_op;           // module id of a layer behind a folded upsampling -> index of its first forward op
    struct BnRun { Mod* bn; Val sums; double cnt; long C; float mom; };
    vector<BnRun> bn_runs;                     // training-mode batch-norms of the forward pass (cg_net_apply_running)
    bool running_pending = false;
};

typedef void* (*alloc_fn_t)(void* user, size_t bytes);
typedef int (*hook_fn_t)(void* user, int what, void* buf, size_t count, int dtype, void* stream);

struct Region { const char* base; size_t bytes; };

struct Net {
    vector<std::unique_ptr<Mod>> mods;
    std::map<std::string, std::unique_ptr<Prog>> progs;
    Prog* last = nullptr;                      // plan of the most recent forward (what backward continues)
    vector<hipStream_t> side; vector<hipEvent_t> side_ev; hipEvent_t fork_ev = nullptr;
    hipEvent_t pack_fork_ev = nullptr, pack_ev = nullptr;   // weight re-packing beside the head of the forward pass (sync_packs)
    hipEvent_t wg_fork_ev[4] = {nullptr, nullptr, nullptr, nullptr}, wg_join_ev[4] = {nullptr, nullptr, nullptr, nullptr};   // weight-gradient streams (option wgrad_stream)
    alloc_fn_t alloc_fn = nullptr; void* alloc_user = nullptr;
    hook_fn_t hook = nullptr; void* hook_user = nullptr;
    void *comm_bn = nullptr, *comm_grad = nullptr; int world = 1; int sync_bn = 1; int bucket_overlap = 0;
    int hook_too = 0;                          // communicator collective AND host hook (cg_net_set_dp bucket_overlap & 2)
    vector<void*> owned;
    vector<Region> regions;
    bool params_dirty = true;
    int qmap[8] = {-1, 0, 0, 0, 0, 0, 0, 0};  // hardware-queue class of stream index t (ensure_streams)
    bool pack_inflight = false;               // sync_packs forked a re-packing that a LATER pass on another stream may still have to wait for
    int defer_running = 0;                     // run-time switch: batch-norm forwards leave the running statistics to cg_net_apply_running
    // cg_net_forward_pair: the second pass's stream (another hardware queue than the caller's), fork / "first pass complete" / join events
    hipStream_t pair_stream = nullptr; hipEvent_t pair_fork_ev = nullptr, pair_a_ev = nullptr, pair_join_ev = nullptr;
    bool pair_pending = false; float* pair_y = nullptr; int pair_ynd = 0, pair_yfmt = 0; long pair_ydims[4] = {0, 0, 0, 0};
    bool fresh_allocs = false;                 // library-owned buffers were allocated + zeroed since the last device synchronise
    // options
    int trace = 0, overlap_groups = 1, defer_wgrad = 1, winograd = 1, share_pool = 1, sampler_shared = 1, view_fuse = 1,
        cat_fuse = 1, stacking = 1, grouped = 1, fusion = 1, fuse_locnet = 1, pack_overlap = 1, head_fuse = 1, wgrad_stream = 1,
        wino_dsplit = 1;       // the F(2x2,3x3) data gradient in K slices when its unsplit launch is <= one workgroup per CU (cg_conv2d_ups2_wino_dgrad_split)
    long wino_min_tiles = 2048;
    int wino22 = 3;                            // F(2x2,2x2) for upsample2 -> conv3x3 above wino_min_tiles: bit 0 forward, 1 data gradient
                                               // (option "winograd22"; the weight gradient is 292 -> 257 us alone but no gain in the step: off)
    std::string trace_log;
    const KTable* K = &kRealTable;
    vector<void*> trace_streams;               // stream handle -> index in trace mode
    char err[512] = {0};
};

struct Run {
    Net* net; Prog* pr;
    hipStream_t st[8];                          // [0] the caller's, [1..3] side streams of branch groups, [4 + s] the weight-gradient stream beside stream s
    const float* x = nullptr; const float* gy = nullptr;
    uint64_t seed = 0, roff = 0; const uint64_t* rbase = nullptr;
    float scale = 1.f;
    int cur = 0;                                // stream index of the op being issued
    void* S(int sidx) const { return (void*)st[sidx]; }
    void* CS() const { return (void*)st[cur]; }
    float* P(const Val& v) const {
        if (v.none) return nullptr;
        if (v.ext == EXT_X) return (float*)((char*)x + v.off);
        if (v.ext == EXT_GY) return (float*)((char*)gy + v.off);
        return v.p;
    }
    void* W() const { return pr->ws[cur]; }
    size_t WB() const { return pr->ws_bytes[cur]; }
};

// ------------------------------------------------------------------------------------------------ tracing
std::string TraceLine::pname(const void* p) {
    if (!p) return "n";
    Net* n = g_cur_net;
    if (n)
        for (size_t k = 0; k < n->regions.size(); ++k) {
            const Region& r = n->regions[k];
            if ((const char*)p >= r.base && (const char*)p < r.base + r.bytes)
                return "r" + std::to_string(k) + "+" + std::to_string((const char*)p - r.base);
        }
    char b[40]; snprintf(b, sizeof b, "?%p", p);
    return b;
}
void TraceLine::ptr(const void* p) { s += "|"; s += pname(p); }
void TraceLine::stream(void* st) {
    Net* n = g_cur_net;
    int k = -1;
    if (n) for (size_t i = 0; i < n->trace_streams.size(); ++i) if (n->trace_streams[i] == st) k = (int)i;
    s += "|s" + std::to_string(k);
}
int TraceLine::done() {
    if (g_cur_net) { g_cur_net->trace_log += s; g_cur_net->trace_log += "\n"; }
    return 0;
}
void trace_note(Net* n, const std::string& s) { if (n->trace) { n->trace_log += s; n->trace_log += "\n"; } }

}  // namespace

// ================================================================================================ the compiler
namespace {

struct GCtx { vector<long> cur; };   // per-branch positions in the counter stream during a lockstep walk

struct Prep { Val x; int wsel; Val out; Geo g; bool ok = true; };
enum { W_PLAIN_F = 0, W_PH_F, W_PLAIN_B, W_PH_B, W_CANON };

static float* wsel(const Mod* m, int sel) {
    switch (sel) {
        case W_PLAIN_F: return m->wf;
        case W_PH_F: return m->wf_ph;
        case W_PLAIN_B: return m->wb ? m->wb : m->w;
        case W_PH_B: return m->wb_ph;
        default: return m->w;
    }
}

struct Compiler {
    Net* net;
    Prog* pr;
    vector<Op>* ops = nullptr;
    int cs = 0;              // stream index launches are emitted on
    long rng = 0;            // counter-stream draws so far, relative to the pass's start
    int dry = 0;             // > 0: shape / draw bookkeeping only (no allocation, no ops, no module state kept)
    bool failed = false;
    int pend[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // deferred weight-gradient reductions queued per stream index
    bool wg_used[4] = {false, false, false, false};   // the weight-gradient stream beside stream s has work to join
    bool wg_unjoined[4] = {false, false, false, false};   // ... that no gradient bucket has waited for yet
    bool acc_pass = false;        // compiling Module:backward (weight gradients deferred) rather than updateGradInput

    Compiler(Net* n, Prog* p) : net(n), pr(p) {}
    const KTable* K() const { return net->K; }
    Mod& M(int id) { return *net->mods[id]; }
    MS& S(const Mod& m) { return pr->ms[m.id]; }
    int err(const char* fmt, ...) {
        va_list ap; va_start(ap, fmt); vsnprintf(net->err, sizeof net->err, fmt, ap); va_end(ap);
        failed = true;
        return 1;
    }

    // ---------------------------------------------------------------------------------------- memory
    void* alloc(size_t bytes) {
        if (dry) return nullptr;
        bytes = (bytes + 255) & ~(size_t)255;
        if (bytes == 0) bytes = 256;
        void* p = nullptr;
        if (net->trace) {
            p = calloc(1, bytes);   // never dereferenced by the recording stubs; a real address keeps regions disjoint
        } else if (net->alloc_fn) {
            p = net->alloc_fn(net->alloc_user, bytes);      // host allocator: zero-initialised device memory
        } else {
            // the memset is ordered on the null stream only: cg_net_forward / _backward synchronise once after compiling (fresh_allocs)
            if (hipMalloc(&p, bytes) != hipSuccess || hipMemset(p, 0, bytes) != hipSuccess) p = nullptr;
            else net->fresh_allocs = true;
        }
        if (!p) { err("cg_net: allocation of %zu bytes failed", bytes); return nullptr; }
        if (!net->alloc_fn || net->trace) net->owned.push_back(p);
        net->regions.push_back(Region{(const char*)p, bytes});
        return p;
    }
    Val anon(std::initializer_list<long> dims, int fmt = PLAIN) {
        long n = 1; for (long x : dims) n *= x;
        Val v = mkval((float*)alloc((size_t)n * 4), dims, fmt);
        return v;
    }
    Val anon_like(const Val& x, int fmt) {
        Val v = x; v.p = (float*)alloc((size_t)x.numel() * 4); v.ext = 0; v.off = 0; v.fmt = fmt; v.ups = 0; v.blk = 0; v.gi = v.gc = 0;
        return v;
    }
    // Module._get: one persistent buffer per (module, role, element count)
    Val buf(const Mod& m, const std::string& role, std::initializer_list<long> dims, int fmt = PLAIN, size_t elem = 4) {
        long n = 1; for (long x : dims) n *= x;
        std::string key = role + "#" + std::to_string(n);
        MS& s = S(m);
        auto it = s.bufs.find(key);
        if (it == s.bufs.end()) {
            Val v = mkval((float*)alloc((size_t)n * elem), dims, fmt);
            s.bufs[key] = v;
            return v;
        }
        Val& b = it->second;
        Val want = mkval(nullptr, dims, fmt);
        if (!b.same_shape(want) || b.fmt != fmt) {
            const bool keep = want.nd && b.nd && want.d[0] == b.d[0];
            Val nb = reshape(b, dims, fmt, 0, keep);
            b = nb;
        }
        return b;
    }
    Val buf_like(const Mod& m, const std::string& role, const Val& like, int fmt) {
        switch (like.nd) {
            case 1: return buf(m, role, {like.d[0]}, fmt);
            case 2: return buf(m, role, {like.d[0], like.d[1]}, fmt);
            case 3: return buf(m, role, {like.d[0], like.d[1], like.d[2]}, fmt);
            default: return buf(m, role, {like.d[0], like.d[1], like.d[2], like.d[3]}, fmt);
        }
    }
    void ws_need(size_t bytes) {   // Workspace.get: grow-only scratch of the current stream
        if (dry) return;
        if (bytes < 4096) bytes = 4096;
        if ((int)pr->ws.size() <= cs) { pr->ws.resize(cs + 1, nullptr); pr->ws_bytes.resize(cs + 1, 0); }
        if (pr->ws_bytes[cs] < bytes) {
            size_t nb = bytes + bytes / 4;
            pr->ws[cs] = alloc(nb);
            pr->ws_bytes[cs] = nb;
        }
    }
    template <class F> void emit(F f) {
        if (dry) return;
        ops->push_back(Op{cs, std::function<int(Run&)>(f)});
    }
    void use_stream(int sidx) { if (!dry && sidx + 1 > pr->nstreams) pr->nstreams = sidx + 1; }

    // ---------------------------------------------------------------------------------------- layout helpers
    Val materialise(const Val& x) {
        if (!x.ups) return x;
        long N = x.d[0], C = x.d[1], H = x.d[2], W = x.d[3];
        Val out = anon({N, C, H, W}, NHWC);
        const KTable* k = K();
        emit([=](Run& c) { return k->upsample2x_forward(c.CS(), c.P(x), c.P(out), (int)N, (int)(H >> 1), (int)(W >> 1), (int)C); });
        return out;
    }
    // NOTE: closures never read Compiler state at run time; the stream index an op was emitted on (Op::sidx) reaches the closure
    // through Run::cur, set by the runner before each op (Run::CS() = that stream, Run::W() its scratch).

    Val as_nhwc(const Val& x, bool keep_ups = false) {
        if (x.fmt == NHWC) return (keep_ups || !x.ups) ? x : materialise(x);
        if (x.nd != 4) { err("cg_net: expected a 4-D feature map"); return x; }
        long N = x.d[0], C = x.d[1], H = x.d[2], W = x.d[3];
        Val out = anon({N, C, H, W}, NHWC);
        const KTable* k = K();
        emit([=](Run& c) { return k->nchw_to_nhwc(c.CS(), c.P(x), c.P(out), (int)N, (int)C, (int)H, (int)W); });
        return out;
    }
    Val as_plain(const Val& x0) {
        if (x0.fmt == PLAIN) return x0;
        Val x = materialise(x0);
        long N = x.d[0], C = x.d[1], H = x.d[2], W = x.d[3];
        Val out = anon({N, C, H, W}, PLAIN);
        const KTable* k = K();
        emit([=](Run& c) { return k->nhwc_to_nchw(c.CS(), c.P(x), c.P(out), (int)N, (int)C, (int)H, (int)W); });
        return out;
    }
    Val match_fmt(const Val& g, const Val& like) {
        if (g.fmt == like.fmt) return g;
        return like.fmt == NHWC ? as_nhwc(g) : as_plain(g);
    }

    // ---------------------------------------------------------------------------------------- stacking of identical branches
    vector<Val> split(const Val& Y, int G) {   // a stacked tensor ([G*N, ...], branch-major) as its G per-branch slices
        vector<Val> out;
        const long n = Y.phys() / G, N = Y.d[0] / G;
        for (int i = 0; i < G; ++i) {
            Val v = Y.at((size_t)i * n * 4);
            v.d[0] = N; v.blk = Y.key(); v.gi = i; v.gc = G;
            out.push_back(v);
        }
        return out;
    }
    bool stacked(const vector<Val>& xs, Val* out) {   // the block behind G per-branch tensors if they are exactly its slices in order
        if (xs.empty() || xs[0].is_tab || xs[0].none || !xs[0].blk || xs[0].gc != (int)xs.size()) return false;
        const Val& x0 = xs[0];
        for (size_t i = 0; i < xs.size(); ++i) {
            const Val& x = xs[i];
            if (x.is_tab || x.none || x.blk != x0.blk || x.gi != (int)i || !x.same_shape(x0) || x.fmt != x0.fmt || x.ups) return false;
        }
        if (out) {
            Val b = x0;   // slice 0 starts where the block starts
            b.d[0] = x0.d[0] * (long)xs.size(); b.blk = 0; b.gi = b.gc = 0;
            *out = b;
        }
        return true;
    }
    void seed_slices(const vector<Mod*>& mods, const std::string& role, const Val& shape_of, int fmt) {
        const int G = (int)mods.size();
        const long n = shape_of.numel();
        std::string key = role + "#" + std::to_string(n);
        vector<Val> cur; bool all = true;
        for (Mod* m : mods) {
            auto it = S(*m).bufs.find(key);
            if (it == S(*m).bufs.end() || !it->second.same_shape(shape_of) || it->second.fmt != fmt) { all = false; break; }
            cur.push_back(it->second);
        }
        if (all && stacked(cur, nullptr)) return;
        Val bs = shape_of; bs.d[0] *= G;
        Val block = buf_like(*mods[0], role + ".block", bs, fmt);
        vector<Val> sl = split(block, G);
        for (int i = 0; i < G; ++i) S(*mods[i]).bufs[key] = sl[i];
    }
    Val restack(const Mod& owner, vector<Val> xs, const Val& like) {   // copy G per-branch tensors into one block
        for (auto& x : xs) x = match_fmt(x, like);
        Val bs = xs[0]; bs.d[0] *= (long)xs.size(); bs.ups = 0;
        Val block = buf_like(owner, "restack.block", bs, xs[0].fmt);
        const size_t nb = (size_t)xs[0].phys() * 4;
        const KTable* k = K();
        for (size_t i = 0; i < xs.size(); ++i) {
            Val dst = block.at(i * nb), src = xs[i];
            emit([=](Run& c) { return k->memcpy_d2d(c.CS(), c.P(dst), c.P(src), nb); });
        }
        return block;
    }

    // ---------------------------------------------------------------------------------------- kernel-side weight copies
    // Allocated when a plan first needs them; refreshed at the start of the next pass after cg_net_params_changed (one batched
    // launch for the plain layers, cg_pack_conv_weight_ups2 (+ Winograd) for layers behind a folded upsampling).
    void need_plain(Mod& m, bool map, long C = 0, long H = 0, long W = 0) {
        if (dry) return;
        const long n = m.kind == K_LINEAR ? m.ia[0] * m.ia[1] : m.ia[0] * m.ia[1] * m.ia[2] * m.ia[3];
        const long taps = map ? H * W : m.kH() * m.kW();
        const int want = map ? 1 : 0;
        if (m.pk_map == want && (!map || (m.map_c == C && m.map_h == H && m.map_w == W))) return;
        if (!m.wf) m.wf = (float*)alloc((size_t)n * 4);
        if (taps > 1 && !m.wb) m.wb = (float*)alloc((size_t)n * 4);
        if (taps <= 1) m.wb = nullptr;     // 1x1 / linear: the canonical [out][in] matrix already is the backward operand
        m.pk_map = want; m.map_c = C; m.map_h = H; m.map_w = W;
        m.dirty_plain = true;
    }
    void need_ups(Mod& m, bool wino) {
        if (dry) return;
        if (ops == &pr->fwd && !pr->ups_first_op.count(m.id)) pr->ups_first_op[m.id] = (int)ops->size();
        if (!m.wf_ph) {
            const size_t n = cg_pack_conv_weight_ups2_floats((int)m.ia[1], (int)m.ia[0], (int)m.kH(), (int)((m.kH() - 1) / 2));
            m.wf_ph = (float*)alloc(n * 4); m.wb_ph = (float*)alloc(n * 4);
            m.dirty_ups = true;
        }
        if (wino && !m.wino) {
            m.wino = true;
            const size_t nu = cg_conv2d_ups2_wino_u_floats((int)m.ia[0], (int)m.ia[1]);
            m.u_fwd = (float*)alloc(nu * 4); m.u_bwd = (float*)alloc(nu * 4);
            m.dirty_ups = true;
        }
    }

    // ---------------------------------------------------------------------------------------- conv / linear preparation
    static bool can_fold_ups(const Mod& m) { return m.kH() == m.kW() && m.kH() % 2 == 1 && m.padH() == m.padW() && m.padH() == (m.kH() - 1) / 2; }
    bool use_wino22(Mod& m, const Val& x) {   // F(2x2,2x2) forward for a lazily upsampled input: 3x3, pad 1, even low-res grid, planes % 128
        if (!(net->winograd && (net->wino22 & 1) && m.kind == K_CONV && x.ups && m.kH() == 3 && m.kW() == 3 && m.padH() == 1 && m.padW() == 1)) return false;
        const long N = x.d[0], Hp = x.d[2] >> 1, Wp = x.d[3] >> 1;
        if (N * Hp * Wp / 4 < net->wino_min_tiles) return false;
        return cg_conv2d_ups2_wino22_supported((int)N, (int)Hp, (int)Wp, (int)m.ia[0], (int)m.ia[1]) != 0;
    }
    bool use_wino(Mod& m, const Val& x) {   // Winograd path for a lazily upsampled input: 5x5, pad 2, even low-res grid, planes % 128
        if (!(net->winograd && m.kind == K_CONV && x.ups && m.kH() == 5 && m.kW() == 5 && m.padH() == 2 && m.padW() == 2)) return false;
        const long N = x.d[0], Hp = x.d[2] >> 1, Wp = x.d[3] >> 1;
        if (N * Hp * Wp / 4 < net->wino_min_tiles) return false;
        return cg_conv2d_ups2_wino_supported((int)N, (int)Hp, (int)Wp, (int)m.ia[0], (int)m.ia[1], 5, 2) != 0;
    }
    static bool epilogue_ok(const Geo& g) {   // the skinny <= 4-plane 3x3 kernels cannot take a fused epilogue
        const bool skinny = g.ups == 0 && g.kH == 3 && g.kW == 3 && g.padH == 1 && g.padW == 1 && (g.Cout == 1 || g.Cout == 3) &&
                            (g.Cin == 64 || g.Cin == 128);
        return !skinny;
    }
    Prep prep_fwd(Mod& m, const Val& in) {
        Prep p;
        MS& s = S(m);
        if (m.kind == K_LINEAR) {
            Val x = in;
            const long o = m.ia[1];
            if (s.map_in && !(x.nd == 4 && x.fmt == NHWC && !x.ups && x.d[1] == s.mc && x.d[2] == s.mh && x.d[3] == s.mw)) s.map_in = false;
            if (s.map_in) {
                const long N = x.d[0];
                need_plain(m, true, s.mc, s.mh, s.mw);
                p.x = x; p.wsel = W_PLAIN_F; p.out = buf(m, "out", {N, o});
                p.g = Geo{(int)N, (int)s.mh, (int)s.mw, (int)s.mc, (int)o, (int)s.mh, (int)s.mw, 0, 0, 0};
                return p;
            }
            x = as_plain(x);
            const long N = x.d[0], i = x.numel() / x.d[0];
            need_plain(m, false);
            p.x = x; p.wsel = W_PLAIN_F; p.out = buf(m, "out", {N, o});
            p.g = Geo{(int)N, 1, 1, (int)i, (int)o, 1, 1, 0, 0, 0};
            return p;
        }
        Val x = as_nhwc(in, true);
        if (x.ups && !can_fold_ups(m)) x = materialise(x);
        const long N = x.d[0], H = x.d[2], W = x.d[3];
        const long Hp = H >> x.ups, Wp = W >> x.ups, Ho = H + 2 * m.padH() - m.kH() + 1, Wo = W + 2 * m.padW() - m.kW() + 1;
        if (x.d[1] != m.ia[0]) { err("cg_net: convolution %d got %ld input planes, expects %ld", m.id, x.d[1], m.ia[0]); }
        if (x.ups) { need_ups(m, false); p.wsel = W_PH_F; } else { need_plain(m, false); p.wsel = W_PLAIN_F; }
        p.x = x;
        p.out = buf(m, "out", {N, m.ia[1], Ho, Wo}, NHWC);
        p.g = Geo{(int)N, (int)Hp, (int)Wp, (int)m.ia[0], (int)m.ia[1], (int)m.kH(), (int)m.kW(), (int)m.padH(), (int)m.padW(), x.ups};
        return p;
    }
    // updateGradInput as a forward call on gradOutput with the backward-packed weights; ok == false: not this form (folded upsampling)
    Prep prep_gin(Mod& m, const Val& go) {
        Prep p;
        MS& s = S(m);
        if (m.kind == K_LINEAR) {
            Val dy = as_plain(go);
            const long N = dy.d[0], o = dy.numel() / dy.d[0], i = m.ia[0];
            p.x = dy;
            if (s.map_in) {   // dx comes out NHWC-flattened: wbT[co][(h*W+w)*C + c] (cg_pack_conv_weight_map)
                p.wsel = W_PLAIN_B; p.out = buf(m, "gin", {N, s.mc, s.mh, s.mw}, NHWC);
            } else {
                p.wsel = W_CANON; p.out = buf(m, "gin", {N, i});
            }
            p.g = Geo{(int)N, 1, 1, (int)o, (int)i, 1, 1, 0, 0, 0};
            return p;
        }
        const Val& x = s.x;
        if (x.ups) { p.ok = false; return p; }
        Val dy = as_nhwc(go);
        const long N = dy.d[0], Ho = dy.d[2], Wo = dy.d[3];
        p.x = dy; p.wsel = W_PLAIN_B;
        p.out = buf(m, "gin", {N, m.ia[0], x.d[2], x.d[3]}, NHWC);
        p.g = Geo{(int)N, (int)Ho, (int)Wo, (int)m.ia[1], (int)m.ia[0], (int)m.kH(), (int)m.kW(), (int)(m.kH() - 1 - m.padH()),
                  (int)(m.kW() - 1 - m.padW()), 0};
        return p;
    }
    struct PrepAcc { Val x, dy; Geo g; };
    PrepAcc prep_acc(Mod& m, const Val& go) {
        PrepAcc p;
        MS& s = S(m);
        p.x = s.x;
        if (m.kind == K_LINEAR) {
            p.dy = as_plain(go);
            if (s.map_in) p.g = Geo{(int)p.x.d[0], (int)s.mh, (int)s.mw, (int)s.mc, (int)m.ia[1], (int)s.mh, (int)s.mw, 0, 0, 0};
            else p.g = Geo{(int)p.x.d[0], 1, 1, (int)(p.x.numel() / p.x.d[0]), (int)m.ia[1], 1, 1, 0, 0, 0};
            return p;
        }
        p.dy = as_nhwc(go);
        const Val& x = s.x;
        p.g = Geo{(int)x.d[0], (int)(x.d[2] >> x.ups), (int)(x.d[3] >> x.ups), (int)m.ia[0], (int)m.ia[1], (int)m.kH(), (int)m.kW(),
                  (int)m.padH(), (int)m.padW(), x.ups};
        return p;
    }

    // ---------------------------------------------------------------------------------------- forward of one module
    Val fwd(Mod& m, const Val& in) {
        const KTable* k = K();
        MS& s = S(m);
        switch (m.kind) {
        case K_SEQ: {
            LocDesc ld;
            if (match_loc(m, in, ld)) return fwd_loc({&m}, in, ld)[0];
            S(m).loc_fused = false;
            return fwd_seq(m, in);
        }
        case K_CONCATTABLE: {
            Val t; t.is_tab = true; t.none = false;
            for (int c : m.kids) t.tab.push_back(fwd(M(c), in));
            s.out = t;
            return t;
        }
        case K_CONCAT: return fwd_concat(m, in);
        case K_LINEAR: case K_CONV: {
            if (m.kind == K_CONV) {
                Val x = as_nhwc(in, true);
                if (use_wino(m, x)) {
                    const long N = x.d[0], Hp = x.d[2] >> 1, Wp = x.d[3] >> 1, Ci = m.ia[0], Co = m.ia[1];
                    need_ups(m, true);
                    Val out = buf(m, "out", {N, Co, x.d[2] + 4 - 5 + 1, x.d[3] + 4 - 5 + 1}, NHWC);
                    Val v = buf(m, "wino_v", {(long)cg_conv2d_ups2_wino_v_floats((int)N, (int)Hp, (int)Wp, (int)Ci)});
                    Mod* mp = &m;
                    emit([=](Run& c) { return k->conv2d_ups2_wino_forward(c.CS(), c.P(x), mp->u_fwd, mp->b, c.P(out), c.P(v), (int)N, (int)Hp, (int)Wp, (int)Ci, (int)Co); });
                    s.x = x; s.out = out; s.use_wino = true;
                    return out;
                }
                s.use_wino = false; s.use_wino22 = false;
            }
            Prep p = prep_fwd(m, in);
            ws_need(cg_conv2d_workspace_bytes(GEO(p.g)));
            Mod* mp = &m;
            emit([=](Run& c) { return k->conv2d_forward(c.CS(), c.P(p.x), wsel(mp, p.wsel), mp->b, c.P(p.out), GEO(p.g), c.W(), c.WB()); });
            s.x = p.x; s.out = p.out;
            return p.out;
        }
        case K_PRELU: {
            Val x = materialise(in);
            Val out = buf_like(m, "out", x, x.fmt);
            const long n = x.phys(); Mod* mp = &m;
            emit([=](Run& c) { return k->prelu_forward(c.CS(), c.P(x), mp->w, c.P(out), n); });
            s.x = x; s.out = out;
            return out;
        }
        case K_LRELU: {
            Val x = materialise(in);
            Val out = buf_like(m, "out", x, x.fmt);
            const long n = x.phys(); const float sl = m.fa[0];
            emit([=](Run& c) { return k->leakyrelu_forward(c.CS(), c.P(x), c.P(out), sl, n); });
            s.x = x; s.out = out;
            return out;
        }
        case K_SIGMOID: {
            Val x = materialise(in);
            Val out = buf_like(m, "out", x, x.fmt);
            const long n = x.phys();
            emit([=](Run& c) { return k->sigmoid_forward(c.CS(), c.P(x), c.P(out), n); });
            s.out = out;
            return out;
        }
        case K_BN: {
            Val x = as_nhwc(in);
            const long N = x.d[0], C = x.d[1], H = x.d[2], W = x.d[3], Mr = N * H * W;
            Val out = buf(m, "out", {N, C, H, W}, NHWC);
            Mod* mp = &m; const float eps = m.fa[0], mom = m.fa[1];
            if (!m.train) {
                emit([=](Run& c) { return k->bn_forward_eval(c.CS(), c.P(x), c.P(out), mp->w, mp->b, mp->rmean, mp->rvar, Mr, (int)C, eps); });
            } else {
                Val sums = buf(m, "sums", {2 * C}, PLAIN, 8), sm = buf(m, "save_mean", {C}), sv = buf(m, "save_std", {C});
                emit([=](Run& c) { return k->bn_stats(c.CS(), c.P(x), Mr, (int)C, (double*)c.P(sums)); });
                emit_allreduce_sum(sums, 2 * C, 1);
                const double cnt = (double)Mr * dp_factor();
                s.count = cnt;
                emit([=](Run& c) { const bool df = c.net->defer_running != 0;
                                   return k->bn_forward(c.CS(), c.P(x), c.P(out), mp->w, mp->b, (const double*)c.P(sums), cnt, Mr, (int)C, eps, mom,
                                                        df ? nullptr : mp->rmean, df ? nullptr : mp->rvar, c.P(sm), c.P(sv)); });
                if (!dry) pr->bn_runs.push_back({mp, sums, cnt, C, mom});
            }
            s.x = x; s.out = out; s.bn_fused = false;
            return out;
        }
        case K_VIEW: {
            s.skip = false;
            Val x = as_plain(in);
            const long N = x.d[0];
            s.in_nd = x.nd; for (int i = 0; i < 4; ++i) s.in_shape[i] = x.d[i];
            Val out;
            if (m.ia[7] == 3) {
                const long C = m.ia[0], H = m.ia[1], W = m.ia[2];
                out = buf(m, "out", {N, C, H, W}, NHWC);
                emit([=](Run& c) { return k->nchw_to_nhwc(c.CS(), c.P(x), c.P(out), (int)N, (int)C, (int)H, (int)W); });
            } else {
                out = reshape(x, {N, m.ia[0]}, PLAIN);
            }
            s.out = out;
            return out;
        }
        case K_COPY: s.out = in; return in;
        case K_TRANSPOSE: {
            Val out;
            if (m.ia[0] == 0) {   // NCHW -> BHWD: with NHWC storage a relabeling of the same memory
                Val x = as_nhwc(in);
                out = reshape(x, {x.d[0], x.d[2], x.d[3], x.d[1]}, PLAIN, 0, true);
            } else {              // BHWD -> NCHW
                if (in.fmt != PLAIN) err("cg_net: nn.Transpose BHWD->NCHW expects a plain tensor");
                out = reshape(in, {in.d[0], in.d[3], in.d[1], in.d[2]}, NHWC, 0, true);
            }
            s.out = out;
            return out;
        }
        case K_UPS: {   // never materialised when a convolution consumes it
            Val x = as_nhwc(in);
            Val out = reshape(x, {x.d[0], x.d[1], 2 * x.d[2], 2 * x.d[3]}, NHWC, 1);
            s.out = out;
            return out;
        }
        case K_AVGPOOL: case K_MAXPOOL: {
            Val x = as_nhwc(in);
            const long N = x.d[0], C = x.d[1], H = x.d[2], W = x.d[3];
            Val out = buf(m, "out", {N, C, H / 2, W / 2}, NHWC);
            const bool mx = m.kind == K_MAXPOOL;
            emit([=](Run& c) { return (mx ? k->maxpool2_forward : k->avgpool2_forward)(c.CS(), c.P(x), c.P(out), (int)N, (int)H, (int)W, (int)C); });
            s.x = x; s.out = out;
            return out;
        }
        case K_SDROP: {
            Val x = as_nhwc(in);
            const long N = x.d[0], C = x.d[1], H = x.d[2], W = x.d[3];
            Val out = buf(m, "out", {N, C, H, W}, NHWC);
            const float p = m.fa[0];
            if (m.train) {
                Val noise = buf(m, "noise", {N, C});
                const long off = rng; rng += N * C;
                emit([=](Run& c) { return k->rng_bernoulli_dev(c.CS(), c.P(noise), N * C, 1.0f - p, 1.0f, c.seed, c.roff + (uint64_t)off, c.rbase); });
                emit([=](Run& c) { return k->mask_mul(c.CS(), c.P(x), c.P(noise), c.P(out), (int)N, H * W, (int)C, 1); });
                s.noise = noise;
            } else {
                const long n = x.phys();
                emit([=](Run& c) { return k->memcpy_d2d(c.CS(), c.P(out), c.P(x), (size_t)n * 4); });
                emit([=](Run& c) { return k->scale(c.CS(), c.P(out), 1.0f - p, n); });
            }
            s.out = out;
            return out;
        }
        case K_DROP: {
            Val x = materialise(in);
            if (!m.train) { s.out = x; return x; }
            const long n = x.phys(); const float p = m.fa[0];
            Val noise = buf_like(m, "noise", x, x.fmt), out = buf_like(m, "out", x, x.fmt);
            const long off = rng; rng += n;
            emit([=](Run& c) { return k->rng_bernoulli_dev(c.CS(), c.P(noise), n, 1.0f - p, 1.0f / (1.0f - p), c.seed, c.roff + (uint64_t)off, c.rbase); });
            emit([=](Run& c) { return k->mask_mul(c.CS(), c.P(x), c.P(noise), c.P(out), 1, n, 1, 0); });
            s.noise = noise; s.out = out;
            return out;
        }
        case K_AFFMAT: {
            Val p = as_plain(in);
            const long N = p.d[0];
            Val out = buf(m, "out", {N, 2, 3});
            const int r = (int)m.ia[0], sc = (int)m.ia[1], tr = (int)m.ia[2];
            emit([=](Run& c) { return k->affine_matrix_forward(c.CS(), c.P(p), c.P(out), (int)N, r, sc, tr); });
            s.p = p; s.out = out;
            return out;
        }
        case K_AFFGRID: {
            const long N = in.d[0], H = m.ia[0], W = m.ia[1];
            Val out = buf(m, "out", {N, H, W, 2});
            Val T = in;
            emit([=](Run& c) { return k->affine_grid_forward(c.CS(), c.P(T), c.P(out), (int)N, (int)H, (int)W); });
            s.out = out;
            return out;
        }
        case K_SAMPLER: {
            if (!in.is_tab || in.tab.size() != 2) { err("cg_net: nn.BilinearSamplerBHWD expects {images, grids}"); return in; }
            Val img = in.tab[0], grid = in.tab[1];
            if (img.fmt != PLAIN || grid.fmt != PLAIN) err("cg_net: sampler inputs must be BHWD tensors");
            const long N = img.d[0], Hi = img.d[1], Wi = img.d[2], C = img.d[3], Ho = grid.d[1], Wo = grid.d[2];
            Val out = buf(m, "out", {N, Ho, Wo, C});
            emit([=](Run& c) { return k->bilinear_sampler_forward(c.CS(), c.P(img), c.P(grid), c.P(out), (int)N, (int)Hi, (int)Wi, (int)C, (int)Ho, (int)Wo); });
            s.in_img = img; s.in_grid = grid; s.out = out; s.has_shared = false;
            return out;
        }
        }
        err("cg_net: module kind %d has no forward", m.kind);
        return in;
    }

    // data-parallel hooks -------------------------------------------------------------------------
    double dp_factor() const { return (net->world > 1 && net->sync_bn) ? (double)net->world : 1.0; }
    void emit_allreduce_sum(const Val& v, long count, int dtype);   // sync-BN: in place SUM over ranks, ordered on the stream

    // ---------------------------------------------------------------------------------------- lockstep forward of sibling modules
    static bool stackable(const Mod& m) {
        switch (m.kind) { case K_LRELU: case K_VIEW: case K_AVGPOOL: case K_MAXPOOL: case K_SDROP: case K_AFFMAT: case K_AFFGRID: return true; default: return false; }
    }
    vector<Val> gfwd_default(const vector<Mod*>& mods, const vector<Val>& ins, GCtx& ctx) {
        // one call per branch, each at its own position in the counter stream
        vector<Val> outs;
        for (size_t b = 0; b < mods.size(); ++b) {
            const long saved = rng; rng = ctx.cur[b];
            outs.push_back(fwd(*mods[b], ins[b]));
            ctx.cur[b] = rng; rng = saved;
        }
        return outs;
    }
    vector<Val> gfwd_stackable(const vector<Mod*>& mods, const vector<Val>& ins, GCtx& ctx) {
        Mod& m0 = *mods[0];
        Val X;
        if (!(net->stacking && stacked(ins, &X))) { S(m0).stk = 0; return gfwd_default(mods, ins, ctx); }
        vector<Val> ys = split(fwd(m0, X), (int)mods.size());
        S(m0).stk = ys[0].blk;
        for (size_t b = 0; b < mods.size(); ++b) S(*mods[b]).out = ys[b];
        return ys;
    }
    bool ran_stacked(Mod& m0) { MS& s = S(m0); return !s.out.none && !s.out.is_tab && s.out.blk && s.out.blk == s.stk; }

    vector<Val> gfwd(const vector<Mod*>& mods, const vector<Val>& ins, GCtx& ctx) {
        Mod& m0 = *mods[0];
        const KTable* k = K();
        const int G = (int)mods.size();
        switch (m0.kind) {
        case K_SEQ: {
            // sibling localisation branches on the SAME input (D32_st3's three transformers): one fused launch for the group
            LocDesc ld;
            bool same = G <= 4 && net->fuse_locnet != 2 && match_loc(m0, ins[0], ld);
            for (int b = 1; same && b < G; ++b) {
                LocDesc lb;
                same = ins[b].key() == ins[0].key() && ins[b].same_shape(ins[0]) && match_loc(*mods[b], ins[b], lb) && lb.S == ld.S && lb.Cin == ld.Cin &&
                       lb.P == ld.P && lb.ur == ld.ur && lb.us == ld.us && lb.ut == ld.ut && lb.slope == ld.slope;
            }
            if (same) return fwd_loc(mods, ins[0], ld);
            for (Mod* m : mods) S(*m).loc_fused = false;
            return gfwd_seq(mods, ins, ctx);
        }
        case K_CONCATTABLE: {
            const size_t nch = m0.kids.size();
            vector<vector<Val>> per_child;
            for (size_t j = 0; j < nch; ++j) {
                vector<Mod*> col; for (Mod* m : mods) col.push_back(&M(m->kids[j]));
                per_child.push_back(gfwd(col, ins, ctx));
            }
            vector<Val> outs;
            for (int b = 0; b < G; ++b) {
                Val t; t.is_tab = true; t.none = false;
                for (size_t j = 0; j < nch; ++j) t.tab.push_back(per_child[j][b]);
                S(*mods[b]).out = t;
                outs.push_back(t);
            }
            return outs;
        }
        case K_AVGPOOL: case K_MAXPOOL: {
            // sibling instances fed the SAME tensor (the localisation nets of D32_st3's three transformer branches all start by
            // pooling the trunk's output, models.lua:843) compute the same thing: one launch, shared result
            bool same = net->fusion && net->share_pool && G > 1 && !ins[0].is_tab;
            for (int b = 1; same && b < G; ++b) same = ins[b].key() == ins[0].key() && ins[b].same_shape(ins[0]) && ins[b].fmt == ins[0].fmt && ins[b].ups == ins[0].ups;
            if (same) {
                Val y = fwd(m0, ins[0]);
                for (Mod* m : mods) { S(*m).x = S(m0).x; S(*m).out = y; }
                S(m0).stk = 0; S(m0).shared_in = true;
                return vector<Val>(G, y);
            }
            S(m0).shared_in = false;
            return gfwd_stackable(mods, ins, ctx);
        }
        case K_SDROP: {
            // stacked form: every branch draws its own [N,C] mask at its own position of the counter stream (into one block),
            // then a single mask multiply runs over the stacked batch
            Val X;
            if (!(net->stacking && stacked(ins, &X))) { S(m0).stk = 0; return gfwd_default(mods, ins, ctx); }
            if (!m0.train) return gfwd_stackable(mods, ins, ctx);
            const long N = ins[0].d[0], C = ins[0].d[1], H = ins[0].d[2], W = ins[0].d[3];
            Val noise = buf(m0, "noise.block", {G * N, C});
            const float p = m0.fa[0];
            for (int b = 0; b < G; ++b) {
                const long off = ctx.cur[b]; ctx.cur[b] += N * C;
                Val nb = noise.at((size_t)b * N * C * 4);
                emit([=](Run& c) { return k->rng_bernoulli_dev(c.CS(), c.P(nb), N * C, 1.0f - p, 1.0f, c.seed, c.roff + (uint64_t)off, c.rbase); });
            }
            Val out = buf_like(m0, "out", X, NHWC);
            emit([=](Run& c) { return k->mask_mul(c.CS(), c.P(X), c.P(noise), c.P(out), (int)(G * N), H * W, (int)C, 1); });
            vector<Val> ys = split(out, G), ns = split(noise, G);
            S(m0).stk = ys[0].blk; S(m0).noise_block = noise;
            for (int b = 0; b < G; ++b) { S(*mods[b]).out = ys[b]; S(*mods[b]).noise = ns[b]; }
            return ys;
        }
        case K_LINEAR: case K_CONV: {
            vector<Prep> preps;
            for (int b = 0; b < G; ++b) preps.push_back(prep_fwd(*mods[b], ins[b]));
            bool same = G <= 4;
            for (int b = 1; b < G; ++b) same = same && preps[b].g == preps[0].g;
            bool wino = false;
            for (int b = 0; b < G; ++b) wino = wino || (mods[b]->kind == K_CONV && preps[b].x.ups && use_wino(*mods[b], preps[b].x));
            if (!same || wino) return gfwd_default(mods, ins, ctx);
            vector<Val> outs; for (auto& p : preps) outs.push_back(p.out);
            if (net->stacking && !stacked(outs, nullptr)) {   // outputs as slices of one block
                seed_slices(mods, "out", preps[0].out, preps[0].out.fmt);
                preps.clear();
                for (int b = 0; b < G; ++b) preps.push_back(prep_fwd(*mods[b], ins[b]));
            }
            const Geo g = preps[0].g;
            ws_need(cg_conv2d_workspace_bytes_grouped(G, GEO(g)));
            vector<Mod*> ms_ = mods;
            emit([=](Run& c) {
                const float *x[4], *w[4], *bi[4]; float* y[4];
                for (int b = 0; b < G; ++b) { x[b] = c.P(preps[b].x); w[b] = wsel(ms_[b], preps[b].wsel); bi[b] = ms_[b]->b; y[b] = c.P(preps[b].out); }
                return k->conv2d_forward_grouped(c.CS(), G, x, w, bi, y, GEO(g), c.W(), c.WB());
            });
            outs.clear();
            for (int b = 0; b < G; ++b) { S(*mods[b]).x = preps[b].x; S(*mods[b]).out = preps[b].out; S(*mods[b]).use_wino = false; S(*mods[b]).use_wino22 = false; outs.push_back(preps[b].out); }
            return outs;
        }
        case K_PRELU: {
            bool ok = net->stacking;
            for (auto& x : ins) ok = ok && !x.is_tab && !x.ups;
            if (ok) seed_slices(mods, "out", ins[0], ins[0].fmt);
            return gfwd_default(mods, ins, ctx);
        }
        case K_SAMPLER: {
            // sibling transformers sampling the SAME image tensor with stacked grids (D32_st3's branches): one launch over the
            // G*N samples; else one launch per branch into the slices of one block
            MS& s0 = S(m0);
            s0.has_shared = false;
            if (net->stacking) {
                const Val &img = ins[0].tab[0], &gr0 = ins[0].tab[1];
                const long N = img.d[0], Hi = img.d[1], Wi = img.d[2], C = img.d[3], Ho = gr0.d[1], Wo = gr0.d[2];
                vector<Val> grids; for (auto& i_ : ins) grids.push_back(i_.tab[1]);
                Val GR;
                bool sh = net->fusion && net->sampler_shared && stacked(grids, &GR) && img.fmt == PLAIN && GR.fmt == PLAIN;
                for (auto& i_ : ins) sh = sh && i_.tab[0].key() == img.key() && i_.tab[0].same_shape(img);
                if (sh) {
                    Val out = buf(m0, "out.block", {G * N, Ho, Wo, C});
                    emit([=](Run& c) { return k->bilinear_sampler_forward_shared(c.CS(), G, c.P(img), c.P(GR), c.P(out), (int)N, (int)Hi, (int)Wi, (int)C, (int)Ho, (int)Wo); });
                    vector<Val> ys = split(out, G);
                    for (int b = 0; b < G; ++b) { S(*mods[b]).out = ys[b]; S(*mods[b]).in_img = ins[b].tab[0]; S(*mods[b]).in_grid = ins[b].tab[1]; }
                    s0.has_shared = true; s0.shared_out = out; s0.shared_grids = GR;
                    return ys;
                }
                Val shp = mkval(nullptr, {N, Ho, Wo, C});
                seed_slices(mods, "out", shp, PLAIN);
            }
            return gfwd_default(mods, ins, ctx);
        }
        default:
            if (stackable(m0)) return gfwd_stackable(mods, ins, ctx);
            return gfwd_default(mods, ins, ctx);
        }
    }

    // ---------------------------------------------------------------------------------------- nn.Sequential: segments
    vector<Seg> plan(const Mod& q) {
        vector<Seg> out;
        const int n = (int)q.kids.size();
        int i = 0;
        while (i < n) {
            Mod& m = M(q.kids[i]);
            int kind = S_ONE, j = i + 1;
            if (net->fusion) {
                Mod* nx = i + 1 < n ? &M(q.kids[i + 1]) : nullptr;
                Mod* nx2 = i + 2 < n ? &M(q.kids[i + 2]) : nullptr;
                Mod* nx3 = i + 3 < n ? &M(q.kids[i + 3]) : nullptr;
                if (m.kind == K_CONV && nx && nx->kind == K_BN && nx->train && nx2 && nx2->kind == K_PRELU) { kind = S_GEMM_BN_ACT; j = i + 3; }
                else if (m.is_gemm() && nx && nx->is_act() && !(nx2 && nx2->is_pool())) { kind = S_GEMM_ACT; j = i + 2; }
                else if (m.kind == K_VIEW && m.ia[7] == 1 && nx && nx->kind == K_LINEAR && net->view_fuse) {
                    if (nx2 && nx2->is_act() && !(nx3 && nx3->is_pool())) { kind = S_VIEW_GEMM_ACT; j = i + 3; }
                    else { kind = S_VIEW_GEMM; j = i + 2; }
                } else if (m.is_act() && nx && nx->is_pool()) {
                    const bool drop = nx2 && nx2->kind == K_SDROP && nx2->train;
                    kind = S_ACT_POOL; j = i + (drop ? 3 : 2);
                } else if (net->head_fuse && net->cat_fuse && m.kind == K_CONCAT && nx && nx->kind == K_SDROP && nx->train) {
                    kind = S_CAT_DROP; j = i + 2;
                } else if (net->head_fuse && m.kind == K_DROP && m.train && nx && nx->kind == K_LINEAR && nx->ia[1] <= 4 && nx2 && nx2->kind == K_SIGMOID) {
                    kind = S_HEAD; j = i + 3;
                }
            }
            out.push_back(Seg{kind, i, j});
            i = j;
        }
        return out;
    }
    Val fwd_seq(Mod& q, const Val& in) {
        Val cur = in;
        vector<Seg> pl = plan(q);
        S(q).ran = pl; S(q).ran_set = true;
        for (const Seg& sg : pl) {
            auto kid = [&](int t) -> Mod* { return &M(q.kids[t]); };
            switch (sg.kind) {
            case S_ONE: cur = fwd(*kid(sg.i), cur); break;
            case S_GEMM_ACT: cur = fwd_gemm_act({kid(sg.i)}, {kid(sg.i + 1)}, {cur}, nullptr)[0]; break;
            case S_ACT_POOL: cur = fwd_act_pool({kid(sg.i)}, {kid(sg.i + 1)}, sg.j - sg.i == 3 ? vector<Mod*>{kid(sg.i + 2)} : vector<Mod*>{}, {cur}, nullptr)[0]; break;
            case S_VIEW_GEMM: cur = fwd_view_gemm({kid(sg.i)}, {kid(sg.i + 1)}, {}, {cur}, nullptr)[0]; break;
            case S_VIEW_GEMM_ACT: cur = fwd_view_gemm({kid(sg.i)}, {kid(sg.i + 1)}, {kid(sg.i + 2)}, {cur}, nullptr)[0]; break;
            case S_CAT_DROP: cur = fwd_concat(*kid(sg.i), cur, kid(sg.i + 1)); break;
            case S_HEAD: cur = fwd_head(*kid(sg.i), *kid(sg.i + 1), *kid(sg.i + 2), cur); break;
            default: cur = fwd_gemm_bn_act(*kid(sg.i), *kid(sg.i + 1), *kid(sg.i + 2), cur); break;
            }
            if (failed) break;
        }
        S(q).out = cur;
        return cur;
    }
    vector<Val> gfwd_seq(const vector<Mod*>& qs, const vector<Val>& ins, GCtx& ctx) {
        vector<Val> cur = ins;
        vector<Seg> pl = plan(*qs[0]);
        for (Mod* q : qs) { S(*q).ran = pl; S(*q).ran_set = true; }
        auto col = [&](int t) { vector<Mod*> c; for (Mod* q : qs) c.push_back(&M(q->kids[t])); return c; };
        for (const Seg& sg : pl) {
            switch (sg.kind) {
            case S_ONE: cur = gfwd(col(sg.i), cur, ctx); break;
            case S_GEMM_ACT: cur = fwd_gemm_act(col(sg.i), col(sg.i + 1), cur, &ctx); break;
            case S_ACT_POOL: cur = fwd_act_pool(col(sg.i), col(sg.i + 1), sg.j - sg.i == 3 ? col(sg.i + 2) : vector<Mod*>{}, cur, &ctx); break;
            case S_VIEW_GEMM: cur = fwd_view_gemm(col(sg.i), col(sg.i + 1), {}, cur, &ctx); break;
            case S_VIEW_GEMM_ACT: cur = fwd_view_gemm(col(sg.i), col(sg.i + 1), col(sg.i + 2), cur, &ctx); break;
            case S_CAT_DROP: case S_HEAD: for (int t = sg.i; t < sg.j; ++t) cur = gfwd(col(t), cur, ctx); break;   // no lockstep form: module by module
            default: {   // not a lockstep case on the path: branch after branch, each at its own stream position
                vector<Val> outs;
                for (size_t b = 0; b < qs.size(); ++b) {
                    const long saved = rng; rng = ctx.cur[b];
                    outs.push_back(fwd_gemm_bn_act(M(qs[b]->kids[sg.i]), M(qs[b]->kids[sg.i + 1]), M(qs[b]->kids[sg.i + 2]), cur[b]));
                    ctx.cur[b] = rng; rng = saved;
                }
                cur = outs;
            } break;
            }
            if (failed) break;
        }
        for (size_t b = 0; b < qs.size(); ++b) S(*qs[b]).out = cur[b];
        return cur;
    }

    // [conv|linear, PReLU|LeakyReLU] x G branches: one (grouped) GEMM launch whose epilogue writes the pre-activation (what the
    // activation's backward needs) and the activation
    vector<Val> fwd_gemm_act(const vector<Mod*>& convs, const vector<Mod*>& acts, const vector<Val>& xs, GCtx* ctx) {
        const KTable* k = K();
        const int G = (int)convs.size();
        vector<Prep> preps;
        for (int b = 0; b < G; ++b) preps.push_back(prep_fwd(*convs[b], xs[b]));
        bool wino = false, ok = G <= 4;
        for (int b = 0; b < G; ++b) {
            wino = wino || (convs[b]->kind == K_CONV && preps[b].x.ups && use_wino(*convs[b], preps[b].x));
            ok = ok && preps[b].g == preps[0].g && epilogue_ok(preps[b].g);
        }
        if (!ok || wino) {
            if (G == 1) return {fwd(*acts[0], fwd(*convs[0], xs[0]))};
            return gfwd(acts, gfwd(convs, xs, *ctx), *ctx);
        }
        if (G > 1 && net->stacking) {
            vector<Val> outs; for (auto& p : preps) outs.push_back(p.out);
            if (!stacked(outs, nullptr)) {
                seed_slices(convs, "out", preps[0].out, preps[0].out.fmt);
                preps.clear();
                for (int b = 0; b < G; ++b) preps.push_back(prep_fwd(*convs[b], xs[b]));
            }
            seed_slices(acts, "out", preps[0].out, preps[0].out.fmt);
        }
        vector<Val> ys;
        for (int b = 0; b < G; ++b) ys.push_back(buf_like(*acts[b], "out", preps[b].out, preps[b].out.fmt));
        const Geo g = preps[0].g;
        const int code = acts[0]->kind == K_PRELU ? 1 : 2;
        const float slope = code == 1 ? 0.f : acts[0]->fa[0];
        ws_need(cg_conv2d_workspace_bytes_grouped(G, GEO(g)));
        vector<Mod*> cv = convs, ac = acts;
        emit([=](Run& c) {
            const float *x[4], *w[4], *bi[4], *al[4]; float *y[4], *ya[4];
            for (int b = 0; b < G; ++b) {
                x[b] = c.P(preps[b].x); w[b] = wsel(cv[b], preps[b].wsel); bi[b] = cv[b]->b; y[b] = c.P(preps[b].out);
                al[b] = ac[b]->w; ya[b] = c.P(ys[b]);
            }
            return k->conv2d_forward_ex(c.CS(), G, x, w, bi, y, GEO(g), code, slope, code == 1 ? al : nullptr, ya, nullptr, c.W(), c.WB());
        });
        for (int b = 0; b < G; ++b) {
            MS& sc = S(*convs[b]); sc.x = preps[b].x; sc.out = preps[b].out; sc.use_wino = false; sc.use_wino22 = false;
            MS& sa = S(*acts[b]); sa.x = preps[b].out; sa.out = ys[b];
        }
        if (G > 1 && stackable(*acts[0])) {   // let the parameter-free activation run its backward as one stacked launch
            vector<Val> po; for (auto& p : preps) po.push_back(p.out);
            Val Xs, Ys;
            if (stacked(po, &Xs) && stacked(ys, &Ys)) { S(*acts[0]).x = Xs; S(*acts[0]).stk = ys[0].blk; }
            else S(*acts[0]).stk = 0;
        }
        return ys;
    }

    // [View(C*H*W), Linear, (activation)] x G branches (models.lua:696-698, 849-851): the linear layer consumes the NHWC map
    vector<Val> fwd_view_gemm(const vector<Mod*>& views, const vector<Mod*>& lins, const vector<Mod*>& acts, vector<Val> xs, GCtx* ctx) {
        const int G = (int)views.size();
        const Val& x0 = xs[0];
        bool ok = !x0.is_tab && x0.nd == 4 && x0.fmt == NHWC && !x0.ups;
        long C = 0, H = 0, W = 0;
        if (ok) {
            C = x0.d[1]; H = x0.d[2]; W = x0.d[3];
            ok = H * W <= 64 && C % 16 == 0 && views[0]->ia[0] == C * H * W && lins[0]->ia[0] == C * H * W;
            for (auto& x : xs) ok = ok && !x.is_tab && x.same_shape(x0) && x.fmt == NHWC && !x.ups;
        }
        for (int b = 0; b < G; ++b) {
            S(*views[b]).skip = ok;
            MS& sl = S(*lins[b]); sl.map_in = ok; sl.mc = C; sl.mh = H; sl.mw = W;
        }
        if (ok) {
            for (int b = 0; b < G; ++b) { MS& sv = S(*views[b]); sv.out = Val(); sv.in_nd = 4; for (int i = 0; i < 4; ++i) sv.in_shape[i] = xs[b].d[i]; }
        } else {
            if (G == 1) xs = {fwd(*views[0], xs[0])}; else xs = gfwd(views, xs, *ctx);
        }
        if (!acts.empty()) return fwd_gemm_act(lins, acts, xs, ctx);
        if (G == 1) return {fwd(*lins[0], xs[0])};
        return gfwd(lins, xs, *ctx);
    }

    // [PReLU|LeakyReLU, Pool 2x2, (SpatialDropout, training)] x G branches in one pass over the stacked input
    vector<Val> fwd_act_pool(const vector<Mod*>& acts, const vector<Mod*>& pools, const vector<Mod*>& drops, const vector<Val>& xs, GCtx* ctx) {
        const KTable* k = K();
        const int G = (int)acts.size();
        Mod &a0 = *acts[0], &p0 = *pools[0];
        Mod* d0 = drops.empty() ? nullptr : drops[0];
        const Val& x0 = xs[0];
        Val X; bool haveX = false;
        long N = 0, C = 0, H = 0, W = 0;
        if (!x0.is_tab && x0.nd == 4 && x0.fmt == NHWC && !x0.ups && G <= 4) {
            N = x0.d[0]; C = x0.d[1]; H = x0.d[2]; W = x0.d[3];
            if (C % 4 == 0 && H % 2 == 0 && W % 2 == 0) {
                if (G == 1) { X = x0; haveX = true; }
                else haveX = net->stacking && stacked(xs, &X);
            }
        }
        if (!haveX) {   // the separate modules
            S(a0).fused = false;
            vector<vector<Mod*>> chain = {acts, pools};
            if (!drops.empty()) chain.push_back(drops);
            vector<Val> cur = xs;
            for (auto& colm : chain) {
                if (G == 1) cur = {fwd(*colm[0], cur[0])}; else cur = gfwd(colm, cur, *ctx);
            }
            return cur;
        }
        const int code = a0.kind == K_PRELU ? 1 : 2;
        const float slope = code == 1 ? 0.f : a0.fa[0];
        Mod& last = d0 ? *d0 : p0;
        Val out = buf(last, "out.fused", {G * N, C, H / 2, W / 2}, NHWC);
        Val mask; bool have_mask = false;
        if (d0) {
            mask = buf(*d0, "noise.block", {G * N, C}); have_mask = true;
            const float p = d0->fa[0];
            if (G == 1) {
                const long off = rng; rng += N * C;
                emit([=](Run& c) { return k->rng_bernoulli_dev(c.CS(), c.P(mask), N * C, 1.0f - p, 1.0f, c.seed, c.roff + (uint64_t)off, c.rbase); });
            } else {
                long offs[4] = {0, 0, 0, 0};
                for (int b = 0; b < G; ++b) { offs[b] = ctx->cur[b]; ctx->cur[b] += N * C; }
                const long o0 = offs[0], o1 = offs[1], o2 = offs[2], o3 = offs[3];
                emit([=](Run& c) {
                    // base-relative positions: the kernel adds *rbase to every one of them
                    return k->rng_bernoulli_dev_grouped(c.CS(), c.P(mask), N * C, G, 1.0f - p, 1.0f, c.seed, c.roff + (uint64_t)o0, G > 1 ? c.roff + (uint64_t)o1 : 0,
                                                        G > 2 ? c.roff + (uint64_t)o2 : 0, G > 3 ? c.roff + (uint64_t)o3 : 0, c.rbase);
                });
            }
        }
        const int pool_max = p0.kind == K_MAXPOOL ? 1 : 0;
        vector<Mod*> ac = acts;
        emit([=](Run& c) {
            const float* al[4];
            for (int b = 0; b < G; ++b) al[b] = ac[b]->w;
            return k->act_pool2_mask_forward(c.CS(), c.P(X), c.P(out), have_mask ? c.P(mask) : nullptr, G, (int)N, (int)H, (int)W, (int)C, code, slope,
                                             code == 1 ? al : nullptr, pool_max);
        });
        vector<Val> outs = G == 1 ? vector<Val>{out} : split(out, G);
        vector<Val> masks;
        if (have_mask) masks = G == 1 ? vector<Val>{mask} : split(mask, G);
        for (int b = 0; b < G; ++b) {
            MS& sa = S(*acts[b]); sa.x = xs[b]; sa.out = Val();
            MS& sp = S(*pools[b]); sp.x = Val(); sp.out = d0 ? Val() : outs[b];
            if (d0) { MS& sd = S(*drops[b]); sd.noise = masks[b]; sd.out = outs[b]; }
        }
        MS& s0 = S(a0);
        s0.fused = true; s0.fG = G; s0.fX = X; s0.fmask = have_mask ? mask : Val(); s0.fN = N; s0.fC = C; s0.fH = H; s0.fW = W;
        return outs;
    }

    // [conv, SpatialBatchNormalization (training), PReLU] (models.lua:206-208, 212-214, 218-220): batch statistics from the GEMM /
    // Winograd epilogue, normalise + activate in one pass; the normalised tensor is not kept (the backward recomputes it)
    Val fwd_gemm_bn_act(Mod& conv, Mod& bn, Mod& act, const Val& in) {
        const KTable* k = K();
        Val x = as_nhwc(in, true);
        const long C = conv.ia[1];
        if (C % 4 != 0) { S(bn).bn_fused = false; return fwd(act, fwd(bn, fwd(conv, in))); }
        long rows = 0; Val part, out;
        Mod *cp = &conv, *bp = &bn, *ap = &act;
        if (use_wino(conv, x)) {
            const long N = x.d[0], Hp = x.d[2] >> 1, Wp = x.d[3] >> 1, Ci = conv.ia[0];
            need_ups(conv, true);
            out = buf(conv, "out", {N, C, x.d[2], x.d[3]}, NHWC);
            Val v = buf(conv, "wino_v", {(long)cg_conv2d_ups2_wino_v_floats((int)N, (int)Hp, (int)Wp, (int)Ci)});
            rows = (long)cg_conv2d_ups2_wino_stats_rows((int)N, (int)Hp, (int)Wp, (int)Ci, (int)C);
            if (rows) part = buf(bn, "stats_part", {rows, 2, C});
            const bool hp = rows != 0;
            emit([=](Run& c) { return k->conv2d_ups2_wino_forward_stats(c.CS(), c.P(x), cp->u_fwd, cp->b, c.P(out), c.P(v), (int)N, (int)Hp, (int)Wp, (int)Ci, (int)C,
                                                                       hp ? c.P(part) : nullptr); });
            S(conv).x = x; S(conv).out = out; S(conv).use_wino = true;
        } else if (use_wino22(conv, x)) {
            // forward in F(2x2,2x2); the layer's state is that of the phase-folded direct path, which its backward takes
            Prep p = prep_fwd(conv, x);
            const long N = x.d[0], Hp = x.d[2] >> 1, Wp = x.d[3] >> 1, Ci = conv.ia[0];
            if (!dry && !conv.u22) {
                conv.u22 = (float*)alloc(cg_conv2d_ups2_wino22_u_floats((int)Ci, (int)C) * 4);
                conv.u22b = (float*)alloc(cg_conv2d_ups2_wino22_u_floats((int)Ci, (int)C) * 4);
                conv.dirty_ups = true;
            }
            out = p.out;
            Val v = buf(conv, "wino22_v", {(long)cg_conv2d_ups2_wino22_v_floats((int)N, (int)Hp, (int)Wp, (int)Ci)});
            rows = (long)cg_conv2d_ups2_wino_stats_rows((int)N, (int)Hp, (int)Wp, (int)Ci, (int)C);
            if (rows) part = buf(bn, "stats_part", {rows, 2, C});
            const bool hp = rows != 0;
            Val px = p.x;
            emit([=](Run& c) { return k->conv2d_ups2_wino22_forward_stats(c.CS(), c.P(px), cp->u22, cp->b, c.P(out), c.P(v), (int)N, (int)Hp, (int)Wp, (int)Ci, (int)C,
                                                                         hp ? c.P(part) : nullptr); });
            S(conv).x = p.x; S(conv).out = p.out; S(conv).use_wino = false; S(conv).use_wino22 = true;
        } else {
            Prep p = prep_fwd(conv, x);
            S(conv).use_wino22 = false;
            rows = epilogue_ok(p.g) ? (long)cg_conv2d_stats_rows(GEO(p.g)) : 0;
            ws_need(cg_conv2d_workspace_bytes(GEO(p.g)));
            out = p.out;
            if (rows) {
                part = buf(bn, "stats_part", {rows, 2, C});
                emit([=](Run& c) {
                    const float *xa[1] = {c.P(p.x)}, *wa[1] = {wsel(cp, p.wsel)}, *ba[1] = {cp->b}; float* ya[1] = {c.P(p.out)};
                    return k->conv2d_forward_ex(c.CS(), 1, xa, wa, ba, ya, GEO(p.g), 0, 0.f, nullptr, nullptr, c.P(part), c.W(), c.WB());
                });
            } else {
                emit([=](Run& c) { return k->conv2d_forward(c.CS(), c.P(p.x), wsel(cp, p.wsel), cp->b, c.P(p.out), GEO(p.g), c.W(), c.WB()); });
            }
            S(conv).x = p.x; S(conv).out = p.out; S(conv).use_wino = false;
        }
        const long N = out.d[0], H = out.d[2], W = out.d[3], Mr = N * H * W;
        Val sums = buf(bn, "sums", {2 * C}, PLAIN, 8), sm = buf(bn, "save_mean", {C}), sv = buf(bn, "save_std", {C});
        if (rows) emit([=](Run& c) { return k->bn_stats_finalize(c.CS(), c.P(part), rows, (int)C, (double*)c.P(sums)); });
        else emit([=](Run& c) { return k->bn_stats(c.CS(), c.P(out), Mr, (int)C, (double*)c.P(sums)); });
        emit_allreduce_sum(sums, 2 * C, 1);
        const double cnt = (double)Mr * dp_factor();
        S(bn).count = cnt;
        Val y = buf_like(act, "out", out, NHWC);
        const float eps = bn.fa[0], mom = bn.fa[1];
        emit([=](Run& c) { const bool df = c.net->defer_running != 0;
                           return k->bn_act_forward(c.CS(), c.P(out), c.P(y), bp->w, bp->b, (const double*)c.P(sums), cnt, Mr, (int)C, eps, mom, df ? nullptr : bp->rmean,
                                                    df ? nullptr : bp->rvar, c.P(sm), c.P(sv), ap->w); });
        if (!dry) pr->bn_runs.push_back({bp, sums, cnt, C, mom});
        MS& sb = S(bn); sb.x = out; sb.out = Val(); sb.bn_fused = true; sb.bnM = Mr; sb.bnC = C;
        MS& sa = S(act); sa.x = Val(); sa.out = y;
        return y;
    }

    // ---------------------------------------------------------------------------------------- fused localisation branch
    // nn.Sequential{ localisation net (models.lua:842-855), AffineTransformMatrixGenerator, AffineGridGeneratorBHWD } (:874-879) as
    // cg_locnet_forward / cg_locnet_backward: one launch each way instead of ~10 / ~12, off the GEMM path (csrc/locnet.hip).
    struct LocDesc { Mod *conv1, *conv2, *lin1, *lin2; int S, Cin, P, ur, us, ut, Hg, Wg; float slope; };
    bool match_loc(Mod& q, const Val& in, LocDesc& d) {
        if (!net->fuse_locnet || !net->fusion || q.kind != K_SEQ || q.kids.size() != 3) return false;
        Mod& L = M(q.kids[0]);
        static const int want[10] = {K_AVGPOOL, K_CONV, K_LRELU, K_CONV, K_LRELU, K_AVGPOOL, K_VIEW, K_LINEAR, K_LRELU, K_LINEAR};
        if (L.kind != K_SEQ || L.kids.size() != 10 || M(q.kids[1]).kind != K_AFFMAT || M(q.kids[2]).kind != K_AFFGRID) return false;
        for (int i = 0; i < 10; ++i) if (M(L.kids[i]).kind != want[i]) return false;
        if (in.is_tab || in.none || in.nd != 4 || in.fmt != NHWC || in.ups || in.d[2] != in.d[3] || in.d[2] % 4) return false;
        d.conv1 = &M(L.kids[1]); d.conv2 = &M(L.kids[3]); d.lin1 = &M(L.kids[7]); d.lin2 = &M(L.kids[9]);
        d.S = (int)(in.d[2] / 2); d.Cin = (int)in.d[1]; d.P = (int)d.lin2->ia[1];
        auto conv_ok = [](const Mod& c, long ci) { return c.ia[0] == ci && c.ia[1] == 16 && c.kW() == 3 && c.kH() == 3 && c.padW() == 1 && c.padH() == 1; };
        if (!conv_ok(*d.conv1, d.Cin) || !conv_ok(*d.conv2, 16)) return false;
        const long K3 = 16L * (d.S / 2) * (d.S / 2);
        if (M(L.kids[6]).ia[7] != 1 || M(L.kids[6]).ia[0] != K3 || d.lin1->ia[0] != K3 || d.lin1->ia[1] != 64 || d.lin2->ia[0] != 64) return false;
        d.slope = M(L.kids[2]).fa[0];
        if (M(L.kids[4]).fa[0] != d.slope || M(L.kids[8]).fa[0] != d.slope) return false;
        const Mod &am = M(q.kids[1]), &ag = M(q.kids[2]);
        d.ur = (int)am.ia[0]; d.us = (int)am.ia[1]; d.ut = (int)am.ia[2]; d.Hg = (int)ag.ia[0]; d.Wg = (int)ag.ia[1];
        if (d.P != d.ur + d.us + 2 * d.ut) return false;
        return cg_locnet_supported(d.S, d.Cin, d.P) != 0;
    }
    static void loc_weights(const vector<Mod*>& qs, Net* n, const float* w[48]) {
        for (size_t b = 0; b < qs.size(); ++b) {
            const Mod& L = *n->mods[qs[b]->kids[0]];
            const Mod *c1 = n->mods[L.kids[1]].get(), *c2 = n->mods[L.kids[3]].get(), *l1 = n->mods[L.kids[7]].get(), *l2 = n->mods[L.kids[9]].get();
            const float* v[12] = {c1->w, c1->b, c2->w, c2->b, l1->w, l1->b, l2->w, l2->b, c1->wf, c1->wb, c2->wf, c2->wb};
            for (int k = 0; k < 12; ++k) w[12 * b + k] = v[k];
        }
    }
    vector<Val> fwd_loc(const vector<Mod*>& qs, const Val& in, const LocDesc& d) {
        const KTable* k = K();
        const int G = (int)qs.size();
        Mod& q0 = *qs[0];
        const long N = in.d[0], S_ = d.S, Cin = d.Cin, K3 = 16L * (S_ / 2) * (S_ / 2);
        Val pooled = buf(q0, "loc.pooled", {G * N, Cin, S_, S_}, NHWC), h1 = buf(q0, "loc.h1", {G * N, 16, S_, S_}, NHWC),
            m2 = buf(q0, "loc.m2", {G * N, 16, S_, S_}, NHWC), h2 = buf(q0, "loc.h2", {G * N, K3}), h3 = buf(q0, "loc.h3", {G * N, 64}),
            prm = buf(q0, "loc.params", {G * N, d.P}), grid = buf(q0, "loc.grid", {G * N, d.Hg, d.Wg, 2});
        for (Mod* q : qs) {   // the two convolutions' packed copies (refreshed with every other layer's after a parameter update)
            const Mod& L = M(q->kids[0]);
            need_plain(M(L.kids[1]), false); need_plain(M(L.kids[3]), false);
        }
        Net* n_ = net; vector<Mod*> qv = qs; const LocDesc dd = d; Val x = in;
        emit([=](Run& c) {
            const float* w[48];
            loc_weights(qv, n_, w);
            return k->locnet_forward(c.CS(), G, (int)N, c.P(x), 1, w, dd.S, dd.Cin, dd.P, dd.ur, dd.us, dd.ut, dd.slope, dd.Hg, dd.Wg, c.P(pooled), c.P(h1),
                                     c.P(m2), c.P(h2), c.P(h3), c.P(prm), c.P(grid));
        });
        vector<Val> outs = G == 1 ? vector<Val>{grid} : split(grid, G);
        for (int b = 0; b < G; ++b) {
            MS& s = S(*qs[b]);
            s.out = outs[b]; s.loc_fused = true; s.locG = G; s.locN = N;
            s.ran_set = false;
        }
        MS& s0 = S(q0);
        s0.x = in;
        return outs;
    }
    vector<Val> bwd_loc(const vector<Mod*>& qs, const vector<Val>& gouts, bool acc) {
        const KTable* k = K();
        const int G = (int)qs.size();
        Mod& q0 = *qs[0];
        MS& s0 = S(q0);
        LocDesc d;
        if (!match_loc(q0, s0.x, d)) { err("cg_net: fused localisation branch lost its shape"); return gouts; }
        const long N = s0.locN, S_ = d.S, Cin = d.Cin, K3 = 16L * (S_ / 2) * (S_ / 2);
        Val GG;
        if (G == 1) GG = gouts[0];
        else if (!stacked(gouts, &GG)) GG = restack(q0, gouts, gouts[0]);
        Val pooled = buf(q0, "loc.pooled", {G * N, Cin, S_, S_}, NHWC), h1 = buf(q0, "loc.h1", {G * N, 16, S_, S_}, NHWC),
            m2 = buf(q0, "loc.m2", {G * N, 16, S_, S_}, NHWC), h2 = buf(q0, "loc.h2", {G * N, K3}), h3 = buf(q0, "loc.h3", {G * N, 64}),
            prm = buf(q0, "loc.params", {G * N, d.P});
        Val ga1 = buf(q0, "loc.ga1", {G * N, 16, S_, S_}, NHWC), ga2 = buf(q0, "loc.ga2", {G * N, 16, S_, S_}, NHWC), g3 = buf(q0, "loc.g3", {G * N, 64}),
            g4 = buf(q0, "loc.g4", {G * N, d.P}), gx = buf(q0, "loc.gx", {G * N, Cin, 2 * S_, 2 * S_}, NHWC);
        Net* n_ = net; vector<Mod*> qv = qs; const LocDesc dd = d;
        emit([=](Run& c) {
            const float* w[48];
            loc_weights(qv, n_, w);
            return k->locnet_backward(c.CS(), G, (int)N, w, dd.S, dd.Cin, dd.P, dd.ur, dd.us, dd.ut, dd.slope, dd.Hg, dd.Wg, c.P(h1), c.P(m2), c.P(h3), c.P(prm),
                                      c.P(GG), c.P(ga1), c.P(ga2), c.P(g3), c.P(g4), c.P(gx));
        });
        if (acc) {
            // weight gradients of the four layers on the GEMM path (grouped over the sibling branches), reductions deferred; their
            // gradOutputs come out of the launch above, so the fork sits behind it
            wg_fork();
            const long P_ = d.P;
            MS* s0p = &s0;
            wg_after_dgrad([this, k, G, N, S_, Cin, K3, P_, pooled, ga1, h1, ga2, h2, g3, h3, g4, n_, qv, s0p]() {
            MS& s0 = *s0p;
            struct W { int li; Val x, dy; Geo g; };
            const W ws_[4] = {
                {1, pooled, ga1, Geo{(int)N, (int)S_, (int)S_, (int)Cin, 16, 3, 3, 1, 1, 0}},
                {3, h1, ga2, Geo{(int)N, (int)S_, (int)S_, 16, 16, 3, 3, 1, 1, 0}},
                {7, h2, g3, Geo{(int)N, 1, 1, (int)K3, 64, 1, 1, 0, 0, 0}},
                {9, h3, g4, Geo{(int)N, 1, 1, 64, (int)P_, 1, 1, 0, 0, 0}}};
            for (int wi = 0; wi < 4; ++wi) {
                const W wv = ws_[wi];
                const bool defer = net->defer_wgrad;
                const size_t need = std::max<size_t>(cg_conv2d_wgrad_workspace_bytes_grouped(G, GEO(wv.g)), 4096);
                void* wsp = nullptr; size_t wsb = 0;
                if (defer) {
                    if (!dry && s0.loc_ws_bytes[wi] < need) { s0.loc_ws[wi] = alloc(need); s0.loc_ws_bytes[wi] = need; }
                    wsp = s0.loc_ws[wi]; wsb = s0.loc_ws_bytes[wi];
                    if (!dry) pend[cs]++;
                } else ws_need(need);
                const size_t xs = (size_t)wv.x.phys() / G * 4, ds = (size_t)wv.dy.phys() / G * 4;
                emit([=](Run& c) {
                    const float *x[4], *d_[4]; float *gw[4], *gb[4];
                    for (int b = 0; b < G; ++b) {
                        const Mod& L = *n_->mods[qv[b]->kids[0]];
                        const Mod* lay = n_->mods[L.kids[wv.li]].get();
                        x[b] = (const float*)((const char*)c.P(wv.x) + b * xs); d_[b] = (const float*)((const char*)c.P(wv.dy) + b * ds);
                        gw[b] = lay->gw; gb[b] = lay->gb;
                    }
                    if (defer) return k->conv2d_wgrad_grouped_deferred(c.CS(), G, x, d_, gw, gb, GEO(wv.g), c.scale, wsp, wsb);
                    return k->conv2d_wgrad_grouped(c.CS(), G, x, d_, gw, gb, GEO(wv.g), c.scale, c.W(), c.WB());
                });
            }
            });
        }
        vector<Val> outs = G == 1 ? vector<Val>{gx} : split(gx, G);
        for (int b = 0; b < G; ++b) S(*qs[b]).gin = outs[b];
        return outs;
    }

    // ---------------------------------------------------------------------------------------- nn.Concat(2)
    std::string signature(const Mod& m) {   // two modules with equal signatures run the same launches with the same geometry
        std::string s = "(" + std::to_string(m.kind);
        for (int i = 0; i < 8; ++i) s += "," + std::to_string(m.ia[i]);
        char b[64]; snprintf(b, sizeof b, ",%a,%a,%d", m.fa[0], m.fa[1], m.train); s += b;
        for (int c : m.kids) s += signature(M(c));
        return s + ")";
    }
    vector<vector<int>> branch_groups(const Mod& q) {
        vector<std::pair<std::string, vector<int>>> by;
        for (size_t i = 0; i < q.kids.size(); ++i) {
            std::string sg = signature(M(q.kids[i]));
            bool found = false;
            for (auto& e : by) if (e.first == sg) { e.second.push_back((int)i); found = true; break; }
            if (!found) by.push_back({sg, {(int)i}});
        }
        vector<vector<int>> groups;
        for (auto& e : by)
            for (size_t k0 = 0; k0 < e.second.size(); k0 += 4) groups.emplace_back(e.second.begin() + k0, e.second.begin() + std::min(k0 + 4, e.second.size()));
        std::sort(groups.begin(), groups.end(), [](const vector<int>& a, const vector<int>& b) { return a[0] < b[0]; });
        return groups;
    }
    // thunks[0] on the current stream, the others forked onto side streams and joined (D32_st3: the long chain of launch-bound
    // kernels of the three transformer branches hides under the big GEMMs of the two-convolution branch)
    void run_groups(const vector<std::function<void()>>& thunks) {
        const KTable* k = K();
        const bool multi = net->overlap_groups && thunks.size() > 1 && cs == 0 && thunks.size() <= 4;
        if (!multi) { for (auto& f : thunks) f(); return; }
        emit([=](Run& c) { return c.net->trace ? (trace_note(c.net, "event|record|fork|s0"), 0) : (hipEventRecord(c.net->fork_ev, (hipStream_t)c.S(0)) == hipSuccess ? 0 : 1); });
        for (size_t t = 1; t < thunks.size(); ++t) {
            const int sidx = (int)t;
            use_stream(sidx);
            cs = sidx;
            emit([=](Run& c) { return c.net->trace ? (trace_note(c.net, "event|wait|fork|s" + std::to_string(sidx)), 0)
                                                    : (hipStreamWaitEvent((hipStream_t)c.S(sidx), c.net->fork_ev, 0) == hipSuccess ? 0 : 1); });
            thunks[t]();
            flush_wgrad();
            emit([=](Run& c) { return c.net->trace ? (trace_note(c.net, "event|record|join" + std::to_string(sidx) + "|s" + std::to_string(sidx)), 0)
                                                    : (hipEventRecord(c.net->side_ev[sidx - 1], (hipStream_t)c.S(sidx)) == hipSuccess ? 0 : 1); });
            cs = 0;
        }
        thunks[0]();
        for (size_t t = 1; t < thunks.size(); ++t) {
            const int sidx = (int)t;
            emit([=](Run& c) { return c.net->trace ? (trace_note(c.net, "event|wait|join" + std::to_string(sidx) + "|s0"), 0)
                                                    : (hipStreamWaitEvent((hipStream_t)c.S(0), c.net->side_ev[sidx - 1], 0) == hipSuccess ? 0 : 1); });
        }
        (void)k;
    }
    void flush_wgrad() {   // reduce the weight-gradient partials queued on the CURRENT stream in one launch
        if (dry || !pend[cs]) return;
        const KTable* k = K();
        emit([=](Run& c) { return k->conv2d_wgrad_flush(c.CS()); });
        pend[cs] = 0;
    }
    // ---- weight gradients beside the data-gradient chain (option wgrad_stream).  accGradParameters of a layer reads the layer's saved
    // input and its gradOutput, and nothing before the optimiser step (or the layer's gradient bucket) reads what it writes; the data
    // gradient is what every layer in front of it waits for.  So in Module:backward the weight-gradient launches (GEMM, partial
    // reductions, the Winograd-domain path, the localisation nets' four) go to stream 4 + s, forked from stream s at the point where
    // gradOutput is complete, and are joined into stream 0 once, at the end of the pass.  The launch-bound kernels between two data
    // gradients (BN backward sums, activation / pooling backward, sampler and localisation backward) then run under a GEMM instead of
    // alone on the chip (profiles/r04_wgrad_stream_ab.txt).
    bool wg_on() const { return net->wgrad_stream && net->fusion && acc_pass && !dry && cs < 4; }
    void wg_fork() {   // call on stream s where the gradOutput the weight gradient reads is complete
        if (!wg_on()) return;
        const int s = cs;
        emit([=](Run& c) { return c.net->trace ? (trace_note(c.net, "event|record|wgfork" + std::to_string(s) + "|s" + std::to_string(s)), 0)
                                                : (hipEventRecord(c.net->wg_fork_ev[s], (hipStream_t)c.S(s)) == hipSuccess ? 0 : 1); });
    }
    int wg_enter() {   // later launches go to the weight-gradient stream; returns the stream index to hand back to wg_leave
        const int s = cs;
        if (!wg_on()) return s;
        use_stream(4 + s);
        cs = 4 + s;
        emit([=](Run& c) { return c.net->trace ? (trace_note(c.net, "event|wait|wgfork" + std::to_string(s) + "|s" + std::to_string(4 + s)), 0)
                                                : (hipStreamWaitEvent((hipStream_t)c.S(4 + s), c.net->wg_fork_ev[s], 0) == hipSuccess ? 0 : 1); });
        wg_used[s] = true; wg_unjoined[s] = true;
        return s;
    }
    void wg_leave(int s) { cs = s; }
    void wg_before_dgrad() { wg_fork(); }                                      // call in front of a layer's data-gradient launch ...
    void wg_after_dgrad(std::function<void()> f) {                             // ... and behind it, with the layer's weight-gradient launches
        const int s_ = wg_enter(); f(); wg_leave(s_);
    }
    void wg_join_all() {   // end of the pass: reductions still queued on the weight-gradient streams, then stream 0 waits for them
        if (dry) return;
        const int back = cs;
        for (int s = 0; s < 4; ++s) {
            if (!wg_used[s]) continue;
            cs = 4 + s;
            flush_wgrad();
            emit([=](Run& c) { return c.net->trace ? (trace_note(c.net, "event|record|wgjoin" + std::to_string(s) + "|s" + std::to_string(4 + s)), 0)
                                                    : (hipEventRecord(c.net->wg_join_ev[s], (hipStream_t)c.S(4 + s)) == hipSuccess ? 0 : 1); });
            cs = 0;
            emit([=](Run& c) { return c.net->trace ? (trace_note(c.net, "event|wait|wgjoin" + std::to_string(s) + "|s0"), 0)
                                                    : (hipStreamWaitEvent((hipStream_t)c.S(0), c.net->wg_join_ev[s], 0) == hipSuccess ? 0 : 1); });
            wg_used[s] = false;
        }
        cs = back;
    }
    long count_draws(Mod& m, const Val& in) {   // counter-stream draws of one forward of `m` (no launches, no state kept)
        Prog tmp; tmp.net = net;
        Prog* save = pr; const long r0 = rng; vector<Op>* so = ops;
        pr = &tmp; ++dry; rng = 0;
        fwd(m, in);
        const long n = rng;
        --dry; pr = save; rng = r0; ops = so;
        return n;
    }
    // drop: the nn.SpatialDropout (training) right behind this nn.Concat in its Sequential, to run inside the concat launch
    Val fwd_concat(Mod& q, const Val& in, Mod* drop = nullptr) {
        const KTable* k = K();
        const int nb = (int)q.kids.size();
        vector<Val> outs(nb);
        if (!net->grouped) {
            for (int i = 0; i < nb; ++i) outs[i] = as_nhwc(fwd(M(q.kids[i]), in));
        } else {
            vector<vector<int>> groups = branch_groups(q);
            vector<long> draws(nb), base(nb);
            long tot = 0;
            for (int i = 0; i < nb; ++i) { draws[i] = count_draws(M(q.kids[i]), in); base[i] = rng + tot; tot += draws[i]; }
            const long end = rng + tot;
            vector<std::function<void()>> thunks;
            for (auto& idxs : groups) {
                thunks.push_back([&, idxs]() {
                    vector<Val> res;
                    if (idxs.size() > 1) {
                        GCtx ctx; vector<Mod*> mods; vector<Val> ins;
                        for (int i : idxs) { ctx.cur.push_back(base[i]); mods.push_back(&M(q.kids[i])); ins.push_back(in); }
                        const long saved = rng;
                        res = gfwd(mods, ins, ctx);
                        rng = saved;
                    } else {
                        const long saved = rng; rng = base[idxs[0]];
                        res = {fwd(M(q.kids[idxs[0]]), in)};
                        rng = saved;
                    }
                    for (size_t t = 0; t < idxs.size(); ++t) outs[idxs[t]] = as_nhwc(res[t]);
                });
            }
            run_groups(thunks);
            rng = end;
        }
        const long N = outs[0].d[0], H = outs[0].d[2], W = outs[0].d[3];
        MS& s = S(q);
        s.sizes.clear();
        long Ct = 0; bool all4 = true;
        for (auto& o : outs) { s.sizes.push_back(o.d[1]); Ct += o.d[1]; all4 = all4 && o.d[1] % 4 == 0; }
        Val out = buf(q, "out", {N, Ct, H, W}, NHWC);
        s.cat_drop = false;
        if (drop && net->fusion && net->cat_fuse && nb <= 4 && all4) {   // concat and the dropout behind it in one launch
            vector<long> sz = s.sizes;
            MS& sd = S(*drop);
            Val noise = buf(*drop, "noise", {N, Ct});
            const long off = rng; rng += N * Ct;
            const float pdrop = drop->fa[0];
            emit([=](Run& c) {
                const float* src[4]; int cc[4];
                for (int i = 0; i < nb; ++i) { src[i] = c.P(outs[i]); cc[i] = (int)sz[i]; }
                return k->concat_channels_dropout(c.CS(), nb, src, cc, c.P(out), c.P(noise), (int)N, H * W, 1.0f - pdrop, 1.0f, c.seed, c.roff + (uint64_t)off, c.rbase);
            });
            s.cat_drop = true;
            s.out = Val();            // the undropped concatenation is never stored
            sd.noise = noise; sd.out = out;
            return out;
        }
        if (net->fusion && net->cat_fuse && nb <= 4 && all4) {   // one launch for all branches
            vector<long> sz = s.sizes;
            emit([=](Run& c) {
                const float* src[4]; int cc[4];
                for (int i = 0; i < nb; ++i) { src[i] = c.P(outs[i]); cc[i] = (int)sz[i]; }
                return k->concat_channels(c.CS(), nb, src, cc, c.P(out), N * H * W);
            });
        } else {
            long off = 0;
            for (int i = 0; i < nb; ++i) {
                const long ci = s.sizes[i], o_ = off; Val src = outs[i];
                emit([=](Run& c) { return k->copy_channels(c.CS(), c.P(src), c.P(out), N * H * W, (int)ci, 0, (int)Ct, (int)o_, (int)ci); });
                off += ci;
            }
        }
        s.out = out;
        if (drop) return fwd(*drop, out);
        return out;
    }
    Val bwd_concat(Mod& q, const Val& in, const Val& go_, bool acc, Mod* drop = nullptr) {
        const KTable* k = K();
        const int nb = (int)q.kids.size();
        MS& s = S(q);
        const bool masked = drop && s.cat_drop;                 // the split launch multiplies by the dropout mask
        Val go = go_;
        if (drop && !masked) go = bwd(*drop, s.out, go_, acc);   // the forward ran the two modules separately
        // channel slices of the gradient, one per branch; the slices of a lockstep group are the parts of one block
        Val g = as_nhwc(go);
        const long N = g.d[0], Ct = g.d[1], H = g.d[2], W = g.d[3];
        vector<Val> dst(nb); vector<bool> have(nb, false);
        vector<vector<int>> groups = net->grouped ? branch_groups(q) : vector<vector<int>>{};
        if (net->grouped && net->stacking) {
            for (auto& idxs : groups) {
                bool eq = idxs.size() > 1;
                for (int i : idxs) eq = eq && s.sizes[i] == s.sizes[idxs[0]];
                if (!eq) continue;
                Val block = buf(q, "gslice_block" + std::to_string(idxs[0]), {(long)idxs.size() * N, s.sizes[idxs[0]], H, W}, NHWC);
                vector<Val> sl = split(block, (int)idxs.size());
                for (size_t t = 0; t < idxs.size(); ++t) { dst[idxs[t]] = sl[t]; have[idxs[t]] = true; }
            }
        }
        bool all4 = true;
        for (int i = 0; i < nb; ++i) {
            if (!have[i]) dst[i] = buf(q, "gslice" + std::to_string(i), {N, s.sizes[i], H, W}, NHWC);
            all4 = all4 && s.sizes[i] % 4 == 0;
        }
        if (masked) {
            vector<long> sz = s.sizes;
            Val noise = S(*drop).noise;
            emit([=](Run& c) {
                float* d_[4]; int cc[4];
                for (int i = 0; i < nb; ++i) { d_[i] = c.P(dst[i]); cc[i] = (int)sz[i]; }
                return k->split_channels_masked(c.CS(), nb, c.P(g), c.P(noise), d_, cc, (int)N, H * W);
            });
            S(*drop).gin = Val();
        } else if (net->fusion && net->cat_fuse && nb <= 4 && all4) {
            vector<long> sz = s.sizes;
            emit([=](Run& c) {
                float* d_[4]; int cc[4];
                for (int i = 0; i < nb; ++i) { d_[i] = c.P(dst[i]); cc[i] = (int)sz[i]; }
                return k->split_channels(c.CS(), nb, c.P(g), d_, cc, N * H * W);
            });
        } else {
            long off = 0;
            for (int i = 0; i < nb; ++i) {
                const long ci = s.sizes[i], o_ = off; Val d_ = dst[i];
                emit([=](Run& c) { return k->copy_channels(c.CS(), c.P(g), c.P(d_), N * H * W, (int)Ct, (int)o_, (int)ci, 0, (int)ci); });
                off += ci;
            }
        }
        vector<Val> grads(nb);
        if (!net->grouped) {
            for (int i = 0; i < nb; ++i) grads[i] = as_nhwc(bwd(M(q.kids[i]), in, dst[i], acc));
        } else {
            vector<std::function<void()>> thunks;
            for (auto& idxs : groups) {
                thunks.push_back([&, idxs]() {
                    vector<Val> res;
                    if (idxs.size() > 1) {
                        GCtx ctx; vector<Mod*> mods; vector<Val> ins, gs;
                        for (int i : idxs) { ctx.cur.push_back(0); mods.push_back(&M(q.kids[i])); ins.push_back(in); gs.push_back(dst[i]); }
                        res = gbwd(mods, ins, gs, acc, ctx);
                    } else {
                        res = {bwd(M(q.kids[idxs[0]]), in, dst[idxs[0]], acc)};
                    }
                    for (size_t t = 0; t < idxs.size(); ++t) grads[idxs[t]] = as_nhwc(res[t]);
                });
            }
            run_groups(thunks);
        }
        // gradInput = ((g0 + g1) + g2) + g3
        Val accv;
        if (net->fusion && net->cat_fuse && nb >= 2 && nb <= 4 && grads[0].phys() % 4 == 0) {
            accv = buf_like(q, "gsum", grads[0], NHWC);
            const long n = accv.phys();
            emit([=](Run& c) {
                const float* src[4];
                for (int i = 0; i < nb; ++i) src[i] = c.P(grads[i]);
                return k->sum_n(c.CS(), nb, src, c.P(accv), n);
            });
        } else {
            accv = buf_like(q, "gsum", grads[0], NHWC);
            const long n = accv.phys();
            Val g0 = grads[0];
            emit([=](Run& c) { return k->memcpy_d2d(c.CS(), c.P(accv), c.P(g0), (size_t)n * 4); });
            for (int i = 1; i < nb; ++i) { Val gi = grads[i]; emit([=](Run& c) { return k->axpy(c.CS(), 1.0f, c.P(gi), c.P(accv), n); }); }
        }
        s.gin = accv;
        return accv;
    }

    // [nn.Dropout (training), nn.Linear(F, O <= 4), nn.Sigmoid]: the discriminator's head (models.lua:699-701) as one launch each way
    Val fwd_head(Mod& drop, Mod& lin, Mod& sig, const Val& in) {
        const KTable* k = K();
        Val x = as_plain(in);
        const long N = x.d[0], F = lin.ia[0], O = lin.ia[1];
        MS &sd = S(drop), &sl = S(lin), &ss = S(sig);
        sd.head_fused = false;
        if (x.nd != 2 || x.d[1] != F || !cg_drop_linear_sigmoid_supported((int)N, (int)F, (int)O)) {
            Val a = fwd(drop, in);
            Val b = fwd(lin, a);
            return fwd(sig, b);
        }
        Val noise = buf_like(drop, "noise", x, PLAIN), xd = buf_like(drop, "out", x, PLAIN);
        Val z = buf(lin, "out", {N, O}), p = buf(sig, "out", {N, O});
        const long off = rng; rng += N * F;
        const float pdrop = drop.fa[0];
        Mod* lp = &lin;
        emit([=](Run& c) {
            if (!lp->w) return cg::fail("cg_net: layer %d has no weight bound (cg_net_bind)", lp->id);
            return k->drop_linear_sigmoid_forward(c.CS(), c.P(x), lp->w, lp->b, c.P(noise), c.P(xd), c.P(z), c.P(p), (int)N, (int)F, (int)O, 1.0f - pdrop,
                                                  1.0f / (1.0f - pdrop), c.seed, c.roff + (uint64_t)off, c.rbase);
        });
        sd.head_fused = true; sd.noise = noise; sd.out = xd; sd.x = x;
        sl.x = xd; sl.out = z; sl.map_in = false;
        ss.out = p;
        return p;
    }
    Val bwd_head(Mod& drop, Mod& lin, Mod& sig, const Val& go, bool acc) {
        const KTable* k = K();
        MS &sd = S(drop), &sl = S(lin), &ss = S(sig);
        Val g = as_plain(go);
        const long N = sd.x.d[0], F = lin.ia[0], O = lin.ia[1];
        Val gi = buf_like(drop, "gin", sd.x, PLAIN);
        Val p = ss.out, xd = sd.out, noise = sd.noise;
        Mod* lp = &lin;
        emit([=](Run& c) {
            return k->drop_linear_sigmoid_backward(c.CS(), c.P(g), c.P(p), c.P(xd), c.P(noise), lp->w, c.P(gi), acc ? lp->gw : nullptr, acc ? lp->gb : nullptr,
                                                   (int)N, (int)F, (int)O, acc ? c.scale : 0.f);
        });
        ss.gin = Val(); sl.gin = Val(); sd.gin = gi;
        return gi;
    }

    // ---------------------------------------------------------------------------------------- backward of one module
    // acc: Module:backward (gradInput + accGradParameters, scaled by Run::scale), else updateGradInput only.
    void wgrad(Mod& m, const Val& go) {   // accGradParameters of a conv / linear layer
        const KTable* k = K();
        MS& s = S(m);
        Mod* mp = &m;
        if (m.kind == K_CONV && s.x.ups && s.use_wino) {
            // Winograd-domain weight gradient from the transformed input the forward of this batch left in wino_v
            Val dy = as_nhwc(go);
            const Val& x = s.x;
            const long N = x.d[0], Hp = x.d[2] >> 1, Wp = x.d[3] >> 1, Ci = m.ia[0], Co = m.ia[1];
            Val v = buf(m, "wino_v", {(long)cg_conv2d_ups2_wino_v_floats((int)N, (int)Hp, (int)Wp, (int)Ci)});
            ws_need(cg_conv2d_ups2_wino_wgrad_workspace_bytes((int)N, (int)Hp, (int)Wp, (int)Ci, (int)Co));
            emit([=](Run& c) { return k->conv2d_ups2_wino_wgrad(c.CS(), c.P(v), c.P(dy), mp->gw, mp->gb, (int)N, (int)Hp, (int)Wp, (int)Ci, (int)Co, c.scale, c.W(), c.WB()); });
            return;
        }
        PrepAcc p = prep_acc(m, go);
        if (net->defer_wgrad && net->fusion) {
            const size_t need = std::max<size_t>(cg_conv2d_wgrad_workspace_bytes(GEO(p.g)), 4096);
            if (!dry && s.wg_ws_bytes < need) { s.wg_ws = alloc(need); s.wg_ws_bytes = need; }
            void* wsp = s.wg_ws; const size_t wsb = s.wg_ws_bytes;
            emit([=](Run& c) {
                const float *xa[1] = {c.P(p.x)}, *da[1] = {c.P(p.dy)}; float *gwa[1] = {mp->gw}, *gba[1] = {mp->gb};
                return k->conv2d_wgrad_grouped_deferred(c.CS(), 1, xa, da, gwa, gba, GEO(p.g), c.scale, wsp, wsb);
            });
            if (!dry) pend[cs]++;
            return;
        }
        ws_need(cg_conv2d_wgrad_workspace_bytes(GEO(p.g)));
        emit([=](Run& c) { return k->conv2d_wgrad(c.CS(), c.P(p.x), c.P(p.dy), mp->gw, mp->gb, GEO(p.g), c.scale, c.W(), c.WB()); });
    }
    Val dgrad(Mod& m, const Val& go) {    // updateGradInput of a conv / linear layer
        const KTable* k = K();
        MS& s = S(m);
        Mod* mp = &m;
        if (m.kind == K_CONV && s.x.ups) {
            // gradient w.r.t. the low-res tensor behind the virtual upsampling (its 2x2 block sum folded in); handed to
            // nn.SpatialUpSamplingNearest as the dual of its lazy output: shape = logical, ups = 1
            Val dy = as_nhwc(go);
            const Val& x = s.x;
            const long N = dy.d[0], Hl = x.d[2], Wl = x.d[3], Hp = Hl >> 1, Wp = Wl >> 1, Ci = m.ia[0], Co = m.ia[1];
            Val lo = buf(m, "gin_lo", {N, Ci, Hp, Wp}, NHWC);
            const int kk = (int)m.kH(), pad = (int)m.padH();
            if (s.use_wino) {
                Val vdy = buf(m, "wino_vdy", {(long)cg_conv2d_ups2_wino_v_floats((int)N, (int)Hp, (int)Wp, (int)(4 * Co))});
                const long np = net->wino_dsplit ? (long)cg_conv2d_ups2_wino_dgrad_part_floats((int)N, (int)Hp, (int)Wp, (int)Ci, (int)Co) : 0;
                if (np) {
                    Val part = buf(m, "wino_dpart", {np});
                    emit([=](Run& c) { return k->conv2d_ups2_wino_dgrad_split(c.CS(), c.P(dy), mp->u_bwd, c.P(lo), c.P(vdy), c.P(part), (int)N, (int)Hp, (int)Wp, (int)Ci, (int)Co); });
                } else
                    emit([=](Run& c) { return k->conv2d_ups2_wino_dgrad(c.CS(), c.P(dy), mp->u_bwd, c.P(lo), c.P(vdy), (int)N, (int)Hp, (int)Wp, (int)Ci, (int)Co); });
            } else if (s.use_wino22 && (net->wino22 & 2) && mp->u22b && cg_conv2d_ups2_wino22_dgrad_supported((int)N, (int)Hp, (int)Wp, (int)Ci, (int)Co)) {
                Val vdy = buf(m, "wino22_vdy", {(long)cg_conv2d_ups2_wino22_dgrad_v_floats((int)N, (int)Hp, (int)Wp, (int)Ci, (int)Co)});
                emit([=](Run& c) { return k->conv2d_ups2_wino22_dgrad(c.CS(), c.P(dy), mp->u22b, c.P(lo), c.P(vdy), (int)N, (int)Hp, (int)Wp, (int)Ci, (int)Co); });
            } else {
                ws_need(cg_conv2d_dgrad_ups2_workspace_bytes((int)N, (int)Hp, (int)Wp, (int)Ci, (int)Co, kk, pad));
                emit([=](Run& c) { return k->conv2d_dgrad_ups2(c.CS(), c.P(dy), mp->wb_ph, c.P(lo), (int)N, (int)Hp, (int)Wp, (int)Ci, (int)Co, kk, pad, c.W(), c.WB()); });
            }
            Val gi = reshape(lo, {N, Ci, Hl, Wl}, NHWC, 1);
            s.gin = gi;
            return gi;
        }
        Prep p = prep_gin(m, go);
        ws_need(cg_conv2d_workspace_bytes(GEO(p.g)));
        emit([=](Run& c) { return k->conv2d_forward(c.CS(), c.P(p.x), wsel(mp, p.wsel), nullptr, c.P(p.out), GEO(p.g), c.W(), c.WB()); });
        s.gin = p.out;
        return p.out;
    }
    Val bwd(Mod& m, const Val& in, const Val& go, bool acc) {
        const KTable* k = K();
        MS& s = S(m);
        Mod* mp = &m;
        switch (m.kind) {
        case K_SEQ:
            if (s.loc_fused && s.locG == 1) return bwd_loc({&m}, {go}, acc)[0];
            return walk_back(m, in, go, acc, false);
        case K_CONCAT: return bwd_concat(m, in, go, acc);
        case K_CONCATTABLE: {
            vector<Val> gs;
            for (size_t j = 0; j < m.kids.size(); ++j) gs.push_back(bwd(M(m.kids[j]), in, go.tab[j], acc));
            return table_sum(m, gs);
        }
        case K_LINEAR: case K_CONV: {
            if (acc) wg_before_dgrad();
            Val gi = dgrad(m, go);
            if (acc) { Mod* mp_ = &m; const Val go_ = go; wg_after_dgrad([this, mp_, go_]() { wgrad(*mp_, go_); }); }
            return gi;
        }
        case K_PRELU: {
            const Val& x = s.x;
            Val dy = match_fmt(go, x);
            Val gi = buf_like(m, "gin", x, x.fmt);
            const long n = x.phys();
            if (acc) ws_need(cg_prelu_backward_workspace_bytes(n));
            emit([=](Run& c) { return k->prelu_backward(c.CS(), c.P(x), c.P(dy), mp->w, c.P(gi), acc ? mp->gw : nullptr, acc ? c.scale : 0.f, n,
                                                        acc ? c.W() : nullptr, acc ? c.WB() : 0); });
            s.gin = gi;
            return gi;
        }
        case K_LRELU: {
            const Val& x = s.x;
            Val dy = match_fmt(go, x);
            Val gi = buf_like(m, "gin", x, x.fmt);
            const long n = x.phys(); const float sl = m.fa[0];
            emit([=](Run& c) { return k->leakyrelu_backward(c.CS(), c.P(x), c.P(dy), c.P(gi), sl, n); });
            s.gin = gi;
            return gi;
        }
        case K_SIGMOID: {
            const Val& y = s.out;
            Val dy = match_fmt(go, y);
            Val gi = buf_like(m, "gin", y, y.fmt);
            const long n = y.phys();
            emit([=](Run& c) { return k->sigmoid_backward(c.CS(), c.P(y), c.P(dy), c.P(gi), n); });
            s.gin = gi;
            return gi;
        }
        case K_BN: {
            if (!m.train) { err("cg_net: batch-norm backward in evaluate() mode is not on the path"); return go; }
            const Val& x = s.x;
            Val dy = as_nhwc(go);
            const long N = x.d[0], C = x.d[1], H = x.d[2], W = x.d[3], Mr = N * H * W;
            Val bs = buf(m, "bsums", {2 * C}, PLAIN, 8), sm = buf(m, "save_mean", {C}), sv = buf(m, "save_std", {C});
            emit([=](Run& c) { return k->bn_backward_stats(c.CS(), c.P(x), c.P(dy), c.P(sm), c.P(sv), Mr, (int)C, (double*)c.P(bs)); });
            Val gs = bs;
            if (net->world > 1 && net->sync_bn) {
                gs = buf(m, "bsums_g", {2 * C}, PLAIN, 8);
                emit([=](Run& c) { return k->memcpy_d2d(c.CS(), c.P(gs), c.P(bs), (size_t)(2 * C) * 8); });
                emit_allreduce_sum(gs, 2 * C, 1);
            }
            Val gi = buf_like(m, "gin", x, NHWC);
            const double cnt = s.count;
            emit([=](Run& c) { return k->bn_backward(c.CS(), c.P(x), c.P(dy), mp->w, c.P(sm), c.P(sv), (const double*)c.P(gs), cnt, (const double*)c.P(bs), Mr, (int)C,
                                                     c.P(gi), acc ? mp->gw : nullptr, acc ? mp->gb : nullptr, acc ? c.scale : 0.f); });
            s.gin = gi;
            return gi;
        }
        case K_VIEW: {
            if (s.skip) { s.gin = go; return go; }   // fused into the nn.Linear behind it, whose gradInput already is the NHWC map
            Val g = as_plain(go);
            Val gi = g; gi.nd = s.in_nd; for (int i = 0; i < 4; ++i) gi.d[i] = s.in_shape[i];
            gi.fmt = PLAIN; gi.blk = 0; gi.gi = gi.gc = 0;
            s.gin = gi;
            return gi;
        }
        case K_COPY: s.gin = go; return go;
        case K_TRANSPOSE: {
            Val gi;
            if (m.ia[0] == 0) {   // forward was NCHW -> BHWD: relabel the BHWD gradient as the NHWC-stored NCHW one
                if (go.fmt != PLAIN) err("cg_net: nn.Transpose backward expects a BHWD tensor");
                gi = reshape(go, {go.d[0], go.d[3], go.d[1], go.d[2]}, NHWC, 0, true);
            } else {
                Val g = as_nhwc(go);
                gi = reshape(g, {g.d[0], g.d[2], g.d[3], g.d[1]}, PLAIN, 0, true);
            }
            s.gin = gi;
            return gi;
        }
        case K_UPS: {
            Val gi;
            if (go.fmt == NHWC && go.ups) {   // the consumer conv already folded the 2x2 block sum
                gi = reshape(go, {go.d[0], go.d[1], go.d[2] / 2, go.d[3] / 2}, NHWC, 0);
            } else {
                Val g = as_nhwc(go);
                const long N = g.d[0], C = g.d[1], H2 = g.d[2], W2 = g.d[3];
                gi = buf(m, "gin", {N, C, H2 / 2, W2 / 2}, NHWC);
                emit([=](Run& c) { return k->upsample2x_backward(c.CS(), c.P(g), c.P(gi), (int)N, (int)(H2 / 2), (int)(W2 / 2), (int)C); });
            }
            s.gin = gi;
            return gi;
        }
        case K_AVGPOOL: case K_MAXPOOL: {
            const Val& x = s.x;
            Val g = as_nhwc(go);
            const long N = x.d[0], C = x.d[1], H = x.d[2], W = x.d[3];
            Val gi = buf_like(m, "gin", x, NHWC);
            if (m.kind == K_AVGPOOL) emit([=](Run& c) { return k->avgpool2_backward(c.CS(), c.P(g), c.P(gi), (int)N, (int)H, (int)W, (int)C); });
            else emit([=](Run& c) { return k->maxpool2_backward(c.CS(), c.P(x), c.P(g), c.P(gi), (int)N, (int)H, (int)W, (int)C); });
            s.gin = gi;
            return gi;
        }
        case K_SDROP: {
            Val g = as_nhwc(go);
            const long N = g.d[0], C = g.d[1], H = g.d[2], W = g.d[3];
            Val gi = buf_like(m, "gin", g, NHWC);
            if (m.train) {
                Val noise = s.noise;
                emit([=](Run& c) { return k->mask_mul(c.CS(), c.P(g), c.P(noise), c.P(gi), (int)N, H * W, (int)C, 1); });
            } else {
                const long n = g.phys(); const float p = m.fa[0];
                emit([=](Run& c) { return k->memcpy_d2d(c.CS(), c.P(gi), c.P(g), (size_t)n * 4); });
                emit([=](Run& c) { return k->scale(c.CS(), c.P(gi), 1.0f - p, n); });
            }
            s.gin = gi;
            return gi;
        }
        case K_DROP: {
            if (!m.train) { s.gin = go; return go; }
            Val g = go;
            Val gi = buf_like(m, "gin", g, g.fmt);
            Val noise = s.noise; const long n = g.phys();
            emit([=](Run& c) { return k->mask_mul(c.CS(), c.P(g), c.P(noise), c.P(gi), 1, n, 1, 0); });
            s.gin = gi;
            return gi;
        }
        case K_AFFMAT: {
            const Val& p = s.p;
            Val gi = buf_like(m, "gin", p, PLAIN);
            const long N = p.d[0]; const int r = (int)m.ia[0], sc = (int)m.ia[1], tr = (int)m.ia[2];
            Val g = go;
            emit([=](Run& c) { return k->affine_matrix_backward(c.CS(), c.P(p), c.P(g), c.P(gi), (int)N, r, sc, tr); });
            s.gin = gi;
            return gi;
        }
        case K_AFFGRID: {
            const long N = go.d[0], H = m.ia[0], W = m.ia[1];
            Val gi = buf(m, "gin", {N, 2, 3});
            Val g = go;
            emit([=](Run& c) { return k->affine_grid_backward(c.CS(), c.P(g), c.P(gi), (int)N, (int)H, (int)W); });
            s.gin = gi;
            return gi;
        }
        case K_SAMPLER: {
            const Val &img = s.in_img, &grid = s.in_grid;
            const long N = img.d[0], Hi = img.d[1], Wi = img.d[2], C = img.d[3], Ho = grid.d[1], Wo = grid.d[2];
            if (go.fmt != PLAIN) err("cg_net: sampler gradOutput must be a BHWD tensor");
            Val gimg = buf_like(m, "gimg", img, PLAIN), ggrid = buf_like(m, "ggrid", grid, PLAIN);
            Val g = go;
            emit([=](Run& c) { return k->bilinear_sampler_backward(c.CS(), c.P(img), c.P(grid), c.P(g), c.P(gimg), c.P(ggrid), (int)N, (int)Hi, (int)Wi, (int)C, (int)Ho, (int)Wo); });
            Val t; t.is_tab = true; t.none = false; t.tab = {gimg, ggrid};
            s.gin = t;
            return t;
        }
        }
        err("cg_net: module kind %d has no backward", m.kind);
        return go;
    }
    Val table_sum(Mod& owner, const vector<Val>& grads) {   // nn.ConcatTable: gradInput = sum of the branches' gradInputs
        const KTable* k = K();
        Val accv; bool have = false;
        for (const Val& g : grads) {
            if (g.none) continue;
            if (!have) { accv = g; have = true; continue; }
            Val a = accv.nd == 4 ? as_nhwc(accv) : accv, b = g.nd == 4 ? as_nhwc(g) : g;
            Val out = buf_like(owner, "sum", a, a.fmt);
            const long n = a.phys();
            emit([=](Run& c) { return k->add(c.CS(), c.P(a), c.P(b), c.P(out), n); });
            accv = out;
        }
        S(owner).gin = accv;
        return accv;
    }

    // ---------------------------------------------------------------------------------------- nn.Sequential backward
    Val walk_back(Mod& q, const Val& in, const Val& go, bool acc, bool root) {
        MS& sq = S(q);
        vector<Seg> pl = sq.ran;
        if (!sq.ran_set) for (int t = 0; t < (int)q.kids.size(); ++t) pl.push_back(Seg{S_ONE, t, t + 1});
        Val cur = go;
        for (int si = (int)pl.size() - 1; si >= 0; --si) {
            const Seg& sg = pl[si];
            auto kid = [&](int t) -> Mod& { return M(q.kids[t]); };
            Val inp = sg.i == 0 ? in : S(kid(sg.i - 1)).out;
            if (sg.kind == S_ACT_POOL && S(kid(sg.i)).fused && S(kid(sg.i)).fG == 1) {
                cur = bwd_act_pool({&kid(sg.i)}, {&kid(sg.i + 1)}, sg.j - sg.i == 3 ? vector<Mod*>{&kid(sg.i + 2)} : vector<Mod*>{}, {cur}, acc)[0];
            } else if (sg.kind == S_GEMM_BN_ACT && S(kid(sg.i + 1)).bn_fused) {
                cur = bwd_gemm_bn_act(kid(sg.i), kid(sg.i + 1), kid(sg.i + 2), inp, cur, acc);
            } else if (sg.kind == S_CAT_DROP) {
                cur = bwd_concat(kid(sg.i), inp, cur, acc, &kid(sg.i + 1));
            } else if (sg.kind == S_HEAD && S(kid(sg.i)).head_fused) {
                cur = bwd_head(kid(sg.i), kid(sg.i + 1), kid(sg.i + 2), cur, acc);
            } else {   // S_ONE, S_GEMM_ACT (both outputs exist), or a chain whose forward ran unfused
                for (int t = sg.j - 1; t >= sg.i; --t) {
                    Val mi = t == sg.i ? inp : S(kid(t - 1)).out;
                    cur = bwd(kid(t), mi, cur, acc);
                }
            }
            if (root && acc) bucket_done(sg.i);
            if (failed) break;
        }
        sq.gin = cur;
        return cur;
    }
    vector<Val> gbwd_seq(const vector<Mod*>& qs, const vector<Val>& ins, const vector<Val>& gouts, bool acc, GCtx& ctx) {
        vector<Val> cur = gouts;
        const int G = (int)qs.size();
        MS& s0 = S(*qs[0]);
        vector<Seg> pl = s0.ran;
        if (!s0.ran_set) for (int t = 0; t < (int)qs[0]->kids.size(); ++t) pl.push_back(Seg{S_ONE, t, t + 1});
        auto col = [&](int t) { vector<Mod*> c; for (Mod* q : qs) c.push_back(&M(q->kids[t])); return c; };
        for (int si = (int)pl.size() - 1; si >= 0; --si) {
            const Seg& sg = pl[si];
            vector<Val> inp;
            if (sg.i == 0) inp = ins; else for (Mod* q : qs) inp.push_back(S(M(q->kids[sg.i - 1])).out);
            bool done = false;
            if (sg.kind == S_ACT_POOL) {
                bool allG = true, all1 = true;
                for (Mod* q : qs) { MS& f = S(M(q->kids[sg.i])); all1 = all1 && f.fused && f.fG == 1; }
                MS& f0 = S(M(qs[0]->kids[sg.i]));
                allG = f0.fused && f0.fG == G;
                if (allG) {
                    cur = bwd_act_pool(col(sg.i), col(sg.i + 1), sg.j - sg.i == 3 ? col(sg.i + 2) : vector<Mod*>{}, cur, acc); done = true;
                } else if (all1) {   // the forward ran branch after branch
                    vector<Val> nx;
                    for (int b = 0; b < G; ++b) {
                        Mod& q = *qs[b];
                        nx.push_back(bwd_act_pool({&M(q.kids[sg.i])}, {&M(q.kids[sg.i + 1])}, sg.j - sg.i == 3 ? vector<Mod*>{&M(q.kids[sg.i + 2])} : vector<Mod*>{},
                                                  {cur[b]}, acc)[0]);
                    }
                    cur = nx; done = true;
                }
            } else if (sg.kind == S_GEMM_BN_ACT && S(M(qs[0]->kids[sg.i + 1])).bn_fused) {
                vector<Val> nx;
                for (int b = 0; b < G; ++b) { Mod& q = *qs[b]; nx.push_back(bwd_gemm_bn_act(M(q.kids[sg.i]), M(q.kids[sg.i + 1]), M(q.kids[sg.i + 2]), inp[b], cur[b], acc)); }
                cur = nx; done = true;
            }
            if (!done) {
                for (int t = sg.j - 1; t >= sg.i; --t) {
                    vector<Val> mi;
                    if (t == sg.i) mi = inp; else for (Mod* q : qs) mi.push_back(S(M(q->kids[t - 1])).out);
                    cur = gbwd(col(t), mi, cur, acc, ctx);
                }
            }
            if (failed) break;
        }
        for (int b = 0; b < G; ++b) S(*qs[b]).gin = cur[b];
        return cur;
    }

    // ---------------------------------------------------------------------------------------- lockstep backward of sibling modules
    vector<Val> gbwd_default(const vector<Mod*>& mods, const vector<Val>& ins, const vector<Val>& gouts, bool acc) {
        vector<Val> out;
        for (size_t b = 0; b < mods.size(); ++b) out.push_back(bwd(*mods[b], ins[b], gouts[b], acc));
        return out;
    }
    vector<Val> gbwd_stackable(const vector<Mod*>& mods, const vector<Val>& ins, const vector<Val>& gouts, bool acc) {
        Mod& m0 = *mods[0];
        if (!ran_stacked(m0)) return gbwd_default(mods, ins, gouts, acc);
        Val X, Gd;
        stacked(ins, &X);
        if (!stacked(gouts, &Gd)) {
            const Val& like = (!S(m0).out.none && !S(m0).out.is_tab) ? S(m0).out : gouts[0];
            Gd = restack(m0, gouts, like);
        }
        vector<Val> gs = split(bwd(m0, X, Gd, false), (int)mods.size());
        for (size_t b = 0; b < mods.size(); ++b) S(*mods[b]).gin = gs[b];
        return gs;
    }
    vector<Val> gbwd(const vector<Mod*>& mods, const vector<Val>& ins, const vector<Val>& gouts, bool acc, GCtx& ctx) {
        Mod& m0 = *mods[0];
        const KTable* k = K();
        const int G = (int)mods.size();
        switch (m0.kind) {
        case K_SEQ:
            if (S(m0).loc_fused && S(m0).locG == G) return bwd_loc(mods, gouts, acc);
            return gbwd_seq(mods, ins, gouts, acc, ctx);
        case K_CONCATTABLE: {
            const size_t nch = m0.kids.size();
            vector<vector<Val>> per_child;
            for (size_t j = 0; j < nch; ++j) {
                vector<Mod*> col; vector<Val> gj;
                for (int b = 0; b < G; ++b) { col.push_back(&M(mods[b]->kids[j])); gj.push_back(gouts[b].tab[j]); }
                per_child.push_back(gbwd(col, ins, gj, acc, ctx));
            }
            vector<Val> blocks; bool all = net->stacking;
            for (auto& c_ : per_child) {
                Val B;
                bool tens = true; for (auto& t : c_) tens = tens && !t.none && !t.is_tab;
                if (all && tens && stacked(c_, &B)) blocks.push_back(B); else all = false;
            }
            if (all) {   // one add per pair of children over the stacked batch
                Val tot = table_sum(m0, blocks);
                vector<Val> outs = split(tot, G);
                for (int b = 0; b < G; ++b) S(*mods[b]).gin = outs[b];
                return outs;
            }
            vector<Val> outs;
            for (int b = 0; b < G; ++b) {
                vector<Val> gs; for (size_t j = 0; j < nch; ++j) gs.push_back(per_child[j][b]);
                outs.push_back(table_sum(*mods[b], gs));
            }
            return outs;
        }
        case K_AVGPOOL: case K_MAXPOOL: {
            MS& s0 = S(m0);
            if (s0.shared_in && m0.kind == K_AVGPOOL) {
                vector<Val> gs; for (auto& g : gouts) gs.push_back(as_nhwc(g));
                Val Gd;
                if (stacked(gs, &Gd)) {   // the average pool's backward does not look at its input: one launch over the stacked gradients
                    const Val& x = s0.x;
                    const long N = x.d[0], C = x.d[1], H = x.d[2], W = x.d[3];
                    Val gi = buf(m0, "gin.shared", {G * N, C, H, W}, NHWC);
                    emit([=](Run& c) { return k->avgpool2_backward(c.CS(), c.P(Gd), c.P(gi), (int)(G * N), (int)H, (int)W, (int)C); });
                    vector<Val> outs = split(gi, G);
                    for (int b = 0; b < G; ++b) S(*mods[b]).gin = outs[b];
                    return outs;
                }
            }
            if (s0.shared_in) return gbwd_default(mods, ins, gouts, acc);
            return gbwd_stackable(mods, ins, gouts, acc);
        }
        case K_SDROP: {
            MS& s0 = S(m0);
            if (!(ran_stacked(m0) && m0.train)) return gbwd_stackable(mods, ins, gouts, acc);
            Val own = s0.noise; s0.noise = s0.noise_block;   // the stacked mask for the one stacked multiply
            vector<Val> r = gbwd_stackable(mods, ins, gouts, acc);
            S(m0).noise = own;
            return r;
        }
        case K_LINEAR: case K_CONV: {
            vector<Prep> gin;
            bool ok = G <= 4;
            for (int b = 0; b < G; ++b) { gin.push_back(prep_gin(*mods[b], gouts[b])); ok = ok && gin[b].ok; }
            for (int b = 1; ok && b < G; ++b) ok = gin[b].g == gin[0].g;
            if (!ok) return gbwd_default(mods, ins, gouts, acc);
            vector<Val> go_; for (auto& p : gin) go_.push_back(p.out);
            if (net->stacking && !stacked(go_, nullptr)) {
                seed_slices(mods, "gin", gin[0].out, gin[0].out.fmt);
                gin.clear();
                for (int b = 0; b < G; ++b) gin.push_back(prep_gin(*mods[b], gouts[b]));
            }
            const Geo g = gin[0].g;
            ws_need(cg_conv2d_workspace_bytes_grouped(G, GEO(g)));
            vector<Mod*> ms_ = mods;
            if (acc) wg_before_dgrad();
            emit([=](Run& c) {
                const float *x[4], *w[4]; float* y[4];
                for (int b = 0; b < G; ++b) { x[b] = c.P(gin[b].x); w[b] = wsel(ms_[b], gin[b].wsel); y[b] = c.P(gin[b].out); }
                return k->conv2d_forward_grouped(c.CS(), G, x, w, nullptr, y, GEO(g), c.W(), c.WB());
            });
            vector<Val> outs;
            for (int b = 0; b < G; ++b) { S(*mods[b]).gin = gin[b].out; outs.push_back(gin[b].out); }
            if (acc) {
                const vector<Val> gouts_ = gouts;
                Mod* m0p = &m0;
                wg_after_dgrad([this, k, G, ms_, gouts_, m0p]() {
                vector<Mod*> mods = ms_;
                const vector<Val>& gouts = gouts_;
                Mod& m0 = *m0p;
                vector<PrepAcc> ap;
                for (int b = 0; b < G; ++b) ap.push_back(prep_acc(*mods[b], gouts[b]));
                const Geo ga = ap[0].g;
                const bool defer = net->defer_wgrad && net->fusion;
                const size_t need = std::max<size_t>(cg_conv2d_wgrad_workspace_bytes_grouped(G, GEO(ga)), 4096);
                MS& s0 = S(m0);
                void* wsp = nullptr; size_t wsb = 0;
                if (defer) {
                    if (!dry && s0.wg_ws_bytes < need) { s0.wg_ws = alloc(need); s0.wg_ws_bytes = need; }
                    wsp = s0.wg_ws; wsb = s0.wg_ws_bytes;
                    if (!dry) pend[cs]++;
                } else ws_need(need);
                emit([=](Run& c) {
                    const float *x[4], *d_[4]; float *gw[4], *gb[4];
                    for (int b = 0; b < G; ++b) { x[b] = c.P(ap[b].x); d_[b] = c.P(ap[b].dy); gw[b] = ms_[b]->gw; gb[b] = ms_[b]->gb; }
                    if (defer) return k->conv2d_wgrad_grouped_deferred(c.CS(), G, x, d_, gw, gb, GEO(ga), c.scale, wsp, wsb);
                    return k->conv2d_wgrad_grouped(c.CS(), G, x, d_, gw, gb, GEO(ga), c.scale, c.W(), c.WB());
                });
                });
            }
            return outs;
        }
        case K_PRELU: {
            if (!net->stacking) return gbwd_default(mods, ins, gouts, acc);
            const Val x0 = S(m0).x;
            vector<Val> xs; for (Mod* m : mods) xs.push_back(S(*m).x);
            Val X;
            const long n = x0.phys();
            if (net->fusion && stacked(xs, &X) && n % 4 == 0 && G <= 4) {
                // one launch for the G modules: stacked x / dy / dx, one slope and one gradient accumulator per group
                vector<Val> gs; for (auto& g : gouts) gs.push_back(match_fmt(g, x0));
                Val Gd;
                if (!stacked(gs, &Gd)) Gd = restack(m0, gs, x0);
                Val bs = x0; bs.d[0] *= G;
                Val dx = buf_like(m0, "gin.gblock", bs, x0.fmt);
                if (acc) ws_need(cg_prelu_backward_grouped_workspace_bytes(G, n));
                vector<Mod*> ms_ = mods;
                emit([=](Run& c) {
                    const float* al[4]; float* ga[4];
                    for (int b = 0; b < G; ++b) { al[b] = ms_[b]->w; ga[b] = ms_[b]->gw; }
                    return k->prelu_backward_grouped(c.CS(), c.P(X), c.P(Gd), al, c.P(dx), acc ? ga : nullptr, c.scale, G, n,
                                                     acc ? c.W() : nullptr, acc ? c.WB() : 0);
                });
                vector<Val> outs = split(dx, G);
                for (int b = 0; b < G; ++b) S(*mods[b]).gin = outs[b];
                return outs;
            }
            seed_slices(mods, "gin", x0, x0.fmt);
            return gbwd_default(mods, ins, gouts, acc);
        }
        case K_SAMPLER: {
            MS& s0 = S(m0);
            bool plain = true; for (auto& g : gouts) plain = plain && g.fmt == PLAIN;
            if (s0.has_shared && !s0.out.none && s0.out.blk && s0.out.blk == s0.shared_out.key() && plain) {
                const Val img = ins[0].tab[0], grids = s0.shared_grids;
                const Val& gr0 = ins[0].tab[1];
                const long N = img.d[0], Hi = img.d[1], Wi = img.d[2], C = img.d[3], Ho = gr0.d[1], Wo = gr0.d[2];
                Val Gd;
                if (!stacked(gouts, &Gd)) Gd = restack(m0, gouts, gouts[0]);
                Val gimg = buf(m0, "gimg.block", {G * N, Hi, Wi, C}), ggrid = buf(m0, "ggrid.block", {G * N, Ho, Wo, 2});
                emit([=](Run& c) { return k->bilinear_sampler_backward_shared(c.CS(), G, c.P(img), c.P(grids), c.P(Gd), c.P(gimg), c.P(ggrid), (int)N, (int)Hi, (int)Wi,
                                                                              (int)C, (int)Ho, (int)Wo); });
                vector<Val> gi = split(gimg, G), gg = split(ggrid, G), res;
                for (int b = 0; b < G; ++b) {
                    Val t; t.is_tab = true; t.none = false; t.tab = {gi[b], gg[b]};
                    S(*mods[b]).gin = t; res.push_back(t);
                }
                return res;
            }
            if (net->stacking) seed_slices(mods, "ggrid", ins[0].tab[1], PLAIN);
            return gbwd_default(mods, ins, gouts, acc);
        }
        default:
            if (stackable(m0)) return gbwd_stackable(mods, ins, gouts, acc);
            return gbwd_default(mods, ins, gouts, acc);
        }
    }

    // ---------------------------------------------------------------------------------------- backward of the fused segments
    vector<Val> bwd_act_pool(const vector<Mod*>& acts, const vector<Mod*>& pools, const vector<Mod*>& drops, const vector<Val>& gouts, bool acc) {
        const KTable* k = K();
        Mod &a0 = *acts[0], &p0 = *pools[0];
        MS& st = S(a0);
        const Val X = st.fX, mask = st.fmask; const int G = st.fG;
        const long N = st.fN, C = st.fC, H = st.fH, W = st.fW;
        vector<Val> gs; for (auto& g : gouts) gs.push_back(as_nhwc(g));
        Val Gd;
        if (G == 1) Gd = gs[0];
        else if (!stacked(gs, &Gd)) Gd = restack(a0, gs, gs[0]);
        Val dx = buf_like(a0, "gin.fused", X, NHWC);
        const int code = a0.kind == K_PRELU ? 1 : 2;
        const float slope = code == 1 ? 0.f : a0.fa[0];
        const bool want = acc && code == 1;
        if (want) ws_need(cg_act_pool2_mask_backward_workspace_bytes(G, (int)N, (int)H, (int)W, (int)C));
        const int pool_max = p0.kind == K_MAXPOOL ? 1 : 0;
        const bool have_mask = !mask.none;
        vector<Mod*> ac = acts;
        emit([=](Run& c) {
            const float* al[4]; float* ga[4];
            for (int b = 0; b < G; ++b) { al[b] = ac[b]->w; ga[b] = ac[b]->gw; }
            return k->act_pool2_mask_backward(c.CS(), c.P(X), c.P(Gd), have_mask ? c.P(mask) : nullptr, c.P(dx), G, (int)N, (int)H, (int)W, (int)C, code, slope,
                                              code == 1 ? al : nullptr, want ? ga : nullptr, c.scale, pool_max, want ? c.W() : nullptr,
                                              want ? c.WB() : 0);
        });
        vector<Val> outs = G == 1 ? vector<Val>{dx} : split(dx, G);
        for (int b = 0; b < G; ++b) {
            S(*acts[b]).gin = outs[b]; S(*pools[b]).gin = Val();
            if (!drops.empty()) S(*drops[b]).gin = Val();
        }
        return outs;
    }
    Val bwd_gemm_bn_act(Mod& conv, Mod& bn, Mod& act, const Val& in, const Val& go, bool acc) {
        const KTable* k = K();
        MS& sb = S(bn);
        const long Mr = sb.bnM, C = sb.bnC;
        const Val x = sb.x;
        Val dy = as_nhwc(go);
        Mod *bp = &bn, *ap = &act;
        Val sm = buf(bn, "save_mean", {C}), sv = buf(bn, "save_std", {C});
        Val b3 = buf(bn, "bsums3", {2 * C + 1}, PLAIN, 8);
        emit([=](Run& c) { return k->bn_act_backward_stats(c.CS(), c.P(x), c.P(dy), c.P(sm), c.P(sv), bp->w, bp->b, ap->w, Mr, (int)C, (double*)c.P(b3)); });
        Val gs = b3;
        if (net->world > 1 && net->sync_bn) {
            gs = buf(bn, "bsums3_g", {2 * C + 1}, PLAIN, 8);
            emit([=](Run& c) { return k->memcpy_d2d(c.CS(), c.P(gs), c.P(b3), (size_t)(2 * C + 1) * 8); });
            emit_allreduce_sum(gs, 2 * C + 1, 1);
        }
        Val dx = buf_like(bn, "gin", x, NHWC);
        const double cnt = sb.count;
        emit([=](Run& c) { return k->bn_act_backward(c.CS(), c.P(x), c.P(dy), bp->w, bp->b, c.P(sm), c.P(sv), ap->w, (const double*)c.P(gs), cnt, (const double*)c.P(b3),
                                                     Mr, (int)C, c.P(dx), acc ? bp->gw : nullptr, acc ? bp->gb : nullptr, acc ? ap->gw : nullptr, c.scale); });
        S(act).gin = Val(); sb.gin = dx;
        return bwd(conv, in, dx, acc);
    }

    // ---------------------------------------------------------------------------------------- data-parallel exchanges
    void bucket_done(int first_module);   // root Sequential, acc pass: every parameter gradient of modules[first_module:] is on the stream
    void buckets_start();                 // start the all-reduce of every complete bucket that has not been started
    int bucket_done_upto = 0, bucket_ready = -1;
};

enum { HOOK_ALLREDUCE_SUM = 0, HOOK_BUCKET_START = 1, HOOK_BUCKETS_FINISH = 2 };

void Compiler::emit_allreduce_sum(const Val& v, long count, int dtype) {
    if (!(net->world > 1 && net->sync_bn)) return;
    emit([=](Run& c) -> int {
        Net* n = c.net;
        void* p = c.P(v);
        if (n->trace) { trace_note(n, "hook|allreduce_sum|" + TraceLine::pname(p) + "|" + std::to_string(count)); return 0; }
        if (n->comm_bn) {   // RCCL on the sync-BN communicator's stream, joined back into the compute stream (no host sync)
            if (cg_comm_allreduce(n->comm_bn, c.CS(), p, (size_t)count, dtype, 0)) return 1;
            if (cg_comm_wait(n->comm_bn, c.CS())) return 1;
            if (!(n->hook_too && n->hook)) return 0;
        }
        if (n->hook) return n->hook(n->hook_user, HOOK_ALLREDUCE_SUM, p, (size_t)count, dtype, c.CS()) ? cg::fail("cg_net: the host hook failed (sync-BN all-reduce of %ld elements)", count) : 0;
        return cg::fail("cg_net: world > 1 with sync-BN but neither a communicator (cg_net_set_dp) nor a host hook is set");
    });
    if (acc_pass && cs == 0) buckets_start();   // a bucket held back for this exchange (bucket_done) goes behind it
}
// Gradient buckets of the root nn.Sequential: each convolution / linear layer with the parameters of the modules
// up to the next one is a contiguous range of the flat gradient vector; as soon as the backward walk has passed it, its all-reduce
// starts on the gradient communicator's stream, under the backward of the layers in front of it.
// Collectives of the two communicators are ordered on the device (comm.hip), and a bucket's all-reduce sits behind its layer's whole weight
// gradient: started right away it would make the sync-BN exchange of the NEXT layer's backward - the first thing on the data-gradient
// chain - wait for that weight gradient (157 us of stall per layer in the one-GPU dry run, profiles/r06_dp_dry_run.txt).  So while a
// batch-norm layer in front still has its exchange to make, a complete bucket is held back and started right behind that exchange.
void Compiler::bucket_done(int first_module) {
    if (dry || !net->bucket_overlap || net->world <= 1) return;
    bucket_ready = first_module;
    bool bn_ahead = false;
    if (net->sync_bn) {
        const Mod& root = *net->mods[0];
        for (int t = 0; t < first_module && t < (int)root.kids.size(); ++t) {
            const Mod& m = *net->mods[root.kids[t]];
            bn_ahead = bn_ahead || (m.kind == K_BN && m.train);
        }
    }
    if (!bn_ahead) buckets_start();
}
void Compiler::buckets_start() {
    if (dry || !net->bucket_overlap || net->world <= 1 || bucket_ready < 0 || bucket_ready >= bucket_done_upto) return;
    const int first_module = bucket_ready;
    for (int t = bucket_done_upto - 1; t >= first_module; --t) {
        for (size_t bi = 0; bi < pr->bucket_first.size(); ++bi) {
            if (pr->bucket_first[bi] != t) continue;
            // the bucket's weight gradients are on the weight-gradient stream (behind everything this stream has issued: fork here), maybe
            // still queued as deferred reductions; its collective starts from there
            wg_fork();
            const int s_ = wg_enter();
            flush_wgrad();
            // ... and on the weight-gradient streams of the branch groups behind it (D32_st3's nn.Concat: streams 5..7), which nothing
            // joins before the end of the pass: the collective's stream waits for what they hold so far (stream 0 does not)
            const int here = cs;
            for (int s = 0; s < 4; ++s) {
                if (!wg_unjoined[s] || 4 + s == here) continue;
                cs = 4 + s;
                flush_wgrad();
                emit([=](Run& c) { return c.net->trace ? (trace_note(c.net, "event|record|wgjoin" + std::to_string(s) + "|s" + std::to_string(4 + s)), 0)
                                                        : (hipEventRecord(c.net->wg_join_ev[s], (hipStream_t)c.S(4 + s)) == hipSuccess ? 0 : 1); });
                cs = here;
                emit([=](Run& c) { return c.net->trace ? (trace_note(c.net, "event|wait|wgjoin" + std::to_string(s) + "|s" + std::to_string(here)), 0)
                                                        : (hipStreamWaitEvent((hipStream_t)c.S(here), c.net->wg_join_ev[s], 0) == hipSuccess ? 0 : 1); });
                wg_unjoined[s] = false;
            }
            wg_unjoined[s_] = false;
            float* ptr = pr->buckets[bi].first; const long cnt = pr->buckets[bi].second;
            emit([=](Run& c) -> int {
                Net* n = c.net;
                if (n->trace) { trace_note(n, "hook|bucket_start|" + TraceLine::pname(ptr) + "|" + std::to_string(cnt)); return 0; }
                if (n->comm_grad) {
                    if (cg_comm_allreduce(n->comm_grad, c.CS(), ptr, (size_t)cnt, 0, 1)) return 1;
                    if (!(n->hook_too && n->hook)) return 0;
                    if (cg_comm_wait(n->comm_grad, c.CS())) return 1;   // the hook's transport reads what the collective wrote
                }
                if (n->hook) return n->hook(n->hook_user, HOOK_BUCKET_START, ptr, (size_t)cnt, 0, c.CS()) ? cg::fail("cg_net: the host hook failed (gradient bucket of %ld elements)", cnt) : 0;
                return cg::fail("cg_net: bucketed all-reduce without a communicator or host hook");
            });
            wg_leave(s_);
        }
    }
    bucket_done_upto = first_module;
}

}  // namespace

// ================================================================================================ runtime + C ABI
namespace cg { extern unsigned long g_opt_epoch; }

namespace {

Net* NET(void* h) { return reinterpret_cast<Net*>(h); }

long param_numel(const Mod& m, int slot) {
    switch (m.kind) {
        case K_LINEAR: return slot == 0 ? m.ia[0] * m.ia[1] : m.ia[1];
        case K_CONV: return slot == 0 ? m.ia[0] * m.ia[1] * m.ia[2] * m.ia[3] : m.ia[1];
        case K_PRELU: return slot == 0 ? 1 : 0;
        case K_BN: return m.ia[0];
        default: return 0;
    }
}
void collect_params(Net* n, const Mod& m, vector<std::pair<float*, long>>& out) {   // gradient tensors, depth-first, weight then bias
    if (m.gw) out.push_back({m.gw, param_numel(m, 0)});
    if (m.gb) out.push_back({m.gb, param_numel(m, 1)});
    for (int c : m.kids) collect_params(n, *n->mods[c], out);
}

// Buffers the library allocated while compiling were zeroed by hipMemset on the NULL stream, which the (non-blocking) streams the
// pass runs on do not wait for: finish the memsets before the first launch can touch them.
int settle_allocs(Net* n) {
    if (n->fresh_allocs && !n->trace) CG_HIP(hipDeviceSynchronize());
    n->fresh_allocs = false;
    return 0;
}

// Side streams come out of the process-wide pool of streams classified by hardware queue (cg::queue_stream, common.h): stream index t of
// a plan runs on queue class kQueueOf[t] relative to the caller's stream - [1..3] the branch groups' streams, [4 + s] the weight-gradient
// stream beside stream s.  Measured on the batch-128 step (profiles/r05_queue_classes.txt; one box, ms per step): what must overlap has to
// sit on DIFFERENT queues - a weight-gradient stream or the host's side stream on the step's own queue costs 0.2-0.4 ms - and beyond that
// the choice is flat within 0.03 ms; D's best was its chain's weight gradients on the branch stream's queue and the branch's on a third
// (5.91 against 5.95-6.2 for the other fifteen).  (The CG_QMAP override of round 5 went with round 6's pruning: the
// assignments were swept there and the choice is flat beyond "what must overlap sits on different queues".)
static const int kQueueOf[8] = {0, 1, 2, 3, 1, 3, 2, 1};
int ensure_streams(Net* n, int nstreams, void* ref) {
    if (n->trace) return 0;
    if ((int)n->side.size() < nstreams - 1) {
        static int slots[4] = {0, 0, 0, 0};
        static std::mutex mu;                 // nets of several host threads share the pool's slot counters
        std::lock_guard<std::mutex> lk(mu);
        if (n->qmap[0] < 0) {
            for (int t = 0; t < 8; ++t) n->qmap[t] = kQueueOf[t];
        }
        while ((int)n->side.size() < nstreams - 1) {
            const int t = (int)n->side.size() + 1;
            const int c = n->qmap[t & 7];
            hipStream_t s = cg::queue_stream((hipStream_t)ref, c, slots[c]++);
            if (!s) return 1;
            hipEvent_t e;
            CG_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            n->side.push_back(s); n->side_ev.push_back(e);
        }
    }
    for (int t = 0; t < 4 && nstreams > 4; ++t)
        if (!n->wg_fork_ev[t]) {
            CG_HIP(hipEventCreateWithFlags(&n->wg_fork_ev[t], hipEventDisableTiming));
            CG_HIP(hipEventCreateWithFlags(&n->wg_join_ev[t], hipEventDisableTiming));
        }
    if (!n->fork_ev) CG_HIP(hipEventCreateWithFlags(&n->fork_ev, hipEventDisableTiming));
    if (!n->pack_ev) {
        CG_HIP(hipEventCreateWithFlags(&n->pack_fork_ev, hipEventDisableTiming));
        CG_HIP(hipEventCreateWithFlags(&n->pack_ev, hipEventDisableTiming));
    }
    return 0;
}

void fill_run(Net* n, Prog* pr, Run& c, void* stream) {
    c.net = n; c.pr = pr;
    c.st[0] = (hipStream_t)stream;
    for (int t = 1; t < 8; ++t) {
        if (n->trace) c.st[t] = (hipStream_t)(uintptr_t)(0x1000 + t);
        else c.st[t] = t - 1 < (int)n->side.size() ? n->side[t - 1] : (hipStream_t)stream;
    }
    if (n->trace) { n->trace_streams.assign(8, nullptr); for (int t = 0; t < 8; ++t) n->trace_streams[t] = (void*)c.st[t]; if (!stream) n->trace_streams[0] = nullptr; }
}

// join_before: index of the first op that needs the weights sync_packs re-packed on the side stream (-1: nothing pending)
int run_ops(Net* n, vector<Op>& ops, Run& c, int join_before = -1) {
    g_cur_net = n;
    int at = 0;
    for (Op& op : ops) {
        if (at++ == join_before) {
            if (n->trace) trace_note(n, "event|wait|packs|all");
            else
                for (int t = 0; t < c.pr->nstreams; ++t)
                    if (t != 1 && hipStreamWaitEvent(c.st[t], n->pack_ev, 0) != hipSuccess) { g_cur_net = nullptr; return cg::fail("cg_net: hipStreamWaitEvent failed"); }
        }
        c.cur = op.sidx;
        const int rc = op.fn(c);
        if (rc) { g_cur_net = nullptr; cg::wgrad_discard_all(); return rc; }   // a failed pass leaves no queued reductions behind
    }
    g_cur_net = nullptr;
    return 0;
}

// Refresh the kernel-side weight copies after a parameter update: every plain layer of the net in ONE launch
// (cg_pack_conv_weight_batch), layers behind a folded upsampling with their own (phase-summed / Winograd) packing.  Only the first
// of those is needed at once: the others are packed on side stream 1 beside the head of the pass (option pack_overlap), and
// *join_before is the forward op in front of which the pass has to wait for them.
int sync_packs(Net* n, Prog* pr, Run& c, int* join_before) {
    *join_before = -1;
    if (n->params_dirty) {
        for (auto& mp : n->mods) { mp->dirty_plain = true; mp->dirty_ups = true; }
        n->params_dirty = false;
    }
    void* stream = c.S(0);
    vector<const float*> w; vector<float*> wf, wb; vector<int> co, ci, kh, kw, mp_;
    vector<Mod*> ups;
    g_cur_net = n;
    for (auto& up : n->mods) {
        Mod& m = *up;
        if (!m.is_gemm()) continue;
        if (m.wf && m.dirty_plain) {
            if (!m.w) return cg::fail("cg_net: layer %d has no weight bound (cg_net_bind)", m.id);
            w.push_back(m.w); wf.push_back(m.wf); wb.push_back(m.wb);
            if (m.pk_map == 1) { co.push_back((int)m.ia[1]); ci.push_back((int)m.map_c); kh.push_back((int)m.map_h); kw.push_back((int)m.map_w); mp_.push_back(1); }
            else if (m.kind == K_LINEAR) { co.push_back((int)m.ia[1]); ci.push_back((int)m.ia[0]); kh.push_back(1); kw.push_back(1); mp_.push_back(0); }
            else { co.push_back((int)m.ia[1]); ci.push_back((int)m.ia[0]); kh.push_back((int)m.kH()); kw.push_back((int)m.kW()); mp_.push_back(0); }
            m.dirty_plain = false;
        }
        if (m.wf_ph && m.dirty_ups) ups.push_back(&m);
    }
    if (!w.empty() && n->K->pack_conv_weight_batch(stream, (int)w.size(), w.data(), wf.data(), wb.data(), co.data(), ci.data(), kh.data(), kw.data(), mp_.data())) return 1;
    auto first_op = [&](const Mod* m) { auto it = pr->ups_first_op.find(m->id); return it == pr->ups_first_op.end() ? -1 : it->second; };
    std::stable_sort(ups.begin(), ups.end(), [&](const Mod* a, const Mod* b) { return first_op(a) < first_op(b); });   // unknown to this plan first
    const bool can_side = n->pack_overlap && c.st[1] != c.st[0];
    bool forked = false;
    for (size_t i = 0; i < ups.size(); ++i) {
        Mod& m = *ups[i];
        const int fo = first_op(&m);
        const bool on_side = can_side && i >= 1 && fo > 0 && fo < (int)pr->fwd.size();
        if (on_side && !forked) {
            forked = true;
            if (n->trace) trace_note(n, "event|record|packs_fork|s0\nevent|wait|packs_fork|s1");
            else if (hipEventRecord(n->pack_fork_ev, c.st[0]) != hipSuccess || hipStreamWaitEvent(c.st[1], n->pack_fork_ev, 0) != hipSuccess)
                return cg::fail("cg_net: cannot fork the packing stream");
        }
        void* st = on_side ? c.S(1) : stream;
        if (n->K->pack_conv_weight_ups2(st, m.w, m.wf_ph, m.wb_ph, (int)m.ia[1], (int)m.ia[0], (int)m.kH(), (int)((m.kH() - 1) / 2))) return 1;
        if (m.wino && n->K->conv2d_ups2_wino_pack(st, m.wf_ph, m.wb_ph, m.u_fwd, m.u_bwd, (int)m.ia[1], (int)m.ia[0])) return 1;
        if (m.u22 && n->K->conv2d_ups2_wino22_pack(st, m.wf_ph, m.wb_ph, m.u22, m.u22b, (int)m.ia[1], (int)m.ia[0])) return 1;
        if (on_side && (*join_before < 0 || fo < *join_before)) *join_before = fo;
        m.dirty_ups = false;
    }
    if (forked) {
        if (n->trace) trace_note(n, "event|record|packs|s1");
        else if (hipEventRecord(n->pack_ev, c.st[1]) != hipSuccess) return cg::fail("cg_net: hipEventRecord failed");
        n->pack_inflight = true;
    } else if ((!w.empty() || !ups.empty()) && !n->trace) {
        // everything was re-packed on the caller's stream (pack_overlap off, fewer than two layers behind an upsampling, no side stream):
        // a LATER pass of this net on ANOTHER stream - the side-by-side generator forward - still has to wait for it
        if (hipEventRecord(n->pack_ev, c.st[0]) != hipSuccess) return cg::fail("cg_net: hipEventRecord failed");
        n->pack_inflight = true;
    }
    g_cur_net = nullptr;
    return 0;
}

std::string prog_key(Net* n, int nd, const long* dims, int fmt) {
    std::string k = std::to_string(fmt) + ":";
    for (int i = 0; i < nd; ++i) k += std::to_string(dims[i]) + "x";
    k += "|t";
    for (auto& m : n->mods) if (m->kind == K_BN || m->kind == K_SDROP || m->kind == K_DROP) k += m->train ? '1' : '0';
    k += "|w" + std::to_string(n->world) + (n->sync_bn ? "s" : "-") + (n->bucket_overlap ? "b" : "-");
    k += "|o" + std::to_string(n->overlap_groups) + std::to_string(n->defer_wgrad) + std::to_string(n->winograd) + std::to_string(n->fusion) +
         std::to_string(n->stacking) + std::to_string(n->grouped) + std::to_string(n->share_pool) + std::to_string(n->sampler_shared) +
         std::to_string(n->view_fuse) + std::to_string(n->cat_fuse) + std::to_string(n->fuse_locnet) + std::to_string(n->head_fuse) +
         std::to_string(n->wgrad_stream) + std::to_string(n->wino22) + std::to_string(n->wino_dsplit) + "m" +
         std::to_string(n->wino_min_tiles);
    return k;
}

}  // namespace

extern "C" {

int cg_net_create(void** net) {
    CG_REQUIRE(net, "cg_net_create: null pointer");
    Net* n = new Net();
    *net = n;
    // CG_NET_OPTIONS="name=value,name=value": plan options of EVERY net of the process, through cg_net_set_option below (same-box A/B runs
    // of bench.py; round 5 folded the eighteen per-option variables - CG_WINOGRAD22, CG_WGRAD_STREAM, CG_FUSION ... - into this one)
    if (const char* e = getenv("CG_NET_OPTIONS")) {
        std::string all(e);
        size_t pos = 0;
        while (pos < all.size()) {
            size_t end = all.find(',', pos);
            if (end == std::string::npos) end = all.size();
            const std::string kv = all.substr(pos, end - pos);
            const size_t eq = kv.find('=');
            if (eq == std::string::npos || eq == 0) { delete n; *net = nullptr; return cg::fail("CG_NET_OPTIONS: expected name=value, got '%s'", kv.c_str()); }
            if (cg_net_set_option(n, kv.substr(0, eq).c_str(), atol(kv.c_str() + eq + 1))) { delete n; *net = nullptr; return 1; }
            pos = end + 1;
        }
    }
    return 0;
}

int cg_net_destroy(void* net) {
    Net* n = NET(net);
    if (!n) return 0;
    for (void* p : n->owned) { if (n->trace) free(p); else (void)hipFree(p); }
    // n->side: pool streams (cg::queue_stream), not ours to destroy
    for (auto e : n->side_ev) (void)hipEventDestroy(e);
    if (n->fork_ev) (void)hipEventDestroy(n->fork_ev);
    if (n->pack_ev) { (void)hipEventDestroy(n->pack_fork_ev); (void)hipEventDestroy(n->pack_ev); }
    if (n->pair_fork_ev) { (void)hipEventDestroy(n->pair_fork_ev); (void)hipEventDestroy(n->pair_a_ev); (void)hipEventDestroy(n->pair_join_ev); }
    for (int t = 0; t < 4; ++t) if (n->wg_fork_ev[t]) { (void)hipEventDestroy(n->wg_fork_ev[t]); (void)hipEventDestroy(n->wg_join_ev[t]); }
    delete n;
    return 0;
}

int cg_net_set_option(void* net, const char* name, long value) {
    Net* n = NET(net);
    CG_REQUIRE(n && name, "cg_net_set_option: null pointer");
    struct { const char* nm; int* p; } tab[] = {
        {"overlap_groups", &n->overlap_groups}, {"defer_wgrad", &n->defer_wgrad}, {"winograd", &n->winograd}, {"share_pool", &n->share_pool},
        {"sampler_shared", &n->sampler_shared}, {"view_fuse", &n->view_fuse}, {"cat_fuse", &n->cat_fuse}, {"stacking", &n->stacking},
        {"grouped", &n->grouped}, {"fusion", &n->fusion}, {"fuse_locnet", &n->fuse_locnet}, {"pack_overlap", &n->pack_overlap},
        {"head_fuse", &n->head_fuse}, {"wgrad_stream", &n->wgrad_stream}, {"wino_dsplit", &n->wino_dsplit}};
    if (!strcmp(name, "trace")) {
        CG_REQUIRE(n->progs.empty(), "cg_net_set_option: trace must be chosen before the first pass");
        n->trace = value != 0; n->K = n->trace ? &kTraceTable : &kRealTable;
        return 0;
    }
    if (!strcmp(name, "defer_running")) { n->defer_running = value != 0; return 0; }   // run-time switch, plans unchanged
    if (!strcmp(name, "winograd_min_tiles")) { n->wino_min_tiles = value; return 0; }
    if (!strcmp(name, "winograd22")) { n->wino22 = (int)value & 3; return 0; }     // bit mask: 1 forward, 2 data gradient
    if (!strcmp(name, "fuse_locnet")) { n->fuse_locnet = (int)value; return 0; }   // 0 off, 1 every localisation branch, 2 ungrouped ones only
    for (auto& t : tab) if (!strcmp(name, t.nm)) { *t.p = value != 0; return 0; }
    return cg::fail("cg_net_set_option: unknown option %s", name);
}

int cg_net_set_allocator(void* net, cg_alloc_fn alloc, void* user) {
    Net* n = NET(net);
    CG_REQUIRE(n, "cg_net_set_allocator: null net");
    n->alloc_fn = alloc; n->alloc_user = user;
    return 0;
}

int cg_net_set_hook(void* net, cg_hook_fn hook, void* user) {
    Net* n = NET(net);
    CG_REQUIRE(n, "cg_net_set_hook: null net");
    n->hook = hook; n->hook_user = user;
    return 0;
}

int cg_net_set_dp(void* net, int world, int sync_bn, void* comm_bn, void* comm_grad, int bucket_overlap) {
    Net* n = NET(net);
    CG_REQUIRE(n && world >= 1, "cg_net_set_dp: bad arguments");
    n->world = world; n->sync_bn = sync_bn != 0; n->comm_bn = comm_bn; n->comm_grad = comm_grad; n->bucket_overlap = (bucket_overlap & 1) != 0;
    n->hook_too = (bucket_overlap & 2) != 0;
    return 0;
}

int cg_net_add(void* net, int parent, int kind, const long* iargs, int niargs, const float* fargs, int nfargs, int* id) {
    Net* n = NET(net);
    CG_REQUIRE(n && id, "cg_net_add: null pointer");
    CG_REQUIRE(kind >= 0 && kind < K_COUNT, "cg_net_add: unknown module kind %d", kind);
    CG_REQUIRE(niargs <= 8 && nfargs <= 4, "cg_net_add: too many arguments");
    CG_REQUIRE(parent < (int)n->mods.size(), "cg_net_add: unknown parent %d", parent);
    CG_REQUIRE((parent < 0) == n->mods.empty(), "cg_net_add: exactly the first module is the root (parent -1)");
    if (parent >= 0) CG_REQUIRE(n->mods[parent]->is_container(), "cg_net_add: parent %d is not a container", parent);
    std::unique_ptr<Mod> m(new Mod());
    m->id = (int)n->mods.size(); m->kind = kind; m->parent = parent;
    for (int i = 0; i < niargs; ++i) m->ia[i] = iargs[i];
    for (int i = 0; i < nfargs; ++i) m->fa[i] = fargs[i];
    if (kind == K_VIEW) m->ia[7] = niargs;
    if (kind == K_CONV) CG_REQUIRE(m->ia[6] <= 1 && m->ia[7] <= 1, "cg_net_add: only stride 1 convolutions are on the path");
    if (kind == K_CONCAT) CG_REQUIRE(m->ia[0] == 2, "cg_net_add: nn.Concat joins on dimension 2 (channels)");
    if (parent >= 0) n->mods[parent]->kids.push_back(m->id);
    *id = m->id;
    n->mods.push_back(std::move(m));
    n->progs.clear(); n->last = nullptr;
    return 0;
}

int cg_net_bind(void* net, int id, int slot, float* param, float* grad) {
    Net* n = NET(net);
    CG_REQUIRE(n && id >= 0 && id < (int)n->mods.size(), "cg_net_bind: unknown module %d", id);
    Mod& m = *n->mods[id];
    if (slot == 0) { m.w = param; m.gw = grad; }
    else if (slot == 1) { m.b = param; m.gb = grad; }
    else if (slot == 2) { m.rmean = param; m.rvar = grad; }
    else return cg::fail("cg_net_bind: slot %d (0 weight, 1 bias, 2 running mean / var)", slot);
    n->params_dirty = true;
    for (auto& kv : n->progs) { kv.second->buckets.clear(); kv.second->bucket_first.clear(); kv.second->have_bwd[1] = false; kv.second->bwd[1].clear(); }
    return 0;
}

int cg_net_set_training(void* net, int id, int train) {
    Net* n = NET(net);
    CG_REQUIRE(n && id < (int)n->mods.size(), "cg_net_set_training: unknown module %d", id);
    if (id < 0) for (auto& m : n->mods) m->train = train != 0;
    else n->mods[id]->train = train != 0;
    return 0;
}

int cg_net_params_changed(void* net) {
    Net* n = NET(net);
    CG_REQUIRE(n, "cg_net_params_changed: null net");
    n->params_dirty = true;
    return 0;
}

int cg_net_trace_region(void* net, const void* base, size_t bytes) {
    Net* n = NET(net);
    CG_REQUIRE(n && base, "cg_net_trace_region: null pointer");
    n->regions.push_back(Region{(const char*)base, bytes});
    return 0;
}

int cg_net_trace_take(void* net, char* out, size_t cap, size_t* len) {
    Net* n = NET(net);
    CG_REQUIRE(n && len, "cg_net_trace_take: null pointer");
    *len = n->trace_log.size();
    if (out && cap >= n->trace_log.size()) { memcpy(out, n->trace_log.data(), n->trace_log.size()); n->trace_log.clear(); }
    return 0;
}

int cg_net_forward(void* net, void* stream, const float* x, int nd, const long* dims, int fmt, uint64_t rng_seed, uint64_t rng_offset,
                   const uint64_t* rng_base, uint64_t* draws, float** y, int* ynd, long* ydims, int* yfmt) {
    Net* n = NET(net);
    CG_REQUIRE(n && x && dims && nd >= 1 && nd <= 4, "cg_net_forward: bad arguments");
    CG_REQUIRE(!n->mods.empty(), "cg_net_forward: empty net");
    const std::string key = prog_key(n, nd, dims, fmt);
    auto it = n->progs.find(key);
    if (it != n->progs.end() && it->second->opt_epoch != cg::g_opt_epoch) {   // cg_set_option moved the dispatch
        if (n->last == it->second.get()) n->last = nullptr;   // a failed recompile must not leave backward a dangling plan
        n->progs.erase(it); it = n->progs.end();
    }
    if (it == n->progs.end()) {
        std::unique_ptr<Prog> pr(new Prog());
        pr->net = n; pr->opt_epoch = cg::g_opt_epoch;
        Val in; in.none = false; in.ext = EXT_X; in.nd = nd; for (int i = 0; i < nd; ++i) in.d[i] = dims[i]; in.fmt = fmt;
        pr->in = in;
        Compiler C(n, pr.get());
        C.ops = &pr->fwd;
        pr->out = C.fwd(*n->mods[0], in);
        if (C.failed) return cg::fail("%s", n->err);
        CG_REQUIRE(!pr->out.is_tab, "cg_net_forward: the root module returns a table");
        pr->out = C.materialise(pr->out);
        pr->draws = C.rng;
        if (ensure_streams(n, std::max(pr->nstreams, n->pack_overlap && pr->ups_first_op.size() >= 2 ? 2 : 1), stream)) return 1;
        if (settle_allocs(n)) return 1;
        it = n->progs.emplace(key, std::move(pr)).first;
    }
    Prog* pr = it->second.get();
    Run c;
    fill_run(n, pr, c, stream);
    int join_before = -1;
    if (sync_packs(n, pr, c, &join_before)) return 1;
    // another pass of this net (on another stream) may have the re-packing in flight: wait for it before the first launch
    if (join_before < 0 && n->pack_inflight && !n->trace && hipStreamWaitEvent(c.st[0], n->pack_ev, 0) != hipSuccess)
        return cg::fail("cg_net_forward: hipStreamWaitEvent failed");
    c.x = x; c.seed = rng_seed; c.roff = rng_offset; c.rbase = rng_base;
    if (run_ops(n, pr->fwd, c, join_before)) return 1;
    n->last = pr;
    pr->running_pending = n->defer_running != 0 && !pr->bn_runs.empty();
    if (draws) *draws = (uint64_t)pr->draws;
    if (y) *y = c.P(pr->out);
    if (ynd) *ynd = pr->out.nd;
    if (ydims) for (int i = 0; i < 4; ++i) ydims[i] = pr->out.d[i];
    if (yfmt) *yfmt = pr->out.fmt;
    return 0;
}

int cg_net_backward(void* net, void* stream, const float* x, const float* gy, int gy_fmt, int acc, float scale, float** gx, int* gnd, long* gdims,
                    int* gfmt) {
    Net* n = NET(net);
    CG_REQUIRE(n && x && gy, "cg_net_backward: null pointer");
    Prog* pr = n->last;
    CG_REQUIRE(pr, "cg_net_backward: no forward pass to continue");
    acc = acc ? 1 : 0;
    if (!pr->have_bwd[acc]) {
        Compiler C(n, pr);
        C.ops = &pr->bwd[acc];
        C.acc_pass = acc != 0;
        pr->bwd[acc].clear();
        Mod& root = *n->mods[0];
        if (acc && n->bucket_overlap && n->world > 1 && root.kind == K_SEQ && root.kids.size() > 1 && pr->buckets.empty()) {
            // contiguous ranges of the flat gradient: a conv / linear layer with the parameters of the modules up to the next one
            bool contiguous = true; float* expect = nullptr;
            for (size_t i = 0; i < root.kids.size(); ++i) {
                vector<std::pair<float*, long>> ps;
                collect_params(n, *n->mods[root.kids[i]], ps);
                long cnt = 0; for (auto& p : ps) cnt += p.second;
                if (n->mods[root.kids[i]]->is_gemm() || pr->buckets.empty()) {
                    if (ps.empty() && pr->buckets.empty()) continue;
                    pr->buckets.push_back({ps.empty() ? expect : ps[0].first, 0}); pr->bucket_first.push_back((int)i);
                }
                for (auto& p : ps) { if (expect && p.first != expect) contiguous = false; expect = p.first + p.second; }
                pr->buckets.back().second += cnt;
            }
            if (!contiguous) { pr->buckets.clear(); pr->bucket_first.clear(); }
            // buckets below 256 KB ride with the one the backward completes next (a collective costs a launch and a latency whatever it
            // carries): G's last convolution (3 459 values) with the 256 -> 128 layer, D's head with Linear(20480, 256), D's first layers
            // in one exchange at the end
            constexpr long kMinBucket = 65536;
            vector<std::pair<float*, long>> mb; vector<int> mf;
            long accn = 0;
            for (int j = (int)pr->buckets.size() - 1; j >= 0; --j) {
                accn += pr->buckets[j].second;
                if (accn >= kMinBucket || j == 0) {
                    mb.insert(mb.begin(), {pr->buckets[j].first, accn}); mf.insert(mf.begin(), pr->bucket_first[j]);
                    accn = 0;
                }
            }
            pr->buckets.swap(mb); pr->bucket_first.swap(mf);
        }
        C.bucket_done_upto = (int)root.kids.size();
        Val go = pr->out; go.ext = EXT_GY; go.off = 0; go.p = nullptr; go.fmt = gy_fmt; go.blk = 0; go.gi = go.gc = 0;
        Val gi = root.kind == K_SEQ ? C.walk_back(root, pr->in, go, acc != 0, true) : C.bwd(root, pr->in, go, acc != 0);
        if (C.failed) return cg::fail("%s", n->err);
        if (acc) { C.buckets_start(); C.flush_wgrad(); C.wg_join_all(); }
        CG_REQUIRE(!gi.is_tab, "cg_net_backward: the root module's gradInput is a table");
        pr->gin[acc] = gi;
        pr->have_bwd[acc] = true;
        if (ensure_streams(n, pr->nstreams, stream)) return 1;
        if (settle_allocs(n)) return 1;
    }
    Run c;
    fill_run(n, pr, c, stream);
    c.x = x; c.gy = gy; c.scale = scale;
    n->pack_inflight = false;      // every forward pass since the last re-packing has waited for it
    if (run_ops(n, pr->bwd[acc], c)) return 1;
    const Val& gi = pr->gin[acc];
    if (gx) *gx = c.P(gi);
    if (gnd) *gnd = gi.nd;
    if (gdims) for (int i = 0; i < 4; ++i) gdims[i] = gi.d[i];
    if (gfmt) *gfmt = gi.fmt + 2 * gi.ups;
    return 0;
}

int cg_net_apply_running(void* net, void* stream) {
    Net* n = NET(net);
    CG_REQUIRE(n, "cg_net_apply_running: null net");
    Prog* pr = n->last;
    if (!pr || !pr->running_pending) return 0;
    pr->running_pending = false;
    g_cur_net = n;
    for (auto& b : pr->bn_runs) {
        if (!b.bn->rmean || !b.bn->rvar) continue;
        if (n->K->bn_running_update(stream, (const double*)b.sums.p, b.cnt, (int)b.C, b.mom, b.bn->rmean, b.bn->rvar)) { g_cur_net = nullptr; return 1; }
    }
    g_cur_net = nullptr;
    return 0;
}

// Two forward passes of ONE net side by side (SURVEY.md 8b "optional coarse entry"): adversarial.lua:232-233 runs MODEL_G on
// the D-step's N/2 noise rows and :185 on the G-step's N rows, and MODEL_G moves only at :262 - both passes read the same parameters.  Pass
// 1 runs on the caller's stream exactly as cg_net_forward would; pass 2 (its input must already be enqueued on `stream`) runs on a
// library stream of ANOTHER hardware queue, leaves the batch-norm running statistics alone and applies its update behind pass 1's, so
// that results, running statistics and the plan cg_net_backward continues (pass 2's) are those of forward(x) followed by forward(x2).
// The caller's stream does not wait for pass 2 until cg_net_pair_join, which also hands out pass 2's output.  draws = both passes'.
int cg_net_forward_pair(void* net, void* stream, const float* x, int nd, const long* dims, int fmt, const float* x2, int nd2, const long* dims2,
                        int fmt2, uint64_t rng_seed, uint64_t rng_offset, const uint64_t* rng_base, uint64_t* draws, float** y, int* ynd,
                        long* ydims, int* yfmt) {
    Net* n = NET(net);
    CG_REQUIRE(n && x && x2 && dims && dims2, "cg_net_forward_pair: null pointer");
    CG_REQUIRE(!n->pair_pending, "cg_net_forward_pair: the previous pair was never joined (cg_net_pair_join)");
    if (n->trace) {      // a trace records the launches of both passes one after the other
        uint64_t d1 = 0, d2 = 0;
        if (cg_net_forward(net, stream, x, nd, dims, fmt, rng_seed, rng_offset, rng_base, &d1, y, ynd, ydims, yfmt)) return 1;
        trace_note(n, "pair|second pass on the pair stream");
        if (cg_net_forward(net, stream, x2, nd2, dims2, fmt2, rng_seed, rng_offset + d1, rng_base, &d2, &n->pair_y, &n->pair_ynd, n->pair_ydims, &n->pair_yfmt)) return 1;
        if (draws) *draws = d1 + d2;
        n->pair_pending = true;
        return 0;
    }
    if (!n->pair_fork_ev) {
        n->pair_stream = cg::queue_stream(cg::S(stream), 2, 7);     // any class but the caller's own measures the same (profiles/r05_queue_classes.txt)
        if (!n->pair_stream) return 1;
        CG_HIP(hipEventCreateWithFlags(&n->pair_fork_ev, hipEventDisableTiming));
        CG_HIP(hipEventCreateWithFlags(&n->pair_a_ev, hipEventDisableTiming));
        CG_HIP(hipEventCreateWithFlags(&n->pair_join_ev, hipEventDisableTiming));
    }
    hipStream_t st = cg::S(stream), side = n->pair_stream;
    CG_HIP(hipEventRecord(n->pair_fork_ev, st));
    CG_HIP(hipStreamWaitEvent(side, n->pair_fork_ev, 0));
    uint64_t d1 = 0, d2 = 0;
    if (cg_net_forward(net, stream, x, nd, dims, fmt, rng_seed, rng_offset, rng_base, &d1, y, ynd, ydims, yfmt)) return 1;
    CG_HIP(hipEventRecord(n->pair_a_ev, st));      // pass 1, its running-statistics update included, is complete on the caller's stream
    const int defer = n->defer_running;
    n->defer_running = 1;
    const int rc = cg_net_forward(net, (void*)side, x2, nd2, dims2, fmt2, rng_seed, rng_offset + d1, rng_base, &d2, &n->pair_y, &n->pair_ynd,
                                  n->pair_ydims, &n->pair_yfmt);
    n->defer_running = defer;
    if (rc) return 1;
    CG_HIP(hipStreamWaitEvent(side, n->pair_a_ev, 0));
    if (cg_net_apply_running(net, (void*)side)) return 1;      // off the caller's chain, but behind pass 1's update: the reference's order
    CG_HIP(hipEventRecord(n->pair_join_ev, side));
    n->pair_pending = true;
    if (draws) *draws = d1 + d2;
    return 0;
}

int cg_net_pair_join(void* net, void* stream, float** y, int* ynd, long* ydims, int* yfmt) {
    Net* n = NET(net);
    CG_REQUIRE(n, "cg_net_pair_join: null net");
    CG_REQUIRE(n->pair_pending, "cg_net_pair_join: no pair in flight");
    if (!n->trace) CG_HIP(hipStreamWaitEvent(cg::S(stream), n->pair_join_ev, 0));
    n->pair_pending = false;
    if (y) *y = n->pair_y;
    if (ynd) *ynd = n->pair_ynd;
    if (ydims) for (int i = 0; i < 4; ++i) ydims[i] = n->pair_ydims[i];
    if (yfmt) *yfmt = n->pair_yfmt;
    return 0;
}

int cg_net_buckets(void* net, int* nbuckets) {
    Net* n = NET(net);
    CG_REQUIRE(n && nbuckets, "cg_net_buckets: null pointer");
    *nbuckets = n->last ? (int)n->last->buckets.size() : 0;
    return 0;
}

int cg_net_module_state(void* net, int id, int which, float** ptr, int* nd, long* dims, int* fmt) {
    Net* n = NET(net);
    CG_REQUIRE(n && ptr && id >= 0 && id < (int)n->mods.size(), "cg_net_module_state: bad arguments");
    *ptr = nullptr;
    if (!n->last) return 0;
    auto it = n->last->ms.find(id);
    if (it == n->last->ms.end()) return 0;
    const Val& v = which == 0 ? it->second.out : which == 1 ? it->second.gin : it->second.noise;
    if (v.none || v.is_tab || v.ext) return 0;
    *ptr = v.p;
    if (nd) *nd = v.nd;
    if (dims) for (int i = 0; i < 4; ++i) dims[i] = v.d[i];
    if (fmt) *fmt = v.fmt + 2 * v.ups;
    return 0;
}

int cg_net_stats(void* net, long* nprograms, long* nlaunch_fwd, long* nlaunch_bwd, size_t* bytes) {
    Net* n = NET(net);
    CG_REQUIRE(n, "cg_net_stats: null net");
    if (nprograms) *nprograms = (long)n->progs.size();
    if (nlaunch_fwd) *nlaunch_fwd = n->last ? (long)n->last->fwd.size() : 0;
    if (nlaunch_bwd) *nlaunch_bwd = n->last ? (long)std::max(n->last->bwd[0].size(), n->last->bwd[1].size()) : 0;
    if (bytes) { size_t b = 0; for (auto& r : n->regions) b += r.bytes; *bytes = b; }
    return 0;
}

// ---- whole-iteration capture: the host runs its training iteration once between begin and end (every launch it makes on `stream`,
// and on streams forked from it by events, is recorded instead of executed), then replays the graph per step.
int cg_graph_begin(void* stream) {
    CG_HIP(hipStreamBeginCapture(cg::S(stream), hipStreamCaptureModeRelaxed));
    return 0;
}
int cg_graph_end(void* stream, void** graph_exec) {
    CG_REQUIRE(graph_exec, "cg_graph_end: null pointer");
    hipGraph_t g = nullptr;
    CG_HIP(hipStreamEndCapture(cg::S(stream), &g));
    hipGraphExec_t ex = nullptr;
    hipError_t e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (e != hipSuccess) return cg::fail("cg_graph_end: hipGraphInstantiate -> %s", hipGetErrorString(e));
    *graph_exec = (void*)ex;
    return 0;
}
int cg_graph_launch(void* graph_exec, void* stream) {
    CG_HIP(hipGraphLaunch((hipGraphExec_t)graph_exec, cg::S(stream)));
    return 0;
}
int cg_graph_destroy(void* graph_exec) {
    if (graph_exec) CG_HIP(hipGraphExecDestroy((hipGraphExec_t)graph_exec));
    return 0;
}

}  // extern "C"
