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
#include <algorithm>
#include <functional>
#include <map>
#include <memory>
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
enum { S_ONE = 0, S_GEMM_ACT, S_ACT_POOL, S_VIEW_GEMM, S_VIEW_GEMM_ACT, S_GEMM_BN_ACT };

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
    bool use_wino = false;
    bool fused = false; int fG = 0; Val fX, fmask; long fN = 0, fC = 0, fH = 0, fW = 0;   // act_pool segment state (on the activation)
    bool bn_fused = false; long bnM = 0, bnC = 0;            // gemm_bn_act segment state (on the batch-norm)
    double count = 0;            // BN: samples behind the statistics (x world under sync-BN)
    vector<long> sizes;          // nn.Concat: channels per branch
    vector<Seg> ran;             // nn.Sequential: the plan its forward ran
    bool ran_set = false;
    void* wg_ws = nullptr; size_t wg_ws_bytes = 0; bool wg_pending = false;   // deferred weight-gradient workspace
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
};

typedef void* (*alloc_fn_t)(void* user, size_t bytes);
typedef int (*hook_fn_t)(void* user, int what, void* buf, size_t count, int dtype, void* stream);

struct Region { const char* base; size_t bytes; };

struct Net {
    vector<std::unique_ptr<Mod>> mods;
    std::map<std::string, std::unique_ptr<Prog>> progs;
    Prog* last = nullptr;                      // plan of the most recent forward (what backward continues)
    vector<hipStream_t> side; vector<hipEvent_t> side_ev; hipEvent_t fork_ev = nullptr;
    alloc_fn_t alloc_fn = nullptr; void* alloc_user = nullptr;
    hook_fn_t hook = nullptr; void* hook_user = nullptr;
    void *comm_bn = nullptr, *comm_grad = nullptr; int world = 1; int sync_bn = 1; int bucket_overlap = 0;
    vector<void*> owned;
    vector<Region> regions;
    bool params_dirty = true;
    // options
    int trace = 0, overlap_groups = 1, defer_wgrad = 1, winograd = 1, share_pool = 1, sampler_shared = 1, view_fuse = 1,
        cat_fuse = 1, stacking = 1, grouped = 1, fusion = 1;
    long wino_min_tiles = 2048;
    std::string trace_log;
    const KTable* K = &kRealTable;
    vector<void*> trace_streams;               // stream handle -> index in trace mode
    char err[512] = {0};
};

struct Run {
    Net* net; Prog* pr;
    hipStream_t st[4];
    const float* x = nullptr; const float* gy = nullptr;
    uint64_t seed = 0, roff = 0; const uint64_t* rbase = nullptr;
    float scale = 1.f;
    void* S(int sidx) const { return (void*)st[sidx]; }
    float* P(const Val& v) const {
        if (v.none) return nullptr;
        if (v.ext == EXT_X) return (float*)((char*)x + v.off);
        if (v.ext == EXT_GY) return (float*)((char*)gy + v.off);
        return v.p;
    }
    void* W(int sidx) const { return pr->ws[sidx]; }
    size_t WB(int sidx) const { return pr->ws_bytes[sidx]; }
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
    int pend[4] = {0, 0, 0, 0};   // deferred weight-gradient reductions queued per stream index
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
            if (hipMalloc(&p, bytes) != hipSuccess) p = nullptr;
            else hipMemset(p, 0, bytes);
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
        emit([=](Run& c) { return k->upsample2x_forward(c.S(cs_of(c)), c.P(x), c.P(out), (int)N, (int)(H >> 1), (int)(W >> 1), (int)C); });
        return out;
    }
    // NOTE: closures must not read Compiler state at run time; the stream index is captured per op by emit() (Op::sidx) and
    // handed to the closure through Run::cur (set by the runner before each op).
    static int cs_of(Run&);

    Val as_nhwc(const Val& x, bool keep_ups = false) {
        if (x.fmt == NHWC) return (keep_ups || !x.ups) ? x : materialise(x);
        if (x.nd != 4) { err("cg_net: expected a 4-D feature map"); return x; }
        long N = x.d[0], C = x.d[1], H = x.d[2], W = x.d[3];
        Val out = anon({N, C, H, W}, NHWC);
        const KTable* k = K();
        emit([=](Run& c) { return k->nchw_to_nhwc(c.S(cs_of(c)), c.P(x), c.P(out), (int)N, (int)C, (int)H, (int)W); });
        return out;
    }
    Val as_plain(const Val& x0) {
        if (x0.fmt == PLAIN) return x0;
        Val x = materialise(x0);
        long N = x.d[0], C = x.d[1], H = x.d[2], W = x.d[3];
        Val out = anon({N, C, H, W}, PLAIN);
        const KTable* k = K();
        emit([=](Run& c) { return k->nhwc_to_nchw(c.S(cs_of(c)), c.P(x), c.P(out), (int)N, (int)C, (int)H, (int)W); });
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
            emit([=](Run& c) { return k->memcpy_d2d(c.S(cs_of(c)), c.P(dst), c.P(src), nb); });
        }
        return block;
    }
