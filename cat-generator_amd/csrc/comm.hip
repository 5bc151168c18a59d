// Collectives of the data-parallel G+D step through the C ABI: RCCL all-reduce over xGMI on a side HIP stream with
// event fork / join against the caller's compute stream (the reference itself is single-GPU:
// train.lua:108-112 selects one device).  One process per GPU; a LuaJIT / C host uses exactly these entry points
// (INTEGRATION.md), cat-generator_amd/parallel.py uses them for every device-side exchange.
//
//   gradients : all-reduce(mean) of the flat GRAD_PARAMETERS vector (or a bucket of it) right after backward, BEFORE
//               penalty / clamp / Adam (the order adversarial.lua:89-112 fixes) - cg_comm_allreduce(..., CG_AVG)
//               returns at once; cg_comm_wait() joins it into the compute stream where the update needs it
//   sync-BN   : all-reduce(sum) of the 2C fp64 batch statistics between cg_bn_stats* and cg_bn_*forward (and of the
//               backward sums), on its OWN communicator so that it never queues behind a gradient bucket in flight
//
// RCCL is bound at run time (dlopen of librccl.so.1, reusing the copy a host such as PyTorch already mapped), so the
// library loads - and every non-collective entry point works - on a machine without RCCL.
#include "common.h"
#include <dlfcn.h>
#include <stdlib.h>

// The handful of RCCL declarations this file needs, stated here so that the library BUILDS on a machine without the RCCL
// headers too (values as in rccl.h / nccl.h 2.x; the functions themselves are bound with dlsym below).
extern "C" {
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclFloat32 = 7, ncclFloat = 7, ncclFloat64 = 8, ncclDouble = 8 } ncclDataType_t;
typedef enum { ncclSum = 0, ncclProd = 1, ncclMax = 2, ncclMin = 3, ncclAvg = 4 } ncclRedOp_t;
}

namespace {

struct Rccl {
    void* h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*GetVersion)(int*) = nullptr;
    bool tried = false;
};
Rccl g_rccl;

bool load_rccl() {
    Rccl& r = g_rccl;
    if (r.tried) return r.h != nullptr;
    r.tried = true;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {   // prefer a copy that is already mapped into the process (one RCCL per process)
        r.h = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
        if (r.h) break;
    }
    if (!r.h)
        for (const char* n : names) {
            r.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (r.h) break;
        }
    if (!r.h) return false;
#define CG_SYM(field, name)                                                   \
    r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.h, name));          \
    if (!r.field) { r.h = nullptr; return false; }
    CG_SYM(GetUniqueId, "ncclGetUniqueId")
    CG_SYM(CommInitRank, "ncclCommInitRank")
    CG_SYM(CommDestroy, "ncclCommDestroy")
    CG_SYM(AllReduce, "ncclAllReduce")
    CG_SYM(Broadcast, "ncclBroadcast")
    CG_SYM(GetErrorString, "ncclGetErrorString")
#undef CG_SYM
    r.GetVersion = reinterpret_cast<decltype(r.GetVersion)>(dlsym(r.h, "ncclGetVersion"));   // optional
    return true;
}

#define CG_NCCL(call)                                                                     \
    do {                                                                                  \
        ncclResult_t r__ = (call);                                                        \
        if (r__ != ncclSuccess)                                                           \
            return cg::fail("%s:%d %s -> %s", __FILE__, __LINE__, #call, g_rccl.GetErrorString(r__)); \
    } while (0)

struct Comm {
    ncclComm_t nccl = nullptr;
    hipStream_t side = nullptr;     // collectives run here
    hipEvent_t fork = nullptr;      // compute stream -> side stream (data produced by the kernels enqueued so far)
    hipEvent_t join = nullptr;      // side stream -> compute stream (last collective enqueued)
    int nranks = 1, rank = 0;
    bool pending = false;
    hipStream_t last = nullptr;     // the stream this communicator's latest collective went to (`join` was recorded there)
};
// One communicator, several streams (the sync-BN sums of the two generator passes of cg_net_forward_pair run on two compute streams): the
// next collective waits for the previous one's `join` whenever the stream changes, so that the communicator's collectives execute in the
// host's program order on every rank whatever the library does about stream changes internally.
int order_on_comm(Comm* c, hipStream_t s) {
    if (c->last && c->last != s) CG_HIP(hipStreamWaitEvent(s, c->join, 0));
    return 0;
}

// Collectives of DIFFERENT communicators on one device are ordered on the device: the side stream of the communicator that
// launches next waits for the join event of the one that launched last.  Two RCCL kernels of different communicators can
// otherwise be in flight together (D's gradient all-reduce under the G-step's generator forward, whose sync-BN sums travel on
// the other communicator), which NCCL / RCCL documents as deadlock-prone when ranks schedule them in different orders.  The
// cost is that a sync-BN exchange may wait for a gradient bucket in flight (~0.3 ms at 8 GPUs); CG_COMM_SERIAL=0 lifts it.
Comm* g_last_comm = nullptr;
bool serial_comms() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("CG_COMM_SERIAL"); v = (e && atoi(e) == 0) ? 0 : 1; }
    return v != 0;
}
// Exchanges of at most 16 KB run ON the compute stream: the sync-BN sums are 2C fp64 values that the caller joins at once, so the side
// stream only added two cross-queue event hops to a latency-bound chain (~50 us of stall per exchange around a 5 us kernel in the one-GPU
// dry run, 6.03 -> 5.93 ms per step: profiles/r06_dp_dry_run.txt).  Gradient buckets (>= 256 KB, joined later) keep the side stream.  The
// order of a communicator's collectives is the host's program order either way.
constexpr size_t kInlineBytes = 16384;
size_t inline_bytes() { return kInlineBytes; }
int order_after_other_comm(Comm* c) {
    if (serial_comms() && g_last_comm && g_last_comm != c) CG_HIP(hipStreamWaitEvent(c->side, g_last_comm->join, 0));
    return 0;
}

}  // namespace

extern "C" {

int cg_comm_available(int* available) {
    CG_REQUIRE(available, "cg_comm_available: null pointer");
    *available = load_rccl() ? 1 : 0;
    return 0;
}

int cg_comm_unique_id(void* id_out, size_t id_bytes) {
    CG_REQUIRE(id_out && id_bytes >= sizeof(ncclUniqueId), "cg_comm_unique_id: need a buffer of %zu bytes", sizeof(ncclUniqueId));
    CG_REQUIRE(load_rccl(), "cg_comm_unique_id: RCCL (librccl.so.1) not found");
    ncclUniqueId id;
    CG_NCCL(g_rccl.GetUniqueId(&id));
    memcpy(id_out, &id, sizeof(id));
    return 0;
}

int cg_comm_init(void** comm, int nranks, int rank, const void* unique_id, size_t id_bytes) {
    CG_REQUIRE(comm && unique_id && id_bytes >= sizeof(ncclUniqueId), "cg_comm_init: null pointer / short id");
    CG_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "cg_comm_init: rank %d of %d", rank, nranks);
    CG_REQUIRE(load_rccl(), "cg_comm_init: RCCL (librccl.so.1) not found");
    Comm* c = new Comm();
    c->nranks = nranks; c->rank = rank;
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    ncclResult_t r = g_rccl.CommInitRank(&c->nccl, nranks, id, rank);   // collective: every rank calls it, current device = its GPU
    if (r != ncclSuccess) { delete c; return cg::fail("cg_comm_init: ncclCommInitRank -> %s", g_rccl.GetErrorString(r)); }
    hipError_t e = hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->fork, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->join, hipEventDisableTiming);
    if (e != hipSuccess) {
        g_rccl.CommDestroy(c->nccl);
        delete c;
        return cg::fail("cg_comm_init: %s", hipGetErrorString(e));
    }
    *comm = c;
    return 0;
}

int cg_comm_destroy(void* comm) {
    if (!comm) return 0;
    Comm* c = static_cast<Comm*>(comm);
    if (g_last_comm == c) g_last_comm = nullptr;
    (void)hipStreamSynchronize(c->side);
    if (c->nccl) g_rccl.CommDestroy(c->nccl);
    if (c->fork) (void)hipEventDestroy(c->fork);
    if (c->join) (void)hipEventDestroy(c->join);
    if (c->side) (void)hipStreamDestroy(c->side);
    delete c;
    return 0;
}

int cg_comm_version(int* version) {
    CG_REQUIRE(version, "cg_comm_version: null pointer");
    *version = 0;
    if (load_rccl() && g_rccl.GetVersion) (void)g_rccl.GetVersion(version);
    return 0;
}

int cg_comm_size(void* comm, int* nranks, int* rank) {
    CG_REQUIRE(comm, "cg_comm_size: null communicator");
    Comm* c = static_cast<Comm*>(comm);
    if (nranks) *nranks = c->nranks;
    if (rank) *rank = c->rank;
    return 0;
}

// In-place all-reduce of buf[0..count) over the communicator's ranks, enqueued on the communicator's side stream behind
// everything `compute_stream` holds so far (on `compute_stream` itself when it is at most 16 KB long).  dtype: 0 fp32, 1 fp64.  op: 0 sum, 1 average.  Returns at once; the result
// may be consumed on `compute_stream` only after cg_comm_wait(comm, compute_stream).
int cg_comm_allreduce(void* comm, void* compute_stream, void* buf, size_t count, int dtype, int op) {
    CG_REQUIRE(comm && buf, "cg_comm_allreduce: null pointer");
    CG_REQUIRE((dtype == 0 || dtype == 1) && (op == 0 || op == 1), "cg_comm_allreduce: dtype %d op %d", dtype, op);
    Comm* c = static_cast<Comm*>(comm);
    if (count == 0) return 0;
    hipStream_t cs = cg::S(compute_stream);
    if (count * (dtype == 0 ? 4 : 8) <= inline_bytes()) {
        if (order_on_comm(c, cs)) return 1;
        c->pending = false;                                                                    // (that wait was the join of any side-stream work)
        if (serial_comms() && g_last_comm && g_last_comm != c) CG_HIP(hipStreamWaitEvent(cs, g_last_comm->join, 0));
        CG_NCCL(g_rccl.AllReduce(buf, buf, count, dtype == 0 ? ncclFloat32 : ncclFloat64, op == 0 ? ncclSum : ncclAvg, c->nccl, cs));
        CG_HIP(hipEventRecord(c->join, cs));   // what another communicator's next collective orders behind; cg_comm_wait has nothing to do
        g_last_comm = c;
        c->last = cs;
        return 0;
    }
    CG_HIP(hipEventRecord(c->fork, cs));
    CG_HIP(hipStreamWaitEvent(c->side, c->fork, 0));
    if (order_on_comm(c, c->side)) return 1;
    if (order_after_other_comm(c)) return 1;
    CG_NCCL(g_rccl.AllReduce(buf, buf, count, dtype == 0 ? ncclFloat32 : ncclFloat64, op == 0 ? ncclSum : ncclAvg, c->nccl, c->side));
    CG_HIP(hipEventRecord(c->join, c->side));
    c->last = c->side;
    c->pending = true;
    g_last_comm = c;
    return 0;
}

// Same for a broadcast from `root` (initial parameters: every rank starts from rank 0's weights).
int cg_comm_broadcast(void* comm, void* compute_stream, void* buf, size_t count, int dtype, int root) {
    CG_REQUIRE(comm && buf, "cg_comm_broadcast: null pointer");
    CG_REQUIRE(dtype == 0 || dtype == 1, "cg_comm_broadcast: dtype %d", dtype);
    Comm* c = static_cast<Comm*>(comm);
    CG_REQUIRE(root >= 0 && root < c->nranks, "cg_comm_broadcast: root %d of %d", root, c->nranks);
    if (count == 0) return 0;
    hipStream_t cs = cg::S(compute_stream);
    CG_HIP(hipEventRecord(c->fork, cs));
    CG_HIP(hipStreamWaitEvent(c->side, c->fork, 0));
    if (order_on_comm(c, c->side)) return 1;
    if (order_after_other_comm(c)) return 1;
    CG_NCCL(g_rccl.Broadcast(buf, buf, count, dtype == 0 ? ncclFloat32 : ncclFloat64, root, c->nccl, c->side));
    CG_HIP(hipEventRecord(c->join, c->side));
    c->last = c->side;
    c->pending = true;
    g_last_comm = c;
    return 0;
}

// Join: `compute_stream` waits (on the device, the host does not block) for every collective enqueued so far.
int cg_comm_wait(void* comm, void* compute_stream) {
    CG_REQUIRE(comm, "cg_comm_wait: null communicator");
    Comm* c = static_cast<Comm*>(comm);
    if (!c->pending) return 0;
    CG_HIP(hipStreamWaitEvent(cg::S(compute_stream), c->join, 0));
    c->pending = false;
    return 0;
}

// Host-side completion (end of a timed region, before reading results back).
int cg_comm_sync(void* comm) {
    CG_REQUIRE(comm, "cg_comm_sync: null communicator");
    CG_HIP(hipStreamSynchronize(static_cast<Comm*>(comm)->side));
    return 0;
}

}  // extern "C"
