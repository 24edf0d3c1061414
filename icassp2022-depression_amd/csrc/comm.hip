// RCCL boundary of the data-parallel path: dep_comm_{unique_id,init,allreduce,destroy}.
//
// One process per GPU; every rank owns a replica of the parameters and a flat fp32 gradient buffer.  The gradient exchange
// is a SUM all-reduce of contiguous ranges of that buffer over xGMI -- one range per recurrent layer, enqueued on a
// dedicated communication stream as soon as the layer's weight gradients are complete (dep_rnn_backward_overlapped in
// api.hip), so the top layer's range travels while the layer below is still in its backward sweep.
//
// librccl is resolved at run time (dlopen / dlsym) the first time a communicator is asked for: a single-GPU run never
// loads it, the library has no link-time dependency on it, and a process that already carries an RCCL (torch.distributed's
// "nccl" backend) shares that copy instead of loading a second one.
#include <dlfcn.h>
#include <string.h>
#include <rccl/rccl.h>
#include "dep_common.h"

namespace {

struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};
RcclApi g_rccl;

bool load_rccl() {
    if (g_rccl.ok) return true;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
        g_rccl.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (g_rccl.handle) break;
    }
    if (!g_rccl.handle) { dep_set_error("librccl not found: %s", dlerror()); return false; }
#define SYM(field, name)                                                                              \
    g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(g_rccl.handle, name));              \
    if (!g_rccl.field) { dep_set_error("librccl: symbol %s missing", name); return false; }
    SYM(GetUniqueId, "ncclGetUniqueId")
    SYM(CommInitRank, "ncclCommInitRank")
    SYM(AllReduce, "ncclAllReduce")
    SYM(CommDestroy, "ncclCommDestroy")
    SYM(GroupStart, "ncclGroupStart")
    SYM(GroupEnd, "ncclGroupEnd")
    SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
    g_rccl.ok = true;
    return true;
}

}  // namespace

struct dep_comm {
    ncclComm_t comm;
    int world, rank, device;
};

#define RCCL_CHECK(call, what)                                                                        \
    do {                                                                                              \
        ncclResult_t r_ = (call);                                                                     \
        if (r_ != ncclSuccess) { dep_set_error("%s: %s", what, g_rccl.GetErrorString(r_)); return DEP_ERR_HIP; } \
    } while (0)

// 1 when librccl resolves in this process (no collective, no GPU work): lets every rank agree on the transport BEFORE
// anybody enters a collective that the others would otherwise wait in for ever.
extern "C" int dep_comm_available(void) { return load_rccl() ? 1 : 0; }

extern "C" int dep_comm_unique_id(void* id_out, size_t bytes) {
    DEP_CHECK_ARG(id_out && bytes >= sizeof(ncclUniqueId));
    if (!load_rccl()) return DEP_ERR_HIP;
    ncclUniqueId id;
    RCCL_CHECK(g_rccl.GetUniqueId(&id), "ncclGetUniqueId");
    memcpy(id_out, &id, sizeof(id));
    return DEP_OK;
}

extern "C" int dep_comm_init(dep_comm** out, int world, int rank, const void* unique_id, size_t id_bytes, int device) {
    DEP_CHECK_ARG(out && unique_id && id_bytes >= sizeof(ncclUniqueId) && world >= 1 && rank >= 0 && rank < world);
    if (!load_rccl()) return DEP_ERR_HIP;
    if (hipSetDevice(device) != hipSuccess) { dep_set_error("dep_comm_init: hipSetDevice(%d) failed", device); return DEP_ERR_HIP; }
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    dep_comm* c = new dep_comm{nullptr, world, rank, device};
    ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, id, rank);
    if (r != ncclSuccess) { dep_set_error("ncclCommInitRank: %s", g_rccl.GetErrorString(r)); delete c; return DEP_ERR_HIP; }
    *out = c;
    return DEP_OK;
}

extern "C" int dep_comm_world(const dep_comm* c) { return c ? c->world : 0; }
extern "C" int dep_comm_rank(const dep_comm* c) { return c ? c->rank : -1; }

// In-place SUM all-reduce of n fp32 values, enqueued on `stream` (nothing synchronises).
extern "C" int dep_comm_allreduce(dep_comm* c, float* buf, long n, void* stream) {
    DEP_CHECK_ARG(c && buf && n > 0);
    dep_olog_add('C', "allreduce", n);
    RCCL_CHECK(g_rccl.AllReduce(buf, buf, (size_t)n, ncclFloat, ncclSum, c->comm, (hipStream_t)stream), "ncclAllReduce");
    return DEP_OK;
}

// Several ranges as ONE grouped RCCL operation (one launch): the layer-0 range and the LayerNorm range of the audio model.
extern "C" int dep_comm_allreduce_ranges(dep_comm* c, float* const* bufs, const long* counts, int nranges, void* stream) {
    DEP_CHECK_ARG(c && bufs && counts && nranges > 0);
    { long tot = 0; for (int i = 0; i < nranges; ++i) tot += counts[i] > 0 ? counts[i] : 0; dep_olog_add('C', "allreduce_ranges", tot); }
    RCCL_CHECK(g_rccl.GroupStart(), "ncclGroupStart");
    for (int i = 0; i < nranges; ++i) {
        if (!bufs[i] || counts[i] <= 0) continue;
        ncclResult_t r = g_rccl.AllReduce(bufs[i], bufs[i], (size_t)counts[i], ncclFloat, ncclSum, c->comm, (hipStream_t)stream);
        if (r != ncclSuccess) { (void)g_rccl.GroupEnd(); dep_set_error("ncclAllReduce: %s", g_rccl.GetErrorString(r)); return DEP_ERR_HIP; }
    }
    RCCL_CHECK(g_rccl.GroupEnd(), "ncclGroupEnd");
    return DEP_OK;
}

extern "C" int dep_comm_destroy(dep_comm* c) {
    if (!c) return DEP_OK;
    if (g_rccl.ok && c->comm) (void)g_rccl.CommDestroy(c->comm);
    delete c;
    return DEP_OK;
}

// Used by dep_rnn_backward_overlapped (api.hip): make `comm_stream` wait for everything enqueued so far on `compute`, then
// all-reduce the range there.  The event is created once per (thread) and reused: record / wait are stream-ordered.
int dep_comm_enqueue_after(dep_comm* c, float* buf, long n, hipStream_t compute, hipStream_t comm_stream) {
    static thread_local hipEvent_t ev = nullptr;
    if (!ev && hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { dep_set_error("hipEventCreate failed"); return DEP_ERR_HIP; }
    if (hipEventRecord(ev, compute) != hipSuccess || hipStreamWaitEvent(comm_stream, ev, 0) != hipSuccess) {
        dep_set_error("dep_comm_enqueue_after: event record / wait failed"); return DEP_ERR_HIP;
    }
    return dep_comm_allreduce(c, buf, n, (void*)comm_stream);
}
