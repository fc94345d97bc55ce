// snmf_tu_rccl.hip -- the process-per-GPU loop's collective on the C side (round 6).  snmf_plan_run_sharded takes the all-reduce as a
// callback; from Python that callback was a ctypes trampoline into torch.distributed: 29 us of host work per iteration, a quarter of
// a 12 500-frame shard's iteration at BASELINE configs[1] on eight ranks (profiles/r05_shard_proxy.txt).  Here the callback is C and
// calls ncclAllReduce on the engine's stream.  RCCL is resolved with dlopen, so libsnmf_hip.so links and loads without it:
// SNMF_RCCL_LIB, else the librccl.so the process already holds (PyTorch's), else the system's.  The communicator is the caller's to
// set up: rank 0 asks for an id (snmf_rccl_get_unique_id), every rank receives it over whatever the host already has (torch.distributed,
// MPI, a file) and calls snmf_rccl_comm_create.  Reference: one sum per iteration of the statistics of src/sparse_nmf.m:215-239 / :248-261.
#include "snmf_internal.h"

#include <dlfcn.h>

namespace {
typedef struct { char internal[128]; } NcclUniqueId;  // rccl.h: NCCL_UNIQUE_ID_BYTES = 128
typedef void* NcclComm;
struct RcclApi {
    void* lib = nullptr;
    int (*GetUniqueId)(NcclUniqueId*) = nullptr;
    int (*CommInitRank)(NcclComm*, int, NcclUniqueId, int) = nullptr;
    int (*CommDestroy)(NcclComm) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, NcclComm, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool ok = false;
};
constexpr int kNcclFloat64 = 8, kNcclSum = 0;  // rccl.h: ncclFloat64 = ncclDouble = 8, ncclSum = 0

RcclApi* rccl() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* env = getenv("SNMF_RCCL_LIB");
        const char* names[] = {env, "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char* n : names) {
            if (!n || !*n) continue;
            // a copy the process already holds first (PyTorch loads its own librccl.so: one RCCL per process)
            api.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
            if (!api.lib) api.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (api.lib) break;
        }
        if (!api.lib) return;
        api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(api.lib, "ncclGetUniqueId");
        api.CommInitRank = (decltype(api.CommInitRank))dlsym(api.lib, "ncclCommInitRank");
        api.CommDestroy = (decltype(api.CommDestroy))dlsym(api.lib, "ncclCommDestroy");
        api.AllReduce = (decltype(api.AllReduce))dlsym(api.lib, "ncclAllReduce");
        api.GetErrorString = (decltype(api.GetErrorString))dlsym(api.lib, "ncclGetErrorString");
        api.ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.AllReduce;
    });
    return &api;
}
const char* rccl_err(int rc) {
    RcclApi* r = rccl();
    return r->GetErrorString ? r->GetErrorString(rc) : "RCCL error";
}
struct RcclComm {
    NcclComm comm = nullptr;
    int device = 0, n_ranks = 1, rank = 0;
};
struct RcclCb {
    RcclComm* c;
    hipStream_t stream;
    int rc;
};
int rccl_all_reduce_cb(double* stats_dev, int64_t len, void* user) {
    RcclCb* u = (RcclCb*)user;
    u->rc = rccl()->AllReduce(stats_dev, stats_dev, (size_t)len, kNcclFloat64, kNcclSum, u->c->comm, u->stream);
    return u->rc == 0 ? 0 : 1;
}
}  // namespace

extern "C" int snmf_rccl_available(void) { return rccl()->ok ? 1 : 0; }

extern "C" int snmf_rccl_get_unique_id(void* id_out, int64_t cap) {
    if (!id_out || cap < (int64_t)sizeof(NcclUniqueId)) return fail(SNMF_ERR_INVALID, "id buffer must hold %zu bytes", sizeof(NcclUniqueId));
    if (!rccl()->ok) return fail(SNMF_ERR_UNSUPPORTED, "librccl.so could not be loaded (SNMF_RCCL_LIB overrides the search)");
    NcclUniqueId id;
    const int rc = rccl()->GetUniqueId(&id);
    if (rc != 0) return fail(SNMF_ERR_INTERNAL, "ncclGetUniqueId: %s", rccl_err(rc));
    std::memcpy(id_out, &id, sizeof id);
    return SNMF_OK;
}

extern "C" int snmf_rccl_comm_create(int32_t device, const void* id_in, int32_t n_ranks, int32_t rank, snmf_rccl_comm** out) {
    if (!id_in || !out) return fail(SNMF_ERR_INVALID, "NULL argument");
    *out = nullptr;
    if (n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(SNMF_ERR_INVALID, "rank %d of %d", rank, n_ranks);
    if (!rccl()->ok) return fail(SNMF_ERR_UNSUPPORTED, "librccl.so could not be loaded (SNMF_RCCL_LIB overrides the search)");
    (void)hipGetLastError();
    HIP_TRY(hipSetDevice(device));
    NcclUniqueId id;
    std::memcpy(&id, id_in, sizeof id);
    RcclComm* c = new RcclComm();
    c->device = device;
    c->n_ranks = n_ranks;
    c->rank = rank;
    const int rc = rccl()->CommInitRank(&c->comm, n_ranks, id, rank);
    if (rc != 0) {
        delete c;
        return fail(SNMF_ERR_INTERNAL, "ncclCommInitRank(rank %d of %d on device %d): %s", rank, n_ranks, device, rccl_err(rc));
    }
    *out = reinterpret_cast<snmf_rccl_comm*>(c);
    return SNMF_OK;
}

extern "C" void snmf_rccl_comm_destroy(snmf_rccl_comm* comm) {
    RcclComm* c = reinterpret_cast<RcclComm*>(comm);
    if (!c) return;
    if (c->comm && rccl()->ok) {
        (void)hipSetDevice(c->device);
        (void)rccl()->CommDestroy(c->comm);
    }
    delete c;
}

extern "C" int snmf_plan_run_sharded_rccl(snmf_plan* pl, int32_t n_iters, double* stats, snmf_rccl_comm* comm, int32_t poll_every,
                                          int32_t finalize, int32_t* iters_done) {
    PLAN_CHECK(pl);
    RcclComm* c = reinterpret_cast<RcclComm*>(comm);
    if (!c || !c->comm) return fail(SNMF_ERR_INVALID, "communicator is NULL");
    if (c->device != pl->ctx->device) return fail(SNMF_ERR_INVALID, "the communicator lives on device %d, the plan on device %d", c->device, pl->ctx->device);
    RcclCb u{c, pl->ctx->stream, 0};
    const int s = snmf_plan_run_sharded(pl, n_iters, stats, rccl_all_reduce_cb, &u, poll_every, finalize, iters_done);
    if (u.rc != 0) return fail(SNMF_ERR_INTERNAL, "ncclAllReduce: %s", rccl_err(u.rc));
    return s;
}
