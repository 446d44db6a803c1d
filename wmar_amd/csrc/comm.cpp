// wmar_comm_*: the exchange step of the sharded job on RCCL (over xGMI inside a node), C ABI.
//
// Reference: generate.py:204 / :304 shard the image list over independent chunk processes and exchange NOTHING; SURVEY section 8(e)
// adds the two steps a one-node job wants -- the finished key table broadcast once (32 MiB for Taming h = 1) and the per-image
// records (codes int64[n, L], counts int32[n], p-values f64[n]) gathered once per step.  These three entry points are what a host in
// any language binds for that; the Python host of this build reaches the same RCCL through torch.distributed("nccl") by default
// (wmar_amd/harness.py) and through this ABI with WMAR_COMM=rccl (wmar_amd/comm.py).
//
// RCCL is resolved at run time (dlopen / dlsym), not at link time: a process that already carries a librccl (PyTorch-ROCm ships its
// own) must keep using exactly that one -- two RCCL builds in one address space break both.
#include <dlfcn.h>

#include <rccl/rccl.h>

#include "common.h"

namespace {

struct Rccl {
    void* h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

Rccl& rccl() {
    static Rccl r;
    if (r.h) return r;
    const char* names[] = {"librccl.so", "librccl.so.1"};
    for (const char* n : names) if (!r.h) r.h = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);    // the copy the process already uses
    for (const char* n : names) if (!r.h) r.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (!r.h) return r;
#define WMAR_SYM(F) r.F = (decltype(r.F))dlsym(r.h, "nccl" #F)
    WMAR_SYM(GetUniqueId); WMAR_SYM(CommInitRank); WMAR_SYM(Broadcast); WMAR_SYM(AllGather); WMAR_SYM(CommDestroy); WMAR_SYM(GetErrorString);
#undef WMAR_SYM
    r.ok = r.GetUniqueId && r.CommInitRank && r.Broadcast && r.AllGather && r.CommDestroy && r.GetErrorString;
    return r;
}

int need_rccl() {
    if (rccl().ok) return WMAR_OK;
    const char* why = dlerror();
    wmar::set_error("wmar_comm: librccl.so could not be loaded (%s)", why ? why : "symbols missing");
    return WMAR_EHIP;
}

#define WMAR_NCCL(expr)                                                                           \
    do {                                                                                          \
        ncclResult_t e_ = (expr);                                                                 \
        if (e_ != ncclSuccess) {                                                                  \
            wmar::set_error("%s failed: %s", #expr, rccl().GetErrorString(e_));                   \
            return WMAR_EHIP;                                                                     \
        }                                                                                         \
    } while (0)

}  // namespace

struct wmar_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
};

extern "C" {

int wmar_comm_unique_id(void* id_out, int64_t id_bytes) {
    WMAR_REQUIRE(id_out && id_bytes >= (int64_t)sizeof(ncclUniqueId), "comm_unique_id: need a buffer of %d bytes", (int)sizeof(ncclUniqueId));
    if (int rc = need_rccl()) return rc;
    ncclUniqueId id;
    WMAR_NCCL(rccl().GetUniqueId(&id));
    memcpy(id_out, &id, sizeof id);
    return WMAR_OK;
}

int wmar_comm_init(const void* id, int64_t id_bytes, int32_t rank, int32_t world, wmar_comm** out) {
    WMAR_REQUIRE(id && out && id_bytes >= (int64_t)sizeof(ncclUniqueId), "comm_init: null argument / short id");
    WMAR_REQUIRE(world >= 1 && rank >= 0 && rank < world, "comm_init: rank %d of %d", rank, world);
    if (int rc = need_rccl()) return rc;
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof uid);
    auto* c = new wmar_comm();
    c->rank = rank; c->world = world;
    ncclResult_t e = rccl().CommInitRank(&c->comm, world, uid, rank);
    if (e != ncclSuccess) { wmar::set_error("ncclCommInitRank failed: %s", rccl().GetErrorString(e)); delete c; return WMAR_EHIP; }
    *out = c;
    return WMAR_OK;
}

int wmar_comm_bcast(wmar_comm* c, void* buf_dev, int64_t bytes, int32_t root, void* stream) {
    WMAR_REQUIRE(c && buf_dev && bytes >= 0 && root >= 0 && root < c->world, "comm_bcast: bad argument");
    if (bytes == 0) return WMAR_OK;
    WMAR_NCCL(rccl().Broadcast(buf_dev, buf_dev, (size_t)bytes, ncclChar, root, c->comm, (hipStream_t)stream));
    return WMAR_OK;
}

int wmar_comm_allgather(wmar_comm* c, const void* send_dev, void* recv_dev, int64_t bytes_per_rank, void* stream) {
    WMAR_REQUIRE(c && send_dev && recv_dev && bytes_per_rank >= 0, "comm_allgather: bad argument");
    if (bytes_per_rank == 0) return WMAR_OK;
    WMAR_NCCL(rccl().AllGather(send_dev, recv_dev, (size_t)bytes_per_rank, ncclChar, c->comm, (hipStream_t)stream));
    return WMAR_OK;
}

int32_t wmar_comm_rank(const wmar_comm* c) { return c ? c->rank : -1; }
int32_t wmar_comm_world(const wmar_comm* c) { return c ? c->world : 0; }

void wmar_comm_destroy(wmar_comm* c) {
    if (!c) return;
    if (c->comm && rccl().ok) (void)rccl().CommDestroy(c->comm);
    delete c;
}

}  // extern "C"
