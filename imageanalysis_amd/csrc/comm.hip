// C-ABI collectives of the hot path: RCCL over xGMI for a caller that is not
// python -- the descriptor all-gather in front of the pair-sharded matching and the all-reduce
// of the camera part of J^T u (+ scalars) inside the point-sharded LSMR.  The python layer uses
// torch.distributed (whose "nccl" backend IS RCCL) for the same two exchanges.
//
// RCCL is bound at run time (dlopen / dlsym): a process that already holds an RCCL (torch
// bundles one) re-uses that copy instead of loading a second one, and libiamx.so needs no RCCL
// to load where no collective is used.
#include <dlfcn.h>

#include "iamx_common.h"

namespace {

struct UniqueId { char internal[128]; };                   // ncclUniqueId (NCCL_UNIQUE_ID_BYTES)
typedef void *Comm;                                        // ncclComm_t
enum { kSum = 0, kUint8 = 1, kFloat64 = 8 };               // ncclSum, ncclUint8, ncclFloat64

struct Api {
    int (*GetUniqueId)(UniqueId *);
    int (*CommInitRank)(Comm *, int, UniqueId, int);
    int (*CommDestroy)(Comm);
    int (*AllGather)(const void *, void *, size_t, int, Comm, hipStream_t);
    int (*AllReduce)(const void *, void *, size_t, int, int, Comm, hipStream_t);
    const char *(*GetErrorString)(int);
    bool ok;
};

Api &api()
{
    static Api a = [] {
        Api x{};
        void *h = nullptr;
        const char *names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char *n : names)                         // a copy the process already mapped
            if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
        for (int i = 0; !h && i < 3; ++i) h = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
        if (!h) return x;
        x.GetUniqueId = reinterpret_cast<decltype(x.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
        x.CommInitRank = reinterpret_cast<decltype(x.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
        x.CommDestroy = reinterpret_cast<decltype(x.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
        x.AllGather = reinterpret_cast<decltype(x.AllGather)>(dlsym(h, "ncclAllGather"));
        x.AllReduce = reinterpret_cast<decltype(x.AllReduce)>(dlsym(h, "ncclAllReduce"));
        x.GetErrorString = reinterpret_cast<decltype(x.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
        x.ok = x.GetUniqueId && x.CommInitRank && x.CommDestroy && x.AllGather && x.AllReduce;
        return x;
    }();
    return a;
}

int rccl_fail(const char *what, int rc)
{
    const char *msg = api().GetErrorString ? api().GetErrorString(rc) : "?";
    return iamx::fail(IAMX_ELAUNCH, "%s: RCCL error %d (%s)", what, rc, msg);
}

}  // namespace

#define IAMX_NEED_RCCL()                                                              \
    do {                                                                              \
        if (!api().ok) return iamx::fail(IAMX_ENODEVICE, "%s: librccl.so not found", __func__); \
    } while (0)

extern "C" int iamx_comm_unique_id(void *id128)
{
    IAMX_REQUIRE(id128, "null pointer");
    IAMX_NEED_RCCL();
    const int rc = api().GetUniqueId(static_cast<UniqueId *>(id128));
    return rc ? rccl_fail("iamx_comm_unique_id", rc) : IAMX_OK;
}

extern "C" int iamx_comm_init(int n_ranks, int rank, const void *id128, void **comm)
{
    IAMX_REQUIRE(id128 && comm, "null pointer");
    IAMX_REQUIRE(n_ranks >= 1 && rank >= 0 && rank < n_ranks, "bad rank / world size");
    IAMX_NEED_RCCL();
    UniqueId id;
    memcpy(&id, id128, sizeof id);
    Comm c = nullptr;
    const int rc = api().CommInitRank(&c, n_ranks, id, rank);
    if (rc) return rccl_fail("iamx_comm_init", rc);
    *comm = c;
    return IAMX_OK;
}

extern "C" int iamx_comm_destroy(void *comm)
{
    IAMX_REQUIRE(comm, "null pointer");
    IAMX_NEED_RCCL();
    const int rc = api().CommDestroy(comm);
    return rc ? rccl_fail("iamx_comm_destroy", rc) : IAMX_OK;
}

extern "C" int iamx_comm_allgather(void *comm, const void *send, void *recv, int64_t bytes_per_rank,
                                   void *stream)
{
    IAMX_REQUIRE(comm && send && recv, "null pointer");
    IAMX_REQUIRE(bytes_per_rank >= 0, "negative size");
    IAMX_NEED_RCCL();
    if (bytes_per_rank == 0) return IAMX_OK;
    const int rc = api().AllGather(send, recv, (size_t)bytes_per_rank, kUint8, comm, iamx::as_stream(stream));
    return rc ? rccl_fail("iamx_comm_allgather", rc) : IAMX_OK;
}

extern "C" int iamx_comm_allreduce_f64(void *comm, double *buf, int64_t n, void *stream)
{
    IAMX_REQUIRE(comm && buf, "null pointer");
    IAMX_REQUIRE(n >= 0, "negative size");
    IAMX_NEED_RCCL();
    if (n == 0) return IAMX_OK;
    const int rc = api().AllReduce(buf, buf, (size_t)n, kFloat64, kSum, comm, iamx::as_stream(stream));
    return rc ? rccl_fail("iamx_comm_allreduce_f64", rc) : IAMX_OK;
}
