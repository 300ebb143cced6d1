// RCCL behind the C-ABI: the collectives of the data-parallel step.
//
//   bm_comm_init / bm_comm_destroy   flashy.distrib.init()              bm/train.py:139
//   bm_comm_reduce_scatter + bm_comm_allgather
//                                    flashy.distrib.sync_model(...)     bm/solver.py:386  (gradient mean:
//                                    reduce-scatter of the flat gradient bucket, Adam on the own shard,
//                                    all-gather of the updated parameters)
//   bm_comm_allgather                whole-node negatives (new vs the reference, README.md:139-143)
//   bm_comm_allreduce                flashy.distrib.average_metrics     bm/solver.py:395, BatchNorm buffers
//
// librccl is opened with dlopen at the first bm_comm_* call (path from BM_RCCL_LIB, else the default
// search path), so libbmhip.so itself loads on a box without RCCL and never mixes symbols with the copy
// PyTorch bundles.  One process per GPU; the 128-byte unique id is created by rank 0
// (bm_comm_unique_id) and handed to the other ranks by the host (TCP store / file), then every rank calls
// bm_comm_init.  All collectives are enqueued on the caller's stream, never synchronised; in-place forms
// follow NCCL's convention (all-gather: send == recv + rank * count; reduce-scatter: recv == send + rank *
// count).
#include "bm_common.h"
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <stdlib.h>
#include <string.h>

namespace {

struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*ReduceScatter)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t,
                                  hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t,
                              hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;          // optional
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;       // optional
};

RcclApi g_api;

int load_rccl() {
    if (g_api.handle) return BM_OK;
    const char* env = getenv("BM_RCCL_LIB");
    const char* candidates[] = {env, "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    void* h = nullptr;
    for (const char* c : candidates) {
        if (!c || !c[0]) continue;
        h = dlopen(c, RTLD_NOW | RTLD_LOCAL);
        if (h) break;
    }
    if (!h) return bm_set_error(BM_ERR_UNSUPPORTED, "bm_comm: cannot dlopen librccl (%s)", dlerror());
#define BM_SYM(FIELD_, NAME_)                                                                  \
    g_api.FIELD_ = reinterpret_cast<decltype(g_api.FIELD_)>(dlsym(h, NAME_));                  \
    if (!g_api.FIELD_) { dlclose(h); return bm_set_error(BM_ERR_UNSUPPORTED, "bm_comm: librccl lacks %s", NAME_); }
    BM_SYM(GetUniqueId, "ncclGetUniqueId")
    BM_SYM(CommInitRank, "ncclCommInitRank")
    BM_SYM(CommDestroy, "ncclCommDestroy")
    BM_SYM(AllGather, "ncclAllGather")
    BM_SYM(ReduceScatter, "ncclReduceScatter")
    BM_SYM(AllReduce, "ncclAllReduce")
    BM_SYM(GetErrorString, "ncclGetErrorString")
#undef BM_SYM
    g_api.CommCount = reinterpret_cast<decltype(g_api.CommCount)>(dlsym(h, "ncclCommCount"));
    g_api.CommUserRank = reinterpret_cast<decltype(g_api.CommUserRank)>(dlsym(h, "ncclCommUserRank"));
    g_api.handle = h;
    return BM_OK;
}

struct BmComm {
    ncclComm_t comm;
    int world, rank;
};

int nccl_fail(const char* what, ncclResult_t r) {
    return bm_set_error(2000 + (int)r, "%s: %s", what, g_api.GetErrorString ? g_api.GetErrorString(r) : "rccl error");
}

}  // namespace

extern "C" int bm_comm_unique_id_bytes(void) { return NCCL_UNIQUE_ID_BYTES; }

// 0 when librccl can be opened and exports every symbol bound above (a LOCAL check: the host agrees on it over
// all ranks before anybody enters the collective ncclCommInitRank)
extern "C" int bm_comm_available(void) { return load_rccl(); }

extern "C" int bm_comm_unique_id(void* out_id) {
    BM_REQUIRE(out_id, "bm_comm_unique_id: null pointer");
    if (int rc = load_rccl()) return rc;
    ncclUniqueId id;
    const ncclResult_t r = g_api.GetUniqueId(&id);
    if (r != ncclSuccess) return nccl_fail("ncclGetUniqueId", r);
    memcpy(out_id, &id, NCCL_UNIQUE_ID_BYTES);
    return BM_OK;
}

extern "C" int bm_comm_init(const void* id, int world, int rank, int device, void** handle) {
    BM_REQUIRE(id && handle, "bm_comm_init: null pointer");
    BM_REQUIRE(world >= 1 && rank >= 0 && rank < world, "bm_comm_init: bad rank %d / world %d", rank, world);
    if (int rc = load_rccl()) return rc;
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return bm_set_error((int)e, "bm_comm_init: hipSetDevice(%d): %s", device, hipGetErrorString(e));
    ncclUniqueId uid;
    memcpy(&uid, id, NCCL_UNIQUE_ID_BYTES);
    BmComm* c = new BmComm{nullptr, world, rank};
    const ncclResult_t r = g_api.CommInitRank(&c->comm, world, uid, rank);
    if (r != ncclSuccess) { delete c; return nccl_fail("ncclCommInitRank", r); }
    *handle = c;
    return BM_OK;
}

extern "C" int bm_comm_destroy(void* handle) {
    if (!handle) return BM_OK;
    BmComm* c = static_cast<BmComm*>(handle);
    const ncclResult_t r = g_api.CommDestroy(c->comm);
    delete c;
    return r == ncclSuccess ? BM_OK : nccl_fail("ncclCommDestroy", r);
}

extern "C" int bm_comm_world(void* handle) { return handle ? static_cast<BmComm*>(handle)->world : 0; }
extern "C" int bm_comm_rank(void* handle) { return handle ? static_cast<BmComm*>(handle)->rank : -1; }

// The communicator's size / this process' rank AS RCCL REPORTS THEM (ncclCommCount / ncclCommUserRank): what a
// multi-GPU run prints next to its numbers, so that "8 ranks" is the library's statement and not the launcher's.
// -1 when the handle is null, the query fails or librccl lacks the symbol.
extern "C" int bm_comm_reported_world(void* handle) {
    if (!handle || !g_api.CommCount) return -1;
    int n = -1;
    return g_api.CommCount(static_cast<BmComm*>(handle)->comm, &n) == ncclSuccess ? n : -1;
}
extern "C" int bm_comm_reported_rank(void* handle) {
    if (!handle || !g_api.CommUserRank) return -1;
    int r = -1;
    return g_api.CommUserRank(static_cast<BmComm*>(handle)->comm, &r) == ncclSuccess ? r : -1;
}

// recv[r * count .. (r + 1) * count) = send of rank r, for every rank (fp32).
extern "C" int bm_comm_allgather(void* handle, const float* send, float* recv, long count, void* stream) {
    BM_REQUIRE(handle && send && recv && count >= 0, "bm_comm_allgather: bad arguments");
    BmComm* c = static_cast<BmComm*>(handle);
    const ncclResult_t r = g_api.AllGather(send, recv, (size_t)count, ncclFloat32, c->comm, (hipStream_t)stream);
    return r == ncclSuccess ? BM_OK : nccl_fail("ncclAllGather", r);
}

// recv[0 .. count) = sum over ranks of send[rank * count .. (rank + 1) * count)   (fp32, `send` holds world * count).
extern "C" int bm_comm_reduce_scatter(void* handle, const float* send, float* recv, long count, void* stream) {
    BM_REQUIRE(handle && send && recv && count >= 0, "bm_comm_reduce_scatter: bad arguments");
    BmComm* c = static_cast<BmComm*>(handle);
    const ncclResult_t r =
        g_api.ReduceScatter(send, recv, (size_t)count, ncclFloat32, ncclSum, c->comm, (hipStream_t)stream);
    return r == ncclSuccess ? BM_OK : nccl_fail("ncclReduceScatter", r);
}

// recv = reduce over ranks of send (fp32); op 0 = sum, 1 = max.  In place when send == recv.
extern "C" int bm_comm_allreduce(void* handle, const float* send, float* recv, long count, int op, void* stream) {
    BM_REQUIRE(handle && send && recv && count >= 0, "bm_comm_allreduce: bad arguments");
    BM_REQUIRE(op == 0 || op == 1, "bm_comm_allreduce: op must be 0 (sum) or 1 (max)");
    BmComm* c = static_cast<BmComm*>(handle);
    const ncclResult_t r = g_api.AllReduce(send, recv, (size_t)count, ncclFloat32, op == 0 ? ncclSum : ncclMax,
                                           c->comm, (hipStream_t)stream);
    return r == ncclSuccess ? BM_OK : nccl_fail("ncclAllReduce", r);
}
