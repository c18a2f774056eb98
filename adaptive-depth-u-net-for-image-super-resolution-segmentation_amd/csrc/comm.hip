// Gradient exchange inside the native boundary (SURVEY 8b `ad_allreduce_bucket`): RCCL all-reduce of one bucket of the
// flat fp32 gradient buffer, in place, on the caller's communication stream.  The reference is single-GPU; the contract
// is SURVEY 8e: data parallelism over patches, sum of the per-rank gradients, 1/world applied by the optimizer.
// RCCL is resolved with dlopen at the first call, so the library itself has no link-time dependency on it (the CPU-only
// checks load libadunet_hip.so without touching RCCL).
#include "common.h"
#include <dlfcn.h>
#include <string.h>

namespace {

struct UniqueId { char bytes[128]; };          // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128), passed by value
typedef int (*get_unique_id_t)(UniqueId*);
typedef int (*comm_init_rank_t)(void**, int, UniqueId, int);
typedef int (*comm_destroy_t)(void*);
typedef int (*all_reduce_t)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef const char* (*get_error_string_t)(int);

struct Rccl {
    void* handle = nullptr;
    get_unique_id_t get_unique_id = nullptr;
    comm_init_rank_t comm_init_rank = nullptr;
    comm_destroy_t comm_destroy = nullptr;
    all_reduce_t all_reduce = nullptr;
    get_error_string_t error_string = nullptr;
};

Rccl* rccl() {
    static Rccl r;
    static bool tried = false;
    if (!tried) {
        tried = true;
        for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
            r.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.handle) break;
        }
        if (r.handle) {
            r.get_unique_id = (get_unique_id_t)dlsym(r.handle, "ncclGetUniqueId");
            r.comm_init_rank = (comm_init_rank_t)dlsym(r.handle, "ncclCommInitRank");
            r.comm_destroy = (comm_destroy_t)dlsym(r.handle, "ncclCommDestroy");
            r.all_reduce = (all_reduce_t)dlsym(r.handle, "ncclAllReduce");
            r.error_string = (get_error_string_t)dlsym(r.handle, "ncclGetErrorString");
        }
    }
    const bool ok = r.handle && r.get_unique_id && r.comm_init_rank && r.comm_destroy && r.all_reduce;
    return ok ? &r : nullptr;
}

int rccl_error(const char* what, int rc) {
    Rccl* r = rccl();
    return ad_set_error(AD_ERR_LAUNCH, "%s: RCCL error %d (%s)", what, rc, r && r->error_string ? r->error_string(rc) : "?");
}

constexpr int kNcclFloat = 7, kNcclSum = 0;    // ncclFloat32, ncclSum

}  // namespace

extern "C" int ad_comm_unique_id(void* id128) {
    AD_REQUIRE(id128, "ad_comm_unique_id: NULL buffer");
    Rccl* r = rccl();
    if (!r) return ad_set_error(AD_ERR_LAUNCH, "ad_comm_unique_id: librccl.so could not be loaded");
    UniqueId id;
    const int rc = r->get_unique_id(&id);
    if (rc) return rccl_error("ncclGetUniqueId", rc);
    memcpy(id128, id.bytes, sizeof(id.bytes));
    return AD_OK;
}

extern "C" int ad_comm_create(const void* id128, int rank, int world, void** comm) {
    AD_REQUIRE(id128 && comm && world > 0 && rank >= 0 && rank < world, "ad_comm_create: bad arguments rank=%d world=%d", rank, world);
    Rccl* r = rccl();
    if (!r) return ad_set_error(AD_ERR_LAUNCH, "ad_comm_create: librccl.so could not be loaded");
    UniqueId id;
    memcpy(id.bytes, id128, sizeof(id.bytes));
    const int rc = r->comm_init_rank(comm, world, id, rank);
    return rc ? rccl_error("ncclCommInitRank", rc) : AD_OK;
}

extern "C" int ad_comm_destroy(void* comm) {
    if (!comm) return AD_OK;
    Rccl* r = rccl();
    if (!r) return ad_set_error(AD_ERR_LAUNCH, "ad_comm_destroy: librccl.so could not be loaded");
    const int rc = r->comm_destroy(comm);
    return rc ? rccl_error("ncclCommDestroy", rc) : AD_OK;
}

extern "C" int ad_allreduce_bucket(void* comm, float* grads, int64_t count, void* stream) {
    AD_REQUIRE(comm && grads && count > 0, "ad_allreduce_bucket: bad arguments");
    Rccl* r = rccl();
    if (!r) return ad_set_error(AD_ERR_LAUNCH, "ad_allreduce_bucket: librccl.so could not be loaded");
    const int rc = r->all_reduce(grads, grads, (size_t)count, kNcclFloat, kNcclSum, comm, (hipStream_t)stream);
    return rc ? rccl_error("ncclAllReduce", rc) : AD_OK;
}
