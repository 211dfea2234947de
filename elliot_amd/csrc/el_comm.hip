// Collectives of the multi-GPU path behind the C ABI (el_comm_init / el_allreduce_rows / el_allgather_topk):
// RCCL over xGMI, one communicator per rank (one process per GPU).  The reference is single-device; these are the three
// exchange steps of the sharded designs in DESIGN.md section 4:
//   el_allreduce_rows       sum of gradient rows over the ranks, in place (item gradients gGi / gBi of the user-sharded step,
//                           the dense gradients of NeuMF / Mult-VAE)
//   el_reduce_scatter_rows  + el_allgather_rows: the "dense" exchange of the item-sharded step (summed user-gradient table
//                           scattered by owner, updated rows gathered back)
//   el_allgather_topk       partial top-k lists of every item shard -> [G, n, k] on every rank (el_topk_merge follows)
// RCCL is bound at RUN time (dlsym of the nccl* entry points in the process, else dlopen("librccl.so.1")): the library has
// no link-time dependency on it, a host that never shards never loads it, and a Python host shares the RCCL that
// torch already mapped.
#include "el_common.h"
#include <dlfcn.h>

typedef struct { char internal[128]; } el_nccl_id;
typedef void* el_nccl_comm;
enum { EL_NCCL_FLOAT32 = 7, EL_NCCL_INT32 = 2, EL_NCCL_INT8 = 0, EL_NCCL_SUM = 0 };   // rccl.h ncclDataType_t / ncclRedOp_t

struct ElNccl {
    int (*GetUniqueId)(el_nccl_id*);
    int (*CommInitRank)(el_nccl_comm*, int, el_nccl_id, int);
    int (*CommDestroy)(el_nccl_comm);
    int (*AllReduce)(const void*, void*, size_t, int, int, el_nccl_comm, hipStream_t);
    int (*AllGather)(const void*, void*, size_t, int, el_nccl_comm, hipStream_t);
    int (*ReduceScatter)(const void*, void*, size_t, int, int, el_nccl_comm, hipStream_t);
    int (*GroupStart)();
    int (*GroupEnd)();
    const char* (*GetErrorString)(int);
    bool ok;
};

static ElNccl* el_nccl() {
    static ElNccl api = [] {
        ElNccl a;
        memset(&a, 0, sizeof(a));
        void* h = RTLD_DEFAULT;
        if (!dlsym(RTLD_DEFAULT, "ncclAllReduce")) {
            h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
            if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
            if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
            if (!h) return a;
        }
#define EL_SYM(field, name) *(void**)(&a.field) = dlsym(h, name)
        EL_SYM(GetUniqueId, "ncclGetUniqueId");
        EL_SYM(CommInitRank, "ncclCommInitRank");
        EL_SYM(CommDestroy, "ncclCommDestroy");
        EL_SYM(AllReduce, "ncclAllReduce");
        EL_SYM(AllGather, "ncclAllGather");
        EL_SYM(ReduceScatter, "ncclReduceScatter");
        EL_SYM(GroupStart, "ncclGroupStart");
        EL_SYM(GroupEnd, "ncclGroupEnd");
        EL_SYM(GetErrorString, "ncclGetErrorString");
#undef EL_SYM
        a.ok = a.GetUniqueId && a.CommInitRank && a.CommDestroy && a.AllReduce && a.AllGather && a.ReduceScatter && a.GroupStart &&
               a.GroupEnd && a.GetErrorString;
        return a;
    }();
    return &api;
}

struct el_comm {
    el_nccl_comm comm;
    int rank, world, device;
};

#define EL_NCCL(call, what)                                                                         \
    do {                                                                                            \
        const int _r = (call);                                                                      \
        if (_r != 0) {                                                                              \
            el_set_error("%s: RCCL error %d (%s)", what, _r, el_nccl()->GetErrorString(_r));        \
            return 1;                                                                               \
        }                                                                                           \
    } while (0)

extern "C" int el_comm_unique_id(void* id128) {
    EL_REQUIRE(id128 != nullptr, "el_comm_unique_id: null buffer");
    EL_REQUIRE(el_nccl()->ok, "el_comm_unique_id: RCCL (librccl.so) is not available in this process");
    el_nccl_id id;
    EL_NCCL(el_nccl()->GetUniqueId(&id), "el_comm_unique_id");
    memcpy(id128, &id, sizeof(id));
    return 0;
}

extern "C" int el_comm_init(el_ctx* ctx, const void* id128, int rank, int world, el_comm** out) {
    if (int rc = el_bind(ctx)) return rc;
    EL_REQUIRE(id128 && out && world >= 1 && rank >= 0 && rank < world, "el_comm_init: bad arguments (rank %d of %d)", rank, world);
    EL_REQUIRE(el_nccl()->ok, "el_comm_init: RCCL (librccl.so) is not available in this process");
    el_nccl_id id;
    memcpy(&id, id128, sizeof(id));
    el_comm* c = new el_comm();
    c->rank = rank, c->world = world, c->device = ctx->device, c->comm = nullptr;
    const int r = el_nccl()->CommInitRank(&c->comm, world, id, rank);
    if (r != 0) {
        el_set_error("el_comm_init: ncclCommInitRank failed: %d (%s)", r, el_nccl()->GetErrorString(r));
        delete c;
        return 1;
    }
    *out = c;
    return 0;
}

extern "C" int el_comm_destroy(el_comm* c) {
    if (!c) return 0;
    if (c->comm && el_nccl()->ok) (void)el_nccl()->CommDestroy(c->comm);
    delete c;
    return 0;
}

extern "C" int el_comm_rank(const el_comm* c, int* rank, int* world) {
    EL_REQUIRE(c && rank && world, "el_comm_rank: null argument");
    *rank = c->rank, *world = c->world;
    return 0;
}

extern "C" int el_allreduce_rows(el_ctx* ctx, el_comm* c, void* stream, float* buf, int64_t count) {
    if (int rc = el_bind(ctx)) return rc;
    EL_REQUIRE(c && buf && count >= 0, "el_allreduce_rows: bad arguments");
    if (count == 0) return 0;
    EL_NCCL(el_nccl()->AllReduce(buf, buf, (size_t)count, EL_NCCL_FLOAT32, EL_NCCL_SUM, c->comm, (hipStream_t)stream), "el_allreduce_rows");
    return 0;
}

extern "C" int el_reduce_scatter_rows(el_ctx* ctx, el_comm* c, void* stream, const float* full, float* own, int64_t count_per_rank) {
    if (int rc = el_bind(ctx)) return rc;
    EL_REQUIRE(c && full && own && count_per_rank >= 0, "el_reduce_scatter_rows: bad arguments");
    if (count_per_rank == 0) return 0;
    EL_NCCL(el_nccl()->ReduceScatter(full, own, (size_t)count_per_rank, EL_NCCL_FLOAT32, EL_NCCL_SUM, c->comm, (hipStream_t)stream),
            "el_reduce_scatter_rows");
    return 0;
}

extern "C" int el_allgather_rows(el_ctx* ctx, el_comm* c, void* stream, const void* part, void* full, int64_t bytes_per_rank) {
    if (int rc = el_bind(ctx)) return rc;
    EL_REQUIRE(c && part && full && bytes_per_rank >= 0, "el_allgather_rows: bad arguments");
    if (bytes_per_rank == 0) return 0;
    EL_NCCL(el_nccl()->AllGather(part, full, (size_t)bytes_per_rank, EL_NCCL_INT8, c->comm, (hipStream_t)stream), "el_allgather_rows");
    return 0;
}

extern "C" int el_allgather_topk(el_ctx* ctx, el_comm* c, void* stream, const int32_t* part_idx, const float* part_val, int64_t n_users,
                                 int32_t k, int32_t* all_idx, float* all_val) {
    if (int rc = el_bind(ctx)) return rc;
    EL_REQUIRE(c && part_idx && part_val && all_idx && all_val && n_users >= 0 && k >= 1, "el_allgather_topk: bad arguments");
    if (n_users == 0) return 0;
    const size_t cnt = (size_t)n_users * (size_t)k;
    EL_NCCL(el_nccl()->GroupStart(), "el_allgather_topk");       // both lists in one RCCL group: one launch, no interleaving hazard
    const int r1 = el_nccl()->AllGather(part_idx, all_idx, cnt, EL_NCCL_INT32, c->comm, (hipStream_t)stream);
    const int r2 = el_nccl()->AllGather(part_val, all_val, cnt, EL_NCCL_FLOAT32, c->comm, (hipStream_t)stream);
    EL_NCCL(el_nccl()->GroupEnd(), "el_allgather_topk");
    EL_NCCL(r1, "el_allgather_topk (indices)");
    EL_NCCL(r2, "el_allgather_topk (scores)");
    return 0;
}
