// comm.hip — the one exchange step of the path behind the C ABI: RCCL (librccl, xGMI) called directly, no PyTorch.
//
// The reference fans SNP column blocks out to a process pool and concatenates the per-block result lists in block
// order (lmm/lmm.py:378-403, SampleIter :413-436).  Here a block lives on one GPU; what crosses GPUs is
//   * once per run: U, d and the rotated y/W from the GPU that ran the eigensolver        -> pg_comm_broadcast_dev
//   * once at the end: the 32-byte result rows of every block, in rank order = SNP order -> pg_comm_allgather_dev
// librccl.so (573 MB) is dlopen'ed on first use, so single-GPU runs never map it.
// Two ways to form a communicator:
//   pg_comm_init_rank : one process per GPU (bench.py ranks, any launcher); rank 0 makes the 128-byte id with
//                       pg_comm_unique_id and hands it to the others by any host channel (pygemma_amd/dist.py: a file/TCP
//                       rendezvous in the standard library)
//   pg_comm_init_all  : one process driving several GPUs with one host thread each (lmm.pygemma(nproc=N), pg_assoc_multi)
#include "common.hpp"

#include <dlfcn.h>
#include <mutex>

namespace pg {

// Types, enum values and prototypes come from the installed header itself (nothing hand-declared: a librccl whose ABI moved
// fails here at build time, not at run time).  Only declarations are used: every entry point is resolved with dlsym, so the
// library is not a link-time dependency.
}  // namespace pg
#include <rccl/rccl.h>
namespace pg {
static_assert(sizeof(ncclUniqueId) == 128, "pg_comm_unique_id / pg_comm_init_rank hand a 128-byte id across the C ABI");

struct Rccl {
    void *handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
};
static Rccl g_rccl;
static std::mutex g_rccl_mu;

static int rccl_load()
{
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.handle) return PG_OK;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const char *nm : names) { h = dlopen(nm, RTLD_NOW | RTLD_LOCAL); if (h) break; }
    if (!h) { set_error("RCCL not found: dlopen(librccl.so.1) failed: %s", dlerror()); return PG_ENOTSUP; }
    Rccl r;
    r.handle = h;
#define PG_SYM(field, name) \
    do { *(void **)(&r.field) = dlsym(h, name); if (!r.field) { set_error("librccl lacks %s", name); dlclose(h); return PG_ENOTSUP; } } while (0)
    PG_SYM(GetUniqueId, "ncclGetUniqueId");
    PG_SYM(CommInitRank, "ncclCommInitRank");
    PG_SYM(CommInitAll, "ncclCommInitAll");
    PG_SYM(CommDestroy, "ncclCommDestroy");
    PG_SYM(GetErrorString, "ncclGetErrorString");
    PG_SYM(Broadcast, "ncclBroadcast");
    PG_SYM(AllGather, "ncclAllGather");
    PG_SYM(AllReduce, "ncclAllReduce");
    PG_SYM(GroupStart, "ncclGroupStart");
    PG_SYM(GroupEnd, "ncclGroupEnd");
#undef PG_SYM
    g_rccl = r;
    return PG_OK;
}

#define PG_NCCL(call)                                                                                   \
    do {                                                                                                \
        ncclResult_t _r = (call);                                                                             \
        if (_r != ncclSuccess) {                                                                        \
            pg::set_error("%s failed: %s (%s:%d)", #call, g_rccl.GetErrorString(_r), __FILE__, __LINE__); \
            return PG_EHIP;                                                                             \
        }                                                                                               \
    } while (0)

}  // namespace pg

using namespace pg;

struct pg_comm {
    pg_ctx *ctx = nullptr;       // device + stream every collective of this communicator is enqueued on
    ncclComm_t comm = nullptr;
    int nranks = 1, rank = 0;
    double *scratch = nullptr;   // 8 bytes of device memory for the barrier
};

extern "C" int pg_comm_unique_id(void *id128)
{
    PG_REQUIRE(id128, "pg_comm_unique_id: NULL argument");
    int rc = rccl_load();
    if (rc) return rc;
    ncclUniqueId id;
    PG_NCCL(g_rccl.GetUniqueId(&id));
    memcpy(id128, id.internal, 128);
    return PG_OK;
}

static int comm_finish(pg_comm *c)
{
    PG_HIP(hipSetDevice(c->ctx->device));
    PG_HIP(hipMalloc(&c->scratch, 8));
    PG_HIP(hipMemsetAsync(c->scratch, 0, 8, c->ctx->stream));
    return PG_OK;
}

extern "C" int pg_comm_init_rank(pg_ctx *ctx, int nranks, int rank, const void *id128, pg_comm **out)
{
    PG_REQUIRE(ctx && id128 && out && nranks >= 1 && rank >= 0 && rank < nranks, "pg_comm_init_rank: bad arguments");
    *out = nullptr;
    int rc = rccl_load();
    if (rc) return rc;
    PG_HIP(hipSetDevice(ctx->device));
    ncclUniqueId id;
    memcpy(id.internal, id128, 128);
    pg_comm *c = new (std::nothrow) pg_comm();
    if (!c) { set_error("pg_comm_init_rank: out of host memory"); return PG_ENOMEM; }
    c->ctx = ctx; c->nranks = nranks; c->rank = rank;
    ncclResult_t r = g_rccl.CommInitRank(&c->comm, nranks, id, rank);
    if (r != ncclSuccess) { set_error("ncclCommInitRank(%d of %d) failed: %s", rank, nranks, g_rccl.GetErrorString(r)); delete c; return PG_EHIP; }
    rc = comm_finish(c);
    if (rc) { g_rccl.CommDestroy(c->comm); delete c; return rc; }
    *out = c;
    return PG_OK;
}

extern "C" int pg_comm_init_all(int ndev, pg_ctx *const *ctxs, pg_comm **out)
{
    PG_REQUIRE(ctxs && out && ndev >= 1 && ndev <= 64, "pg_comm_init_all: bad arguments");
    for (int g = 0; g < ndev; g++) { out[g] = nullptr; PG_REQUIRE(ctxs[g], "pg_comm_init_all: NULL context %d", g); }
    int rc = rccl_load();
    if (rc) return rc;
    std::vector<int> devs(ndev);
    std::vector<ncclComm_t> comms(ndev, nullptr);
    for (int g = 0; g < ndev; g++) devs[g] = ctxs[g]->device;
    PG_NCCL(g_rccl.CommInitAll(comms.data(), ndev, devs.data()));
    // every RCCL handle gets its owner before any step that can fail: on a failure the whole set is destroyed here and
    // the caller receives no handles (the communicators behind a failing index used to leak)
    std::vector<pg_comm *> made(ndev, nullptr);
    bool oom = false;
    for (int g = 0; g < ndev; g++) {
        made[g] = new (std::nothrow) pg_comm();
        if (!made[g]) { oom = true; continue; }
        made[g]->ctx = ctxs[g]; made[g]->comm = comms[g]; made[g]->nranks = ndev; made[g]->rank = g;
    }
    rc = PG_OK;
    if (oom) { set_error("pg_comm_init_all: out of host memory"); rc = PG_ENOMEM; }
    for (int g = 0; g < ndev && !rc; g++) rc = comm_finish(made[g]);
    if (rc) {
        for (int g = 0; g < ndev; g++) {
            if (made[g]) { (void)pg_comm_destroy(made[g]); }
            else if (comms[g]) { (void)g_rccl.CommDestroy(comms[g]); }
        }
        return rc;
    }
    for (int g = 0; g < ndev; g++) out[g] = made[g];
    return PG_OK;
}

extern "C" int pg_comm_destroy(pg_comm *c)
{
    if (!c) return PG_OK;
    (void)hipSetDevice(c->ctx->device);
    (void)hipStreamSynchronize(c->ctx->stream);
    if (c->comm) (void)g_rccl.CommDestroy(c->comm);
    if (c->scratch) (void)hipFree(c->scratch);
    delete c;
    return PG_OK;
}

extern "C" int pg_comm_size(const pg_comm *c) { return c ? c->nranks : 0; }
extern "C" int pg_comm_rank(const pg_comm *c) { return c ? c->rank : -1; }

extern "C" int pg_comm_broadcast_dev(pg_comm *c, void *buf, size_t bytes, int root)
{
    PG_REQUIRE(c && (buf || bytes == 0) && root >= 0 && root < c->nranks, "pg_comm_broadcast_dev: bad arguments");
    if (bytes == 0) return PG_OK;
    PG_HIP(hipSetDevice(c->ctx->device));
    PG_NCCL(g_rccl.Broadcast(buf, buf, bytes, ncclUint8, root, c->comm, c->ctx->stream));
    return PG_OK;
}

extern "C" int pg_comm_allgather_dev(pg_comm *c, const void *send, void *recv, size_t bytes_per_rank)
{
    PG_REQUIRE(c && ((send && recv) || bytes_per_rank == 0), "pg_comm_allgather_dev: bad arguments");
    if (bytes_per_rank == 0) return PG_OK;
    PG_HIP(hipSetDevice(c->ctx->device));
    PG_NCCL(g_rccl.AllGather(send, recv, bytes_per_rank, ncclUint8, c->comm, c->ctx->stream));
    return PG_OK;
}

extern "C" int pg_comm_allreduce_f64_dev(pg_comm *c, double *buf, size_t count, int op_max)
{
    PG_REQUIRE(c && (buf || count == 0), "pg_comm_allreduce_f64_dev: bad arguments");
    if (count == 0) return PG_OK;
    PG_HIP(hipSetDevice(c->ctx->device));
    PG_NCCL(g_rccl.AllReduce(buf, buf, count, ncclFloat64, op_max ? ncclMax : ncclSum, c->comm, c->ctx->stream));
    return PG_OK;
}

extern "C" int pg_comm_barrier(pg_comm *c)
{
    PG_REQUIRE(c, "pg_comm_barrier: NULL communicator");
    PG_HIP(hipSetDevice(c->ctx->device));
    PG_NCCL(g_rccl.AllReduce(c->scratch, c->scratch, 1, ncclFloat64, ncclSum, c->comm, c->ctx->stream));
    PG_HIP(hipStreamSynchronize(c->ctx->stream));
    return PG_OK;
}

extern "C" int pg_comm_group_start(void)
{
    int rc = rccl_load();
    if (rc) return rc;
    PG_NCCL(g_rccl.GroupStart());
    return PG_OK;
}
extern "C" int pg_comm_group_end(void)
{
    int rc = rccl_load();
    if (rc) return rc;
    PG_NCCL(g_rccl.GroupEnd());
    return PG_OK;
}
