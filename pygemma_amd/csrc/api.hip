// api.hip — context management, memory helpers and host-pointer conveniences of the C ABI
// (include/pygemma_hip.h).  No exceptions cross the ABI; errors go to a thread-local string.
#include "common.hpp"

#include <algorithm>
#include <string>
#include <thread>
#include <vector>

#include <cmath>
#include <functional>
#include <new>

namespace pg {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int ensure(pg_ctx *ctx, void **ptr, size_t *have, size_t need)
{
    if (*have >= need && *ptr) return PG_OK;
    if (*ptr) {
        PG_HIP(hipStreamSynchronize(ctx->stream));
        PG_HIP(hipFree(*ptr));
        *ptr = nullptr; *have = 0;
    }
    hipError_t e = hipMalloc(ptr, need);
    if (e != hipSuccess) {
        set_error("hipMalloc(%zu bytes) failed: %s", need, hipGetErrorString(e));
        *ptr = nullptr;
        return PG_ENOMEM;
    }
    *have = need;
    return PG_OK;
}

// numpy's float32 add.reduce over a contiguous array of length n: chunks of 8192 (the ufunc buffer
// size) summed sequentially; inside a chunk pairwise_sum: n<8 plain loop, n<=128 eight strided
// accumulators, else split at n/2 rounded down to a multiple of 8.  The plan lists the leaves and
// the internal nodes (sorted by height so that one height can be evaluated in parallel).
int build_npsum_plan(pg_ctx *ctx, int64_t n)
{
    NpSumPlan &pl = ctx->plan;
    if (pl.n == n) return PG_OK;
    struct Node { int l, r, h; };
    std::vector<int> leaf;               // start,len pairs
    std::vector<Node> nodes;             // internal nodes, value index = -(k+1) until remapped
    std::vector<int> chunk_roots;        // encoded: >=0 leaf index, <0 internal -(k+1)
    std::function<int(int64_t, int64_t, int *)> rec = [&](int64_t start, int64_t len, int *height) -> int {
        if (len <= 128) {
            leaf.push_back((int)start); leaf.push_back((int)len);
            *height = 0;
            return (int)(leaf.size() / 2 - 1);
        }
        int64_t n2 = len / 2;
        n2 -= n2 % 8;
        int hl, hr;
        int a = rec(start, n2, &hl), b = rec(start + n2, len - n2, &hr);
        *height = (hl > hr ? hl : hr) + 1;
        nodes.push_back({a, b, *height});
        return -(int)nodes.size();
    };
    for (int64_t off = 0; off < n; off += 8192) {
        int64_t k = n - off < 8192 ? n - off : 8192;
        int h;
        chunk_roots.push_back(rec(off, k, &h));
    }
    const int n_leaf = (int)(leaf.size() / 2), n_node = (int)nodes.size();
    // order internal nodes by height (stable), remap references
    std::vector<int> order(n_node), pos(n_node);
    int maxh = 0;
    for (auto &nd : nodes) maxh = nd.h > maxh ? nd.h : maxh;
    std::vector<int> level(maxh + 1, 0);
    {
        int w = 0;
        for (int h = 1; h <= maxh; h++) {
            level[h - 1] = w;
            for (int k = 0; k < n_node; k++) if (nodes[k].h == h) { order[w] = k; pos[k] = w; w++; }
        }
        level[maxh] = w;
    }
    auto val_index = [&](int enc) { return enc >= 0 ? enc : n_leaf + pos[-enc - 1]; };
    std::vector<int> node_lr(2 * (size_t)(n_node > 0 ? n_node : 1));
    for (int w = 0; w < n_node; w++) {
        node_lr[2 * w] = val_index(nodes[order[w]].l);
        node_lr[2 * w + 1] = val_index(nodes[order[w]].r);
    }
    std::vector<int> chunks;
    for (int enc : chunk_roots) chunks.push_back(val_index(enc));

    PG_HIP(hipStreamSynchronize(ctx->stream));
    if (pl.d_leaf) { (void)hipFree(pl.d_leaf); (void)hipFree(pl.d_node); (void)hipFree(pl.d_level); (void)hipFree(pl.d_chunk); }
    pl = NpSumPlan{};
    PG_HIP(hipMalloc(&pl.d_leaf, leaf.size() * 4));
    PG_HIP(hipMalloc(&pl.d_node, node_lr.size() * 4));
    PG_HIP(hipMalloc(&pl.d_level, level.size() * 4));
    PG_HIP(hipMalloc(&pl.d_chunk, chunks.size() * 4));
    PG_HIP(hipMemcpy(pl.d_leaf, leaf.data(), leaf.size() * 4, hipMemcpyHostToDevice));
    PG_HIP(hipMemcpy(pl.d_node, node_lr.data(), node_lr.size() * 4, hipMemcpyHostToDevice));
    PG_HIP(hipMemcpy(pl.d_level, level.data(), level.size() * 4, hipMemcpyHostToDevice));
    PG_HIP(hipMemcpy(pl.d_chunk, chunks.data(), chunks.size() * 4, hipMemcpyHostToDevice));
    pl.n = n; pl.n_leaf = n_leaf; pl.n_node = n_node; pl.n_level = maxh; pl.n_chunk = (int)chunks.size();
    return PG_OK;
}

}  // namespace pg

using namespace pg;

extern "C" const char *pg_last_error(void) { return g_err; }
extern "C" const char *pg_version(void) { return "pygemma_hip 0.1.0 (gfx950)"; }

extern "C" int pg_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

static int ctx_create_impl(int device, hipStream_t stream, bool own, pg_ctx **out)
{
    PG_REQUIRE(out != nullptr, "pg_ctx_create: out is NULL");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        set_error("pg_ctx_create: no HIP device visible (the MI355X path has no CPU fallback)");
        return PG_ENODEV;
    }
    PG_REQUIRE(device >= 0 && device < ndev, "pg_ctx_create: device %d out of range (0..%d)", device, ndev - 1);
    PG_HIP(hipSetDevice(device));
    pg_ctx *c = new (std::nothrow) pg_ctx();
    if (!c) { set_error("pg_ctx_create: out of host memory"); return PG_ENOMEM; }
    c->device = device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) c->num_cu = prop.multiProcessorCount;
    if (own) {
        hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (e != hipSuccess) { set_error("hipStreamCreate failed: %s", hipGetErrorString(e)); delete c; return PG_EHIP; }
    } else c->stream = stream;
    c->own_stream = own;
    *out = c;
    return PG_OK;
}
extern "C" int pg_ctx_create(int device, pg_ctx **out) { return ctx_create_impl(device, nullptr, true, out); }
extern "C" int pg_ctx_create_on_stream(int device, void *hip_stream, pg_ctx **out)
{
    return ctx_create_impl(device, (hipStream_t)hip_stream, false, out);
}
extern "C" void pg_ctx_destroy(pg_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->fixed) (void)hipFree(ctx->fixed);
    if (ctx->tabs) (void)hipFree(ctx->tabs);
    if (ctx->stats) (void)hipFree(ctx->stats);
    if (ctx->scratch) (void)hipFree(ctx->scratch);
    if (ctx->plan.d_leaf) { (void)hipFree(ctx->plan.d_leaf); (void)hipFree(ctx->plan.d_node); (void)hipFree(ctx->plan.d_level); (void)hipFree(ctx->plan.d_chunk); }
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}
extern "C" int pg_ctx_sync(pg_ctx *ctx)
{
    PG_REQUIRE(ctx, "pg_ctx_sync: NULL ctx");
    PG_HIP(hipSetDevice(ctx->device));
    PG_HIP(hipStreamSynchronize(ctx->stream));
    return PG_OK;
}
extern "C" int pg_ctx_device(const pg_ctx *ctx) { return ctx ? ctx->device : -1; }

extern "C" int pg_malloc(pg_ctx *ctx, size_t bytes, void **dptr)
{
    PG_REQUIRE(ctx && dptr, "pg_malloc: NULL argument");
    PG_HIP(hipSetDevice(ctx->device));
    hipError_t e = hipMalloc(dptr, bytes ? bytes : 1);
    if (e != hipSuccess) { set_error("hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e)); *dptr = nullptr; return PG_ENOMEM; }
    return PG_OK;
}
extern "C" int pg_free(pg_ctx *ctx, void *dptr)
{
    PG_REQUIRE(ctx, "pg_free: NULL ctx");
    if (!dptr) return PG_OK;
    PG_HIP(hipSetDevice(ctx->device));
    PG_HIP(hipStreamSynchronize(ctx->stream));
    PG_HIP(hipFree(dptr));
    return PG_OK;
}
extern "C" int pg_memcpy_h2d(pg_ctx *ctx, void *dst, const void *src, size_t bytes)
{
    PG_REQUIRE(ctx && (bytes == 0 || (dst && src)), "pg_memcpy_h2d: NULL argument");
    PG_HIP(hipSetDevice(ctx->device));
    PG_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    PG_HIP(hipStreamSynchronize(ctx->stream));
    return PG_OK;
}
extern "C" int pg_memcpy_d2h(pg_ctx *ctx, void *dst, const void *src, size_t bytes)
{
    PG_REQUIRE(ctx && (bytes == 0 || (dst && src)), "pg_memcpy_d2h: NULL argument");
    PG_HIP(hipSetDevice(ctx->device));
    PG_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    PG_HIP(hipStreamSynchronize(ctx->stream));
    return PG_OK;
}
extern "C" int pg_memset(pg_ctx *ctx, void *dst, int value, size_t bytes)
{
    PG_REQUIRE(ctx && (bytes == 0 || dst), "pg_memset: NULL argument");
    PG_HIP(hipSetDevice(ctx->device));
    PG_HIP(hipMemsetAsync(dst, value, bytes, ctx->stream));
    return PG_OK;
}

extern "C" int pg_memcpy2d_h2d(pg_ctx *ctx, void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t height)
{
    PG_REQUIRE(ctx && dst && src && dpitch >= width && spitch >= width, "pg_memcpy2d_h2d: bad arguments");
    PG_HIP(hipSetDevice(ctx->device));
    PG_HIP(hipMemcpy2DAsync(dst, dpitch, src, spitch, width, height, hipMemcpyHostToDevice, ctx->stream));
    PG_HIP(hipStreamSynchronize(ctx->stream));
    return PG_OK;
}
extern "C" int pg_event_create(pg_ctx *ctx, void **event)
{
    PG_REQUIRE(ctx && event, "pg_event_create: NULL argument");
    PG_HIP(hipSetDevice(ctx->device));
    hipEvent_t e;
    PG_HIP(hipEventCreate(&e));
    *event = (void *)e;
    return PG_OK;
}
extern "C" int pg_event_destroy(pg_ctx *ctx, void *event)
{
    PG_REQUIRE(ctx, "pg_event_destroy: NULL ctx");
    if (event) PG_HIP(hipEventDestroy((hipEvent_t)event));
    return PG_OK;
}
extern "C" int pg_event_record(pg_ctx *ctx, void *event)
{
    PG_REQUIRE(ctx && event, "pg_event_record: NULL argument");
    PG_HIP(hipEventRecord((hipEvent_t)event, ctx->stream));
    return PG_OK;
}
extern "C" int pg_event_elapsed_ms(pg_ctx *ctx, void *start, void *stop, float *ms)
{
    PG_REQUIRE(ctx && start && stop && ms, "pg_event_elapsed_ms: NULL argument");
    PG_HIP(hipEventSynchronize((hipEvent_t)stop));
    PG_HIP(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
    return PG_OK;
}

// host-pointer convenience around pg_assoc_dev: columns [0, p) of a host matrix X in the reference layout (n rows, row
// stride ldX floats); the outputs are host arrays of length p
static int assoc_host_block(pg_ctx *ctx, int64_t n, int c, int64_t p, const float *d, const float *Wr, const float *yr,
                            const float *X, int64_t ldX, int grid, float *beta, float *se, float *tau, float *lambda, double *F,
                            double *pval, unsigned long long *stats2)
{
    if (p == 0) return PG_OK;
    PG_HIP(hipSetDevice(ctx->device));
    const int64_t ldx = (n + 63) / 64 * 64;
    float *dd = nullptr, *dW = nullptr, *dy = nullptr, *dX = nullptr, *dXr = nullptr, *dout = nullptr;
    double *dF = nullptr;
    unsigned long long *dstats = nullptr;
    int rc = PG_OK;
    auto cleanup = [&]() {
        (void)hipStreamSynchronize(ctx->stream);
        for (void *q : {(void *)dd, (void *)dW, (void *)dy, (void *)dX, (void *)dXr, (void *)dout, (void *)dF, (void *)dstats})
            if (q) (void)hipFree(q);
    };
#define PG_TRY(call) do { hipError_t _e = (call); if (_e != hipSuccess) { set_error("%s failed: %s", #call, hipGetErrorString(_e)); cleanup(); return (_e == hipErrorOutOfMemory) ? PG_ENOMEM : PG_EHIP; } } while (0)
    PG_TRY(hipMalloc(&dd, n * 4));
    PG_TRY(hipMalloc(&dW, (size_t)n * (c > 0 ? c : 1) * 4));
    PG_TRY(hipMalloc(&dy, n * 4));
    PG_TRY(hipMalloc(&dX, (size_t)n * p * 4));
    PG_TRY(hipMalloc(&dXr, (size_t)p * ldx * 4));
    PG_TRY(hipMalloc(&dout, (size_t)p * 4 * 4));
    PG_TRY(hipMalloc(&dF, (size_t)p * 2 * 8));
    PG_TRY(hipMalloc(&dstats, 16));
    PG_TRY(hipMemsetAsync(dstats, 0, 16, ctx->stream));
    PG_TRY(hipMemcpyAsync(dd, d, n * 4, hipMemcpyHostToDevice, ctx->stream));
    PG_TRY(hipMemcpyAsync(dW, Wr, (size_t)n * c * 4, hipMemcpyHostToDevice, ctx->stream));
    PG_TRY(hipMemcpyAsync(dy, yr, n * 4, hipMemcpyHostToDevice, ctx->stream));
    PG_TRY(hipMemcpy2DAsync(dX, (size_t)p * 4, X, (size_t)ldX * 4, (size_t)p * 4, (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    rc = pg_transpose_dev(ctx, n, p, dX, p, dXr, ldx);
    if (!rc) rc = pg_assoc_dev(ctx, n, c, p, dd, dW, dy, dXr, ldx, grid, dout, dout + p, dout + 2 * p, dout + 3 * p, dF,
                               pval ? dF + p : nullptr, dstats);
    if (rc) { cleanup(); return rc; }
    PG_TRY(hipMemcpyAsync(beta, dout, p * 4, hipMemcpyDeviceToHost, ctx->stream));
    PG_TRY(hipMemcpyAsync(se, dout + p, p * 4, hipMemcpyDeviceToHost, ctx->stream));
    PG_TRY(hipMemcpyAsync(tau, dout + 2 * p, p * 4, hipMemcpyDeviceToHost, ctx->stream));
    PG_TRY(hipMemcpyAsync(lambda, dout + 3 * p, p * 4, hipMemcpyDeviceToHost, ctx->stream));
    PG_TRY(hipMemcpyAsync(F, dF, p * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (pval) PG_TRY(hipMemcpyAsync(pval, dF + p, p * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (stats2) PG_TRY(hipMemcpyAsync(stats2, dstats, 16, hipMemcpyDeviceToHost, ctx->stream));
    PG_TRY(hipStreamSynchronize(ctx->stream));
#undef PG_TRY
    cleanup();
    return PG_OK;
}

extern "C" int pg_assoc(pg_ctx *ctx, int64_t n, int c, int64_t p, const float *d, const float *Wr, const float *yr,
                        const float *X, int grid, float *beta, float *se, float *tau, float *lambda, double *F,
                        double *pval, unsigned long long *stats2)
{
    PG_REQUIRE(ctx && d && Wr && yr && X && beta && se && tau && lambda && F, "pg_assoc: NULL argument");
    PG_REQUIRE(n >= 2 && p >= 0 && c >= 0, "pg_assoc: bad shape");
    return assoc_host_block(ctx, n, c, p, d, Wr, yr, X, p, grid, beta, se, tau, lambda, F, pval, stats2);
}

// The same over several GPUs of this node: contiguous SNP blocks of ceil(p/ngpu) columns like the reference's
// SampleIter (lmm/lmm.py:427-434), one host thread + one context per GPU, results written in SNP order.  SNPs are
// independent, so there is no device-to-device exchange: the gather is each GPU's copy-out into its slice of the outputs.
extern "C" int pg_assoc_multi(int ngpu, int64_t n, int c, int64_t p, const float *d, const float *Wr, const float *yr,
                              const float *X, int grid, float *beta, float *se, float *tau, float *lambda, double *F, double *pval)
{
    PG_REQUIRE(d && Wr && yr && X && beta && se && tau && lambda && F, "pg_assoc_multi: NULL argument");
    PG_REQUIRE(n >= 2 && p >= 0 && c >= 0 && ngpu >= 1, "pg_assoc_multi: bad arguments");
    int have = 0;
    if (hipGetDeviceCount(&have) != hipSuccess || have < 1) { set_error("pg_assoc_multi: no GPU visible"); return PG_ENODEV; }
    const int G = (int)std::min<int64_t>(std::min(ngpu, have), std::max<int64_t>(p, 1));
    const int64_t cols = (p + G - 1) / G;
    std::vector<int> rcs(G, PG_OK);
    std::vector<std::string> msgs(G);
    std::vector<std::thread> th;
    for (int g = 0; g < G; g++) {
        th.emplace_back([&, g]() {
            const int64_t a = g * cols, b = std::min<int64_t>(p, a + cols);
            if (a >= b) return;
            pg_ctx *ctx = nullptr;
            int rc = pg_ctx_create(g, &ctx);
            if (!rc) {
                rc = assoc_host_block(ctx, n, c, b - a, d, Wr, yr, X + a, p, grid, beta + a, se + a, tau + a, lambda + a, F + a,
                                      pval ? pval + a : nullptr, nullptr);
                if (rc) msgs[g] = pg_last_error();      // thread-local text: carry it to the caller's thread
                pg_ctx_destroy(ctx);
            } else msgs[g] = pg_last_error();
            rcs[g] = rc;
        });
    }
    for (auto &t : th) t.join();
    for (int g = 0; g < G; g++)
        if (rcs[g]) { set_error("pg_assoc_multi: GPU %d: %s", g, msgs[g].c_str()); return rcs[g]; }
    return PG_OK;
}
