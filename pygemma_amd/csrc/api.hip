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
    {
        hipError_t e = hipMalloc(&pl.d_leaf, leaf.size() * 4);
        if (e == hipSuccess) e = hipMalloc(&pl.d_node, node_lr.size() * 4);
        if (e == hipSuccess) e = hipMalloc(&pl.d_level, level.size() * 4);
        if (e == hipSuccess) e = hipMalloc(&pl.d_chunk, chunks.size() * 4);
        if (e == hipSuccess) e = hipMemcpy(pl.d_leaf, leaf.data(), leaf.size() * 4, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(pl.d_node, node_lr.data(), node_lr.size() * 4, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(pl.d_level, level.data(), level.size() * 4, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(pl.d_chunk, chunks.data(), chunks.size() * 4, hipMemcpyHostToDevice);
        if (e != hipSuccess) {   // no partial plan is left behind
            set_error("build_npsum_plan: %s", hipGetErrorString(e));
            for (void *q : {(void *)pl.d_leaf, (void *)pl.d_node, (void *)pl.d_level, (void *)pl.d_chunk}) if (q) (void)hipFree(q);
            pl = NpSumPlan{};
            return e == hipErrorOutOfMemory ? PG_ENOMEM : PG_EHIP;
        }
    }
    pl.n = n; pl.n_leaf = n_leaf; pl.n_node = n_node; pl.n_level = maxh; pl.n_chunk = (int)chunks.size();
    return PG_OK;
}

}  // namespace pg

using namespace pg;

extern "C" const char *pg_last_error(void) { return g_err; }
extern "C" const char *pg_version(void) { return "pygemma_hip 0.2.0 (gfx950)"; }

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
    if (ctx->arena) (void)hipFree(ctx->arena);
    if (ctx->hpin) (void)hipHostFree(ctx->hpin);
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

extern "C" int pg_mem_info(pg_ctx *ctx, size_t *free_bytes, size_t *total_bytes)
{
    PG_REQUIRE(ctx && free_bytes && total_bytes, "pg_mem_info: NULL argument");
    PG_HIP(hipSetDevice(ctx->device));
    PG_HIP(hipMemGetInfo(free_bytes, total_bytes));
    return PG_OK;
}

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

// ---- S1: pinned host memory and asynchronous copies (streaming SNP batches / eigenvectors from the host, BASELINE configs 4-5;
// the reference's eigen=False caller reads raw float32 .bin files, experiments/large_gwas/run_pygemma.py:33-65) -------------
extern "C" int pg_host_alloc(pg_ctx *ctx, size_t bytes, void **hptr)
{
    PG_REQUIRE(ctx && hptr, "pg_host_alloc: NULL argument");
    PG_HIP(hipSetDevice(ctx->device));
    hipError_t e = hipHostMalloc(hptr, bytes ? bytes : 1, hipHostMallocPortable);   // portable: usable from every GPU's context
    if (e != hipSuccess) { set_error("hipHostMalloc(%zu) failed: %s", bytes, hipGetErrorString(e)); *hptr = nullptr; return PG_ENOMEM; }
    return PG_OK;
}
extern "C" int pg_host_free(pg_ctx *ctx, void *hptr)
{
    PG_REQUIRE(ctx, "pg_host_free: NULL ctx");
    if (!hptr) return PG_OK;
    PG_HIP(hipSetDevice(ctx->device));
    PG_HIP(hipHostFree(hptr));
    return PG_OK;
}
extern "C" int pg_host_register(pg_ctx *ctx, void *hptr, size_t bytes)
{
    PG_REQUIRE(ctx && hptr && bytes > 0, "pg_host_register: bad arguments");
    PG_HIP(hipSetDevice(ctx->device));
    hipError_t e = hipHostRegister(hptr, bytes, hipHostRegisterPortable);
    if (e != hipSuccess) { set_error("hipHostRegister(%zu bytes) failed: %s", bytes, hipGetErrorString(e)); (void)hipGetLastError(); return PG_EHIP; }
    return PG_OK;
}
extern "C" int pg_host_unregister(pg_ctx *ctx, void *hptr)
{
    PG_REQUIRE(ctx && hptr, "pg_host_unregister: bad arguments");
    PG_HIP(hipSetDevice(ctx->device));
    PG_HIP(hipHostUnregister(hptr));
    return PG_OK;
}
extern "C" int pg_memcpy_h2d_async(pg_ctx *ctx, void *dst, const void *src, size_t bytes)
{
    PG_REQUIRE(ctx && (bytes == 0 || (dst && src)), "pg_memcpy_h2d_async: NULL argument");
    PG_HIP(hipSetDevice(ctx->device));
    PG_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    return PG_OK;
}
extern "C" int pg_memcpy_d2h_async(pg_ctx *ctx, void *dst, const void *src, size_t bytes)
{
    PG_REQUIRE(ctx && (bytes == 0 || (dst && src)), "pg_memcpy_d2h_async: NULL argument");
    PG_HIP(hipSetDevice(ctx->device));
    PG_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    return PG_OK;
}
extern "C" int pg_memcpy2d_h2d_async(pg_ctx *ctx, void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t height)
{
    PG_REQUIRE(ctx && dst && src && dpitch >= width && spitch >= width, "pg_memcpy2d_h2d_async: bad arguments");
    PG_HIP(hipSetDevice(ctx->device));
    PG_HIP(hipMemcpy2DAsync(dst, dpitch, src, spitch, width, height, hipMemcpyHostToDevice, ctx->stream));
    return PG_OK;
}
extern "C" int pg_memcpy_d2d_async(pg_ctx *ctx, void *dst, const void *src, size_t bytes)
{
    PG_REQUIRE(ctx && (bytes == 0 || (dst && src)), "pg_memcpy_d2d_async: NULL argument");
    PG_HIP(hipSetDevice(ctx->device));
    PG_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return PG_OK;
}
// host-side gather of a column window (height rows of width bytes, row stride spitch) into a (pinned) staging buffer with
// nthreads copy threads: the pageable -> pinned leg of a streamed batch, off the Python thread (ctypes drops the GIL)
extern "C" int pg_stage_rows(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t height, int nthreads)
{
    PG_REQUIRE(dst && src && dpitch >= width && spitch >= width, "pg_stage_rows: bad arguments");
    const int T = (int)std::max<size_t>(1, std::min<size_t>((size_t)std::max(nthreads, 1), height));
    auto work = [&](int t) {
        const size_t r0 = height * t / T, r1 = height * (t + 1) / T;
        for (size_t r = r0; r < r1; r++) memcpy((char *)dst + r * dpitch, (const char *)src + r * spitch, width);
    };
    if (T == 1) { work(0); return PG_OK; }
    std::vector<std::thread> th;
    for (int t = 0; t < T; t++) th.emplace_back(work, t);
    for (auto &x : th) x.join();
    return PG_OK;
}
extern "C" int pg_event_sync(pg_ctx *ctx, void *event)
{
    PG_REQUIRE(ctx && event, "pg_event_sync: NULL argument");
    PG_HIP(hipSetDevice(ctx->device));
    PG_HIP(hipEventSynchronize((hipEvent_t)event));
    return PG_OK;
}
extern "C" int pg_stream_wait_event(pg_ctx *ctx, void *event)
{
    PG_REQUIRE(ctx && event, "pg_stream_wait_event: NULL argument");
    PG_HIP(hipSetDevice(ctx->device));
    PG_HIP(hipStreamWaitEvent(ctx->stream, (hipEvent_t)event, 0));
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
    PG_HIP(hipSetDevice(ctx->device));
    if (event) PG_HIP(hipEventDestroy((hipEvent_t)event));
    return PG_OK;
}
extern "C" int pg_event_record(pg_ctx *ctx, void *event)
{
    PG_REQUIRE(ctx && event, "pg_event_record: NULL argument");
    PG_HIP(hipSetDevice(ctx->device));
    PG_HIP(hipEventRecord((hipEvent_t)event, ctx->stream));
    return PG_OK;
}
extern "C" int pg_event_elapsed_ms(pg_ctx *ctx, void *start, void *stop, float *ms)
{
    PG_REQUIRE(ctx && start && stop && ms, "pg_event_elapsed_ms: NULL argument");
    PG_HIP(hipSetDevice(ctx->device));
    PG_HIP(hipEventSynchronize((hipEvent_t)stop));
    PG_HIP(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
    return PG_OK;
}

// host-pointer convenience around pg_assoc_dev: columns [0, p) of a host matrix X in the reference layout (n rows, row
// stride ldX floats) -> one device result block of `cols` >= p rows laid out [F | pval | beta | se | tau | lambda]
// (cols f64, cols f64, 4 x cols f32 = 32 bytes per SNP: the row block the ranks exchange).  Enqueues; does not synchronise.
struct HostBlock {
    float *dd = nullptr, *dW = nullptr, *dy = nullptr, *dX = nullptr, *dXr = nullptr;
    unsigned long long *dstats = nullptr;
    void release(pg_ctx *ctx)
    {
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);
        for (void *q : {(void *)dd, (void *)dW, (void *)dy, (void *)dX, (void *)dXr, (void *)dstats})
            if (q) (void)hipFree(q);
        *this = HostBlock{};
    }
};
static int assoc_host_block(pg_ctx *ctx, HostBlock &hb, int64_t n, int c, int64_t p, const float *d, const float *Wr, const float *yr,
                            const float *X, int64_t ldX, int grid, char *res, int64_t cols, bool want_p)
{
    if (p == 0) return PG_OK;
    PG_HIP(hipSetDevice(ctx->device));
    const int64_t ldx = (n + 63) / 64 * 64;
#define PG_TRY(call) do { hipError_t _e = (call); if (_e != hipSuccess) { set_error("%s failed: %s", #call, hipGetErrorString(_e)); return (_e == hipErrorOutOfMemory) ? PG_ENOMEM : PG_EHIP; } } while (0)
    PG_TRY(hipMalloc(&hb.dd, n * 4));
    PG_TRY(hipMalloc(&hb.dW, (size_t)n * (c > 0 ? c : 1) * 4));
    PG_TRY(hipMalloc(&hb.dy, n * 4));
    PG_TRY(hipMalloc(&hb.dX, (size_t)n * p * 4));
    PG_TRY(hipMalloc(&hb.dXr, (size_t)p * ldx * 4));
    PG_TRY(hipMalloc(&hb.dstats, 16));
    PG_TRY(hipMemsetAsync(hb.dstats, 0, 16, ctx->stream));
    PG_TRY(hipMemcpyAsync(hb.dd, d, n * 4, hipMemcpyHostToDevice, ctx->stream));
    PG_TRY(hipMemcpyAsync(hb.dW, Wr, (size_t)n * c * 4, hipMemcpyHostToDevice, ctx->stream));
    PG_TRY(hipMemcpyAsync(hb.dy, yr, n * 4, hipMemcpyHostToDevice, ctx->stream));
    PG_TRY(hipMemcpy2DAsync(hb.dX, (size_t)p * 4, X, (size_t)ldX * 4, (size_t)p * 4, (size_t)n, hipMemcpyHostToDevice, ctx->stream));
#undef PG_TRY
    int rc = pg_transpose_dev(ctx, n, p, hb.dX, p, hb.dXr, ldx);
    double *F = (double *)res, *pv = F + cols;
    float *f4 = (float *)(pv + cols);
    if (!rc) rc = pg_assoc_dev(ctx, n, c, p, hb.dd, hb.dW, hb.dy, hb.dXr, ldx, grid, f4, f4 + cols, f4 + 2 * cols, f4 + 3 * cols, F,
                               want_p ? pv : nullptr, hb.dstats);
    return rc;
}
// rows [0, cnt) of one result block (device, `cols` rows) -> the caller's host arrays
static int copy_block_out(pg_ctx *ctx, const char *res, int64_t cols, int64_t cnt, float *beta, float *se, float *tau, float *lambda,
                          double *F, double *pval)
{
    if (cnt <= 0) return PG_OK;
    const double *dF = (const double *)res, *dp = dF + cols;
    const float *f4 = (const float *)(dp + cols);
    PG_HIP(hipMemcpyAsync(F, dF, cnt * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (pval) PG_HIP(hipMemcpyAsync(pval, dp, cnt * 8, hipMemcpyDeviceToHost, ctx->stream));
    PG_HIP(hipMemcpyAsync(beta, f4, cnt * 4, hipMemcpyDeviceToHost, ctx->stream));
    PG_HIP(hipMemcpyAsync(se, f4 + cols, cnt * 4, hipMemcpyDeviceToHost, ctx->stream));
    PG_HIP(hipMemcpyAsync(tau, f4 + 2 * cols, cnt * 4, hipMemcpyDeviceToHost, ctx->stream));
    PG_HIP(hipMemcpyAsync(lambda, f4 + 3 * cols, cnt * 4, hipMemcpyDeviceToHost, ctx->stream));
    return PG_OK;
}

extern "C" int pg_assoc(pg_ctx *ctx, int64_t n, int c, int64_t p, const float *d, const float *Wr, const float *yr,
                        const float *X, int grid, float *beta, float *se, float *tau, float *lambda, double *F,
                        double *pval, unsigned long long *stats2)
{
    PG_REQUIRE(ctx && d && Wr && yr && X && beta && se && tau && lambda && F, "pg_assoc: NULL argument");
    PG_REQUIRE(n >= 2 && p >= 0 && c >= 0, "pg_assoc: bad shape");
    if (p == 0) return PG_OK;
    PG_HIP(hipSetDevice(ctx->device));
    HostBlock hb;
    char *res = nullptr;
    hipError_t e = hipMalloc(&res, (size_t)p * 32);
    if (e != hipSuccess) { set_error("pg_assoc: hipMalloc(%zu) failed: %s", (size_t)p * 32, hipGetErrorString(e)); return PG_ENOMEM; }
    int rc = assoc_host_block(ctx, hb, n, c, p, d, Wr, yr, X, p, grid, res, p, pval != nullptr);
    if (!rc) rc = copy_block_out(ctx, res, p, p, beta, se, tau, lambda, F, pval);
    if (!rc && stats2 && hipMemcpyAsync(stats2, hb.dstats, 16, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) { set_error("pg_assoc: stats copy failed"); rc = PG_EHIP; }
    if (!rc && hipStreamSynchronize(ctx->stream) != hipSuccess) { set_error("pg_assoc: stream failed: %s", hipGetErrorString(hipGetLastError())); rc = PG_EHIP; }
    hb.release(ctx);
    (void)hipFree(res);
    return rc;
}

// The same over several GPUs of this node: contiguous SNP blocks of ceil(p/ngpu) columns like the reference's
// SampleIter (lmm/lmm.py:427-434), one host thread + one context per GPU.  Every GPU writes its block of 32-byte result rows
// (padded to `cols` rows); one RCCL all-gather (pg_comm_*, xGMI) assembles the blocks in rank order = SNP order — the
// device-side equivalent of the reference's ordered concatenation of the per-block lists (lmm/lmm.py:393,401) — and GPU 0's
// copy goes to the caller's arrays.
extern "C" int pg_assoc_multi(int ngpu, int64_t n, int c, int64_t p, const float *d, const float *Wr, const float *yr,
                              const float *X, int grid, float *beta, float *se, float *tau, float *lambda, double *F, double *pval)
{
    PG_REQUIRE(d && Wr && yr && X && beta && se && tau && lambda && F, "pg_assoc_multi: NULL argument");
    PG_REQUIRE(n >= 2 && p >= 0 && c >= 0 && ngpu >= 1, "pg_assoc_multi: bad arguments");
    int have = 0;
    if (hipGetDeviceCount(&have) != hipSuccess || have < 1) { set_error("pg_assoc_multi: no GPU visible"); return PG_ENODEV; }
    if (p == 0) return PG_OK;
    const int G = (int)std::min<int64_t>(std::min(ngpu, have), p);
    const int64_t cols = (p + G - 1) / G;
    std::vector<pg_ctx *> ctxs(G, nullptr);
    std::vector<pg_comm *> comms(G, nullptr);
    std::vector<char *> res(G, nullptr), all(G, nullptr);
    std::vector<HostBlock> hbs(G);
    std::vector<int> rcs(G, PG_OK);
    std::vector<std::string> msgs(G);
    int rc = PG_OK;
    auto teardown = [&]() {
        for (int g = 0; g < G; g++) {
            if (!ctxs[g]) continue;
            hbs[g].release(ctxs[g]);
            if (comms[g]) pg_comm_destroy(comms[g]);
            (void)hipSetDevice(ctxs[g]->device);
            if (res[g]) (void)hipFree(res[g]);
            if (all[g]) (void)hipFree(all[g]);
            pg_ctx_destroy(ctxs[g]);
        }
    };
    for (int g = 0; g < G && !rc; g++) {
        rc = pg_ctx_create(g, &ctxs[g]);
        if (!rc && hipMalloc(&res[g], (size_t)cols * 32) != hipSuccess) { set_error("pg_assoc_multi: GPU %d: out of memory", g); rc = PG_ENOMEM; }
        if (!rc && hipMalloc(&all[g], (size_t)G * cols * 32) != hipSuccess) { set_error("pg_assoc_multi: GPU %d: out of memory", g); rc = PG_ENOMEM; }
        if (!rc && hipMemsetAsync(res[g], 0, (size_t)cols * 32, ctxs[g]->stream) != hipSuccess) { set_error("pg_assoc_multi: memset failed"); rc = PG_EHIP; }
    }
    if (!rc) rc = pg_comm_init_all(G, ctxs.data(), comms.data());
    if (rc) { teardown(); return rc; }
    std::vector<std::thread> th;
    for (int g = 0; g < G; g++) {
        th.emplace_back([&, g]() {
            const int64_t a = g * cols, b = std::min<int64_t>(p, a + cols);
            int r = PG_OK;
            if (a < b) r = assoc_host_block(ctxs[g], hbs[g], n, c, b - a, d, Wr, yr, X + a, p, grid, res[g], cols, pval != nullptr);
            // every rank enters the collective, also after a local failure (its block then holds zeros), so no rank is left waiting
            const int r2 = pg_comm_allgather_dev(comms[g], res[g], all[g], (size_t)cols * 32);
            if (!r) r = r2;
            if (!r && hipStreamSynchronize(ctxs[g]->stream) != hipSuccess) { set_error("stream failed: %s", hipGetErrorString(hipGetLastError())); r = PG_EHIP; }
            if (r) msgs[g] = pg_last_error();      // thread-local text: carry it to the caller's thread
            rcs[g] = r;
        });
    }
    for (auto &t : th) t.join();
    for (int g = 0; g < G && !rc; g++)
        if (rcs[g]) { set_error("pg_assoc_multi: GPU %d: %s", g, msgs[g].c_str()); rc = rcs[g]; }
    if (!rc) {
        (void)hipSetDevice(ctxs[0]->device);
        for (int g = 0; g < G && !rc; g++) {
            const int64_t a = g * cols, b = std::min<int64_t>(p, a + cols);
            rc = copy_block_out(ctxs[0], all[0] + (size_t)g * cols * 32, cols, b - a, beta + a, se + a, tau + a, lambda + a, F + a,
                                pval ? pval + a : nullptr);
        }
        if (!rc && hipStreamSynchronize(ctxs[0]->stream) != hipSuccess) { set_error("pg_assoc_multi: copy-out failed"); rc = PG_EHIP; }
    }
    teardown();
    return rc;
}
