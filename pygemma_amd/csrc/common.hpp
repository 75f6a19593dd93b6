// common.hpp — context, error plumbing and small device helpers shared by the HIP translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/pygemma_hip.h"

namespace pg {

void set_error(const char *fmt, ...);

#define PG_HIP(call)                                                                             \
    do {                                                                                         \
        hipError_t _e = (call);                                                                  \
        if (_e != hipSuccess) {                                                                  \
            pg::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), __FILE__, __LINE__); \
            return PG_EHIP;                                                                      \
        }                                                                                        \
    } while (0)

#define PG_REQUIRE(cond, ...)            \
    do {                                 \
        if (!(cond)) {                   \
            pg::set_error(__VA_ARGS__);  \
            return PG_EINVAL;            \
        }                                \
    } while (0)

// numpy float32 sum schedule for a given n (see assoc.hip: device_logdet_H)
struct NpSumPlan {
    int64_t n = -1;
    int n_leaf = 0, n_node = 0, n_level = 0, n_chunk = 0;
    int *d_leaf = nullptr;   // [n_leaf][2] start, len
    int *d_node = nullptr;   // [n_node][2] left, right  (indices into the value array; leaves first)
    int *d_level = nullptr;  // [n_level+1] offsets into d_node, by height
    int *d_chunk = nullptr;  // [n_chunk] root value index of each 8192-chunk
};

}  // namespace pg

struct pg_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = true;
    int num_cu = 256;
    // assoc scratch (grown on demand)
    void *fixed = nullptr; size_t fixed_bytes = 0;   // packed rows d,w..,y
    void *tabs = nullptr;  size_t tabs_bytes = 0;    // htab + fixed grams + ldH + t1
    unsigned long long *stats = nullptr;
    unsigned *eval_trace = nullptr;                  // caller-owned (pg_assoc_set_eval_trace)
    pg::NpSumPlan plan;
    // generic scratch
    void *scratch = nullptr; size_t scratch_bytes = 0;
    // the eigensolver's work arena, kept between solves up to 16 GiB (n <= ~14 000): a hipMalloc / hipFree pair
    // per solve was followed, in one solve out of three, by a kernel launch that blocked the host for 25 - 60 ms (r4, HIP API trace)
    void *arena = nullptr; size_t arena_bytes = 0;
    // page-locked host staging of the divide & conquer's per-level tables (grown on demand, freed with the context): from pageable
    // std::vectors every one of its ~8 small copies per level blocked the host for 50 - 100 us (5.5 ms of idle GPU per solve at n = 10 000)
    void *hpin = nullptr; size_t hpin_bytes = 0;
};

namespace pg {
int ensure(pg_ctx *ctx, void **ptr, size_t *have, size_t need);
int build_npsum_plan(pg_ctx *ctx, int64_t n);
}  // namespace pg
