// rotate.hip — H2: X <- U' X on fp32 MFMA (placeholder until the GEMM lands in this round)
#include "common.hpp"
extern "C" int pg_rotate_dev(pg_ctx *, int64_t, int64_t, const float *, const float *, float *, int64_t)
{
    pg::set_error("pg_rotate_dev: not built yet");
    return PG_ENOTSUP;
}
