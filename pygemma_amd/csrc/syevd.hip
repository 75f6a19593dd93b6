// syevd.hip — H1: eigendecomposition of K in fp64 (placeholder until it lands in this round)
#include "common.hpp"
extern "C" int pg_syevd_dev(pg_ctx *, int64_t, const float *, float *, float *, double *, double *)
{
    pg::set_error("pg_syevd_dev: not built yet");
    return PG_ENOTSUP;
}
