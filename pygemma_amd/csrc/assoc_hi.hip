// assoc_hi.hip — the association kernel's instantiations for c = 16..PG_MAX_COVARIATES (second translation unit of assoc.hip,
// so that the parts compile in parallel; the reference's covariate benchmark goes up to c = 26,
// experiments/animal_gwas/benchmark_pygemma.py:238-255).
#define PG_ASSOC_PART 1
#include "assoc.hip"
