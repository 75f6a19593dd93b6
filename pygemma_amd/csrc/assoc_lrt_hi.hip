// assoc_lrt_hi.hip — LRT instantiations of the association kernel, c = 16..PG_MAX_COVARIATES (fourth translation unit of assoc.hip).
#define PG_ASSOC_PART 3
#include "assoc.hip"
