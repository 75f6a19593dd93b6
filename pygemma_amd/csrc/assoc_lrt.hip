// assoc_lrt.hip — the association kernel with the ML lambda search of the LRT appended (N2; lmm/lmm.py:22-84, 277-300),
// instantiations c = 0..15 (third translation unit of assoc.hip).
#define PG_ASSOC_PART 2
#include "assoc.hip"
