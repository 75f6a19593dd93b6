"""Where one workgroup of the stage-2 back-transformation (bt2_apply4_kernel) spends its cycles: s_memtime at the phase boundaries of slab 10 of
the LAST launch (one block per trip, the workgroup alone on its CU), n = 10 000.  Needs: tools/build_variant.sh bt2t sb2 -DPG_BT2_TIME, PYGEMMA_HIP_LIB=.../lib_dev/bt2t/..."""
import sys, ctypes as C
import numpy as np
sys.path.insert(0, '/root/repo')
from pygemma_amd import _lib, ops
n = 10000
L = _lib.load(); L.pgx_bt2_time.argtypes = [C.c_void_p]; L.pgx_bt2_time.restype = C.c_int
rng = np.random.default_rng(0); A = rng.standard_normal((n, n)); A = (A + A.T) / 2
with _lib.Context(0) as ctx:
    ops.syevd(A, ctx=ctx); ops.syevd(A, ctx=ctx)
    buf = (C.c_longlong * 16)(); assert L.pgx_bt2_time(buf) == 0
t = np.array(buf[:], dtype=np.int64)
print("cycles: slab in %d | block 1 %d (its first product %d) | block 2 %d | block 3 %d | block 4 %d | slab out %d | total %d" % (
    t[1] - t[0], t[2] - t[1], t[8] - t[1], t[3] - t[2], t[4] - t[3], t[6] - t[4], t[7] - t[6], t[7] - t[0]))
print("MFMA cycles of one block per wavefront: 184 x 64 = 11 776 (80 in the first product, 104 in the second)")
