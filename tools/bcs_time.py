"""Where a step of the stationary bulge chasing spends its cycles: s_memtime at the phase boundaries of workgroup 40 (thread 0), sweeps 2000-5999.
Needs the stamped build: tools/build_variant.sh bcs sb2 -DPG_BCS_TIME, PYGEMMA_HIP_LIB=pygemma_amd/lib_dev/bcs/libpygemma_hip.so.  usage: bcs_time.py [n]"""
import sys, ctypes as C
import numpy as np
sys.path.insert(0, '/root/repo')
from pygemma_amd import _lib, ops
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
L = _lib.load()
L.pgx_bcs_time.argtypes = [C.c_void_p, C.c_int]; L.pgx_bcs_time.restype = C.c_int
rng = np.random.default_rng(0)
A = rng.standard_normal((n, n)); A = (A + A.T) / 2
with _lib.Context(0) as ctx:
    ops.syevd(A, ctx=ctx)                 # warm
    assert L.pgx_bcs_time(None, 1) == 0
    ops.syevd(A, ctx=ctx)
    buf = (C.c_longlong * 12)()
    assert L.pgx_bcs_time(buf, 0) == 0
t = np.array(buf[:7], dtype=np.float64) / 4000.0
names = ["loop + last barrier", "wait for messages + barrier", "right-apply (dot, barrier, update)", "reflector + send + barrier", "w = E'v, p = Dv + barrier",
         "reductions + barrier", "rank updates + top row out"]
print("cycles of s_memtime per step (workgroup 40, mean over 4000 sweeps): total %.0f" % t.sum())
for nm, v in zip(names, t): print("  %-38s %7.0f" % (nm, v))
