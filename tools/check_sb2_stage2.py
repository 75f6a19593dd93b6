"""Stage 2 alone (bulge chasing, optionally + Q2 applied to I) on a random band matrix. usage: check_sb2_stage2.py n [withZ]"""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygemma_amd import _lib
import scipy.linalg as sl
L = _lib.load(); ctx = _lib.Context(0)
n = int(sys.argv[1]); withZ = len(sys.argv) > 2
rng = np.random.default_rng(5)
A = rng.standard_normal((n, n)); A = A + A.T
i, j = np.indices((n, n)); Bm = np.where(np.abs(i - j) <= 64, A, 0.0)
lam = np.linalg.eigvalsh(Bm); nrm = np.abs(Bm).max()
dB = ctx.to_device(Bm); dd, de = ctx.alloc(n * 8), ctx.alloc(n * 8)
dZ = ctx.to_device(np.eye(n)) if withZ else None
flags = (C.c_int * 4)()
for rep in range(2):
    if withZ: dZ.upload(np.eye(n))
    t = time.time()
    _lib.check(L.pgx_sb2_stage2_dev(ctx.handle, n, dB.ptr, dd.ptr, de.ptr, dZ.ptr if withZ else None, flags), "stage2")
    dt = time.time() - t
d = dd.download((n,), np.float64); e = de.download((n,), np.float64)[:n - 1]
print(f"n={n} NWG={os.environ.get('PG_BC_NWG')} stage 2: {dt:.3f} s flags {list(flags)} eig diff {np.abs(sl.eigvalsh_tridiagonal(d, e) - lam).max() / nrm:.2e}", flush=True)
if withZ:
    Q2 = dZ.download((n, n), np.float64); T = np.diag(d) + np.diag(e, 1) + np.diag(e, -1)
    print(f"   |Q2'Q2-I| {np.abs(Q2.T @ Q2 - np.eye(n)).max():.2e}  |Q2 T Q2' - B| {np.abs(Q2 @ T @ Q2.T - Bm).max() / nrm:.2e}", flush=True)
