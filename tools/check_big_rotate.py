"""One-off large-n check of the genotype rotation (offsets beyond 2^31 bytes): sampled outputs against fp64 dot products.
usage: check_big_rotate.py [n] [p]"""
import sys, time, ctypes as C
import numpy as np
sys.path.insert(0, '/root/repo')
from pygemma_amd import _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
p = int(sys.argv[2]) if len(sys.argv) > 2 else 512
L = _lib.load(); ctx = _lib.Context(0)
rng = np.random.default_rng(0)
t = time.time()
U = rng.standard_normal((n, n), dtype=np.float32) / np.float32(np.sqrt(n))     # not orthogonal: any matrix exercises the GEMM
X = rng.binomial(2, rng.uniform(0.05, 0.5, p), size=(n, p)).astype(np.float32)
X[rng.random((n, p)) < 0.01] = 0.4321                                            # one imputed value per column -> indicator pass too
print(f"inputs {time.time()-t:.1f} s", flush=True)
ldx = (n + 63) // 64 * 64
dU, dX = ctx.to_device(U), ctx.to_device(X)
dprep = ctx.alloc(L.pg_geno_prep_bytes(n)); dwork = ctx.alloc(L.pg_geno_work_bytes(n, p)); dXr = ctx.alloc(p * ldx * 4)
_lib.check(L.pg_geno_prep_dev(ctx.handle, n, dU.ptr, n, dprep.ptr), "prep")
flag = C.c_int(0)
t = time.time()
_lib.check(L.pg_rotate_geno_dev(ctx.handle, n, p, dprep.ptr, dX.ptr, p, dXr.ptr, ldx, dwork.ptr, C.byref(flag)), "rot"); ctx.sync()
print(f"rotate_geno n={n} p={p}: is_geno={flag.value}  {time.time()-t:.3f} s", flush=True)
Xr = dXr.download((p, ldx), np.float32)
worst = 0.0
for g in [0, 1, p // 2, p - 1]:
    for k in [0, 1, 12345 % n, n // 2, n - 2, n - 1]:
        ref = float(U[:, k].astype(np.float64) @ X[:, g].astype(np.float64))
        scale = float(np.abs(U[:, k]).astype(np.float64) @ np.abs(X[:, g]).astype(np.float64))
        worst = max(worst, abs(Xr[g, k] - ref) / scale)
print(f"max |err| / sum|u||x| over 24 sampled entries: {worst:.2e}  (fp32 accumulation class: ~1e-7)", flush=True)
assert flag.value == 1 and worst < 5e-6 and (Xr[:, n:] == 0).all()
print("ok")
