"""Device-resident timing of pg_rotate_dev. usage: bench_rotate.py n p"""
import sys, time
import numpy as np
sys.path.insert(0, '/root/repo')
from pygemma_amd import _lib
n, p = int(sys.argv[1]), int(sys.argv[2])
L = _lib.load(); ctx = _lib.Context(0)
rng = np.random.default_rng(0)
U = rng.standard_normal((n, n), dtype=np.float32); X = rng.standard_normal((n, p), dtype=np.float32)
ldx = (n + 63)//64*64
dU, dX = ctx.to_device(U), ctx.to_device(X); dXr = ctx.alloc(p*ldx*4)
def run():
    _lib.check(L.pg_rotate_dev(ctx.handle, n, p, dU.ptr, n, dX.ptr, p, dXr.ptr, ldx), "rot"); ctx.sync()
run(); ts=[]
for _ in range(5):
    t=time.time(); run(); ts.append(time.time()-t)
t=min(ts); print(f"rotate n={n} p={p}: {t*1e3:.1f} ms  {2*n*n*p/t/1e12:.1f} TFLOP/s  {p/t:.0f} SNPs/s")
