"""Device-resident timing of pg_assoc_lrt_dev against pg_assoc_dev at the bench shape. usage: bench_lrt.py [n] [p] [c]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygemma_amd import _lib, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
p = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
c = int(sys.argv[3]) if len(sys.argv) > 3 else 5
L = _lib.load(); ctx = _lib.Context(0)
rp = synth.fast_rotated_panel(n, p, c)
ldx = (n + 63) // 64 * 64
Xr = np.zeros((p, ldx), np.float32); Xr[:, :n] = rp["X"].T
dd, dW, dy, dX = ctx.to_device(rp["d"]), ctx.to_device(rp["W"]), ctx.to_device(rp["Y"]), ctx.to_device(Xr)
o4, o8 = ctx.alloc(16 * p), ctx.alloc(48 * p)
def wald():
    _lib.check(L.pg_assoc_dev(ctx.handle, n, c, p, dd.ptr, dW.ptr, dy.ptr, dX.ptr, ldx, 0, o4.ptr, o4.ptr + 4*p, o4.ptr + 8*p, o4.ptr + 12*p, o8.ptr, o8.ptr + 8*p, None), "assoc"); ctx.sync()
def lrt():
    _lib.check(L.pg_assoc_lrt_dev(ctx.handle, n, c, p, dd.ptr, dW.ptr, dy.ptr, dX.ptr, ldx, 0, o4.ptr, o4.ptr + 4*p, o4.ptr + 8*p, o4.ptr + 12*p,
                                  o8.ptr, o8.ptr + 8*p, o8.ptr + 16*p, o8.ptr + 24*p, o8.ptr + 32*p, o8.ptr + 40*p), "lrt"); ctx.sync()
for name, fn in (("Wald only (pg_assoc_dev)", wald), ("Wald + LRT (pg_assoc_lrt_dev)", lrt)):
    fn(); ts = []
    for _ in range(3):
        t = time.time(); fn(); ts.append(time.time() - t)
    print(f"{name:32s} n={n} p={p} c={c}: {min(ts)*1e3:.2f} ms -> {p/min(ts)/1e6:.2f} M SNPs/s", flush=True)
