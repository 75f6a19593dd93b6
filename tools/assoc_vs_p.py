"""Time of pg_assoc_dev against the number of SNPs in the launch (wave-per-SNP grid: how the time steps with the number of
rounds of resident wavefronts).  usage: assoc_vs_p.py n c pmax step"""
import sys, time
import numpy as np
sys.path.insert(0, '/root/repo')
from pygemma_amd import _lib, synth
n, c, pmax, step = (int(a) for a in sys.argv[1:5])
L = _lib.load(); ctx = _lib.Context(0)
rp = synth.fast_rotated_panel(n, pmax, c)
ldx = (n + 63) // 64 * 64
Xr = np.zeros((pmax, ldx), np.float32); Xr[:, :n] = rp["X"].T
dd, dW, dy, dX = ctx.to_device(rp["d"]), ctx.to_device(rp["W"]), ctx.to_device(rp["Y"]), ctx.to_device(Xr)
out = ctx.alloc(pmax * 16); F = ctx.alloc(pmax * 16)
def run(p):
    _lib.check(L.pg_assoc_dev(ctx.handle, n, c, p, dd.ptr, dW.ptr, dy.ptr, dX.ptr, ldx, 0, out.ptr, out.ptr + 4*p, out.ptr + 8*p, out.ptr + 12*p, F.ptr, F.ptr + 8*p, None), "assoc")
    ctx.sync()
run(pmax)
for p in list(range(step, pmax + 1, step)):
    ts = []
    for _ in range(4):
        t = time.time(); run(p); ts.append(time.time() - t)
    print(f"p={p:6d}: {min(ts)*1e3:7.3f} ms  {min(ts)/p*1e9:7.1f} ns/SNP  rounds of 2048 waves: {p/2048:.2f}", flush=True)
