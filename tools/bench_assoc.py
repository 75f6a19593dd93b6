"""Quick device-resident timing of pg_assoc_dev (assoc only).  usage: bench_assoc.py n p c [grid] [null]"""
import sys, time, ctypes as C
import numpy as np
sys.path.insert(0, '/root/repo')
from pygemma_amd import _lib, synth
n, p, c = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
grid = int(sys.argv[4]) if len(sys.argv) > 4 else 0
null = int(sys.argv[5]) if len(sys.argv) > 5 else 0
L = _lib.load()
ctx = _lib.Context(0)
rp = synth.fast_rotated_panel(n, p, c, null=bool(null))
ldx = (n + 63) // 64 * 64
Xr = np.zeros((p, ldx), np.float32); Xr[:, :n] = rp["X"].T
dd, dW, dy, dX = ctx.to_device(rp["d"]), ctx.to_device(rp["W"]), ctx.to_device(rp["Y"]), ctx.to_device(Xr)
out = ctx.alloc(p * 16); F = ctx.alloc(p * 16); st = ctx.alloc(16)
def run():
    L.pg_memset(ctx.handle, st.ptr, 0, 16)
    _lib.check(L.pg_assoc_dev(ctx.handle, n, c, p, dd.ptr, dW.ptr, dy.ptr, dX.ptr, ldx, grid, out.ptr, out.ptr + 4*p, out.ptr + 8*p, out.ptr + 12*p, F.ptr, F.ptr + 8*p, st.ptr), "assoc")
    ctx.sync()
run()
ts = []
for _ in range(3):
    t = time.time(); run(); ts.append(time.time() - t)
s = st.download((2,), np.uint64)
lam = out.download((p,), np.float32, 12*p)
print(f"n={n} p={p} c={c} grid={grid} null={null}: best {min(ts)*1e3:.1f} ms -> {p/min(ts):.0f} SNPs/s; evals/SNP fast={s[0]/p:.2f} full={s[1]/p:.2f}; lambda median {np.median(lam):.3g} boundary frac {(lam<=1.0001e-5).mean():.2f}")
