"""A/B helper for association-kernel work: bit-parity against the oracle (kernel order) at one c, then the timing at the bench
shape.  The library under test comes from PYGEMMA_HIP_LIB (tools/build_dev.sh).  usage: ab_assoc.py [c] [n] [p]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygemma_amd import _lib, ops, synth
from oracle import oracle as O
c = int(sys.argv[1]) if len(sys.argv) > 1 else 5
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
p = int(sys.argv[3]) if len(sys.argv) > 3 else 16384
print("lib:", _lib.LIB_PATH, flush=True)
ctx = _lib.Context(0)
bad = 0
for seed, (nn, pp, null) in enumerate([(777, 300, False), (130, 64, True), (1000, 96, False), (2049, 40, False)]):
    rp = synth.fast_rotated_panel(nn, pp, c, seed=seed + 3, null=null)
    for grid in (False, True):
        g = ops.assoc(rp["d"], rp["W"], rp["Y"], rp["X"], grid=grid, ctx=ctx)
        o = O.calculate(rp["d"], rp["Y"], rp["W"], rp["X"], grid=grid, order=1, nthreads=16)
        for col in ("beta", "se_beta", "tau", "lambda", "F_wald"):
            a, b = np.ascontiguousarray(g[col]), np.ascontiguousarray(o[col].astype(g[col].dtype))
            nb = int((a.view(np.uint8) != b.view(np.uint8)).reshape(len(a), -1).any(axis=1).sum())
            if nb:
                bad += nb; print("MISMATCH", nn, pp, grid, col, nb, flush=True)
print("parity vs oracle(order=1):", "OK" if bad == 0 else f"{bad} rows differ", flush=True)
L = _lib.load()
rp = synth.fast_rotated_panel(n, p, c)
ldx = (n + 63) // 64 * 64
Xr = np.zeros((p, ldx), np.float32); Xr[:, :n] = rp["X"].T
dd, dW, dy, dX = ctx.to_device(rp["d"]), ctx.to_device(rp["W"]), ctx.to_device(rp["Y"]), ctx.to_device(Xr)
out = ctx.alloc(p * 16); F = ctx.alloc(p * 16); st = ctx.alloc(16)
def run(grid=0):
    L.pg_memset(ctx.handle, st.ptr, 0, 16)
    _lib.check(L.pg_assoc_dev(ctx.handle, n, c, p, dd.ptr, dW.ptr, dy.ptr, dX.ptr, ldx, grid, out.ptr, out.ptr + 4*p, out.ptr + 8*p, out.ptr + 12*p, F.ptr, F.ptr + 8*p, st.ptr), "assoc")
    ctx.sync()
for grid in (0, 1):
    run(grid); ts = []
    for _ in range(4):
        t = time.time(); run(grid); ts.append(time.time() - t)
    s = st.download((2,), np.uint64)
    print(f"n={n} p={p} c={c} grid={grid}: best {min(ts)*1e3:.2f} ms -> {p/min(ts)/1e6:.2f} M SNPs/s; evals/SNP fast={s[0]/p:.2f} full={s[1]/p:.2f}", flush=True)
