"""The lower-triangle update C -= A'B at m = 9984 for a range of depths K: K = 8 is (almost) only the C tile's read-modify-write —
what the memory system gives this access pattern — larger K add products.  usage: bench_update_k.py"""
import sys, time, ctypes as C
import numpy as np
sys.path.insert(0, '/root/repo')
from pygemma_amd import _lib
L = _lib.load(); ctx = _lib.Context(0)
rng = np.random.default_rng(0)
m = 9984; ldt = m + 128
dC = ctx.to_device(np.zeros((m, m)))
tiles = (m // 128) * (m // 128 + 1) // 2
for K in (8, 16, 32, 64, 128, 256, 512):
    dA = ctx.to_device(rng.standard_normal((K, ldt)))
    for beta in (1.0, 0.0):
        run = lambda: _lib.check(L.pgx_dgemm_ex_dev(ctx.handle, 1 | 4, 0, m, m, K, -1.0e-3, dA.ptr, ldt, dA.ptr, ldt, beta, dC.ptr, m), "dgemm")
        run(); ctx.sync(); best = 1e9
        for _ in range(3):
            t = time.time()
            for _ in range(20): run()
            ctx.sync(); best = min(best, (time.time() - t) / 20)
        byts = tiles * 128 * 128 * 8 * (2 if beta else 1)
        fl = 2.0 * tiles * 128 * 128 * K
        print(f"K={K:4d} beta={beta}: {best*1e3:7.3f} ms   C traffic {byts/best/1e12:5.2f} TB/s   {fl/best/1e12:5.1f} TF = {fl/best/78.6e12:.2f}", flush=True)
    dA.free()
