"""Timing of pg_kinship_geno_dev (standardisation + lower-triangle syrk). usage: bench_kinship.py n p"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygemma_amd import _lib
n, p = int(sys.argv[1]), int(sys.argv[2])
L = _lib.load(); ctx = _lib.Context(0)
rng = np.random.default_rng(0)
G = rng.integers(0, 3, size=(n, p)).astype(np.float32)
dG, dK = ctx.to_device(G), ctx.alloc(n * n * 4)
def run():
    _lib.check(L.pg_kinship_geno_dev(ctx.handle, n, p, dG.ptr, p, 1, dK.ptr), "kinship"); ctx.sync()
run(); ts = []
for _ in range(3):
    t = time.time(); run(); ts.append(time.time() - t)
t = min(ts)
print(f"kinship n={n} p={p}: {t*1e3:.2f} ms = {n*n*p/t/1e12:.1f} TFLOP/s on the lower triangle (n^2 p flop), fp32 MFMA peak 157.3")
