"""Sampled fp64 invariants of pg_syevd_dev at sizes where a host reference solve is out of reach (the checks of
tests/test_gpu_fullsize.py::test_syevd_n20000_sampled_fp64_invariants for any n). usage: check_syevd_big.py n   (PG_SYEVD_STAGES=1|2 forces a path)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygemma_amd import _lib
n = int(sys.argv[1])
L = _lib.load(); ctx = _lib.Context(0)
rng = np.random.default_rng(n)
m = n // 4
G = rng.standard_normal((n, m), dtype=np.float32)
K = (G @ G.T) / np.float32(m)
K[np.arange(n), np.arange(n)] += np.float32(0.25)          # full rank: a quarter-rank Gram matrix plus a ridge
del G
dK, d64, U64 = ctx.to_device(K), ctx.alloc(n * 8), ctx.alloc(n * n * 8)
for rep in range(2):
    t = time.time()
    _lib.check(L.pg_syevd_dev(ctx.handle, n, dK.ptr, None, None, d64.ptr, U64.ptr), "pg_syevd_dev")
    print(f"syevd n={n}: {time.time() - t:.3f} s (run {rep})", flush=True)
ev = d64.download((n,), np.float64)
idx = np.concatenate([[0, 1, n - 2, n - 1], rng.choice(n, 60, replace=False)])
cols = np.concatenate([idx, rng.choice(n, 192, replace=False)])
Ufull = U64.download((n, n), np.float64)
V = Ufull[:, idx]; sub = Ufull[:, cols]
del Ufull
K64 = np.tril(K).astype(np.float64); K64 = K64 + np.tril(K64, -1).T
res = np.abs(K64 @ V - V * ev[idx][None, :]).max() / np.abs(ev).max()
orth = np.abs(sub.T @ sub - np.eye(sub.shape[1])).max()
tr = abs(ev.sum() - np.trace(K64)) / np.trace(K64)
print(f"n={n}: residual (64 sampled pairs) {res:.2e}  orthonormality (256 sampled columns) {orth:.2e}  trace {tr:.2e}  ascending {bool((np.diff(ev) >= 0).all())}", flush=True)
