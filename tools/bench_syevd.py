"""Timing + invariants of pg_syevd_dev. usage: bench_syevd.py n [check]"""
import sys, time
import numpy as np
sys.path.insert(0, '/root/repo')
from pygemma_amd import _lib
n = int(sys.argv[1]); check = len(sys.argv) > 2
L = _lib.load(); ctx = _lib.Context(0)
rng = np.random.default_rng(1)
p_k = 2 * n
G = rng.binomial(2, rng.uniform(0.05, 0.5, p_k), size=(n, p_k)).astype(np.float32)
G = (G - G.mean(0)) / G.std(0)
K = (G @ G.T / p_k).astype(np.float32)
dK = ctx.to_device(K); dev = ctx.alloc(n*4); dU = ctx.alloc(n*n*4); d64 = ctx.alloc(n*8); U64 = ctx.alloc(n*n*8)
for rep in range(2):
    t = time.time()
    _lib.check(L.pg_syevd_dev(ctx.handle, n, dK.ptr, dev.ptr, dU.ptr, d64.ptr, U64.ptr), "syevd")
    print(f"syevd n={n}: {time.time()-t:.3f} s (run {rep})", flush=True)
if check:
    ev = d64.download((n,), np.float64); U = U64.download((n, n), np.float64)
    K64 = np.tril(K.astype(np.float64)); K64 = K64 + np.tril(K64, -1).T
    t = time.time(); orth = np.abs(U.T @ U - np.eye(n)).max()
    res = np.linalg.norm(K64 - (U * ev) @ U.T) / np.linalg.norm(K64)
    print(f"orth {orth:.2e}  residual {res:.2e}  (check {time.time()-t:.1f}s)", flush=True)
    t = time.time(); ref = np.linalg.eigvalsh(K64); tl = time.time() - t
    print(f"eig err {np.abs(ev-ref).max()/np.abs(ref).max():.2e}; host LAPACK eigvalsh (values only) {tl:.1f} s", flush=True)
