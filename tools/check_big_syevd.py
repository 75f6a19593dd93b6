"""One-off large-n check of pg_syevd_dev (BASELINE config 5 size): a random symmetric matrix, invariants on samples.
usage: check_big_syevd.py [n]"""
import sys, time
import numpy as np
sys.path.insert(0, '/root/repo')
from pygemma_amd import _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
L = _lib.load(); ctx = _lib.Context(0)
rng = np.random.default_rng(5)
t = time.time()
K = rng.standard_normal((n, n), dtype=np.float32)
K *= np.float32(1.0 / np.sqrt(n))
K[np.arange(n), np.arange(n)] += np.linspace(0.0, 3.0, n, dtype=np.float32)      # spread the spectrum a little
print(f"inputs {time.time()-t:.1f} s (only the lower triangle of K is read)", flush=True)
dK = ctx.to_device(K); dev = ctx.alloc(n * 4); dU = ctx.alloc(n * n * 4); d64 = ctx.alloc(n * 8)
t = time.time()
_lib.check(L.pg_syevd_dev(ctx.handle, n, dK.ptr, dev.ptr, dU.ptr, d64.ptr, None), "syevd")
print(f"syevd n={n}: {time.time()-t:.1f} s", flush=True)
ev = d64.download((n,), np.float64)
assert (np.diff(ev) >= 0).all()
U = dU.download((n, n), np.float32)
# symmetric matrix from the lower triangle, applied to sampled eigenvectors without forming it: K_sym v = L v + L' v - diag v
idx = np.array([0, 1, n // 3, n // 2, n - 2, n - 1])
V = U[:, idx].astype(np.float64)
Lo = np.tril(K)                      # float32, 10 GB at n=50k
KV = Lo.astype(np.float32) @ V.astype(np.float32)            # float32 products are enough for a 1e-5 check
KV = KV.astype(np.float64) + (Lo.T @ V.astype(np.float32)).astype(np.float64) - np.diag(K).astype(np.float64)[:, None] * V
res = np.abs(KV - V * ev[idx][None, :]).max(axis=0) / max(abs(ev[0]), abs(ev[-1]))
print("residual max|Kv - lambda v| / |lambda|max for sampled pairs:", res, flush=True)
sub = U[:, rng.choice(n, 64, replace=False)].astype(np.float64)
orth = np.abs(sub.T @ sub - np.eye(64)).max()
print(f"orthogonality of 64 random eigenvectors: {orth:.2e}; trace: sum(ev) {ev.sum():.12e} vs tr(K) {np.diag(K).astype(np.float64).sum():.12e}")
assert res.max() < 5e-5 and orth < 1e-5
print("ok")
