"""End-to-end lmm.pygemma wall time from host arrays (config 3 shape by default). usage: bench_e2e.py n p c [grid]"""
import sys, time
import numpy as np
sys.path.insert(0, '/root/repo')
from pygemma_amd import synth
from pygemma import lmm
n, p, c = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
grid = bool(int(sys.argv[4])) if len(sys.argv) > 4 else False
rng = np.random.default_rng(0)
t = time.time()
GK = synth.genotypes(rng, n, 2 * n)
K = (GK @ GK.T / (2 * n)).astype(np.float32)
X = synth.genotypes(rng, n, p)
W = np.concatenate([np.ones((n, 1)), rng.standard_normal((n, c - 1))], axis=1).astype(np.float32)
y = (0.2 * X[:, 0] + GK @ (rng.standard_normal(2 * n) * np.sqrt(0.5 / (2 * n))).astype(np.float32) + rng.standard_normal(n) * np.sqrt(0.5)).astype(np.float32).reshape(-1, 1)
print(f"inputs: {time.time()-t:.1f} s", flush=True)
for rep in range(2):
    t = time.time()
    df = lmm.pygemma(y, X, W, K, grid=grid, verbose=1)
    dt = time.time() - t
    print(f"pygemma(n={n}, p={p}, c={c}, grid={grid}): {dt:.2f} s wall -> {p/dt:.0f} SNPs/s end-to-end (incl. eigh + all host<->device copies)", flush=True)
print(df.head(3))
