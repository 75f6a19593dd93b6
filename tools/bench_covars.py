"""The reference's covariate-scaling benchmark (experiments/animal_gwas/benchmark_pygemma.py:238-255: mouse HS1940 shape, n = 1940,
p = 12 226 SNPs, PCS+1 = c in {1, 6, 11, 16, 21, 26} covariates; BASELINE.md quotes its runtime-vs-covariates plot: 55 s at c=1 ...
305 s at c=26, nproc=1, eigh included) on one GPU through lmm.pygemma, synthetic genotypes/K of that shape (the real ones are not
bundled).  usage: bench_covars.py [n] [p]"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygemma_amd import lmm, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1940
p = int(sys.argv[2]) if len(sys.argv) > 2 else 12226
ref_plot = {1: 55, 6: 78, 11: 120, 16: 165, 21: 230, 26: 305}       # seconds, read off the reference's plot
rng = np.random.default_rng(1940)
GK = synth.genotypes(rng, n, 2 * n)
K = (GK @ GK.T / (2 * n)).astype(np.float32)
X = rng.binomial(2, rng.uniform(0.05, 0.5, p), size=(n, p)).astype(np.float32)
y = (0.3 * X[:, :1] + GK @ (rng.standard_normal((2 * n, 1)) * np.sqrt(0.5 / (2 * n))) + rng.standard_normal((n, 1)) * np.sqrt(0.5)).astype(np.float32)
rows = []
lmm.pygemma(y, X[:, :64], np.ones((n, 1), np.float32), K)           # warm-up (library load, first allocations)
for c in (1, 6, 11, 16, 21, 26):
    W = np.concatenate([np.ones((n, 1)), rng.standard_normal((n, c - 1))], axis=1).astype(np.float32)
    st = {}
    t = time.time(); df = lmm.pygemma(y, X, W, K, stats=st); dt = time.time() - t
    assert np.isfinite(df["beta"].to_numpy()).all()
    rows.append({"c": c, "seconds": dt, "snp_loop_seconds": st["seconds"], "snps_per_s": p / dt, "reference_plot_seconds": ref_plot[c], "speedup_vs_reference_plot": ref_plot[c] / dt})
    print(rows[-1], flush=True)
print(json.dumps({"shape": {"n": n, "p": p}, "what": "lmm.pygemma(Y, X, W, K) wall time incl. eigh, one MI355X, Brent path", "rows": rows}))
