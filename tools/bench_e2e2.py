"""End-to-end lmm.pygemma from host arrays with the streaming counters. usage: bench_e2e2.py n p c"""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygemma_amd import synth, lmm
n, p, c = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
rng = np.random.default_rng(0)
GK = synth.genotypes(rng, n, n)
K = (GK @ GK.T / n).astype(np.float32)
X = rng.binomial(2, 0.3, size=(n, p)).astype(np.float32)
W = np.concatenate([np.ones((n, 1)), rng.standard_normal((n, c - 1))], axis=1).astype(np.float32)
y = (GK @ (rng.standard_normal(n) * np.sqrt(0.5 / n)).astype(np.float32) + rng.standard_normal(n) * np.sqrt(0.5)).astype(np.float32).reshape(-1, 1)
Xp = lmm.pinned_empty(X.shape, np.float32); Xp[:] = X
X8 = X.astype(np.int8)
for tag, Xin in (("pageable", X), ("pinned", Xp), ("pageable", X), ("int8 pageable", X8), ("int8 pageable", X8)):
    st = {}
    t = time.time(); df = lmm.pygemma(y, Xin, W, K, stats=st); dt = time.time() - t
    print(tag, f"{dt:.3f} s wall;", {k: (round(v, 4) if isinstance(v, float) else v) for k, v in st.items()}, flush=True)
