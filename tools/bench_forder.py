"""lmm.pygemma from a Fortran-ordered (SNP-major) float32 X against the C-ordered matrix, and the host transposition it avoids."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygemma_amd import synth, lmm
n, p, c = 10000, 100000, 5
rng = np.random.default_rng(0)
GK = synth.genotypes(rng, n, n)
K = (GK @ GK.T / n).astype(np.float32)
W = np.concatenate([np.ones((n, 1)), rng.standard_normal((n, c - 1))], axis=1).astype(np.float32)
y = (GK @ (rng.standard_normal(n) * np.sqrt(0.5 / n)).astype(np.float32) + rng.standard_normal(n) * np.sqrt(0.5)).astype(np.float32).reshape(-1, 1)
Xc = rng.binomial(2, 0.3, size=(n, p)).astype(np.float32)
Xf = np.asfortranarray(Xc)
t = time.time(); tmp = np.ascontiguousarray(Xf); print(f"host transposition np.ascontiguousarray(X_F) of {Xf.nbytes / 1e9:.1f} GB: {time.time() - t:.2f} s"); del tmp
for tag, X in (("C-ordered", Xc), ("F-ordered", Xf), ("C-ordered", Xc), ("F-ordered", Xf)):
    st = {}
    t = time.time(); df = lmm.pygemma(y, X, W, K, stats=st); dt = time.time() - t
    print(f"{tag}: wall {dt:.3f} s; loop {st['seconds']:.4f} s; batches {st['batches']}", flush=True)
dr = np.random.default_rng(1).gamma(0.5, 2.0, n).astype(np.float32)
for tag, X in (("eigen=False C-ordered", Xc), ("eigen=False F-ordered", Xf)):
    st = {}
    t = time.time(); df = lmm.pygemma(y, X, W, dr, eigen=False, stats=st); dt = time.time() - t
    print(f"{tag}: wall {dt:.3f} s; loop {st['seconds']:.4f} s", flush=True)
