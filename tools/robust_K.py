"""Degenerate relatedness matrices through the whole pipeline (lmm.pygemma, eigen=True): rank-deficient K, duplicated samples,
block-diagonal K, K = 0, K scaled by 1e-6 / 1e6 — against an fp64 pipeline (numpy eigh + rotation + oracle in the reference's order).
Degenerate eigenspaces have no unique basis, but the statistics do not depend on the basis: they must agree."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygemma_amd import lmm, synth
from oracle import oracle as O
rng = np.random.default_rng(5)
n, p, c = 401, 64, 3
G = rng.binomial(2, 0.3, size=(n, 4 * n)).astype(np.float64); G = (G - G.mean(0)) / np.maximum(G.std(0), 1e-9)
def kin(Z): return (Z @ Z.T / Z.shape[1])
Ks = {"regular": kin(G), "rank deficient (p_k = n/4)": kin(G[:, : n // 4])}
Gd = G.copy(); Gd[n // 2:] = Gd[: n - n // 2]; Ks["duplicated samples"] = kin(Gd)
B = np.zeros((n, n)); s = 0
while s < n:
    b = min(int(rng.integers(2, 30)), n - s); A = rng.standard_normal((b, 3 * b)); B[s:s + b, s:s + b] = A @ A.T / (3 * b); s += b
Ks["block diagonal"] = B; Ks["zero"] = np.zeros((n, n)); Ks["regular * 1e-6"] = kin(G) * 1e-6; Ks["regular * 1e6"] = kin(G) * 1e6
Ks["identity"] = np.eye(n)
X = rng.binomial(2, 0.25, size=(n, p)).astype(np.float32)
W = np.concatenate([np.ones((n, 1)), rng.standard_normal((n, c - 1))], axis=1).astype(np.float32)
bad = 0
for name, K in Ks.items():
    K32 = K.astype(np.float32)
    K64 = np.tril(K32.astype(np.float64)); K64 = K64 + np.tril(K64, -1).T
    d0, U0 = np.linalg.eigh(K64)
    dmax = max(np.abs(d0).max(), 1e-300)
    g = U0 @ (np.sqrt(np.maximum(d0, 0) / dmax) * rng.standard_normal(n))
    y = (0.4 * X[:, 0] + 0.7 * g + 0.7 * rng.standard_normal(n)).astype(np.float32).reshape(-1, 1)
    for kw in ({}, {"grid": True}):
        df = lmm.pygemma(y, X, W, K32, **kw)
        d, U = np.linalg.eigh(K64)
        rot = lambda A: (U.T @ A.astype(np.float64)).astype(np.float32)
        tr = O.calculate(np.maximum(d, 0).astype(np.float32), rot(y), rot(W), rot(X), grid=bool(kw.get("grid")), order=0, nthreads=4)
        b, t, se = df["beta"].to_numpy().astype(np.float64), tr["beta"].astype(np.float64), tr["se_beta"].astype(np.float64)
        ok = np.isfinite(t) & np.isfinite(b)
        z = np.abs(b[ok] - t[ok]) / se[ok]
        lam_rel = np.abs(df["lambda"].to_numpy()[ok] - tr["lambda"][ok]) / np.maximum(tr["lambda"][ok], 1e-5)
        flag = "" if (ok.all() and z.max() < 2e-2) else "  <-- CHECK"
        bad += bool(flag)
        print(f"{name:28s} {str(kw):16s} finite {int(ok.sum())}/{p}  max |dbeta|/se {z.max():.2e}  median {np.median(z):.2e}  median |dlam|/lam {np.median(lam_rel):.1e}{flag}", flush=True)
print("problems:", bad)
