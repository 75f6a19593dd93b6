"""Deterministic synthetic (n, p, c) association panels (SURVEY.md §8d / BASELINE.md §3).

Genotypes G[i,g] ~ Binomial(2, maf_g), maf_g ~ U(0.05, 0.5), column-standardised;
K = G_K G_K' / p_K from an independent SNP set (full rank, PSD, realistic spectrum);
W = [1 | N(0,1)^(c-1)];  y = 0.2*g_0 + G_K b + e  (h2 = 0.5)  or a pure-noise phenotype.
Pure NumPy: this is input generation, not part of the measured path.
"""
import numpy as np

SEED = 20241115


def genotypes(rng, n, p, dtype=np.float32):
    maf = rng.uniform(0.05, 0.5, size=p)
    G = rng.binomial(2, maf, size=(n, p)).astype(np.float64)
    mu = G.mean(axis=0)
    sd = G.std(axis=0)
    sd[sd == 0] = 1.0
    return ((G - mu) / sd).astype(dtype)


def panel(n, p, c, seed=SEED, null=False, p_k=None, h2=0.5):
    """Returns dict(Y (n,1), X (n,p), W (n,c), K (n,n)) float32 — raw (un-rotated) inputs of lmm.pygemma."""
    rng = np.random.default_rng(seed)
    p_k = p_k or 2 * n
    GK = genotypes(rng, n, p_k, np.float64)
    K = (GK @ GK.T) / p_k
    X = genotypes(rng, n, p)
    W = np.concatenate([np.ones((n, 1)), rng.standard_normal((n, max(c - 1, 0)))], axis=1)[:, :c]
    if null:
        y = rng.standard_normal(n)
    else:
        b = rng.standard_normal(p_k) * np.sqrt(h2 / p_k)
        y = 0.2 * X[:, 0].astype(np.float64) + GK @ b + rng.standard_normal(n) * np.sqrt(1 - h2)
    return {"Y": y.reshape(-1, 1).astype(np.float32), "X": X, "W": W.astype(np.float32), "K": K.astype(np.float32)}


def rotated_panel(n, p, c, seed=SEED, null=False, h2=0.5):
    """Eigen-basis inputs (the reference's eigen=False entry, lmm/lmm.py:164-167): d (n,), Xr (n,p), Yr, Wr.
    Rotation done in float64 on the host (input preparation for kernels that start at that boundary)."""
    raw = panel(n, p, c, seed, null, h2=h2)
    d, U = np.linalg.eigh(raw["K"].astype(np.float64))
    d = np.maximum(d, 0.0).astype(np.float32)
    rot = lambda A: (U.T @ A.astype(np.float64)).astype(np.float32)
    return {"d": d, "X": rot(raw["X"]), "Y": rot(raw["Y"]), "W": rot(raw["W"])}


def fast_rotated_panel(n, p, c, seed=SEED, null=False):
    """Cheap large-shape eigen-basis inputs for throughput runs: a synthetic spectrum (bulk + a few
    large eigenvalues, like a GRM's) and Gaussian rotated columns with a planted polygenic component,
    so the Brent/Newton evaluation counts are in the realistic (signal) regime. O(n p)."""
    rng = np.random.default_rng(seed)
    d = np.sort(rng.gamma(2.0, 0.5, size=n)).astype(np.float32)
    d[-5:] *= np.array([3, 5, 8, 12, 40], np.float32)
    d[: max(1, n // 100)] = 0.0
    X = rng.standard_normal((n, p), dtype=np.float32)
    W = rng.standard_normal((n, c), dtype=np.float32)
    if null:
        y = rng.standard_normal(n, dtype=np.float32)
    else:
        lam = 1.0
        y = (rng.standard_normal(n) * np.sqrt(0.5 * (lam * d.astype(np.float64) + 1.0))).astype(np.float32)
        y += 0.05 * X[:, 0]
    return {"d": d, "X": X, "Y": y.reshape(-1, 1), "W": W}
