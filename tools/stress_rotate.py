"""Randomised shape sweep of the genotype / split-plane rotation against an fp64 rotation (all three input classes).
Bound: float32's componentwise 4 * 2^-24 sqrt(n) sum|x||u|, plus — for genotype blocks on the int8 kernel, which holds U as a 24-bit fixed
point per eigenvector — that representation's own worst case 2^-23 max_i|u_ik| sum_i|x_i| (it shows on sparse raw x at tiny n: a rare variant's
U'x is one or two entries of U, and an entry 100x below its column's largest is 7 bits coarser than in float32)."""
import sys
import numpy as np
sys.path.insert(0, '/root/repo')
from pygemma_amd import _lib, ops
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
ctx = _lib.Context(0)
worst = worst_raw = 0.0
import os
i8 = os.environ.get('PG_GENO_I8', '1') != '0'
for it in range(60):
    n = int(rng.integers(1, 1400)); p = int(rng.integers(1, 900))
    if it < 6:
        n, p = [(1, 1), (2, 3), (63, 257), (64, 256), (65, 255), (129, 513)][it]
    U = rng.standard_normal((n, n)).astype(np.float32) / np.float32(np.sqrt(n))
    kind = it % 3
    G = rng.binomial(2, rng.uniform(0.05, 0.5, p), size=(n, p)).astype(np.float64)
    if kind == 1:
        m = rng.random((n, p)) < 0.03
        G[m] = np.nan
        mu = np.where(np.isnan(np.nanmean(np.where(m.all(0), 0.0, G), axis=0)), 0.0, np.nanmean(np.where(m.all(0), 0.0, G), axis=0))
        G = np.where(np.isnan(G), mu[None, :], G)
    if kind == 2:
        G = G + rng.uniform(-0.4, 0.4, G.shape)
    X = G.astype(np.float32)
    got, ok = ops.rotate_geno(U, X, ctx=ctx)
    assert ok in (1, 2), (n, p, kind, ok)
    exact = (U.astype(np.float64).T @ X.astype(np.float64)).T
    bound = np.abs(X.astype(np.float64)).T @ np.abs(U.astype(np.float64)) + 1e-300
    rep = (2.0 ** -23 * np.abs(X.astype(np.float64)).sum(0)[:, None] * np.abs(U.astype(np.float64)).max(0)[None, :]) if (ok == 1 and i8) else 0.0
    aerr = np.abs(got[:, :n] - exact)
    err = (np.maximum(aerr - rep, 0.0) / bound).max()
    worst = max(worst, err / (2.0 ** -24 * max(np.sqrt(n), 1.0)))
    worst_raw = max(worst_raw, (aerr / bound).max() / (2.0 ** -24 * max(np.sqrt(n), 1.0)))
    assert err <= 4 * 2.0 ** -24 * max(np.sqrt(n), 1.0), (n, p, kind, ok, err)
    assert (got[:, n:] == 0).all()
print(f"60 shapes ok; worst error beyond the representation term = {worst:.2f} x 2^-24 sqrt(n) sum|x||u|; without that allowance {worst_raw:.2f} x")
