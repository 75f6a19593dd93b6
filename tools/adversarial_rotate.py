"""Extreme magnitudes through pg_rotate_auto_dev against an fp64 rotation.  Error bound per output: float32's own, 2^-24 per product
and per partial sum, i.e. err <= C 2^-24 sqrt(n) sum_i |x_i||u_i| with C = 4 (the bound tools/stress_rotate.py uses)."""
import sys
import numpy as np
sys.path.insert(0, '/root/repo')
from pygemma_amd import _lib, ops
rng = np.random.default_rng(3)
ctx = _lib.Context(0)
n = 777
Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
U = Q.astype(np.float32)
geno = rng.binomial(2, 0.3, size=(n, 40)).astype(np.float64)
norm = rng.standard_normal((n, 40))
blocks = {
    "genotypes": geno, "genotypes * 1e20": geno * 1e20, "genotypes * 1e-20": geno * 1e-20, "genotypes * 1e-40 (denormal step)": geno * 1e-40,
    "genotypes - 1e6 (large offset)": geno - 1e6, "genotypes mixed scales per column": geno * (10.0 ** rng.integers(-30, 30, 40))[None, :],
    "dosages": geno + rng.uniform(-0.3, 0.3, geno.shape), "normal * 1e30": norm * 1e30, "normal * 1e-30": norm * 1e-30,
    "normal * 1e-40 (denormals)": norm * 1e-40, "normal mixed scales per column": norm * (10.0 ** rng.integers(-35, 35, 40))[None, :],
    "normal mixed scales per element": norm * (10.0 ** rng.integers(-20, 20, norm.shape)),
    "near float32 max": norm * 1e38 / np.abs(norm).max() * 3.0, "all zero": np.zeros((n, 40)), "constant columns": np.ones((n, 40)) * rng.standard_normal(40)[None, :],
    "one +inf": np.where((np.arange(n)[:, None] == 5) & (np.arange(40)[None, :] == 7), np.inf, norm),
    "one NaN": np.where((np.arange(n)[:, None] == 5) & (np.arange(40)[None, :] == 7), np.nan, geno),
}
bad = 0
for name, Xd in blocks.items():
    with np.errstate(over="ignore"):
        X = Xd.astype(np.float32)
    got, path = ops.rotate_auto(U, X, ctx=ctx)
    got = got[:, :n].astype(np.float64)
    X64 = X.astype(np.float64)
    with np.errstate(invalid="ignore", over="ignore"):
        exact = (U.astype(np.float64).T @ X64).T
        bound = np.abs(X64).T @ np.abs(U.astype(np.float64))
    fin = np.isfinite(exact)
    nonfin_same = ((np.isnan(got) == np.isnan(exact)) | ~np.isfinite(exact))[~fin].all() if (~fin).any() else True
    cols_fin = np.isfinite(X64).all(0)            # columns without NaN/inf must be untouched by a neighbour's
    err = np.abs(got - exact)[cols_fin] / np.maximum(bound[cols_fin], 1e-300)
    worst = err.max() / (2.0 ** -24 * np.sqrt(n)) if err.size else 0.0
    denorm_floor = (np.abs(got - exact)[cols_fin] <= 1.5e-45 * n).all()      # outputs below float32's denormal spacing
    ok = np.isfinite(got[cols_fin]).all() and (worst <= 4.0 or denorm_floor) and nonfin_same
    bad += not ok
    print(f"{'ok ' if ok else 'BAD'} {name:38s} path {path}  worst error {worst:8.3f} x 2^-24 sqrt(n) sum|x||u|  finite cols finite: {bool(np.isfinite(got[cols_fin]).all())}  non-finite pattern ok: {bool(nonfin_same)}", flush=True)
print("problems:", bad)
sys.exit(1 if bad else 0)
