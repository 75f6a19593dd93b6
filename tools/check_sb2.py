"""Stage-by-stage check of the two-stage tridiagonalisation (csrc/sb2.hip) on the GPU against fp64 NumPy / LAPACK:
  stage 1 (pgx_sb2_stage1_dev): the band matrix keeps K's spectrum; Q1 (applied to I) is orthogonal and Q1 B Q1' = K
  stage 2 (pgx_sb2_stage2_dev): the tridiagonal keeps the band's spectrum; Q2 orthogonal and Q2 T Q2' = B
  full    (pg_syevd_dev, PG_SYEVD_STAGES=2): orthogonality, residual, eigenvalues vs LAPACK
usage: check_sb2.py [n ...]      (every printed figure is relative to |K|; anything above ~1e-13 is a defect)"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygemma_amd import _lib  # noqa: E402

L = _lib.load()
ctx = _lib.Context(0)
B = 64


def band_of(A):
    n = A.shape[0]
    i, j = np.indices((n, n))
    Bm = np.where(np.abs(i - j) <= B, np.tril(A), 0.0)
    Bm = np.tril(Bm)
    return Bm + np.tril(Bm, -1).T


def run(n, seed=1):
    rng = np.random.default_rng(seed)
    G = rng.standard_normal((n, 2 * n))
    K = (G @ G.T / (2 * n)).astype(np.float32)
    K64 = np.tril(K.astype(np.float64)); K64 = K64 + np.tril(K64, -1).T
    nrm = np.abs(K64).max()
    lam = np.linalg.eigvalsh(K64)
    flags = (C.c_int * 4)()
    dK = ctx.to_device(K)
    # ---- stage 1
    dA = ctx.alloc(n * n * 8)
    dZ = ctx.to_device(np.eye(n))
    t = time.time()
    _lib.check(L.pgx_sb2_stage1_dev(ctx.handle, n, dK.ptr, dA.ptr, dZ.ptr, flags), "stage1")
    t1 = time.time() - t
    A = dA.download((n, n), np.float64)
    Q1 = dZ.download((n, n), np.float64)
    Bm = band_of(A)
    print(f"n={n} stage 1: flags {list(flags)}  eig diff {np.abs(np.linalg.eigvalsh(Bm) - lam).max() / nrm:.2e}  "
          f"|Q1'Q1-I| {np.abs(Q1.T @ Q1 - np.eye(n)).max():.2e}  |Q1 B Q1' - K| {np.abs(Q1 @ Bm @ Q1.T - K64).max() / nrm:.2e}  ({t1:.2f} s with hook overheads)",
          flush=True)
    # ---- stage 2 on that band
    dB = ctx.to_device(Bm)
    dd, de = ctx.alloc(n * 8), ctx.alloc(n * 8)
    dZ.upload(np.eye(n))
    t = time.time()
    _lib.check(L.pgx_sb2_stage2_dev(ctx.handle, n, dB.ptr, dd.ptr, de.ptr, dZ.ptr, flags), "stage2")
    t2 = time.time() - t
    d = dd.download((n,), np.float64); e = de.download((n,), np.float64)[:n - 1]
    Q2 = dZ.download((n, n), np.float64)
    T = np.diag(d) + np.diag(e, 1) + np.diag(e, -1)
    import scipy.linalg as sl
    lamT = sl.eigvalsh_tridiagonal(d, e)
    print(f"n={n} stage 2: flags {list(flags)}  eig diff {np.abs(lamT - lam).max() / nrm:.2e}  |Q2'Q2-I| {np.abs(Q2.T @ Q2 - np.eye(n)).max():.2e}  "
          f"|Q2 T Q2' - B| {np.abs(Q2 @ T @ Q2.T - Bm).max() / nrm:.2e}  ({t2:.2f} s with hook overheads)", flush=True)
    for b in (dA, dZ, dB, dd, de):
        b.free()
    # ---- full solver, both paths
    for stages in ("2", "1"):
        os.environ["PG_SYEVD_STAGES"] = stages
        dev, dU, d64, U64 = ctx.alloc(n * 4), ctx.alloc(n * n * 4), ctx.alloc(n * 8), ctx.alloc(n * n * 8)
        best = 1e9
        for rep in range(2):
            t = time.time()
            _lib.check(L.pg_syevd_dev(ctx.handle, n, dK.ptr, dev.ptr, dU.ptr, d64.ptr, U64.ptr), "syevd")
            best = min(best, time.time() - t)
        ev = d64.download((n,), np.float64); U = U64.download((n, n), np.float64)
        print(f"n={n} full, {stages}-stage: {best:.3f} s  |U'U-I| {np.abs(U.T @ U - np.eye(n)).max():.2e}  "
              f"residual {np.linalg.norm(K64 - (U * ev) @ U.T) / np.linalg.norm(K64):.2e}  eig diff {np.abs(ev - lam).max() / nrm:.2e}", flush=True)
        for b in (dev, dU, d64, U64):
            b.free()
    os.environ.pop("PG_SYEVD_STAGES", None)
    dK.free()


if __name__ == "__main__":
    for n in [int(a) for a in sys.argv[1:]] or [300, 1000]:
        run(n)
