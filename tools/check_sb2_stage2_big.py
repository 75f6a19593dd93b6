"""Stage 2 alone (bulge chasing) on a random band matrix at sizes around the limits of the stationary kernel: the tridiagonal's spectrum
against scipy.linalg.eig_banded of the band (O(n^2 b) on the host). usage: check_sb2_stage2_big.py n [n ...]"""
import ctypes as C, os, sys, time
import numpy as np
import scipy.linalg as sl
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygemma_amd import _lib
L = _lib.load(); ctx = _lib.Context(0)
for n in map(int, sys.argv[1:]):
    rng = np.random.default_rng(n)
    b = 64
    ab = rng.standard_normal((b + 1, n))              # lower band storage: ab[i, j] = A[j + i, j]
    for i in range(1, b + 1):
        ab[i, n - i:] = 0.0
    lam = sl.eig_banded(ab, lower=True, eigvals_only=True)
    Bm = np.zeros((n, n))
    for i in range(b + 1):
        idx = np.arange(n - i)
        Bm[idx + i, idx] = ab[i, :n - i]
        Bm[idx, idx + i] = ab[i, :n - i]
    dB, dd, de = ctx.to_device(Bm), ctx.alloc(n * 8), ctx.alloc(n * 8)
    del Bm
    flags = (C.c_int * 4)()
    t = time.time()
    _lib.check(L.pgx_sb2_stage2_dev(ctx.handle, n, dB.ptr, dd.ptr, de.ptr, None, flags), "stage2")
    dt = time.time() - t
    d = dd.download((n,), np.float64); e = de.download((n,), np.float64)[:n - 1]
    err = np.abs(sl.eigvalsh_tridiagonal(d, e) - lam).max() / np.abs(lam).max()
    print(f"n={n} row blocks {(n - 1 + 63) // 64}: stage 2 {dt:.3f} s flags {list(flags)} eigenvalue difference {err:.2e}", flush=True)
    for buf in (dB, dd, de):
        buf.free()
