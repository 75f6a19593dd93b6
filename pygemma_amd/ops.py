"""Array-level wrappers over the C ABI (host NumPy in / out).  HIP path only."""
import ctypes as C

import numpy as np

from . import _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def assoc(d, Wr, yr, Xr, grid=False, ctx=None, want_p=True, return_stats=False):
    """calculate((eigenVals, Y, W, X_block, grid)) (lmm/lmm.py:461) on the GPU.
    d (n,), Wr (n,c), yr (n,) or (n,1), Xr (n,p) in the REFERENCE layout, all in the eigenbasis."""
    L = _lib.load()
    own = ctx is None
    ctx = ctx or _lib.Context(0)
    try:
        d, Wr, yr, Xr = _f32(d), _f32(Wr), _f32(np.asarray(yr).reshape(-1)), _f32(Xr)
        n, c = Wr.shape
        p = Xr.shape[1]
        assert d.shape == (n,) and yr.shape == (n,) and Xr.shape[0] == n
        beta, se, tau, lam = (np.empty(p, np.float32) for _ in range(4))
        F, pv = np.empty(p, np.float64), np.empty(p, np.float64)
        stats = np.zeros(2, np.uint64)
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        _lib.check(L.pg_assoc(ctx.handle, n, c, p, vp(d), vp(Wr), vp(yr), vp(Xr), int(bool(grid)), vp(beta), vp(se),
                              vp(tau), vp(lam), vp(F), vp(pv) if want_p else None, vp(stats)), "pg_assoc")
        out = {"beta": beta, "se_beta": se, "tau": tau, "lambda": lam.astype(np.float64), "F_wald": F,
               "p_wald": pv if want_p else None}
        if return_stats:
            out["n_evals"] = stats.astype(np.int64)
        return out
    finally:
        if own:
            ctx.close()


def fdist_sf(F, dfd, ctx=None):
    L = _lib.load()
    own = ctx is None
    ctx = ctx or _lib.Context(0)
    try:
        F = np.ascontiguousarray(F, np.float64).ravel()
        dF = ctx.to_device(F)
        dP = ctx.alloc(F.nbytes)
        _lib.check(L.pg_fdist_sf_dev(ctx.handle, F.size, dF.ptr, float(dfd), dP.ptr), "pg_fdist_sf_dev")
        ctx.sync()
        return dP.download(F.shape, np.float64)
    finally:
        if own:
            ctx.close()


def rotate(U, X, ctx=None, ldx=None):
    """X <- U' X (lmm/lmm.py:243-246) on the GPU; U (n,n) eigenvectors in columns, X (n,p).
    Returns the SNP-major rotated block (p, ldx) float32."""
    L = _lib.load()
    own = ctx is None
    ctx = ctx or _lib.Context(0)
    try:
        U, X = _f32(U), _f32(X)
        n, p = X.shape
        ldx = ldx or (n + 63) // 64 * 64
        dU, dX = ctx.to_device(U), ctx.to_device(X)
        dXr = ctx.alloc(p * ldx * 4)
        _lib.check(L.pg_rotate_dev(ctx.handle, n, p, dU.ptr, n, dX.ptr, p, dXr.ptr, ldx), "pg_rotate_dev")
        ctx.sync()
        out = dXr.download((p, ldx), np.float32)
        for b in (dU, dX, dXr):
            b.free()
        return out
    finally:
        if own:
            ctx.close()


def syevd(K, ctx=None, want64=False):
    """scipy.linalg.eigh(K) (lmm/lmm.py:152) on the GPU: lower triangle of K (n,n) float32 ->
    (evals f32 ascending clamped >= 0, U f32 with eigenvector j in column j[, evals f64, U f64])."""
    L = _lib.load()
    own = ctx is None
    ctx = ctx or _lib.Context(0)
    try:
        K = _f32(K)
        n = K.shape[0]
        assert K.shape == (n, n)
        dK = ctx.to_device(K)
        dev, dU = ctx.alloc(n * 4), ctx.alloc(n * n * 4)
        d64 = ctx.alloc(n * 8) if want64 else None
        U64 = ctx.alloc(n * n * 8) if want64 else None
        _lib.check(L.pg_syevd_dev(ctx.handle, n, dK.ptr, dev.ptr, dU.ptr, d64.ptr if want64 else None,
                                  U64.ptr if want64 else None), "pg_syevd_dev")
        ctx.sync()
        out = [dev.download((n,), np.float32), dU.download((n, n), np.float32)]
        if want64:
            out += [d64.download((n,), np.float64), U64.download((n, n), np.float64)]
        for b in (dK, dev, dU, d64, U64):
            if b is not None:
                b.free()
        return tuple(out)
    finally:
        if own:
            ctx.close()


def rotate_geno(U, X, ctx=None, ldx=None):
    """Genotype fast path of X <- U'X (pg_rotate_geno_dev).  Returns (Xr (p, ldx) float32, True) when every column of X
    takes <= 3 equally spaced values, else (None, False)."""
    L = _lib.load()
    own = ctx is None
    ctx = ctx or _lib.Context(0)
    try:
        U, X = _f32(U), _f32(X)
        n, p = X.shape
        ldx = ldx or (n + 63) // 64 * 64
        dU, dX = ctx.to_device(U), ctx.to_device(X)
        dprep = ctx.alloc(L.pg_geno_prep_bytes(n))
        dwork = ctx.alloc(L.pg_geno_work_bytes(n, p))
        dXr = ctx.alloc(p * ldx * 4)
        _lib.check(L.pg_geno_prep_dev(ctx.handle, n, dU.ptr, n, dprep.ptr), "pg_geno_prep_dev")
        ok = C.c_int(0)
        _lib.check(L.pg_rotate_geno_dev(ctx.handle, n, p, dprep.ptr, dX.ptr, p, dXr.ptr, ldx, dwork.ptr, C.byref(ok)), "pg_rotate_geno_dev")
        ctx.sync()
        out = dXr.download((p, ldx), np.float32) if ok.value else None
        for b in (dU, dX, dprep, dwork, dXr):
            b.free()
        return out, int(ok.value)     # 1: genotype-valued block, 2: general finite block (X split in two fp16 planes)
    finally:
        if own:
            ctx.close()


def rotate_auto(U, X, ctx=None):
    """pg_rotate_auto_dev: the rotation of a float32 block with the path (genotype fp16x2 / split planes / fp32 MFMA for NaN
    blocks) chosen on the device.  Returns (Xr (p, ldx) float32, path) with path 1 / 2 / 0 like pg_rotate_geno_dev's flag."""
    L = _lib.load()
    own = ctx is None
    ctx = ctx or _lib.Context(0)
    try:
        U, X = _f32(U), _f32(X)
        n, p = X.shape
        ldx = (n + 63) // 64 * 64
        dU, dX = ctx.to_device(U), ctx.to_device(X)
        dprep, dwork = ctx.alloc(L.pg_geno_prep_bytes(n)), ctx.alloc(L.pg_geno_work_bytes(n, p))
        dXr, dpath = ctx.alloc(p * ldx * 4), ctx.alloc(4)
        _lib.check(L.pg_geno_prep_dev(ctx.handle, n, dU.ptr, n, dprep.ptr), "pg_geno_prep_dev")
        _lib.check(L.pg_rotate_auto_dev(ctx.handle, n, p, dU.ptr, n, dprep.ptr, dX.ptr, p, dXr.ptr, ldx, dwork.ptr, dpath.ptr),
                   "pg_rotate_auto_dev")
        ctx.sync()
        out = dXr.download((p, ldx), np.float32), int(dpath.download((1,), np.int32)[0])
        for b in (dU, dX, dprep, dwork, dXr, dpath):
            b.free()
        return out
    finally:
        if own:
            ctx.close()
