"""The model-level functions the reference's own tests call directly (tests/test_pygemma.py:256-294), evaluated on
the GPU through the C ABI — same names, argument order and return shapes as pygemma_model.pyx:

    precompute_mat(lam, eigenVals, W, Y, full)                     pyx:880   -> dict (reference keys)
    calc_lambda_restricted(eigenVals, Y, W, precompute, grid)      pyx:64    -> float
    calc_beta_vg_ve_restricted_overload(eigenVals, W, x, lam, Y)   pyx:1514  -> (beta, 0.0, se_beta, tau)
    newton(lam, eigenVals, Y, W, precompute, lambda_min, lambda_max) pyx:1349 -> float
    likelihood_restricted_lambda_overload(...)                     pyx:1813
    likelihood_derivative1_restricted_lambda_overload(...)         pyx:1656
    likelihood_derivative2_restricted_lambda_overload(...)         pyx:1675
    wrapper_likelihood_derivative1_restricted_lambda(lam, eigenVals, Y, W)   pyx:1631

`W` is the reference's `np.c_[W, x]` wherever the reference takes it that way (the SNP is its last column).
Single calls, made for checking fixtures: one wavefront each.  No CPU fallback.
"""
import ctypes as C

import numpy as np

from . import _lib

__all__ = ["precompute_mat", "calc_lambda_restricted", "calc_beta_vg_ve_restricted_overload", "newton", "calc_lambda",
           "likelihood_lambda", "likelihood_derivative1_lambda", "likelihood_derivative2_lambda",
           "likelihood_restricted_lambda_overload", "likelihood_derivative1_restricted_lambda_overload",
           "likelihood_derivative2_restricted_lambda_overload", "wrapper_likelihood_derivative1_restricted_lambda"]

MIN_VAL = np.float32(1e-35)   # pyx:39


def _f32(a):
    return np.ascontiguousarray(a, np.float32)


def _ctx(ctx):
    return (ctx, False) if ctx is not None else (_lib.Context(0), True)


def precompute_mat(lam, eigenVals, W, Y, full=False, ctx=None):
    """pyx:880.  W (n, c) with the SNP as last column, Y (n,1).  Entries the reference leaves undefined
    (its arrays come from np.empty) are NaN here."""
    L = _lib.load()
    d, Wx, y = _f32(eigenVals).reshape(-1), _f32(W), _f32(np.asarray(Y).reshape(-1))
    n, ctot = Wx.shape
    m = ctot + 1
    ctx, own = _ctx(ctx)
    try:
        dd, dW, dy = ctx.to_device(d), ctx.to_device(Wx), ctx.to_device(y)
        dP, dQ, dR = (ctx.alloc(m * m * m * 4) for _ in range(3))
        dv, ds = ctx.alloc(5 * m * 4), ctx.alloc(8 * 4)
        _lib.check(L.pg_precompute_mat_dev(ctx.handle, n, ctot, float(np.float32(lam)), dd.ptr, dW.ptr, dy.ptr, int(bool(full)),
                                           dP.ptr, dQ.ptr, dR.ptr, dv.ptr, ds.ptr), "pg_precompute_mat_dev")
        ctx.sync()
        P3, Q3, R3 = (b.download((m, m, m), np.float32) for b in (dP, dQ, dR))
        vecs, scal = dv.download((5, m), np.float32), ds.download((8,), np.float32)
        for b in (dd, dW, dy, dP, dQ, dR, dv, ds):
            b.free()
    finally:
        if own:
            ctx.close()
    out = {"wjt_Pi_wk": P3, "wjt_Pi_Pi_wk": Q3[:ctot, :, :ctot], "tr_Pi": vecs[3], "yt_Pi_y": vecs[0], "yt_Pi_Pi_y": vecs[1],
           "logdet_Wt_W": 0.0, "logdet_Wt_H_inv_W": float(scal[0]), "logdet_H": float(scal[1])}
    if full:
        out.update({"wjt_Pi_Pi_Pi_wk": R3[:ctot, :, :ctot], "tr_Pi_Pi": vecs[4], "yt_Pi_Pi_Pi_y": vecs[2]})
    out["_d1"], out["_d2"], out["_logl"] = scal[2], scal[3], scal[4]   # the three scalars at the last level (extra)
    out["_sum_h"], out["_sum_h2"] = scal[5], scal[6]                    # un-projected traces sum h, sum h^2 (the ML functions)
    return out


def _scalars(n, c, lam=1.0, yPy=1.0, yPPy=1.0, yPPPy=1.0, trP=0.0, trPP=0.0, ldH=0.0, ld=0.0, ctx=None):
    L = _lib.load()
    ctx, own = _ctx(ctx)
    try:
        a = np.array([lam, yPy, yPPy, yPPPy, trP, trPP, ldH, ld], np.float32)
        da, do = ctx.to_device(a), ctx.alloc(3 * 4)
        _lib.check(L.pg_reml_scalars_dev(ctx.handle, int(n), int(c), da.ptr, do.ptr), "pg_reml_scalars_dev")
        ctx.sync()
        out = do.download((3,), np.float32)
        da.free(); do.free()
        return out
    finally:
        if own:
            ctx.close()


def likelihood_restricted_lambda_overload(lam, n, c, yt_Px_y, logdet_H, logdet_Wt_W, logdet_Wt_H_inv_W, ctx=None):
    """pyx:1813.  logdet_Wt_W is 0.0 on the live path (pyx:1047); any other value is not supported."""
    if float(logdet_Wt_W) != 0.0:
        raise NotImplementedError("logdet_Wt_W is 0.0 everywhere on the reference's live path (pyx:1047)")
    return np.float32(_scalars(n, c, lam=lam, yPy=yt_Px_y, ldH=logdet_H, ld=logdet_Wt_H_inv_W, ctx=ctx)[0])


def likelihood_derivative1_restricted_lambda_overload(lam, n, c, yt_Px_y, yt_Px_Px_y, tr_Px, ctx=None):
    """pyx:1656"""
    return np.float32(_scalars(n, c, lam=lam, yPy=yt_Px_y, yPPy=yt_Px_Px_y, trP=tr_Px, ctx=ctx)[1])


def likelihood_derivative2_restricted_lambda_overload(lam, n, c, yt_Px_y, yt_Px_Px_y, yt_Px_Px_Px_y, tr_Px, tr_Px_Px, ctx=None):
    """pyx:1675"""
    return np.float32(_scalars(n, c, lam=lam, yPy=yt_Px_y, yPPy=yt_Px_Px_y, yPPPy=yt_Px_Px_Px_y, trP=tr_Px, trPP=tr_Px_Px,
                               ctx=ctx)[2])


def wrapper_likelihood_derivative1_restricted_lambda(lam, eigenVals, Y, W, ctx=None):
    """pyx:1631: precompute_mat(full=False) then the first derivative at the last level."""
    r = precompute_mat(lam, eigenVals, W, Y, full=False, ctx=ctx)
    return np.float32(r["_d1"])


def newton(lam, eigenVals, Y, W, precompute=True, lambda_min=1e-5, lambda_max=1e5, ctx=None):
    """pyx:1349 (the precompute=True branch, the only live one)."""
    if not precompute:
        raise NotImplementedError("newton(precompute=False) is dead code in the reference's live path (pyx:1371)")
    L = _lib.load()
    d, Wx, y = _f32(eigenVals).reshape(-1), _f32(W), _f32(np.asarray(Y).reshape(-1))
    n, ctot = Wx.shape
    ctx, own = _ctx(ctx)
    try:
        dd, dW, dy, do = ctx.to_device(d), ctx.to_device(Wx), ctx.to_device(y), ctx.alloc(4)
        _lib.check(L.pg_newton_dev(ctx.handle, n, ctot, float(np.float32(lam)), float(np.float32(lambda_min)),
                                   float(np.float32(lambda_max)), dd.ptr, dW.ptr, dy.ptr, do.ptr), "pg_newton_dev")
        ctx.sync()
        out = do.download((1,), np.float32)[0]
        for b in (dd, dW, dy, do):
            b.free()
        return float(out)
    finally:
        if own:
            ctx.close()


def calc_lambda_restricted(eigenVals, Y, W, precompute=True, grid=False, ctx=None):
    """pyx:64: the REML lambda for one SNP (last column of W).  Returns a Python float (a widened float32)."""
    if not precompute:
        raise NotImplementedError("calc_lambda_restricted(precompute=False) is not on the reference's live path (pyx:135)")
    from . import ops
    Wx = _f32(W)
    if Wx.shape[1] < 2:
        raise ValueError("W must hold at least one covariate column plus the SNP column")
    r = ops.assoc(eigenVals, Wx[:, :-1], Y, Wx[:, -1:], grid=grid, ctx=ctx, want_p=False)
    return float(r["lambda"][0])


def calc_beta_vg_ve_restricted_overload(eigenVals, W, x, lam, Y, ctx=None):
    """pyx:1514: one precompute_mat(full=False) at `lam`, then the float32 scalar statements of pyx:1529-1537."""
    Wf, xf = _f32(W), _f32(np.asarray(x).reshape(-1, 1))
    n, c = Wf.shape
    r = precompute_mat(lam, eigenVals, np.ascontiguousarray(np.c_[Wf, xf]), Y, full=False, ctx=ctx)
    P = r["wjt_Pi_wk"]
    with np.errstate(all="ignore"):
        beta = np.float32(P[c + 1, c, c] / P[c, c, c])                                 # f32 / f32
        ytPxy = np.float32(r["yt_Pi_y"][c + 1])
        # np.sqrt(f32) stays f32; max(f32, MIN_VAL) is an f32; np.sqrt(int) is a float64 (pyx:1533)
        se_beta = np.float32(np.float64(np.sqrt(np.float64(ytPxy))) /
                             (np.float64(np.sqrt(max(P[c, c, c], MIN_VAL))) * np.sqrt(np.float64(n - c - 1))))
        tau = np.float32(np.float32(n - c - 1) / ytPxy)
    return np.float32(beta), 0.0, np.float32(se_beta), np.float32(tau)


# ---- N2 at the model level: the ML functions the reference's calc_lambda is made of (lmm/lmm.py:22-84; pyx:1542-1603) ----------------
def _ml_scalars(lam, eigenVals, Y, W, full, ctx=None):
    """(likelihood_lambda, likelihood_derivative1_lambda, likelihood_derivative2_lambda) at `lam` for the model Y ~ W (every column
    of W projected out, the SNP — if there is one — being its last column): the quadratic forms come from precompute_mat's sweeps on
    the device, the scalars from pg_ml_scalars_dev (the statements the LRT kernel runs).  d2 is NaN unless full."""
    L = _lib.load()
    Wx = _f32(W)
    n, ctot = Wx.shape
    ctx, own = _ctx(ctx)
    try:
        r = precompute_mat(lam, eigenVals, Wx, Y, full=full, ctx=ctx)
        yPy, yPPy = r["yt_Pi_y"][ctot], r["yt_Pi_Pi_y"][ctot]
        yPPPy = r["yt_Pi_Pi_Pi_y"][ctot] if full else np.float32(0)
        args = np.array([np.float32(lam), yPy, yPPy, yPPPy, r["_sum_h"], r["_sum_h2"] if full else 0.0, r["logdet_H"]], np.float32)
        da, do = ctx.to_device(args), ctx.alloc(12)
        _lib.check(L.pg_ml_scalars_dev(ctx.handle, n, da.ptr, do.ptr), "pg_ml_scalars_dev")
        ctx.sync()
        out = do.download((3,), np.float32)
        da.free(); do.free()
        if not full:
            out[2] = np.nan
        return out
    finally:
        if own:
            ctx.close()


def likelihood_lambda(lam, eigenVals, Y, W, ctx=None):
    """pyx:1542: the ML log-likelihood at lam (np.float32)."""
    return np.float32(_ml_scalars(lam, eigenVals, Y, W, False, ctx)[0])


def likelihood_derivative1_lambda(lam, eigenVals, Y, W, ctx=None):
    """pyx:1567."""
    return np.float32(_ml_scalars(lam, eigenVals, Y, W, False, ctx)[1])


def likelihood_derivative2_lambda(lam, eigenVals, Y, W, ctx=None):
    """pyx:1586."""
    return np.float32(_ml_scalars(lam, eigenVals, Y, W, True, ctx)[2])


def calc_lambda(eigenVals, Y, W, ctx=None):
    """lmm/lmm.py:22-84: the ML lambda of the model Y ~ W — per decade of [1e-5, 1e5] with a sign change of dlogL/dlambda one root by
    scipy's brentq(rtol=0.1, maxiter=5000) refined by scipy's newton(fprime=d2, rtol=1e-5, maxiter=10), the two ends as further
    candidates, the one with the largest likelihood_lambda returned.  Same SciPy calls as the reference, the functions evaluated on
    the device.  (lmm.pygemma(..., lrt=True) runs this search inside the association kernel for every SNP at once.)"""
    from scipy import optimize
    ctx, own = _ctx(ctx)
    try:
        d1 = lambda l: likelihood_derivative1_lambda(l, eigenVals, Y, W, ctx)                    # noqa: E731
        d2 = lambda l: likelihood_derivative2_lambda(l, eigenVals, Y, W, ctx)                    # noqa: E731
        roots = [np.power(10.0, -5.0), np.power(10.0, 5.0)]
        f1 = None
        for k in np.arange(-5.0, 5.0, 1.0, dtype=np.float32):
            lambda0, lambda1 = 10.0 ** k, 10.0 ** (k + 1.0)
            f0 = d1(lambda0) if f1 is None else f1
            f1 = d1(lambda1)
            if np.sign(f0) * np.sign(f1) < 0:
                root = optimize.brentq(f=d1, a=lambda0, b=lambda1, rtol=0.1, maxiter=5000, disp=False)
                root = optimize.newton(func=d1, x0=root, rtol=1e-5, fprime=d2, maxiter=10, disp=False)
                roots.append(root)
        ll = [likelihood_lambda(lam, eigenVals, Y, W, ctx) for lam in roots]
        return roots[int(np.argmax(ll))]
    finally:
        if own:
            ctx.close()
