"""ctypes loader for the CPU oracle (oracle/pygemma_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (pygemma_amd/) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None

f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "pygemma_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B" if force else "-s"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.orc_np_logf.restype = C.c_float
        L.orc_np_logf.argtypes = [C.c_float]
        L.orc_np_sum_f32.restype = C.c_float
        L.orc_np_sum_f32.argtypes = [f32p, C.c_long]
        L.orc_logdet_H.restype = C.c_float
        L.orc_logdet_H.argtypes = [C.c_float, f32p, C.c_long]
        L.orc_hinv.restype = None
        L.orc_hinv.argtypes = [C.c_float, f32p, C.c_long, f64p]
        L.orc_precompute_mat.restype = C.c_int
        L.orc_precompute_mat.argtypes = [C.c_float, f32p, f32p, f32p, C.c_long, C.c_int, C.c_int, C.c_int] + [f32p] * 10
        L.orc_logl.restype = C.c_float
        L.orc_logl.argtypes = [C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float]
        L.orc_d1.restype = C.c_float
        L.orc_d1.argtypes = [C.c_float, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float]
        L.orc_d2.restype = C.c_float
        L.orc_d2.argtypes = [C.c_float, C.c_int, C.c_int] + [C.c_float] * 5
        L.orc_pow10f.restype = C.c_float
        L.orc_pow10f.argtypes = [C.c_int]
        L.orc_fdist_sf.restype = C.c_double
        L.orc_fdist_sf.argtypes = [C.c_double, C.c_double]
        L.orc_calculate.restype = C.c_int
        L.orc_calculate.argtypes = [f32p, f32p, f32p, f32p, C.c_long, C.c_long, C.c_long, C.c_int, C.c_long,
                                    C.c_int, C.c_int, C.c_int, f32p, f32p, f32p, f32p, f64p, f64p, i64p]
        L.orc_calc_lambda_restricted.restype = C.c_float
        L.orc_calc_lambda_restricted.argtypes = [f32p, f32p, f32p, C.c_long, C.c_int, C.c_int, C.c_int, i64p]
        L.orc_newton.restype = C.c_float
        L.orc_newton.argtypes = [C.c_float, f32p, f32p, f32p, C.c_long, C.c_int, C.c_float, C.c_float, C.c_int,
                                 C.POINTER(C.c_int)]
        L.orc_wrapper_d1.restype = C.c_float
        L.orc_wrapper_d1.argtypes = [C.c_float, f32p, f32p, f32p, C.c_long, C.c_int, C.c_int]
        L.orc_ml_logl.restype = C.c_float
        L.orc_ml_logl.argtypes = [C.c_int, C.c_float, C.c_float]
        L.orc_ml_d1.restype = C.c_float
        L.orc_ml_d1.argtypes = [C.c_float, C.c_int, C.c_float, C.c_float, C.c_float]
        L.orc_ml_d2.restype = C.c_float
        L.orc_ml_d2.argtypes = [C.c_float, C.c_int] + [C.c_float] * 5
        L.orc_calc_lambda_ml.restype = C.c_float
        L.orc_calc_lambda_ml.argtypes = [f32p, f32p, f32p, C.c_long, C.c_int, C.c_int, f32p]
        L.orc_ml_functions.restype = None
        L.orc_ml_functions.argtypes = [C.c_float, f32p, f32p, f32p, C.c_long, C.c_int, C.c_int, f32p]
        L.orc_calculate_lrt.restype = C.c_int
        L.orc_calculate_lrt.argtypes = [f32p, f32p, f32p, f32p, C.c_long, C.c_long, C.c_long, C.c_int, C.c_long, C.c_int, C.c_int,
                                        f32p, f32p, f32p, f32p, f32p, f64p]
        L.orc_chi2_sf1.restype = C.c_double
        L.orc_chi2_sf1.argtypes = [C.c_double]
        L.orc_max_threads.restype = C.c_int
        L.orc_brentq.restype = C.c_double
        L.orc_rotate.restype = None
        L.orc_rotate.argtypes = [f32p, f32p, C.c_long, C.c_long, f32p, C.c_long]
        _lib = L
    return _lib


BRENT_FN = C.CFUNCTYPE(C.c_double, C.c_double, C.c_void_p)


def brentq(f, a, b, xtol=2e-12, rtol=0.1, maxiter=100):
    """orc_brentq driven by a Python callable (fuzz tests against scipy.optimize.brentq)."""
    L = lib()
    cb = BRENT_FN(lambda x, _u: float(f(x)))
    fc, it, st = C.c_int(0), C.c_int(0), C.c_int(0)
    L.orc_brentq.argtypes = [BRENT_FN, C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int,
                             C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    r = L.orc_brentq(cb, None, a, b, xtol, rtol, maxiter, C.byref(fc), C.byref(it), C.byref(st))
    return r, fc.value, it.value, st.value


def _c(a, dt=np.float32):
    return np.ascontiguousarray(a, dtype=dt)


def precompute_mat(lam, d, Wx, y, full=True, order=0):
    """Mirror of precompute_mat(lam, eigenVals, W, Y, full) (pyx:880). Returns the reference's dict keys;
    entries the reference leaves undefined are NaN."""
    L = lib()
    d, Wx, y = _c(d), _c(Wx), _c(np.asarray(y).reshape(-1))
    n, ctot = Wx.shape
    m = ctot + 1
    P3, Q3, R3 = (np.empty((m, m, m), np.float32) for _ in range(3))
    yPy, yPPy, yPPPy, trP, trPP = (np.full(m, np.nan, np.float32) for _ in range(5))
    ld, ldH = np.zeros(1, np.float32), np.zeros(1, np.float32)
    rc = L.orc_precompute_mat(np.float32(lam), d, Wx, y, n, ctot, int(full), int(order),
                              P3, Q3, R3, yPy, yPPy, yPPPy, trP, trPP, ld, ldH)
    assert rc == 0
    out = {"wjt_Pi_wk": P3, "wjt_Pi_Pi_wk": Q3[:ctot, :, :ctot], "yt_Pi_y": yPy, "yt_Pi_Pi_y": yPPy,
           "tr_Pi": trP, "logdet_Wt_W": 0.0, "logdet_Wt_H_inv_W": float(ld[0]), "logdet_H": float(ldH[0])}
    if full:
        out.update({"wjt_Pi_Pi_Pi_wk": R3[:ctot, :, :ctot], "yt_Pi_Pi_Pi_y": yPPPy, "tr_Pi_Pi": trPP})
    return out


def calculate(d, y, W, X, grid=False, order=0, nthreads=1, snp_major=False, pvals=True):
    """Mirror of calculate((eigenVals, Y, W, X_block, grid)) (lmm:461). X is (n,p) like the reference
    unless snp_major (then (p,n)). Returns dict of arrays + evaluation counts."""
    L = lib()
    d, y, W, X = _c(d), _c(np.asarray(y).reshape(-1)), _c(W), _c(X)
    n, c = W.shape
    if snp_major:
        p = X.shape[0]; lde, lds = 1, n
    else:
        p = X.shape[1]; lde, lds = p, 1
    beta, se, tau, lam = (np.empty(p, np.float32) for _ in range(4))
    F, pv = np.empty(p, np.float64), np.empty(p, np.float64)
    ne = np.zeros(2, np.int64)
    rc = L.orc_calculate(d, y, W, X, lde, lds, n, c, p, int(grid), int(order), int(nthreads),
                         beta, se, tau, lam, F, pv, ne)
    assert rc == 0
    return {"beta": beta, "se_beta": se, "tau": tau, "lambda": lam.astype(np.float64), "F_wald": F,
            "p_wald": pv if pvals else None, "n_evals": ne}


def rotate(U, X, ldx=None):
    """U.T @ X (lmm:243-246) as a k-ordered f32 fma chain; returns SNP-major (p, ldx)."""
    L = lib()
    U, X = _c(U), _c(X)
    n, p = X.shape
    ldx = ldx or (n + 63) // 64 * 64
    out = np.empty((p, ldx), np.float32)
    L.orc_rotate(U, X, n, p, out, ldx)
    return out


def calc_lambda_ml(d, y, Wx, order=0):
    """Mirror of calc_lambda(eigenVals, Y, W) (lmm/lmm.py:22-84): (lambda_ML, likelihood_lambda at it)."""
    L = lib()
    d, y, Wx = _c(d), _c(np.asarray(y).reshape(-1)), _c(Wx)
    n, ctot = Wx.shape
    ll = np.zeros(1, np.float32)
    lam = L.orc_calc_lambda_ml(d, y, Wx, n, ctot, int(order), ll)
    return float(lam), float(ll[0])


def ml_functions(lam, d, y, Wx, order=0):
    """likelihood_lambda, likelihood_derivative1_lambda, likelihood_derivative2_lambda (pyx:1542-1603) at lam."""
    L = lib()
    d, y, Wx = _c(d), _c(np.asarray(y).reshape(-1)), _c(Wx)
    n, ctot = Wx.shape
    out = np.zeros(3, np.float32)
    L.orc_ml_functions(np.float32(lam), d, y, Wx, n, ctot, int(order), out)
    return out


def calculate_lrt(d, y, W, X, order=0, nthreads=1):
    """The LRT columns the reference sketches (lmm/lmm.py:277-300): per SNP lambda_alt = calc_lambda(d, Y, [W, x]),
    l_alt = likelihood at it; l_null the same for W alone; D_lrt = 2 (l_alt - l_null); p_lrt = chi2(1).sf(D_lrt)."""
    L = lib()
    d, y, W, X = _c(d), _c(np.asarray(y).reshape(-1)), _c(W), _c(X)
    n, c = W.shape
    p = X.shape[1]
    la, lam, D = (np.empty(p, np.float32) for _ in range(3))
    ln, lam0 = np.zeros(1, np.float32), np.zeros(1, np.float32)
    pl = np.empty(p, np.float64)
    rc = L.orc_calculate_lrt(d, y, W, X, p, 1, n, c, p, int(order), int(nthreads), la, lam, ln, lam0, D, pl)
    assert rc == 0
    return {"l_alt": la, "lambda_alt": lam, "l_null": float(ln[0]), "lambda_null": float(lam0[0]), "D_lrt": D, "p_lrt": pl}
