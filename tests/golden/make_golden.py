#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by CALLING the real reference.

Runs in the build container only (needs /root/reference and the scratch build made by
oracle/build_ref.py, default /tmp/pygemma_ref).  What is committed is data: inputs and the
reference's outputs (+ the versions that produced them).  No reference source is copied.

    python oracle/build_ref.py && python tests/golden/make_golden.py
"""
import io
import contextlib
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = os.environ.get("PYGEMMA_REF", "/tmp/pygemma_ref")
sys.path.insert(0, REF)

warnings.filterwarnings("ignore")
with contextlib.redirect_stdout(io.StringIO()):
    from pygemma import lmm as ref  # the real reference  # noqa: E402
import scipy  # noqa: E402
import scipy.optimize  # noqa: E402
import scipy.stats  # noqa: E402
import pandas  # noqa: E402
import Cython  # noqa: E402

from pygemma_amd import synth  # noqa: E402

VERS = np.array([f"numpy {np.__version__}", f"scipy {scipy.__version__}", f"pandas {pandas.__version__}",
                 f"cython {Cython.__version__}", "reference rlangefe/pygemma @ 2024-11-15"])

LAMS = [1e-5, 1e-3, 0.2, 5.0, 400.0, 1e3, 1e5]  # tests/test_pygemma.py:253 + the search boundaries


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        return fn(*a, **k)


def defined_mask(m):
    """[row, level, col] entries precompute_mat defines: row >= col >= level... i.e. the lower
    triangle of block [i:, i, i:] (the rest is np.empty garbage / never-written upper part)."""
    msk = np.zeros((m, m, m), bool)
    for lvl in range(m):
        for r in range(lvl, m):
            for c in range(lvl, r + 1):
                msk[r, lvl, c] = True
    return msk


def gen_precompute():
    out = {"versions": VERS, "lams": np.array(LAMS, np.float32)}
    cases = [(64, 1), (64, 5), (200, 10), (500, 5)]
    out["cases"] = np.array(cases)
    for ci, (n, c) in enumerate(cases):
        rp = synth.rotated_panel(n, 3, c, seed=1000 + ci)
        d, Wx, y = rp["d"], np.ascontiguousarray(np.c_[rp["W"], rp["X"][:, 0]]), rp["Y"]
        out[f"c{ci}_d"], out[f"c{ci}_Wx"], out[f"c{ci}_y"] = d, Wx, y
        m = Wx.shape[1] + 1
        msk = defined_mask(m)
        for li, lam in enumerate(LAMS):
            for full in (0, 1):
                r = quiet(ref.precompute_mat, lam, d, Wx, y, full=bool(full))
                key = f"c{ci}_l{li}_f{full}_"
                P3 = np.where(msk, r["wjt_Pi_wk"], np.nan).astype(np.float32)
                out[key + "P3"] = P3
                out[key + "Q3"] = np.where(msk[:m - 1, :, :m - 1], r["wjt_Pi_Pi_wk"], np.nan).astype(np.float32)
                out[key + "yPy"], out[key + "yPPy"], out[key + "trP"] = r["yt_Pi_y"], r["yt_Pi_Pi_y"], r["tr_Pi"]
                out[key + "ld"] = np.float32(r["logdet_Wt_H_inv_W"])
                out[key + "ldH"] = np.float32(r["logdet_H"])
                ctot = m - 1
                if full:
                    out[key + "R3"] = np.where(msk[:m - 1, :, :m - 1], r["wjt_Pi_Pi_Pi_wk"], np.nan).astype(np.float32)
                    out[key + "yPPPy"], out[key + "trPP"] = r["yt_Pi_Pi_Pi_y"], r["tr_Pi_Pi"]
                    out[key + "d2"] = np.float32(ref.likelihood_derivative2_restricted_lambda_overload(
                        lam, n, ctot, r["yt_Pi_y"][ctot], r["yt_Pi_Pi_y"][ctot], r["yt_Pi_Pi_Pi_y"][ctot],
                        r["tr_Pi"][ctot], r["tr_Pi_Pi"][ctot]))
                out[key + "d1"] = np.float32(ref.likelihood_derivative1_restricted_lambda_overload(
                    lam, n, ctot, r["yt_Pi_y"][ctot], r["yt_Pi_Pi_y"][ctot], r["tr_Pi"][ctot]))
                out[key + "logl"] = np.float32(ref.likelihood_restricted_lambda_overload(
                    lam, n, ctot, r["yt_Pi_y"][ctot], r["logdet_H"], r["logdet_Wt_W"], r["logdet_Wt_H_inv_W"]))
    np.savez_compressed(os.path.join(HERE, "precompute_mat.npz"), **out)
    print("precompute_mat.npz", len(out), "arrays")


def run_panel(name, n, p, c, seed, null, extra=None, h2=0.5):
    rp = synth.rotated_panel(n, p, c, seed=seed, null=null, h2=h2)
    d, X, Y, W = rp["d"], rp["X"], rp["Y"], rp["W"]
    out = {"versions": VERS, "d": d, "X": X, "Y": Y, "W": W}
    for grid in (False, True):
        df = quiet(ref.pygemma, Y, X, W, d, snps=[f"rs{i}" for i in range(p)], grid=grid, eigen=False, nproc=1)
        tag = "grid" if grid else "brent"
        assert list(df.columns) == ["beta", "se_beta", "tau", "lambda", "F_wald", "p_wald", "SNPs"], df.columns
        for col in ["beta", "se_beta", "tau", "lambda", "F_wald", "p_wald"]:
            out[f"{tag}_{col}"] = df[col].to_numpy()
        out[f"{tag}_dtypes"] = np.array([str(t) for t in df.dtypes])
        # calc_lambda_restricted alone (pyx:64)
        lam = np.array([quiet(ref.calc_lambda_restricted, d, Y, np.ascontiguousarray(np.c_[W, X[:, g]]), grid=grid)
                        for g in range(p)], np.float64)
        out[f"{tag}_calc_lambda"] = lam
    # wrapper d1 on the decade grid and newton from the decade mid-points, first 40 SNPs (pyx:1631, pyx:1349)
    ks = np.arange(-5, 6)
    g40 = min(p, 40)
    d1tab = np.zeros((g40, ks.size), np.float32)
    nwt = np.zeros((g40, 10), np.float32)
    for g in range(g40):
        Wx = np.ascontiguousarray(np.c_[W, X[:, g]])
        for j, k in enumerate(ks):
            lam = np.float32(10.0 ** float(k))
            d1tab[g, j] = ref.wrapper_likelihood_derivative1_restricted_lambda(lam, d, Y, Wx)
        for j, k in enumerate(ks[:-1]):
            l0, l1 = np.float32(10.0 ** float(k)), np.float32(10.0 ** float(k + 1))
            start = np.float32(3.0) * l0
            nwt[g, j] = quiet(ref.newton, start, d, Y, Wx, precompute=True, lambda_min=l0, lambda_max=l1)
    out["d1_decades"], out["newton_from_3e_k"] = d1tab, nwt
    if extra:
        out.update(extra)
    np.savez_compressed(os.path.join(HERE, name), **out)
    nb = (out["brent_lambda"] <= 1.0001e-5).sum(), (out["brent_lambda"] >= 9.999e4).sum()
    print(name, "n,p,c =", n, p, c, "boundary-lambda rows (lo,hi):", nb)


def gen_brentq_fuzz():
    rng = np.random.default_rng(7)
    rows = []
    # family: f(x) = a3 (x-r)^3 + a1 (x-r) + a0*sin(w x) evaluated at float32(x), result float32 (as at pyx:1631)
    for t in range(400):
        k = rng.integers(-5, 5)
        a, b = 10.0 ** k, 10.0 ** (k + 1)
        r = a + (b - a) * rng.uniform(0.01, 0.99)
        a3, a1, a0, w = rng.uniform(-2, 2), rng.uniform(0.1, 3), rng.uniform(-0.05, 0.05), rng.uniform(0.1, 20)
        sgn = rng.choice([-1.0, 1.0])

        def f(x, a3=a3, a1=a1, a0=a0, w=w, r=r, sgn=sgn, s=(b - a)):
            x = np.float32(x)
            u = (np.float64(x) - r) / s
            return float(np.float32(sgn * (a3 * u ** 3 + a1 * u + a0 * np.sin(w * u))))
        if f(a) * f(b) >= 0:
            continue
        root, info = scipy.optimize.brentq(f, a, b, rtol=0.1, maxiter=100, full_output=True, disp=False)
        rows.append([a, b, r, a3, a1, a0, w, sgn, root, info.function_calls, info.iterations])
    np.savez_compressed(os.path.join(HERE, "brentq_fuzz.npz"), versions=VERS, rows=np.array(rows, np.float64),
                        columns=np.array("a b r a3 a1 a0 w sgn root funcalls iterations".split()))
    print("brentq_fuzz.npz", len(rows), "cases")


def gen_fdist():
    F = np.concatenate([[0.0, 1e-300, 1e-12, 1e-6], np.logspace(-4, 3.5, 160), [5e3, 2e4, 1e5]])
    dfd = np.array([3.0, 10.0, 57.0, 292.0, 394.0, 1938.0, 1994.0, 9994.0, 9989.0, 49994.0])
    FF, DD = np.meshgrid(F, dfd, indexing="ij")
    sf = scipy.stats.f.sf(FF, 1, DD)
    np.savez_compressed(os.path.join(HERE, "fdist_sf.npz"), versions=VERS, F=FF, dfd=DD, sf=sf)
    print("fdist_sf.npz", sf.shape, "min p", sf[sf > 0].min())


def gen_nplog():
    rng = np.random.default_rng(3)
    x = np.concatenate([np.exp(rng.uniform(0, 16, 200000)), np.linspace(1, 3, 50000)]).astype(np.float32)
    ns = [1, 7, 8, 9, 63, 64, 127, 128, 129, 255, 256, 257, 1000, 1940, 2000, 4097, 10000, 50000]
    sums = np.array([np.log(x[:k]).sum() for k in ns], np.float32)
    np.savez_compressed(os.path.join(HERE, "np_log_f32.npz"), versions=VERS, x=x[:60000], logx=np.log(x[:60000]),
                        ns=np.array(ns), sums=sums, xsum=x[:50000])
    print("np_log_f32.npz")


def gen_eigen_true():
    """Tier C fixture: a full eigen=True run of the reference (float32 eigh + sgemm inside)."""
    n, p, c = 300, 120, 3
    raw = synth.panel(n, p, c, seed=4242)
    out = {"versions": VERS, **raw}
    for grid in (False, True):
        df = quiet(ref.pygemma, raw["Y"], raw["X"], raw["W"], raw["K"], grid=grid, eigen=True, nproc=1)
        for col in ["beta", "se_beta", "tau", "lambda", "F_wald", "p_wald"]:
            out[("grid_" if grid else "brent_") + col] = df[col].to_numpy()
    np.savez_compressed(os.path.join(HERE, "eigen_true_n300.npz"), **out)
    print("eigen_true_n300.npz")


def gen_eigen_true_big(n, p=256, c=5, seed=9001):
    """Tier C at the configs' n: the REAL reference with eigen=True (float32 ssyevr + sgemm inside) on inputs that
    synth.exact_panel regenerates bit for bit from the seed on the GPU box.  Stored: the seed, Y, W, a checksum of K, the reference's
    six columns (Brent and grid) and an fp64 'truth' pipeline (host dsyevd + fp64 rotation + the oracle in the reference's order)
    for beta, se_beta, p_wald, so that the GPU test needs no n = 10 000 host eigendecomposition."""
    import time
    from oracle import oracle as O
    ex = synth.exact_panel(n, p, c, seed=seed)
    X, K, Cm = ex["X"], ex["K"], ex["C"]
    rng = np.random.default_rng(seed + 1)
    W = np.concatenate([np.ones((n, 1)), rng.standard_normal((n, c - 1))], axis=1).astype(np.float32)
    pk = Cm.shape[1]
    b = rng.standard_normal(pk) * np.sqrt(0.5 / (pk * 0.4))
    x0 = X[:, 0].astype(np.float64)
    y = 0.25 * (x0 - x0.mean()) / max(x0.std(), 1e-9) + Cm.astype(np.float64) @ b + rng.standard_normal(n) * np.sqrt(0.5)
    Y = y.astype(np.float32).reshape(-1, 1)
    del Cm
    out = {"versions": VERS, "seed": np.int64(seed), "n": np.int64(n), "p": np.int64(p), "c": np.int64(c), "Y": Y, "W": W,
           "K_sum": np.float64(K.astype(np.float64).sum()), "K_trace": np.float64(np.trace(K.astype(np.float64)))}
    for grid in (False, True):
        t = time.time()
        df = quiet(ref.pygemma, Y, X, W, K, grid=grid, eigen=True, nproc=1)
        print(f"  reference n={n} grid={grid}: {time.time() - t:.1f} s", flush=True)
        for col in ["beta", "se_beta", "tau", "lambda", "F_wald", "p_wald"]:
            out[("grid_" if grid else "brent_") + col] = df[col].to_numpy()
    t = time.time()
    K64 = np.tril(K.astype(np.float64)); K64 = K64 + np.tril(K64, -1).T
    d, U = np.linalg.eigh(K64)
    print(f"  fp64 eigh n={n}: {time.time() - t:.1f} s", flush=True)
    rot = lambda A: (U.T @ A.astype(np.float64)).astype(np.float32)
    dY, dW, dX = rot(Y), rot(W), rot(X)
    d32 = np.maximum(d, 0).astype(np.float32)
    for grid in (False, True):
        truth = O.calculate(d32, dY, dW, dX, grid=grid, order=0, nthreads=8)
        for col in ["beta", "se_beta", "p_wald", "lambda"]:
            out[("truth_grid_" if grid else "truth_brent_") + col] = np.asarray(truth[col])
    name = f"eigen_true_exact_n{n}.npz"
    np.savez_compressed(os.path.join(HERE, name), **out)
    print(name, os.path.getsize(os.path.join(HERE, name)), "bytes")


def gen_eigen_true_imputed():
    """Tier C fixtures for the two other input classes of the rotation: raw 0/1/2 calls with missing entries imputed by the
    column mean (what the reference's callers feed, experiments/benchmarks/benchmarks.py:233-244) and continuous dosages
    (experiments/wtccc/run_pygemma_imputed.py).  K is regenerated by the test from the seed (its checksum is stored)."""
    n, p, c, seed = 600, 150, 5, 777
    rng = np.random.default_rng(seed)
    GK = synth.genotypes(rng, n, 2 * n)
    K = (GK @ GK.T / (2 * n)).astype(np.float32)
    G = rng.binomial(2, rng.uniform(0.05, 0.5, p), size=(n, p)).astype(np.float64)
    miss = rng.random((n, p)) < 0.02
    miss[:, ::11] = False
    Gm = np.where(miss, np.nan, G)
    mu = np.nanmean(Gm, axis=0)
    X_imp = np.where(miss, mu[None, :], G).astype(np.float32)
    X_dos = np.clip(G + rng.normal(0, 0.15, G.shape), 0, 2).astype(np.float32)
    W = np.concatenate([np.ones((n, 1)), rng.standard_normal((n, c - 1))], axis=1).astype(np.float32)
    y = (0.25 * G[:, 0] + GK @ (rng.standard_normal(2 * n) * np.sqrt(0.5 / (2 * n))) + rng.standard_normal(n) * np.sqrt(0.5))
    y = y.astype(np.float32).reshape(-1, 1)
    out = {"versions": VERS, "seed": np.int64(seed), "K_sum": np.float64(K.astype(np.float64).sum()), "X_imp": X_imp, "X_dos": X_dos,
           "W": W, "Y": y}
    for tag, X in (("imp", X_imp), ("dos", X_dos)):
        df = quiet(ref.pygemma, y, X, W, K, grid=False, eigen=True, nproc=1)
        for col in ["beta", "se_beta", "tau", "lambda", "F_wald", "p_wald"]:
            out[f"{tag}_{col}"] = df[col].to_numpy()
    np.savez_compressed(os.path.join(HERE, "eigen_true_imputed_n600.npz"), **out)
    print("eigen_true_imputed_n600.npz")


def gen_mouse():
    """config #1 shape: real phenotype column of data/mouse_hs1940.pheno.txt (NA -> mean, as
    experiments/animal_gwas/run_gwas.py:83), SYNTHETIC genotypes / K (the real ones are not bundled,
    .MISSING_LARGE_BLOBS), intercept-only W.  A 48-SNP slice keeps the fixture small."""
    ph = np.genfromtxt("/root/reference/data/mouse_hs1940.pheno.txt", missing_values="NA", filling_values=np.nan)
    y = ph[:, 0].copy()
    y[np.isnan(y)] = np.nanmean(y)
    n = y.size
    raw = synth.panel(n, 48, 1, seed=1940)
    d, U = np.linalg.eigh(raw["K"].astype(np.float64))
    d = np.maximum(d, 0).astype(np.float32)
    rot = lambda A: (U.T @ A.astype(np.float64)).astype(np.float32)
    Xr, Yr, Wr = rot(raw["X"]), rot(y.reshape(-1, 1)), rot(raw["W"])
    out = {"versions": VERS, "d": d, "X": Xr, "Y": Yr, "W": Wr, "y_raw": y.astype(np.float32)}
    for grid in (False, True):
        df = quiet(ref.pygemma, Yr, Xr, Wr, d, grid=grid, eigen=False, nproc=1)
        for col in ["beta", "se_beta", "tau", "lambda", "F_wald", "p_wald"]:
            out[("grid_" if grid else "brent_") + col] = df[col].to_numpy()
    np.savez_compressed(os.path.join(HERE, "mouse_hs1940_synthG.npz"), **out)
    print("mouse_hs1940_synthG.npz n =", n)


if __name__ == "__main__":
    which = sys.argv[1:] or ["precompute", "panels", "brentq", "fdist", "nplog", "eigen", "mouse"]
    if "imputed" in which:
        gen_eigen_true_imputed()
    if "precompute" in which:
        gen_precompute()
    if "panels" in which:
        run_panel("panel_signal_n400_c5.npz", 400, 300, 5, seed=11, null=False)
        run_panel("panel_null_n400_c5.npz", 400, 300, 5, seed=12, null=True)
        run_panel("panel_signal_n257_c1.npz", 257, 120, 1, seed=13, null=False)
        run_panel("panel_null_n320_c10.npz", 320, 120, 10, seed=14, null=True)
        run_panel("panel_weak_n300_c3.npz", 300, 200, 3, seed=15, null=False, h2=0.04)
    if "brentq" in which:
        gen_brentq_fuzz()
    if "fdist" in which:
        gen_fdist()
    if "nplog" in which:
        gen_nplog()
    if "eigen" in which:
        gen_eigen_true()
    if "mouse" in which:
        gen_mouse()


def gen_reference_test_matrices():
    """The reference's own test inputs: generate_test_matrices(n=1000, covars=10, seed=42) (tests/test_pygemma.py:195-212)
    pushed through the live-path functions of its function list (:258-295) at lam in {1e-3, 5, 400, 1e3, 1e5} (:253).
    W there ends up holding x TWICE (np.c_[W, x] at :225 and again at :258-): c = 12 covariates, the SNP collinear with
    one of them — the degenerate case in which the MIN_VAL pivot clamps (pyx:939-961) decide the numbers."""
    np.random.seed(42)
    n, covars = 1000, 10
    K = np.random.uniform(size=(n, n))
    K = np.abs(np.tril(K) + np.tril(K, -1).T)
    K = np.dot(K, K.T)
    eigenVals, U = np.linalg.eig(K)
    eigenVals = np.maximum(0, eigenVals)
    W = np.random.rand(n, covars)
    W = np.c_[W, np.ones(n)]
    x = np.random.choice([0, 1, 2], size=(n, 1), replace=True)
    Y = np.random.rand(n, 1).reshape(-1, 1)
    x, Y, W = x.astype(np.float32), Y.astype(np.float32), W.astype(np.float32)
    eigenVals, U = np.real(eigenVals).astype(np.float32), np.real(U).astype(np.float32)
    W = np.c_[W, x]
    xr = (U.T @ x).reshape(-1)
    Yr = U.T @ Y
    Wr = np.ascontiguousarray(U.T @ W)
    out = {"versions": VERS, "d": eigenVals, "x": xr, "Y": Yr, "W": Wr, "lams": np.array([1e-3, 5.0, 400, 1e3, 1e5], np.float32)}
    Wx = np.ascontiguousarray(np.c_[Wr, xr])
    c = Wx.shape[1]
    m = c + 1
    msk = defined_mask(m)
    for li, lam in enumerate([1e-3, 5.0, 400, 1e3, 1e5]):
        for full in (0, 1):
            r = quiet(ref.precompute_mat, lam, eigenVals, Wx, Yr, bool(full))
            key = f"l{li}_f{full}_"
            out[key + "P3"] = np.where(msk, r["wjt_Pi_wk"], np.nan).astype(np.float32)
            out[key + "yPy"], out[key + "yPPy"], out[key + "trP"] = r["yt_Pi_y"], r["yt_Pi_Pi_y"], r["tr_Pi"]
            out[key + "ld"], out[key + "ldH"] = np.float32(r["logdet_Wt_H_inv_W"]), np.float32(r["logdet_H"])
            if full:
                out[key + "yPPPy"], out[key + "trPP"] = r["yt_Pi_Pi_Pi_y"], r["tr_Pi_Pi"]
        b = quiet(ref.calc_beta_vg_ve_restricted_overload, eigenVals, Wr, xr.reshape(-1, 1), np.float32(lam), Yr)
        out[f"l{li}_beta"] = np.array([b[0], b[2], b[3]], np.float32)
        out[f"l{li}_d1"] = np.float32(ref.wrapper_likelihood_derivative1_restricted_lambda(np.float32(lam), eigenVals, Yr, Wx))
        out[f"l{li}_newton"] = np.float32(quiet(ref.newton, lam, eigenVals, Yr, Wx, True))
    out["calc_lambda_restricted"] = np.float64(quiet(ref.calc_lambda_restricted, eigenVals, Yr, Wx))
    out["calc_lambda_restricted_grid"] = np.float64(quiet(ref.calc_lambda_restricted, eigenVals, Yr, Wx, grid=True))
    df = quiet(ref.pygemma, Yr, xr.reshape(-1, 1), Wr, eigenVals, eigen=False, nproc=1)
    for col in ["beta", "se_beta", "tau", "lambda", "F_wald", "p_wald"]:
        out["df_" + col] = df[col].to_numpy()
    np.savez_compressed(os.path.join(HERE, "reference_test_matrices.npz"), **out)
    print("reference_test_matrices.npz: lambda", out["calc_lambda_restricted"], "beta row", [df[c].to_numpy()[0] for c in df.columns])


if __name__ == "__main__" and "refmat" in sys.argv[1:]:
    gen_reference_test_matrices()


def gen_lrt():
    """N2: the ML functions and the LRT the reference sketches (lmm/lmm.py:22-84, 277-300; pyx:1542-1603), by CALLING the
    real reference: likelihood_lambda / likelihood_derivative1_lambda / likelihood_derivative2_lambda on a lambda list,
    calc_lambda per SNP ([W, x]) and for the null model (W), and the D_lrt / p_lrt the commented-out lines would compute
    (l from likelihood_lambda at calc_lambda's root — the sketch's likelihood(lambda, tau=n/yPy, beta_GLS) is that number
    analytically).  Panels are regenerated from synth.rotated_panel (inputs stored too)."""
    out = {"versions": VERS, "lams": np.array(LAMS, np.float32)}
    for name, (n, p, c, seed, null, h2) in {"sig": (400, 160, 5, 21, False, 0.5), "weak": (300, 120, 3, 22, False, 0.04),
                                              "null": (320, 80, 10, 23, True, 0.5), "c1": (257, 60, 1, 24, False, 0.5)}.items():
        rp = synth.rotated_panel(n, p, c, seed=seed, null=null, h2=h2)
        d, Y, W, X = rp["d"], rp["Y"], np.ascontiguousarray(rp["W"]), np.ascontiguousarray(rp["X"])
        for k, v in (("d", d), ("Y", Y), ("W", W), ("X", X)):
            out[f"{name}_{k}"] = v
        lam0 = quiet(ref.calc_lambda, d, Y, W)
        l0 = np.float32(ref.likelihood_lambda(np.float32(lam0), d, Y, W))
        out[f"{name}_lambda_null"], out[f"{name}_l_null"] = np.float64(lam0), l0
        la, ll = np.empty(p, np.float64), np.empty(p, np.float32)
        for g in range(p):
            Wx = np.ascontiguousarray(np.c_[W, X[:, g]]).astype(np.float32)
            la[g] = quiet(ref.calc_lambda, d, Y, Wx)
            ll[g] = ref.likelihood_lambda(np.float32(la[g]), d, Y, Wx)
        D = (2 * (ll - l0)).astype(np.float32)                              # lmm.py:283 on np.float32 scalars
        out[f"{name}_lambda_alt"], out[f"{name}_l_alt"], out[f"{name}_D_lrt"] = la, ll, D
        out[f"{name}_p_lrt"] = 1 - scipy.stats.chi2.cdf(x=D.astype(np.float64), df=1)   # lmm.py:300
        out[f"{name}_p_lrt_sf"] = scipy.stats.chi2.sf(D.astype(np.float64), 1)
        # the three ML scalars on the first two SNPs (+ the null model) over the lambda list
        fn = np.empty((3, len(LAMS), 3), np.float32)
        for si, Wx in enumerate([W, np.c_[W, X[:, 0]].astype(np.float32), np.c_[W, X[:, 1]].astype(np.float32)]):
            Wx = np.ascontiguousarray(Wx)
            for li, lam in enumerate(LAMS):
                fn[si, li] = [ref.likelihood_lambda(np.float32(lam), d, Y, Wx), ref.likelihood_derivative1_lambda(np.float32(lam), d, Y, Wx),
                              ref.likelihood_derivative2_lambda(np.float32(lam), d, Y, Wx)]
        out[f"{name}_ml_functions"] = fn
        print(name, "lambda_null", lam0, "l_null", l0, "D_lrt max", D.max(), "boundary roots", int((la <= 1.01e-5).sum()), int((la >= 9.9e4).sum()))
    np.savez_compressed(os.path.join(HERE, "lrt_panels.npz"), **out)


if __name__ == "__main__" and "lrt" in sys.argv[1:]:
    gen_lrt()


def gen_degenerate():
    """The degenerate panels of synth.degenerate_panels() (zero / collinear / non-finite SNPs, eigenvalues that are zero, inf, NaN
    or negative, degenerate y and W) through the real reference's live path, lmm.pygemma(..., eigen=False), for both lambda
    paths.  A case on which the reference raises is stored as such (its message), not skipped."""
    out = {"versions": VERS}
    tags = []
    for k, (tag, d, W, y, X) in enumerate(synth.degenerate_panels()):
        tags.append(tag)
        for nm, v in (("d", d), ("W", W), ("y", y), ("X", X)):
            out[f"c{k}_{nm}"] = v
        for grid in (False, True):
            key = f"c{k}_{'grid' if grid else 'brent'}"
            try:
                df = quiet(ref.pygemma, y.reshape(-1, 1), X, W, d, grid=grid, eigen=False, nproc=1)
                for col in ["beta", "se_beta", "tau", "lambda", "F_wald", "p_wald"]:
                    out[f"{key}_{col}"] = df[col].to_numpy()
                out[f"{key}_raised"] = np.array("")
                print(f"{tag:22s} grid={grid}: {int(np.isnan(df['beta'].to_numpy()).sum())} NaN rows of {len(df)}")
            except Exception as ex:
                out[f"{key}_raised"] = np.array(f"{type(ex).__name__}: {ex}"[:300])
                print(f"{tag:22s} grid={grid}: RAISED {type(ex).__name__}: {str(ex)[:120]}")
    out["tags"] = np.array(tags)
    np.savez_compressed(os.path.join(HERE, "degenerate_panels.npz"), **out)


if __name__ == "__main__" and "degenerate" in sys.argv[1:]:
    gen_degenerate()


if __name__ == "__main__" and "eigen_big" in sys.argv[1:]:
    for nn in (2000, 10000):
        gen_eigen_true_big(nn)


def _crc(a):
    import zlib
    return np.int64(zlib.crc32(np.ascontiguousarray(a).tobytes()))


def gen_tier_a_big():
    """Tier A at the configs' n: the REAL reference with eigen=False (experiments/large_gwas/run_pygemma.py:57-65 is the
    caller of that entry, lmm/lmm.py:164-167) on synth.fast_rotated_panel(n, p, c, seed) inputs, which the GPU box regenerates from the
    seed (CRC-32 of d, X, Y, W stored and asserted there).  Stored: seed, shape, the six output columns per lambda path."""
    import time
    cases = [("n10000_c5", 10000, 128, 5, 910005, (False, True)),     # configs[2]
             ("n10000_c10", 10000, 128, 10, 910010, (False, True)),   # configs[3], per-GPU kernel shape
             ("n50000_c5", 50000, 16, 5, 950005, (True,))]            # configs[4]: grid path only
    out = {"versions": VERS, "cases": np.array([c[0] for c in cases])}
    for tag, n, p, c, seed, grids in cases:
        rp = synth.fast_rotated_panel(n, p, c, seed=seed)
        out[f"{tag}_shape"] = np.array([n, p, c, seed], np.int64)
        out[f"{tag}_crc"] = np.array([_crc(rp["d"]), _crc(rp["X"]), _crc(rp["Y"]), _crc(rp["W"])], np.int64)
        for grid in grids:
            t = time.time()
            df = quiet(ref.pygemma, rp["Y"], rp["X"], rp["W"], rp["d"], grid=grid, eigen=False, nproc=1)
            print(f"  reference {tag} grid={grid}: {time.time() - t:.1f} s", flush=True)
            for col in ["beta", "se_beta", "tau", "lambda", "F_wald", "p_wald"]:
                out[f"{tag}_{'grid' if grid else 'brent'}_{col}"] = df[col].to_numpy()
    name = "tier_a_config_sizes.npz"
    np.savez_compressed(os.path.join(HERE, name), **out)
    print(name, os.path.getsize(os.path.join(HERE, name)), "bytes")


if __name__ == "__main__" and "tier_a_big" in sys.argv[1:]:
    gen_tier_a_big()
