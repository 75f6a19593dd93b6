"""Pins the CPU oracle (oracle/pygemma_oracle.c) to golden vectors emitted by the REAL reference
(tests/golden/make_golden.py).  CPU-only.  Bar: bit-exact on every f32/f64 column except p_wald
(third-party scipy.stats.f.sf: 1e-9 relative)."""
import os

import numpy as np
import pytest

from oracle import oracle as O

G = os.path.join(os.path.dirname(__file__), "golden")


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint32) if a.dtype == np.float32 else a.view(np.uint64)


def test_numpy_f32_log_and_pairwise_sum_bit_exact():
    """pyx:972 logdet_H uses numpy's float32 log and float32 pairwise sum: reproduce both bit-for-bit
    (fixture from the build container + the live numpy of whatever host runs this test)."""
    L = O.lib()
    z = np.load(os.path.join(G, "np_log_f32.npz"))
    x = z["x"][:30000]
    mine = np.array([L.orc_np_logf(float(v)) for v in x], np.float32)
    assert (bits(mine) == bits(z["logx"][:30000])).all()
    xs = z["xsum"]
    for k, s in zip(z["ns"], z["sums"]):
        t = np.ascontiguousarray(np.log(xs[:k]))
        assert np.float32(L.orc_np_sum_f32(t, int(k))) == s, k
    rng = np.random.default_rng(5)
    v = rng.uniform(0, 12, 70001).astype(np.float32)
    for n in (5, 100, 8191, 8192, 8193, 10000, 16385, 50000, 70001):
        a = np.ascontiguousarray(v[:n])
        assert np.float32(L.orc_np_sum_f32(a, n)) == a.sum(), n


def _cmp(a, b, allow=0):
    a = np.asarray(a, np.float32)
    b = np.asarray(b, np.float32)
    msk = ~np.isnan(b)
    return int((bits(a[msk]) != bits(b[msk])).sum()), int(msk.sum())


def test_precompute_mat_and_scalars_vs_reference():
    """precompute_mat dicts (pyx:880) for n in {64,200,500}, c in {1,5,10}, the lambda list of
    tests/test_pygemma.py:253 plus the boundaries.  The quadratic forms the live path reads (yPy, yPPy,
    trP, logdets at every level; P at every level) must be bit-exact.  Q/R blocks at lambda=1e5 are
    ill-conditioned (cancellation amplifies the reference's own BLAS summation-order noise): bounded."""
    L = O.lib()
    z = np.load(os.path.join(G, "precompute_mat.npz"))
    exact_bad = 0
    loose = []
    for ci, (n, c) in enumerate(z["cases"]):
        d, Wx, y = z[f"c{ci}_d"], z[f"c{ci}_Wx"], z[f"c{ci}_y"]
        ctot = Wx.shape[1]
        for li, lam in enumerate(z["lams"]):
            for full in (0, 1):
                k = f"c{ci}_l{li}_f{full}_"
                r = O.precompute_mat(lam, d, Wx, y, full=bool(full), order=0)
                for mine, key in ((r["wjt_Pi_wk"], "P3"), (r["yt_Pi_y"], "yPy"), (r["yt_Pi_Pi_y"], "yPPy"),
                                  (r["tr_Pi"], "trP"), ([r["logdet_Wt_H_inv_W"]], "ld"), ([r["logdet_H"]], "ldH")):
                    bad, tot = _cmp(mine, z[k + key])
                    exact_bad += bad
                pairs = [(r["wjt_Pi_Pi_wk"], "Q3")]
                if full:
                    pairs += [(r["wjt_Pi_Pi_Pi_wk"], "R3"), (r["yt_Pi_Pi_Pi_y"], "yPPPy"), (r["tr_Pi_Pi"], "trPP")]
                for mine, key in pairs:
                    bad, tot = _cmp(mine, z[k + key])
                    if lam < 1e4:
                        assert bad <= 4, (k, key, bad)   # isolated 1-ulp flips only
                        ref = z[k + key]; msk = ~np.isnan(ref)
                        np.testing.assert_allclose(np.asarray(mine, np.float32)[msk], ref[msk], rtol=2e-7)
                    else:
                        loose.append(bad / tot)
                # scalar functions on the REFERENCE's own quadratic forms: must be bit-exact
                d1 = L.orc_d1(lam, int(n), ctot, z[k + "yPy"][ctot], z[k + "yPPy"][ctot], z[k + "trP"][ctot])
                assert bits(np.float32(d1)) == bits(z[k + "d1"])
                ll = L.orc_logl(int(n), ctot, z[k + "yPy"][ctot], z[k + "ldH"], 0.0, z[k + "ld"])
                assert bits(np.float32(ll)) == bits(z[k + "logl"])
                if full:
                    d2 = L.orc_d2(lam, int(n), ctot, z[k + "yPy"][ctot], z[k + "yPPy"][ctot], z[k + "yPPPy"][ctot],
                                  z[k + "trP"][ctot], z[k + "trPP"][ctot])
                    assert bits(np.float32(d2)) == bits(z[k + "d2"])
    assert exact_bad == 0


PANELS = ["panel_signal_n400_c5", "panel_null_n400_c5", "panel_signal_n257_c1", "panel_null_n320_c10",
          "panel_weak_n300_c3", "mouse_hs1940_synthG"]


@pytest.mark.parametrize("name", PANELS)
@pytest.mark.parametrize("grid", [False, True])
@pytest.mark.parametrize("order", [0, 1])
def test_calculate_vs_reference_dataframe(name, grid, order):
    """lmm.pygemma(..., eigen=False) DataFrames of the reference (lmm:87-411 -> calculate lmm:461):
    beta, se_beta, tau (f32), lambda, F_wald (f64 holding f32-precision values) bit-exact on >= 99 % of
    rows (here: all); p_wald within 1e-9 relative.  order=1 is the HIP kernels' summation order."""
    z = np.load(os.path.join(G, name + ".npz"))
    tag = "grid" if grid else "brent"
    r = O.calculate(z["d"], z["Y"], z["W"], z["X"], grid=grid, order=order, nthreads=4)
    p = len(r["beta"])
    rowbad = np.zeros(p, bool)
    for col in ["beta", "se_beta", "tau", "lambda", "F_wald"]:
        ref = z[f"{tag}_{col}"]
        mine = r[col].astype(ref.dtype)
        rowbad |= bits(mine) != bits(ref)
    assert rowbad.mean() <= 0.01, rowbad.sum()
    ref = z[f"{tag}_p_wald"]
    np.testing.assert_allclose(r["p_wald"], ref, rtol=1e-9, atol=0)
    if rowbad.any():   # the rare flipped rows stay within the Tier-A tolerances
        for col, tol in (("beta", 1e-4), ("se_beta", 1e-4), ("lambda", 2e-5)):
            np.testing.assert_allclose(r[col][rowbad], z[f"{tag}_{col}"][rowbad], rtol=tol)


@pytest.mark.parametrize("name", PANELS[:5])
def test_calc_lambda_d1_newton_vs_reference(name):
    """calc_lambda_restricted (pyx:64), wrapper d1 on the decade grid (pyx:1631), newton (pyx:1349)."""
    L = O.lib()
    z = np.load(os.path.join(G, name + ".npz"))
    d, Y, W, X = z["d"], np.ascontiguousarray(z["Y"].reshape(-1)), z["W"], z["X"]
    n, c = W.shape
    ne = np.zeros(2, np.int64)
    for grid, tag in ((0, "brent"), (1, "grid")):
        ref = z[f"{tag}_calc_lambda"]
        mine = np.array([L.orc_calc_lambda_restricted(d, Y, np.ascontiguousarray(np.c_[W, X[:, g]]), n, c + 1, grid, 0, ne)
                         for g in range(X.shape[1])], np.float64)
        assert (mine != ref).mean() <= 0.01
    d1 = z["d1_decades"]
    nw = z["newton_from_3e_k"]
    import ctypes
    it = ctypes.c_int(0)
    for g in range(d1.shape[0]):
        Wx = np.ascontiguousarray(np.c_[W, X[:, g]])
        for j, k in enumerate(range(-5, 6)):
            lam = np.float32(10.0 ** float(k))
            v = np.float32(L.orc_wrapper_d1(lam, d, Y, Wx, n, c + 1, 0))
            assert bits(v) == bits(d1[g, j]), (g, k)
        for j, k in enumerate(range(-5, 5)):
            l0, l1 = np.float32(10.0 ** float(k)), np.float32(10.0 ** float(k + 1))
            v = np.float32(L.orc_newton(np.float32(3.0) * l0, d, Y, Wx, n, c + 1, l0, l1, 0, ctypes.byref(it)))
            assert bits(v) == bits(nw[g, j]) or abs(float(v) - float(nw[g, j])) <= 2e-5 * abs(float(nw[g, j])), (g, k)


def test_brentq_vs_scipy_fixture():
    """Port of scipy.optimize.brentq (third-party, SciPy 1.15.3) vs SciPy's own roots + call counts."""
    z = np.load(os.path.join(G, "brentq_fuzz.npz"))
    for a, b, r, a3, a1, a0, w, sgn, root, fc, it in z["rows"]:
        s = b - a

        def f(x):
            x = np.float32(x)
            u = (np.float64(x) - r) / s
            return float(np.float32(sgn * (a3 * u ** 3 + a1 * u + a0 * np.sin(w * u))))
        got, gfc, git, st = O.brentq(f, a, b)
        assert st == 0
        assert got == root and gfc == int(fc) and git == int(it), (a, b, got, root)


def test_fdist_sf_vs_scipy_fixture():
    """scipy.stats.f.sf(F, 1, dfd) (lmm:482), down to p ~ 1e-300."""
    L = O.lib()
    z = np.load(os.path.join(G, "fdist_sf.npz"))
    F, D, sf = z["F"].ravel(), z["dfd"].ravel(), z["sf"].ravel()
    mine = np.array([L.orc_fdist_sf(float(f), float(d)) for f, d in zip(F, D)])
    ok = sf > 1e-300
    np.testing.assert_allclose(mine[ok], sf[ok], rtol=1e-9)
    assert (mine[~ok] <= 1e-299).all()


def test_reference_output_schema_pin():
    """experiments/large_gwas/output.txt:1 header == the column order the golden DataFrames carry; dtypes
    float32 x3, float64 x3, object."""
    z = np.load(os.path.join(G, "panel_signal_n400_c5.npz"))
    assert list(z["brent_dtypes"]) == ["float32", "float32", "float32", "float64", "float64", "float64", "object"]
    assert z["brent_lambda"].dtype == np.float64 and z["brent_beta"].dtype == np.float32
    # lambda column holds float32-precision values widened to float64 (e.g. 9.999999747378752e-06)
    lam = z["brent_lambda"]
    assert (lam.astype(np.float32).astype(np.float64) == lam).all()


def test_reference_own_test_matrices_degenerate_collinear_snp():
    """The reference's own test inputs (tests/test_pygemma.py:195-212,:253-295): n=1000, 12 covariates of which one IS the
    SNP (x appears twice in W*), so the x pivot is ~0 and the MIN_VAL clamps (pyx:939-961) decide beta/se.  The live-path
    functions must still reproduce the reference: quadratic forms, d1, newton, lambda, and the DataFrame row."""
    import ctypes
    L = O.lib()
    z = np.load(os.path.join(G, "reference_test_matrices.npz"))
    d, xr, Y, W = z["d"], z["x"], np.ascontiguousarray(z["Y"].reshape(-1)), z["W"]
    Wx = np.ascontiguousarray(np.c_[W, xr])
    n, ctot = Wx.shape
    nbad = 0
    for li, lam in enumerate(z["lams"]):
        for full in (0, 1):
            r = O.precompute_mat(lam, d, Wx, Y, full=bool(full), order=0)
            k = f"l{li}_f{full}_"
            for mine, key in ((r["yt_Pi_y"], "yPy"), (r["yt_Pi_Pi_y"], "yPPy"), (r["tr_Pi"], "trP")):
                bad, tot = _cmp(mine, z[k + key])
                nbad += bad
        v = np.float32(L.orc_wrapper_d1(np.float32(lam), d, Y, Wx, n, ctot, 0))
        # cancellation noise of the collinear column reaches d1 through yPPy: a few per cent, never the sign
        assert np.sign(v) == np.sign(z[f"l{li}_d1"]) and abs(float(v) - float(z[f"l{li}_d1"])) <= 5e-2 * abs(float(z[f"l{li}_d1"]))
    assert nbad <= 40          # a collinear column makes the last two levels pure cancellation noise; the others agree
    ne = np.zeros(2, np.int64)
    for grid, key in ((0, "calc_lambda_restricted"), (1, "calc_lambda_restricted_grid")):
        lam = L.orc_calc_lambda_restricted(d, Y, Wx, n, ctot, grid, 0, ne)
        assert np.float64(np.float32(lam)) == z[key]
    r = O.calculate(d, Y, W, xr.reshape(-1, 1), grid=False, order=0, nthreads=1)
    assert r["lambda"][0] == z["df_lambda"][0]
    for col in ("beta", "se_beta", "tau"):
        a, b = float(r[col][0]), float(z["df_" + col][0])
        assert np.isfinite(a) == np.isfinite(b)
        # the x pivot is the difference of two ~equal numbers: agreement to a few per cent is all that is defined here
        assert abs(a - b) <= (1e-4 if col == "tau" else 5e-2) * abs(b)


# ---- N2: ML functions + LRT (tolerance parity: the reference's quadratic forms come from float32 NumPy helpers) ---------------
@pytest.mark.parametrize("name", ["sig", "weak", "null", "c1"])
@pytest.mark.parametrize("order", [0, 1])
def test_lrt_oracle_vs_reference_fixtures(name, order):
    """calc_lambda (lmm/lmm.py:22-84) + likelihood_lambda (pyx:1542-1562) per SNP and for the null model, D_lrt and p_lrt as the
    commented-out lines would form them (lmm.py:277-300): oracle vs the real reference's own functions (tests/golden/lrt_panels.npz).
    Tolerances = the reference's float32 noise: log-likelihoods within 2 float32 ulp, D within 2 quanta, lambda 5e-5."""
    z = np.load(os.path.join(G, "lrt_panels.npz"))
    d, Y, W, X = (z[f"{name}_{k}"] for k in "dYWX")
    r = O.calculate_lrt(d, Y, W, X, order=order, nthreads=4)
    ulp = np.spacing(np.float32(abs(float(z[f"{name}_l_null"]))))
    assert abs(r["l_null"] - float(z[f"{name}_l_null"])) <= 2 * ulp
    assert abs(r["lambda_null"] / float(z[f"{name}_lambda_null"]) - 1) <= 5e-5
    assert np.abs(r["l_alt"].astype(np.float64) - z[f"{name}_l_alt"]).max() <= 2 * ulp
    assert np.abs(r["D_lrt"].astype(np.float64) - z[f"{name}_D_lrt"]).max() <= 4 * ulp + 1e-6
    assert np.abs(r["lambda_alt"] / z[f"{name}_lambda_alt"] - 1).max() <= 5e-5
    ref_p = z[f"{name}_p_lrt_sf"]
    assert np.abs(r["p_lrt"] / ref_p - 1).max() <= 5e-3
    big = z[f"{name}_p_lrt"] > 1e-10                 # the reference's 1 - cdf form is the same number where it has digits
    np.testing.assert_allclose(z[f"{name}_p_lrt"][big], ref_p[big], rtol=1e-5)


def test_ml_scalar_functions_vs_reference_fixtures():
    """likelihood_lambda / likelihood_derivative1_lambda / likelihood_derivative2_lambda (pyx:1542-1603) on the lambda list:
    logL agrees to float32 rounding wherever the reference's float32 projector is well conditioned (lambda <= 400; 2e-6 at 1e3); the
    derivatives, which the reference forms by cancellation in float32, within 1e-3 for 1e-3 < lambda < 1e4."""
    z = np.load(os.path.join(G, "lrt_panels.npz"))
    lams = z["lams"]
    for name in ["sig", "weak", "null", "c1"]:
        d, Y, W, X = (z[f"{name}_{k}"] for k in "dYWX")
        fn = z[f"{name}_ml_functions"]
        for si, Wx in enumerate([W, np.c_[W, X[:, 0]], np.c_[W, X[:, 1]]]):
            Wx = np.ascontiguousarray(Wx, np.float32)
            for li, lam in enumerate(lams):
                o = O.ml_functions(lam, d, Y, Wx)
                if lam <= 400:
                    assert abs(o[0] - fn[si, li, 0]) <= 3 * np.spacing(np.float32(abs(fn[si, li, 0]))), (name, si, lam)
                elif lam <= 1e3:     # the reference's float32 inverse of W'H^-1 W starts to lose digits
                    assert abs(o[0] / fn[si, li, 0] - 1) <= 2e-6, (name, si, lam)
                if 1e-3 < lam < 1e4:
                    assert abs(o[1] / fn[si, li, 1] - 1) <= 1e-3 and abs(o[2] / fn[si, li, 2] - 1) <= 2e-3, (name, si, lam, o, fn[si, li])


from pygemma_amd.synth import DEGENERATE_SINGULAR_CASES as _SINGULAR_CASES, DEGENERATE_SINGULAR_SNPS as _SINGULAR_SNPS  # noqa: E402


@pytest.mark.parametrize("order", [0, 1])
@pytest.mark.parametrize("grid", [False, True])
def test_degenerate_panels_vs_reference(grid, order):
    """synth.degenerate_panels() through the real reference's lmm.pygemma(eigen=False) (fixture made by make_golden.py degenerate;
    the reference never raises on them: it returns rows): same NaN rows and, in the reference's own summation order, the same bits,
    with eigenvalues clamped at 0 as lmm/lmm.py:166-167 does before the scan."""
    z = np.load(os.path.join(G, "degenerate_panels.npz"))
    compared = 0
    for k, tag in enumerate(z["tags"]):
        d, W, y, X = (z[f"c{k}_{nm}"] for nm in "dWyX")
        key = f"c{k}_{'grid' if grid else 'brent'}"
        assert str(z[f"{key}_raised"]) == ""
        o = O.calculate(np.maximum(np.float32(0.0), d), y, W, X, grid=grid, order=order, nthreads=4)
        keep = np.ones(X.shape[1], bool)
        if not grid or order == 1:      # order 0 on the grid path agrees even on the singular designs
            keep[list(_SINGULAR_SNPS[order])] = False
            if tag in _SINGULAR_CASES[order]:
                keep[:] = False
        for col in ("beta", "se_beta", "tau", "lambda", "F_wald", "p_wald"):
            r = z[f"{key}_{col}"]
            a = np.asarray(o[col]).astype(r.dtype)
            assert (np.isnan(a) == np.isnan(r)).all(), (tag, col)           # the same rows are NaN, singular designs included
            a, r = a[keep], r[keep]
            if order == 0 and col != "p_wald":
                same = (a == r) | (np.isnan(a) & np.isnan(r))
            else:       # kernel order: float32 last-bit differences allowed, as on the regular panels; p through another betainc
                same = np.isclose(a, r, rtol=1e-8 if order == 0 else 2e-4, atol=0) | (np.isnan(a) & np.isnan(r)) | (a == r)
            assert same.all(), (tag, col, np.nonzero(~same)[0], a[~same][:3], r[~same][:3])
            compared += int(keep.sum())
    assert compared >= 15 * 11 * 6


def _tier_a_cases():
    z = np.load(os.path.join(G, "tier_a_config_sizes.npz"))
    return z, [str(t) for t in z["cases"]]


def _tier_a_inputs(z, tag):
    """Regenerates the fixture's inputs from its seed and checks them byte for byte (CRC-32) against what the reference was given."""
    import zlib
    from pygemma_amd import synth
    n, p, c, seed = (int(v) for v in z[f"{tag}_shape"])
    rp = synth.fast_rotated_panel(n, p, c, seed=seed)
    crc = [zlib.crc32(np.ascontiguousarray(rp[k]).tobytes()) for k in ("d", "X", "Y", "W")]
    assert crc == [int(v) for v in z[f"{tag}_crc"]], "synth.fast_rotated_panel no longer regenerates the fixture's inputs"
    return rp, n, p, c


@pytest.mark.parametrize("tag", ["n10000_c5", "n10000_c10", "n50000_c5"])
def test_oracle_equals_the_reference_at_the_configs_n(tag):
    """Tier A at BASELINE.json's sizes: the reference itself (eigen=False, lmm/lmm.py:164-167; caller
    experiments/large_gwas/run_pygemma.py:57-65) was run at n = 10 000 (c = 5, 10; Brent and grid) and n = 50 000 (grid); the oracle in
    the reference's summation order reproduces every row bit for bit, p within 1e-9."""
    z, _ = _tier_a_cases()
    rp, n, p, c = _tier_a_inputs(z, tag)
    for path in ("brent", "grid"):
        if f"{tag}_{path}_beta" not in z.files:
            continue
        orc = O.calculate(rp["d"], rp["Y"], rp["W"], rp["X"], grid=(path == "grid"), order=0, nthreads=8)
        for col in ("beta", "se_beta", "tau", "lambda", "F_wald"):
            ref = z[f"{tag}_{path}_{col}"]
            bad = int((bits(orc[col].astype(ref.dtype)) != bits(ref)).sum())
            assert bad == 0, (tag, path, col, bad)
        np.testing.assert_allclose(orc["p_wald"], z[f"{tag}_{path}_p_wald"], rtol=1e-9, atol=0)
