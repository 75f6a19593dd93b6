"""Model-level functions on the GPU (the secondary surface tests/test_pygemma.py:256-294 of the reference uses)
against the reference's golden outputs and, bit for bit, against the oracle in the kernels' summation order."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def bits(a):
    return np.asarray(a, np.float32).view(np.uint32)


def _same(a, b):
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    return ((bits(a) == bits(b)) | (np.isnan(a) & np.isnan(b)))


@pytest.fixture(scope="module")
def ctx():
    from pygemma_amd import _lib
    c = _lib.Context(0)
    yield c
    c.close()


def test_precompute_mat_vs_reference_and_oracle(ctx):
    """precompute_mat dicts for n in {64,200,500}, c in {1,5,10}, 7 lambdas, full in {False,True}: every level of every
    array bit-identical to the oracle (order=1); against the reference's own arrays the quantities the live path reads
    are bit-exact and the Q/R blocks agree to 2e-7 away from the ill-conditioned lambda=1e5 end (same bar as the
    oracle's own golden test)."""
    from pygemma_amd import lmm
    from oracle import oracle as O
    z = np.load(os.path.join(G, "precompute_mat.npz"))
    n_exact = n_bad = 0
    for ci, (n, c) in enumerate(z["cases"]):
        d, Wx, y = z[f"c{ci}_d"], z[f"c{ci}_Wx"], z[f"c{ci}_y"]
        ctot = Wx.shape[1]
        for li, lam in enumerate(z["lams"]):
            for full in (0, 1):
                k = f"c{ci}_l{li}_f{full}_"
                r = lmm.precompute_mat(lam, d, Wx, y, full=bool(full), ctx=ctx)
                o = O.precompute_mat(lam, d, Wx, y, full=bool(full), order=1)
                keys = ["wjt_Pi_wk", "wjt_Pi_Pi_wk", "yt_Pi_y", "yt_Pi_Pi_y", "tr_Pi"] + \
                       (["wjt_Pi_Pi_Pi_wk", "yt_Pi_Pi_Pi_y", "tr_Pi_Pi"] if full else [])
                for key in keys:
                    assert _same(r[key], o[key]).all(), (k, key)
                assert bits(r["logdet_H"]) == bits(o["logdet_H"]) and r["logdet_Wt_W"] == 0.0
                # device log() vs glibc log(): the f32 accumulator may differ in its last bit, rarely
                assert abs(r["logdet_Wt_H_inv_W"] - o["logdet_Wt_H_inv_W"]) <= 1.2e-7 * abs(o["logdet_Wt_H_inv_W"]) + 1e-30
                # against the reference
                for mine, key in ((r["wjt_Pi_wk"], "P3"), (r["yt_Pi_y"], "yPy"), (r["yt_Pi_Pi_y"], "yPPy"), (r["tr_Pi"], "trP"),
                                  ([r["logdet_H"]], "ldH")):
                    ref = np.asarray(z[k + key], np.float32).reshape(-1)
                    mine = np.asarray(mine, np.float32).reshape(-1)
                    msk = ~np.isnan(ref)
                    same = _same(mine[msk], ref[msk])
                    n_exact += int(same.sum()); n_bad += int((~same).sum())
                    np.testing.assert_allclose(mine[msk], ref[msk], rtol=3e-7)
                if lam < 1e4:
                    pairs = [(r["wjt_Pi_Pi_wk"], "Q3")] + ([(r["wjt_Pi_Pi_Pi_wk"], "R3"), (r["yt_Pi_Pi_Pi_y"], "yPPPy"),
                                                            (r["tr_Pi_Pi"], "trPP")] if full else [])
                    for mine, key in pairs:
                        ref = z[k + key]
                        msk = ~np.isnan(ref)
                        np.testing.assert_allclose(np.asarray(mine, np.float32)[msk], ref[msk], rtol=2e-7)
                # scalars at the last level: d1 and logL from the device's own forms
                if bits(r["yt_Pi_y"][ctot]) == bits(z[k + "yPy"][ctot]) and bits(r["yt_Pi_Pi_y"][ctot]) == bits(z[k + "yPPy"][ctot]) \
                        and bits(r["tr_Pi"][ctot]) == bits(z[k + "trP"][ctot]):
                    assert bits(r["_d1"]) == bits(z[k + "d1"]), k
    assert n_bad <= 0.002 * (n_exact + n_bad), (n_bad, n_exact)


def test_overload_scalars_on_reference_forms(ctx):
    """The three *_overload scalars evaluated on the REFERENCE's quadratic forms must reproduce its d1/d2/logL bits."""
    from pygemma_amd import lmm
    z = np.load(os.path.join(G, "precompute_mat.npz"))
    for ci, (n, c) in enumerate(z["cases"]):
        ctot = z[f"c{ci}_Wx"].shape[1]
        for li, lam in enumerate(z["lams"]):
            k = f"c{ci}_l{li}_f1_"
            yPy, yPPy, yPPPy = z[k + "yPy"][ctot], z[k + "yPPy"][ctot], z[k + "yPPPy"][ctot]
            trP, trPP = z[k + "trP"][ctot], z[k + "trPP"][ctot]
            d1 = lmm.likelihood_derivative1_restricted_lambda_overload(lam, int(n), ctot, yPy, yPPy, trP, ctx=ctx)
            d2 = lmm.likelihood_derivative2_restricted_lambda_overload(lam, int(n), ctot, yPy, yPPy, yPPPy, trP, trPP, ctx=ctx)
            ll = lmm.likelihood_restricted_lambda_overload(lam, int(n), ctot, yPy, z[k + "ldH"], 0.0, z[k + "ld"], ctx=ctx)
            assert bits(d1) == bits(z[k + "d1"]) and bits(d2) == bits(z[k + "d2"]), k
            # logL ends in a double log(): device libm vs glibc may differ in the last bit of the f32 result
            assert abs(float(ll) - float(z[k + "logl"])) <= 1.2e-7 * abs(float(z[k + "logl"])), k


@pytest.mark.parametrize("panel", ["panel_signal_n400_c5", "panel_signal_n257_c1", "panel_null_n320_c10"])
def test_newton_wrapper_d1_calc_lambda_vs_reference(panel, ctx):
    """wrapper d1 on the decade grid, newton from 3e^k inside each decade, calc_lambda_restricted (brent and grid),
    calc_beta_vg_ve_restricted_overload at the reference's lambda — first SNPs of the golden panels."""
    from pygemma_amd import lmm
    z = np.load(os.path.join(G, panel + ".npz"))
    d, X, Y, W = z["d"], z["X"], z["Y"], z["W"]
    d1g, nwg = z["d1_decades"], z["newton_from_3e_k"]
    ks = np.arange(-5, 6)
    for g in range(6):
        Wx = np.ascontiguousarray(np.c_[W, X[:, g]])
        for j, kk in enumerate(ks):
            lam = np.float32(10.0 ** float(kk))
            assert bits(lmm.wrapper_likelihood_derivative1_restricted_lambda(lam, d, Y, Wx, ctx=ctx)) == bits(d1g[g, j]), (g, j)
        for j, kk in enumerate(ks[:-1]):
            l0, l1 = np.float32(10.0 ** float(kk)), np.float32(10.0 ** float(kk + 1))
            got = lmm.newton(np.float32(3.0) * l0, d, Y, Wx, precompute=True, lambda_min=l0, lambda_max=l1, ctx=ctx)
            assert bits(got) == bits(nwg[g, j]), (g, j, got, nwg[g, j])
        assert lmm.calc_lambda_restricted(d, Y, Wx, ctx=ctx) == z["brent_calc_lambda"][g]
        assert lmm.calc_lambda_restricted(d, Y, Wx, grid=True, ctx=ctx) == z["grid_calc_lambda"][g]
        b, zero, se, tau = lmm.calc_beta_vg_ve_restricted_overload(d, W, X[:, g:g + 1], np.float32(z["brent_lambda"][g]), Y, ctx=ctx)
        assert zero == 0.0
        assert bits(b) == bits(z["brent_beta"][g]) and bits(se) == bits(z["brent_se_beta"][g]) and bits(tau) == bits(z["brent_tau"][g])


def test_model_functions_on_reference_own_test_matrices(ctx):
    """The function list of the reference's own test script (tests/test_pygemma.py:253-295) on its own inputs
    (generate_test_matrices: n=1000, 12 covariates, the SNP collinear with one of them): every model-level function runs
    through the GPU and reproduces the oracle bit for bit; against the reference the well-defined quantities agree and the
    cancellation-dominated ones (x pivot ~ 0) agree to the few per cent that is all that is defined there."""
    from pygemma_amd import lmm
    from oracle import oracle as O
    z = np.load(os.path.join(G, "reference_test_matrices.npz"))
    d, xr, Y, W = z["d"], z["x"], z["Y"].reshape(-1, 1), z["W"]
    Wx = np.ascontiguousarray(np.c_[W, xr])
    n, ctot = Wx.shape
    assert ctot == 13
    for li, lam in enumerate(z["lams"]):
        for full in (0, 1):
            r = lmm.precompute_mat(lam, d, Wx, Y, full=bool(full), ctx=ctx)
            o = O.precompute_mat(lam, d, Wx, Y, full=bool(full), order=1)
            for key in ["wjt_Pi_wk", "wjt_Pi_Pi_wk", "yt_Pi_y", "yt_Pi_Pi_y", "tr_Pi"] + (["wjt_Pi_Pi_Pi_wk", "yt_Pi_Pi_Pi_y", "tr_Pi_Pi"] if full else []):
                assert _same(r[key], o[key]).all(), (li, full, key)
            # levels before the collinear column enters are well conditioned: equal to the reference to float32 rounding
            k = f"l{li}_f{full}_"
            np.testing.assert_allclose(r["yt_Pi_y"][:ctot - 1], z[k + "yPy"][:ctot - 1], rtol=3e-7)
            np.testing.assert_allclose(r["tr_Pi"][:ctot - 1], z[k + "trP"][:ctot - 1], rtol=3e-7)
        v = lmm.wrapper_likelihood_derivative1_restricted_lambda(lam, d, Y, Wx, ctx=ctx)
        assert np.sign(v) == np.sign(z[f"l{li}_d1"]) and abs(float(v) - float(z[f"l{li}_d1"])) <= 5e-2 * abs(float(z[f"l{li}_d1"]))
        got = lmm.newton(lam, d, Y, Wx, precompute=True, ctx=ctx)
        ref = float(z[f"l{li}_newton"])
        assert np.isfinite(got) == np.isfinite(ref) and abs(got - ref) <= 5e-2 * abs(ref) + 1e-12, (li, got, ref)
    assert lmm.calc_lambda_restricted(d, Y, Wx, ctx=ctx) == z["calc_lambda_restricted"]
    assert lmm.calc_lambda_restricted(d, Y, Wx, grid=True, ctx=ctx) == z["calc_lambda_restricted_grid"]


def test_ml_functions_and_calc_lambda_vs_reference_fixtures():
    """N2 at the model level: lmm.likelihood_lambda / likelihood_derivative1_lambda / likelihood_derivative2_lambda (pyx:1542-1603)
    and lmm.calc_lambda (lmm/lmm.py:22-84) against what the real reference's functions returned (tests/golden/lrt_panels.npz), at the
    tolerances of the oracle's own comparison (tests/test_oracle_golden.py): the reference forms these in float32."""
    import os
    from pygemma import lmm
    from pygemma_amd import _lib
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "lrt_panels.npz"))
    lams = z["lams"]
    with _lib.Context(0) as ctx:
        for name in ["sig", "c1"]:
            d, Y, W, X = (z[f"{name}_{k}"] for k in "dYWX")
            fn = z[f"{name}_ml_functions"]
            for si, Wx in enumerate([W, np.c_[W, X[:, 0]]]):
                Wx = np.ascontiguousarray(Wx, np.float32)
                for li, lam in enumerate(lams):
                    o = (lmm.likelihood_lambda(lam, d, Y, Wx, ctx=ctx), lmm.likelihood_derivative1_lambda(lam, d, Y, Wx, ctx=ctx),
                         lmm.likelihood_derivative2_lambda(lam, d, Y, Wx, ctx=ctx))
                    if lam <= 400:
                        assert abs(o[0] - fn[si, li, 0]) <= 3 * np.spacing(np.float32(abs(fn[si, li, 0]))), (name, si, lam)
                    elif lam <= 1e3:
                        assert abs(o[0] / fn[si, li, 0] - 1) <= 2e-6, (name, si, lam)
                    if 1e-3 < lam < 1e4:
                        assert abs(o[1] / fn[si, li, 1] - 1) <= 1e-3 and abs(o[2] / fn[si, li, 2] - 1) <= 2e-3, (name, si, lam, o, fn[si, li])
            # calc_lambda: the null model and the first SNPs of the alternative
            ulp = np.spacing(np.float32(abs(float(z[f"{name}_l_null"]))))
            lam0 = lmm.calc_lambda(d, Y, W, ctx=ctx)
            assert abs(lam0 / float(z[f"{name}_lambda_null"]) - 1) <= 5e-5
            assert abs(float(lmm.likelihood_lambda(np.float32(lam0), d, Y, W, ctx=ctx)) - float(z[f"{name}_l_null"])) <= 2 * ulp
            for g in range(3):
                Wx = np.ascontiguousarray(np.c_[W, X[:, g]], np.float32)
                lam = lmm.calc_lambda(d, Y, Wx, ctx=ctx)
                assert abs(lam / z[f"{name}_lambda_alt"][g] - 1) <= 5e-5, (name, g)
                assert abs(float(lmm.likelihood_lambda(np.float32(lam), d, Y, Wx, ctx=ctx)) - float(z[f"{name}_l_alt"][g])) <= 2 * ulp
