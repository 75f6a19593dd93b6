"""GPU edge cases of the per-SNP operator: every supported covariate count, tiny / ragged n, single SNP, empty block,
unsupported c, NaN rows — GPU must equal the oracle (kernel summation order) bit-for-bit wherever results are finite."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint32) if a.dtype == np.float32 else a.view(np.uint64)


@pytest.fixture(scope="module")
def ctx():
    from pygemma_amd import _lib
    c = _lib.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("n,p,c", [(40, 5, 1), (65, 70, 2), (130, 33, 4), (96, 20, 6), (150, 24, 7), (128, 16, 8), (200, 12, 9),
                                   (90, 9, 11), (257, 10, 12), (64, 1, 3), (1000, 7, 5), (300, 6, 10), (120, 5, 13), (333, 9, 14),
                                   (200, 4, 15), (129, 3, 16), (500, 6, 17), (260, 5, 18), (192, 2, 19), (400, 7, 20),
                                   # the reference's covariate benchmark runs PCS+1 in {1,6,11,16,21,26} (experiments/animal_gwas/benchmark_pygemma.py:238-255)
                                   (300, 5, 21), (222, 3, 23), (450, 6, 26), (193, 2, 29), (384, 4, 30)])
@pytest.mark.parametrize("grid", [False, True])
def test_all_covariate_counts_and_ragged_shapes(n, p, c, grid, ctx):
    from oracle import oracle as O
    from pygemma_amd import ops, synth
    rp = synth.rotated_panel(n, p, c, seed=n * 31 + c, null=(c % 2 == 0), h2=0.3)
    got = ops.assoc(rp["d"], rp["W"], rp["Y"], rp["X"], grid=grid, ctx=ctx)
    orc = O.calculate(rp["d"], rp["Y"], rp["W"], rp["X"], grid=grid, order=1, nthreads=4)
    for col in ["beta", "se_beta", "tau", "lambda", "F_wald"]:
        a, b = got[col], orc[col].astype(got[col].dtype)
        assert (bits(a) == bits(b)).all(), (col, np.nonzero(bits(a) != bits(b))[0][:5])
    np.testing.assert_allclose(got["p_wald"], orc["p_wald"], rtol=1e-9)


def test_empty_block_and_bad_arguments(ctx):
    import ctypes as C
    from pygemma_amd import _lib, ops, synth
    L = _lib.load()
    rp = synth.rotated_panel(64, 4, 2, seed=3)
    # p = 0: nothing to do, success
    z = np.zeros(1, np.float32)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    d, W, y = rp["d"], np.ascontiguousarray(rp["W"]), np.ascontiguousarray(rp["Y"].reshape(-1))
    assert L.pg_assoc(ctx.handle, 64, 2, 0, vp(d), vp(W), vp(y), vp(z), 0, vp(z), vp(z), vp(z), vp(z), vp(z), vp(z), None) == 0
    # unsupported number of covariates: loud, with a message, no crash
    with pytest.raises(_lib.PgError, match="covariates not supported"):
        ops.assoc(rp["d"], np.ones((64, 31), np.float32), rp["Y"], rp["X"], ctx=ctx)
    # NULL pointer
    assert L.pg_assoc(ctx.handle, 64, 2, 4, None, vp(W), vp(y), vp(z), 0, vp(z), vp(z), vp(z), vp(z), vp(z), vp(z), None) < 0
    assert b"NULL" in L.pg_last_error()


def test_nan_snp_gives_nan_row_not_an_error(ctx):
    """lmm/lmm.py:484-493 semantics: non-finite per-SNP results come back as rows, the scan goes on."""
    from oracle import oracle as O
    from pygemma_amd import ops, synth
    rp = synth.rotated_panel(128, 6, 2, seed=9)
    X = rp["X"].copy()
    X[:, 2] = np.nan          # a NaN SNP
    X[:, 4] = 0.0             # a monomorphic (all-zero after centring) SNP
    got = ops.assoc(rp["d"], rp["W"], rp["Y"], X, ctx=ctx)
    orc = O.calculate(rp["d"], rp["Y"], rp["W"], X, grid=False, order=1, nthreads=2)
    assert np.isnan(got["beta"][2]) and np.isnan(orc["beta"][2])
    ok = [0, 1, 3, 5]
    assert (bits(got["beta"][ok]) == bits(orc["beta"][ok])).all()
    for col in ["beta", "se_beta", "tau", "F_wald"]:       # the degenerate SNP: same non-finite pattern and same bits
        a, b = got[col][4:5], orc[col][4:5].astype(got[col].dtype)
        assert (np.isnan(a) == np.isnan(b)).all() and (bits(a)[~np.isnan(a)] == bits(b)[~np.isnan(b)]).all(), col


def test_reference_test_matrices_collinear_snp_c12(ctx):
    """The reference's own test inputs (tests/test_pygemma.py:195-212): c = 12 covariates one of which is the SNP itself.
    Pivot clamps decide the row; the GPU must still agree bit-for-bit with the oracle in kernel order, select the
    reference's lambda, and land within the noise band of the reference's (cancellation-dominated) beta."""
    import os
    from oracle import oracle as O
    from pygemma_amd import ops
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_test_matrices.npz"))
    X = np.ascontiguousarray(np.repeat(z["x"].reshape(-1, 1), 5, axis=1))
    got = ops.assoc(z["d"], z["W"], z["Y"], X, ctx=ctx)
    orc = O.calculate(z["d"], z["Y"], z["W"], X, grid=False, order=1, nthreads=2)
    for col in ["beta", "se_beta", "tau", "lambda", "F_wald"]:
        a, b = got[col], orc[col].astype(got[col].dtype)
        assert (bits(a) == bits(b)).all(), col
    assert got["lambda"][0] == z["df_lambda"][0]
    assert abs(got["beta"][0] - z["df_beta"][0]) <= 5e-2 * abs(z["df_beta"][0])
    assert abs(got["tau"][0] - z["df_tau"][0]) <= 1e-4 * abs(z["df_tau"][0])


@pytest.mark.parametrize("grid", [False, True])
def test_degenerate_inputs_same_bits_and_same_nans_as_oracle(grid, ctx):
    """Nothing raises, hangs or differs on degenerate inputs: every column equals the oracle (kernel order) bit for bit, NaN where
    the oracle has NaN (tools/adversarial_assoc.py is the same sweep with a report per case)."""
    from oracle import oracle as O
    from pygemma_amd import ops, synth
    for tag, d, W, y, X in synth.degenerate_panels():
        got = ops.assoc(d, W, y, X, grid=grid, ctx=ctx)
        orc = O.calculate(d, y, W, X, grid=grid, order=1, nthreads=4)
        for col in ("beta", "se_beta", "tau", "lambda", "F_wald"):
            a, b = np.asarray(got[col]), np.asarray(orc[col])
            a = a.astype(b.dtype)
            same = (a == b) | (np.isnan(a) & np.isnan(b))
            assert same.all(), (tag, col, np.nonzero(~same)[0][:5], a[~same][:3], b[~same][:3])
        a, b = got["p_wald"], orc["p_wald"]
        assert (np.isclose(a, b, rtol=1e-8, atol=0) | (np.isnan(a) & np.isnan(b))).all(), (tag, "p_wald")


@pytest.mark.parametrize("grid", [False, True])
def test_degenerate_panels_through_lmm_vs_reference_fixture(grid):
    """The same panels through lmm.pygemma(eigen=False) against what the REAL reference returned for them
    (tests/golden/degenerate_panels.npz): negative eigenvalues clamped like lmm/lmm.py:166-167, the same rows NaN, no exception,
    and on the well-posed rows the reference's numbers (float32 last-bit tolerance of the kernels' summation order)."""
    import os
    from pygemma import lmm
    from pygemma_amd.synth import DEGENERATE_SINGULAR_CASES as _SINGULAR_CASES, DEGENERATE_SINGULAR_SNPS as _SINGULAR_SNPS
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "degenerate_panels.npz"))
    for k, tag in enumerate(z["tags"]):
        d, W, y, X = (z[f"c{k}_{nm}"] for nm in "dWyX")
        key = f"c{k}_{'grid' if grid else 'brent'}"
        df = lmm.pygemma(y.reshape(-1, 1), X, W, d, eigen=False, grid=grid)
        keep = np.ones(X.shape[1], bool)
        keep[list(_SINGULAR_SNPS[1])] = False
        if tag in _SINGULAR_CASES[1]:
            keep[:] = False
        for col in ("beta", "se_beta", "tau", "lambda", "F_wald", "p_wald"):
            r, a = z[f"{key}_{col}"].astype(np.float64), df[col].to_numpy().astype(np.float64)
            assert (np.isnan(a) == np.isnan(r)).all(), (tag, col)
            a, r = a[keep], r[keep]
            same = np.isclose(a, r, rtol=2e-4, atol=0) | (np.isnan(a) & np.isnan(r)) | (a == r)
            assert same.all(), (tag, col, np.nonzero(~same)[0], a[~same][:3], r[~same][:3])
