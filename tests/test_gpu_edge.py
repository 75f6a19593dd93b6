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


def test_every_entry_point_rejects_bad_arguments_before_any_launch(ctx):
    """The C ABI never throws or faults on misuse ('Error behaviour'): NULL pointers, empty or negative shapes,
    leading dimensions smaller than the rows they describe and unsupported covariate counts come back as a negative code with a
    message — checked on the host, before any kernel could read out of bounds — and the context keeps working afterwards."""
    import ctypes as C
    from pygemma_amd import _lib, ops, synth
    L = _lib.load()
    n, p, c = 64, 8, 2
    buf = ctx.alloc(1 << 20)                      # one valid device buffer stands in for every non-NULL pointer argument
    b, h = buf.ptr, ctx.handle
    host = np.zeros(4096, np.float64)
    hp = host.ctypes.data
    ok_ld = 64
    bad = [
        ("pg_syevd_dev n=0", lambda: L.pg_syevd_dev(h, 0, b, b, b, None, None)),
        ("pg_syevd_dev NULL K", lambda: L.pg_syevd_dev(h, n, None, b, b, None, None)),
        ("pg_syevd_dev no outputs", lambda: L.pg_syevd_dev(h, n, b, None, b, None, None)),
        ("pg_syevd_dev n too large", lambda: L.pg_syevd_dev(h, 70000, b, b, b, None, None)),
        ("pg_rotate_dev ldX < p", lambda: L.pg_rotate_dev(h, n, p, b, n, b, p - 1, b, ok_ld)),
        ("pg_rotate_dev ldU < n", lambda: L.pg_rotate_dev(h, n, p, b, n - 1, b, p, b, ok_ld)),
        ("pg_rotate_dev ldx < n", lambda: L.pg_rotate_dev(h, n, p, b, n, b, p, b, n - 1)),
        ("pg_rotate_dev p = 0", lambda: L.pg_rotate_dev(h, n, 0, b, n, b, p, b, ok_ld)),
        ("pg_rotate_dev NULL U", lambda: L.pg_rotate_dev(h, n, p, None, n, b, p, b, ok_ld)),
        ("pg_rotate_dev NULL ctx", lambda: L.pg_rotate_dev(None, n, p, b, n, b, p, b, ok_ld)),
        ("pg_geno_prep_dev n=0", lambda: L.pg_geno_prep_dev(h, 0, b, n, b)),
        ("pg_geno_prep_dev ldU < n", lambda: L.pg_geno_prep_dev(h, n, b, n - 1, b)),
        ("pg_rotate_auto_dev ldX < p", lambda: L.pg_rotate_auto_dev(h, n, p, b, n, b, b, p - 1, b, ok_ld, b, None)),
        ("pg_rotate_auto_dev NULL work", lambda: L.pg_rotate_auto_dev(h, n, p, b, n, b, b, p, b, ok_ld, None, None)),
        ("pg_rotate_auto_i8_dev n<0", lambda: L.pg_rotate_auto_i8_dev(h, -1, p, b, b, 0, p, b, ok_ld, b, None)),
        ("pg_rotate_bed_dev ldb too small", lambda: L.pg_rotate_bed_dev(h, n, p, b, b, n // 4 - 1, 0, b, ok_ld, b)),
        ("pg_transpose_dev ldx < n", lambda: L.pg_transpose_dev(h, n, p, b, p, b, n - 1)),
        ("pg_kinship_geno_dev p=0", lambda: L.pg_kinship_geno_dev(h, n, 0, b, p, 1, b)),
        ("pg_kinship_geno_dev ldG < p", lambda: L.pg_kinship_geno_dev(h, n, p, b, p - 1, 1, b)),
        ("pg_assoc_dev n=1", lambda: L.pg_assoc_dev(h, 1, c, p, b, b, b, b, ok_ld, 0, b, b, b, b, b, b, None)),
        ("pg_assoc_dev ldx < n", lambda: L.pg_assoc_dev(h, n, c, p, b, b, b, b, n - 1, 0, b, b, b, b, b, b, None)),
        ("pg_assoc_dev p < 0", lambda: L.pg_assoc_dev(h, n, c, -3, b, b, b, b, ok_ld, 0, b, b, b, b, b, b, None)),
        ("pg_assoc_dev c = 0", lambda: L.pg_assoc_dev(h, n, 0, p, b, b, b, b, ok_ld, 0, b, b, b, b, b, b, None)),
        ("pg_assoc_dev c = 31", lambda: L.pg_assoc_dev(h, n, 31, p, b, b, b, b, ok_ld, 0, b, b, b, b, b, b, None)),
        ("pg_assoc_dev n - c - 1 = 0", lambda: L.pg_assoc_dev(h, 4, 3, p, b, b, b, b, ok_ld, 0, b, b, b, b, b, b, None)),
        ("pg_assoc_dev NULL F", lambda: L.pg_assoc_dev(h, n, c, p, b, b, b, b, ok_ld, 0, b, b, b, b, None, b, None)),
        ("pg_assoc_lrt_dev NULL D_lrt", lambda: L.pg_assoc_lrt_dev(h, n, c, p, b, b, b, b, ok_ld, 0, b, b, b, b, b, b, b, b, None, b)),
        ("pg_assoc_lrt_dev c = 31", lambda: L.pg_assoc_lrt_dev(h, n, 31, p, b, b, b, b, ok_ld, 0, b, b, b, b, b, b, b, b, b, b)),
        ("pg_fdist_sf_dev count < 0", lambda: L.pg_fdist_sf_dev(h, -1, b, 10.0, b)),
        ("pg_memcpy2d_h2d_async pitch < width", lambda: L.pg_memcpy2d_h2d_async(h, b, 16, hp, 8, 16, 4)),
        ("pg_stage_rows NULL dst", lambda: L.pg_stage_rows(None, 16, hp, 16, 16, 4, 1)),
        ("pg_host_register 0 bytes", lambda: L.pg_host_register(h, hp, 0)),
        ("pg_comm_broadcast_dev NULL comm", lambda: L.pg_comm_broadcast_dev(None, b, 16, 0)),
        ("pg_comm_init_rank rank >= nranks", lambda: L.pg_comm_init_rank(h, 2, 2, hp, C.byref(C.c_void_p()))),
        ("pg_ctx_create device out of range", lambda: L.pg_ctx_create(4096, C.byref(C.c_void_p()))),
        ("pg_zkzt_dev q = 0", lambda: L.pg_zkzt_dev(h, n, 0, b, 0, n, b, 0, n, b, n)),
        ("pg_zkzt_dev ldz < q", lambda: L.pg_zkzt_dev(h, n, 8, b, 0, 7, b, 0, 8, b, n)),
        ("pg_zkzt_dev ldo < n", lambda: L.pg_zkzt_dev(h, n, 8, b, 0, 8, b, 0, 8, b, n - 1)),
        ("pg_zkzt_dev NULL K", lambda: L.pg_zkzt_dev(h, n, 8, b, 0, 8, None, 0, 8, b, n)),
        ("pg_assoc_warm n = 1", lambda: L.pg_assoc_warm(h, 1, c)),
        ("pg_assoc_warm c = 0", lambda: L.pg_assoc_warm(h, n, 0)),
        ("pg_assoc_warm c too large", lambda: L.pg_assoc_warm(h, n, 1000)),
    ]
    for name, call in bad:
        rc = call()
        assert rc < 0, (name, rc)
        assert len(L.pg_last_error()) > 0, name
    assert L.pg_assoc_dev(h, n, c, 0, b, b, b, b, ok_ld, 0, b, b, b, b, b, b, None) == 0      # an empty block is not an error
    assert L.pg_assoc_warm(h, n, c) == 0 and L.pg_assoc_warm(h, n, c) == 0                     # set-up ahead of time, twice: no harm
    buf.free()
    # the context still computes
    rp = synth.rotated_panel(n, p, c, seed=3)
    got = ops.assoc(rp["d"], rp["W"], rp["Y"], rp["X"], ctx=ctx)
    assert np.isfinite(got["beta"]).all()
