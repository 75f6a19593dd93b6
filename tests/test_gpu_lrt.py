"""N2 on the GPU: lmm.pygemma(..., lrt=True) — the likelihood-ratio columns the reference sketches
(lmm/lmm.py:137-141, 277-300) from its ML functions (lmm/lmm.py:22-84; pyx:1542-1603) — against the oracle in kernel
order (bit-level expectations) and against fixtures made by calling the real reference (float32-noise tolerances)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint32) if a.dtype == np.float32 else a.view(np.uint64)


@pytest.mark.parametrize("name", ["sig", "weak", "null", "c1"])
def test_lrt_columns_vs_reference_and_oracle(name):
    from oracle import oracle as O
    from pygemma import lmm
    z = np.load(os.path.join(G, "lrt_panels.npz"))
    d, Y, W, X = (z[f"{name}_{k}"] for k in "dYWX")
    p = X.shape[1]
    df = lmm.pygemma(Y, X, W, d, eigen=False, lrt=True)
    assert list(df.columns) == ["beta", "se_beta", "tau", "lambda", "F_wald", "p_wald", "l_alt", "l_null", "D_lrt", "p_lrt"]
    # the default columns are untouched by the option
    base = lmm.pygemma(Y, X, W, d, eigen=False)
    for col in ("beta", "se_beta", "tau", "lambda", "F_wald", "p_wald"):
        assert (bits(df[col].to_numpy()) == bits(base[col].to_numpy())).all(), col
    # reference fixtures: float32-noise tolerances (see tests/test_oracle_golden.py)
    ulp = np.spacing(np.float32(abs(float(z[f"{name}_l_null"]))))
    assert (df["l_null"].to_numpy() == df["l_null"].to_numpy()[0]).all()
    assert abs(df["l_null"].to_numpy()[0] - float(z[f"{name}_l_null"])) <= 2 * ulp
    assert np.abs(df["l_alt"].to_numpy() - z[f"{name}_l_alt"]).max() <= 2 * ulp
    assert np.abs(df["D_lrt"].to_numpy() - z[f"{name}_D_lrt"]).max() <= 4 * ulp + 1e-6
    assert np.abs(df["p_lrt"].to_numpy() / z[f"{name}_p_lrt_sf"] - 1).max() <= 5e-3
    # oracle in the kernels' summation order: same statements on the same quadratic forms
    orc = O.calculate_lrt(d, Y, W, X, order=1, nthreads=8)
    same = bits(df["l_alt"].to_numpy().astype(np.float32)) == bits(orc["l_alt"])
    assert same.mean() >= 0.97, same.mean()                      # device log() vs glibc: last-ulp differences flip a few f32 roundings
    assert np.abs(df["l_alt"].to_numpy() - orc["l_alt"]).max() <= ulp
    assert abs(df["l_null"].to_numpy()[0] - orc["l_null"]) <= ulp
    np.testing.assert_allclose(df["p_lrt"].to_numpy(), orc["p_lrt"], rtol=2e-3)
    assert ((df["D_lrt"].to_numpy() > 0) | (df["p_lrt"].to_numpy() == 1.0)).all()


def test_lrt_grid_path_and_streamed_batches(monkeypatch):
    """lrt=True composes with grid=True and with batch streaming; Wald and LRT p-values tell the same story on a signal panel."""
    from pygemma_amd import lmm, synth
    monkeypatch.setattr(lmm, "_BATCH_SNPS", 256)
    rp = synth.rotated_panel(320, 700, 4, seed=77)
    a = lmm.pygemma(rp["Y"], rp["X"], rp["W"], rp["d"], eigen=False, lrt=True, grid=True)
    b = lmm.pygemma(rp["Y"], rp["X"], rp["W"], rp["d"], eigen=False, lrt=True)
    for col in ("l_alt", "l_null", "D_lrt", "p_lrt"):          # the ML search does not depend on the REML search's mode
        assert (bits(a[col].to_numpy()) == bits(b[col].to_numpy())).all(), col
    lw, ll = -np.log10(b["p_wald"].to_numpy()), -np.log10(b["p_lrt"].to_numpy())
    assert np.corrcoef(lw, ll)[0, 1] >= 0.99
    assert b["p_lrt"].to_numpy()[0] < 1e-3 and b["p_wald"].to_numpy()[0] < 1e-3     # the planted causal SNP


@pytest.mark.parametrize("n,p,c", [(300, 40, 12), (257, 24, 17), (320, 16, 22)])
def test_lrt_wide_covariate_counts_vs_oracle(n, p, c):
    """The LRT instantiations for slot-chunked shapes (c >= 11) and the second translation unit (c >= 16), null model included
    (c - 1 covariates + the last one as the SNP): GPU vs the oracle in kernel order."""
    from oracle import oracle as O
    from pygemma_amd import lmm, synth
    rp = synth.rotated_panel(n, p, c, seed=n + c, h2=0.4)
    df = lmm.pygemma(rp["Y"], rp["X"], rp["W"], rp["d"], eigen=False, lrt=True)
    orc = O.calculate_lrt(rp["d"], rp["Y"], rp["W"], rp["X"], order=1, nthreads=8)
    ulp = np.spacing(np.float32(abs(orc["l_null"])))
    assert abs(df["l_null"].to_numpy()[0] - orc["l_null"]) <= ulp
    assert np.abs(df["l_alt"].to_numpy() - orc["l_alt"]).max() <= ulp
    assert (bits(df["l_alt"].to_numpy().astype(np.float32)) == bits(orc["l_alt"])).mean() >= 0.9
    np.testing.assert_allclose(df["p_lrt"].to_numpy(), orc["p_lrt"], rtol=5e-3)
    base = lmm.pygemma(rp["Y"], rp["X"], rp["W"], rp["d"], eigen=False)
    for col in ("beta", "se_beta", "tau", "lambda", "F_wald", "p_wald"):
        assert (bits(df[col].to_numpy()) == bits(base[col].to_numpy())).all(), col


def test_lrt_on_degenerate_panels_same_nan_rows_as_oracle():
    """synth.degenerate_panels() with lrt=True: no exception, the oracle's NaN rows, l_alt / l_null within one float32 ulp of the
    oracle in the kernels' order wherever finite (tools/adversarial_lrt.py prints the same per case)."""
    from oracle import oracle as O
    from pygemma import lmm
    from pygemma_amd import synth
    for tag, d, W, y, X in synth.degenerate_panels():
        df = lmm.pygemma(y.reshape(-1, 1), X, W, d, eigen=False, lrt=True)
        o = O.calculate_lrt(np.maximum(np.float32(0), d), y, W, X, order=1, nthreads=4)
        for col in ("l_alt", "D_lrt", "p_lrt"):
            a, b = df[col].to_numpy().astype(np.float64), np.asarray(o[col], np.float64)
            assert (np.isnan(a) == np.isnan(b)).all(), (tag, col)
        a, b = df["l_alt"].to_numpy().astype(np.float64), np.asarray(o["l_alt"], np.float64)
        fin = np.isfinite(a) & np.isfinite(b)
        assert (np.abs(a[fin] - b[fin]) <= np.spacing(np.abs(b[fin]).astype(np.float32))).all(), tag
        assert (a[~fin & ~np.isnan(a)] == b[~fin & ~np.isnan(a)]).all(), tag           # infinities on the same rows, same sign
