"""GPU tests of the drop-in entry point lmm.pygemma (through the C ABI)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint32) if a.dtype == np.float32 else a.view(np.uint64)


@pytest.mark.parametrize("name", ["panel_signal_n400_c5", "panel_weak_n300_c3", "mouse_hs1940_synthG"])
@pytest.mark.parametrize("grid", [False, True])
def test_pygemma_eigen_false_matches_reference_dataframe(name, grid):
    """Tier A through the public API: schema, dtypes and all six columns vs the reference's own DataFrame."""
    from pygemma import lmm
    z = np.load(os.path.join(G, name + ".npz"))
    p = z["X"].shape[1]
    snps = [f"rs{i}" for i in range(p)]
    df = lmm.pygemma(z["Y"], z["X"], z["W"], z["d"], snps=snps, grid=grid, eigen=False, nproc=1)
    assert list(df.columns) == ["beta", "se_beta", "tau", "lambda", "F_wald", "p_wald", "SNPs"]
    assert [str(t) for t in df.dtypes] == ["float32", "float32", "float32", "float64", "float64", "float64", "object"]
    assert list(df["SNPs"]) == snps and len(df) == p
    tag = "grid" if grid else "brent"
    rowbad = np.zeros(p, bool)
    for col in ["beta", "se_beta", "tau", "lambda", "F_wald"]:
        rowbad |= bits(df[col].to_numpy()) != bits(z[f"{tag}_{col}"])
    assert rowbad.mean() <= 0.01
    np.testing.assert_allclose(df["p_wald"].to_numpy(), z[f"{tag}_p_wald"], rtol=1e-8)


def test_pygemma_float64_inputs_and_no_snps_column():
    from pygemma import lmm
    z = np.load(os.path.join(G, "panel_signal_n257_c1.npz"))
    df = lmm.pygemma(z["Y"].astype(np.float64), z["X"].astype(np.float64), z["W"].astype(np.float64), z["d"].astype(np.float64),
                     eigen=False)
    assert list(df.columns) == ["beta", "se_beta", "tau", "lambda", "F_wald", "p_wald"]
    assert (bits(df["beta"].to_numpy()) == bits(z["brent_beta"])).all()


def test_pygemma_nan_check_raises_only_when_enabled():
    from pygemma import lmm
    z = np.load(os.path.join(G, "panel_signal_n257_c1.npz"))
    X = z["X"].copy(); X[3, 5] = np.nan
    with pytest.raises(ValueError, match="NaNs present in data"):
        lmm.pygemma(z["Y"], X, z["W"], z["d"], eigen=False, disable_checks=False)
    df = lmm.pygemma(z["Y"], X, z["W"], z["d"], eigen=False)      # default: NaN row comes back, nothing raised
    assert np.isnan(df["beta"].to_numpy()[5]) and np.isfinite(df["beta"].to_numpy()[4])


@pytest.mark.parametrize("grid", [False, True])
def test_pygemma_full_pipeline_tier_c(grid):
    """Tier C (eigen=True): build and reference are each compared with an fp64 'truth' pipeline (host dsyevd +
    fp64 rotation, same lambda rule); the build must not be further from the truth than the reference's own
    float32 pipeline is, and p-values agree with the reference within 1e-3 on >= 99 % of SNPs."""
    from oracle import oracle as O
    from pygemma import lmm
    z = np.load(os.path.join(G, "eigen_true_n300.npz"))
    Y, X, W, K = z["Y"], z["X"], z["W"], z["K"]
    n, p = X.shape
    df = lmm.pygemma(Y, X, W, K, grid=grid, eigen=True, nproc=1)
    K64 = np.tril(K.astype(np.float64)); K64 = K64 + np.tril(K64, -1).T
    d, U = np.linalg.eigh(K64)
    rot = lambda A: (U.T @ A.astype(np.float64)).astype(np.float32)
    truth = O.calculate(np.maximum(d, 0).astype(np.float32), rot(Y), rot(W), rot(X), grid=grid, order=0, nthreads=4)
    tag = "grid_" if grid else "brent_"
    for col in ["beta", "se_beta", "p_wald"]:
        t = truth[col].astype(np.float64)
        eb = np.abs(df[col].to_numpy().astype(np.float64) - t) / np.abs(t)
        er = np.abs(z[tag + col].astype(np.float64) - t) / np.abs(t)
        assert np.median(eb) <= max(1.5 * np.median(er), 5e-7), (col, np.median(eb), np.median(er))
        assert np.quantile(eb, 0.99) <= max(2.0 * np.quantile(er, 0.99), 5e-6), (col, eb.max(), er.max())
    relp = np.abs(df["p_wald"].to_numpy() - z[tag + "p_wald"]) / z[tag + "p_wald"]
    assert (relp <= 1e-3).mean() >= 0.99


def test_pygemma_from_packed_bed_matches_float_input(tmp_path):
    """N4: a PLINK .bed image (2 bits per call, missing calls present) handed to lmm.pygemma gives the results of the
    float32 matrix the reference's callers would build from it (pysnptools count_A1=False + mean imputation)."""
    from pygemma_amd import lmm, synth
    from pygemma_amd.bed import PackedBed, write_bed
    rng = np.random.default_rng(11)
    n, p, c = 301, 700, 3                      # n not a multiple of 4: the last byte of every SNP record is partial
    G = rng.binomial(2, rng.uniform(0.05, 0.5, p), size=(n, p)).astype(np.float64)
    G[rng.random((n, p)) < 0.01] = np.nan
    G[:, 5] = np.where(np.isnan(G[:, 5]), 0, G[:, 5])       # a SNP without missing calls
    G[rng.random(n) < 0.6, 7] = np.nan                      # a SNP with most calls missing
    G[:, 9] = np.nan                                        # all calls missing: no called genotype to average
    prefix = str(tmp_path / "toy")
    write_bed(prefix, G)
    bed = PackedBed.open(prefix + ".bed", count_A1=False)
    assert bed.shape == (n, p) and len(bed.snps) == p
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)      # nanmean of the all-missing column
        Xf = bed.to_float()
        mu = np.nanmean(G, axis=0)
    np.testing.assert_array_equal(Xf, np.where(np.isnan(G), mu[None, :], G).astype(np.float32))
    ok = np.ones(p, bool); ok[9] = False                    # the all-missing SNP is all-NaN on the float path, constant on the device
    GK = synth.genotypes(rng, n, 2 * n)
    K = (GK @ GK.T / (2 * n)).astype(np.float32)
    W = np.concatenate([np.ones((n, 1)), rng.standard_normal((n, c - 1))], axis=1).astype(np.float32)
    y = (0.3 * np.nan_to_num(G[:, 0]) + rng.standard_normal(n)).astype(np.float32).reshape(-1, 1)
    a = lmm.pygemma(y, bed, W, K, snps=bed.snps)
    b = lmm.pygemma(y, Xf, W, K, snps=bed.snps)
    assert list(a.columns) == list(b.columns) and (a["SNPs"] == b["SNPs"]).all()
    # both sides run the same fp16x2 GEMM on the same codes; they differ only in where the imputed mean is formed (device fp64 mean
    # of the called genotypes vs numpy's nanmean rounded to float32): a few ulp of float32, nothing like a wrong divisor would give
    for col in ("beta", "se_beta", "tau", "F_wald"):
        np.testing.assert_allclose(a[col].to_numpy()[ok], b[col].to_numpy()[ok], rtol=2e-5, atol=1e-7, err_msg=col)
    np.testing.assert_allclose(a["p_wald"].to_numpy()[ok], b["p_wald"].to_numpy()[ok], rtol=1e-4, atol=1e-12)
    assert (np.abs(a["lambda"].to_numpy()[ok] / b["lambda"].to_numpy()[ok] - 1) <= 1e-3).mean() >= 0.99
    # the rotated block itself, against an fp64 rotation of the host-decoded, mean-imputed matrix (1e-6 of the column norm)
    from pygemma_amd import _lib
    L = _lib.load()
    with _lib.Context(0) as ctx:
        ev, U32, ev64, U64 = __import__("pygemma_amd.ops", fromlist=["syevd"]).syevd(K, ctx=ctx, want64=True)
        ldx = (n + 63) // 64 * 64
        dU = ctx.to_device(U32); dprep = ctx.alloc(L.pg_geno_prep_bytes(n)); dwork = ctx.alloc(L.pg_geno_work_bytes(n, p))
        _lib.check(L.pg_geno_prep_dev(ctx.handle, n, dU.ptr, n, dprep.ptr), "prep")
        dB = ctx.to_device(np.ascontiguousarray(bed.data)); dXr = ctx.alloc(p * ldx * 4)
        _lib.check(L.pg_rotate_bed_dev(ctx.handle, n, p, dprep.ptr, dB.ptr, bed.data.shape[1], 0, dXr.ptr, ldx, dwork.ptr), "rotate_bed")
        ctx.sync()
        Xr = dXr.download((p, ldx), np.float32)[:, :n]
    X64 = np.where(np.isnan(G), mu[None, :], G)
    ref = (U32.astype(np.float64).T @ X64[:, ok]).T
    err = np.abs(Xr[ok] - ref).max(axis=1) / np.linalg.norm(X64[:, ok], axis=0)
    assert err.max() <= 1e-6, err.max()
    # A1 dosage = 2 - A2 dosage: beta flips sign, the test statistic does not change
    a1 = lmm.pygemma(y, PackedBed.open(prefix, count_A1=True), W, K)
    np.testing.assert_allclose(a1["beta"].to_numpy()[ok], -a["beta"].to_numpy()[ok], rtol=2e-4, atol=1e-6)
    np.testing.assert_allclose(a1["F_wald"].to_numpy()[ok], a["F_wald"].to_numpy()[ok], rtol=5e-4, atol=1e-6)


def test_checkpoint_restart(tmp_path, monkeypatch):
    """N4: finished SNP batches are written under `checkpoint`; a rerun restores them, recomputes only what is missing and
    returns the same frame; a directory from another run is refused."""
    import os
    from pygemma_amd import lmm, synth
    monkeypatch.setattr(lmm, "_BATCH_SNPS", 512)
    rp = synth.rotated_panel(200, 1500, 2, seed=4)
    d, X, Y, W = rp["d"], rp["X"], rp["Y"], rp["W"]
    ck = str(tmp_path / "ck")
    ref = lmm.pygemma(Y, X, W, d, eigen=False)
    a = lmm.pygemma(Y, X, W, d, eigen=False, checkpoint=ck)
    parts = sorted(f for f in os.listdir(ck) if f.startswith("part_"))
    assert len(parts) == 3 and os.path.exists(os.path.join(ck, "manifest.json"))
    os.remove(os.path.join(ck, parts[1]))
    # poison a surviving part: a restart must take it from disk, not recompute it
    z = dict(np.load(os.path.join(ck, parts[0])))
    z["beta"] = z["beta"] + 1.0
    np.savez(os.path.join(ck, parts[0]), **z)
    b = lmm.pygemma(Y, X, W, d, eigen=False, checkpoint=ck)
    for col in ("se_beta", "tau", "lambda", "F_wald", "p_wald"):
        assert (a[col].to_numpy() == ref[col].to_numpy()).all() and (b[col].to_numpy() == ref[col].to_numpy()).all()
    assert (a["beta"].to_numpy() == ref["beta"].to_numpy()).all()
    assert (b["beta"].to_numpy()[:512] == ref["beta"].to_numpy()[:512] + 1.0).all()
    assert (b["beta"].to_numpy()[512:] == ref["beta"].to_numpy()[512:]).all()
    with pytest.raises(ValueError):
        lmm.pygemma(Y, X[:, :700], W, d, eigen=False, checkpoint=ck)
    # another genotype matrix of the same shape, or other covariates with the same c, must be refused, not mixed in
    X2 = X.copy(); X2[:, 100] += 1.0
    with pytest.raises(ValueError, match="different genotypes"):
        lmm.pygemma(Y, X2, W, d, eigen=False, checkpoint=ck)
    with pytest.raises(ValueError, match="manifest"):
        lmm.pygemma(Y, X, W * np.float32(1.5), d, eigen=False, checkpoint=ck)


@pytest.mark.parametrize("dtype", [np.int8, np.uint8, np.float64])
def test_pygemma_int8_genotypes_match_float_input(dtype):
    """8-bit and float64 genotype matrices go to the device as they are (the reference would cast them to float32 first, lmm.py:121-122):
    same results as the float32 call; a block that is not genotype-valued (0..5 dosages) takes the cast + fp32 path."""
    from pygemma_amd import lmm, synth
    rng = np.random.default_rng(21)
    n, p, c = 257, 600, 2
    G = rng.binomial(2, rng.uniform(0.05, 0.5, p), size=(n, p)).astype(dtype)
    GK = synth.genotypes(rng, n, 2 * n)
    K = (GK @ GK.T / (2 * n)).astype(np.float32)
    W = np.concatenate([np.ones((n, 1)), rng.standard_normal((n, c - 1))], axis=1).astype(np.float32)
    y = (0.3 * G[:, 0] + rng.standard_normal(n)).astype(np.float32).reshape(-1, 1)
    a = lmm.pygemma(y, G, W, K)
    b = lmm.pygemma(y, G.astype(np.float32), W, K)
    for col in ("beta", "se_beta", "tau", "lambda", "F_wald", "p_wald"):
        assert (a[col].to_numpy() == b[col].to_numpy()).all(), col     # same codes, same planes, same kernel: identical
    G2 = rng.integers(0, 6, size=(n, p)).astype(dtype)                   # not genotype-valued
    if dtype == np.float64:
        G2 = G2 + rng.uniform(-0.3, 0.3, G2.shape)                        # dosages with more digits than float32 holds
        G2[7, 3] = np.nan                                                 # and a NaN: that block takes the cast + fp32 kernel
    a = lmm.pygemma(y, G2, W, K)
    b = lmm.pygemma(y, G2.astype(np.float32), W, K)
    for col in ("beta", "se_beta", "tau", "lambda", "F_wald", "p_wald"):
        x, z = a[col].to_numpy(), b[col].to_numpy()
        assert ((x == z) | (np.isnan(x) & np.isnan(z))).all(), col


@pytest.mark.parametrize("tag", ["imp", "dos"])
def test_pygemma_tier_c_imputed_and_dosage_inputs(tag):
    """Tier C on the two other input classes of the fp16 rotation, against runs of the real reference: raw 0/1/2 calls with
    mean-imputed missing entries (genotype path + indicator pass) and continuous dosages (X split into two fp16 planes).
    Same bar as above: not further from an fp64 truth pipeline than the reference's own float32 pipeline is."""
    from oracle import oracle as O
    from pygemma import lmm
    from pygemma_amd import synth
    z = np.load(os.path.join(G, "eigen_true_imputed_n600.npz"))
    X, W, Y = z["X_" + tag], z["W"], z["Y"]
    n, p = X.shape
    rng = np.random.default_rng(int(z["seed"]))
    GK = synth.genotypes(rng, n, 2 * n)
    K = (GK @ GK.T / (2 * n)).astype(np.float32)
    assert float(K.astype(np.float64).sum()) == float(z["K_sum"])          # the generator's K, regenerated
    df = lmm.pygemma(Y, X, W, K, eigen=True, nproc=1)
    K64 = np.tril(K.astype(np.float64)); K64 = K64 + np.tril(K64, -1).T
    d, U = np.linalg.eigh(K64)
    rot = lambda A: (U.T @ A.astype(np.float64)).astype(np.float32)
    truth = O.calculate(np.maximum(d, 0).astype(np.float32), rot(Y), rot(W), rot(X), grid=False, order=0, nthreads=4)
    for col in ["beta", "se_beta", "p_wald"]:
        t = truth[col].astype(np.float64)
        eb = np.abs(df[col].to_numpy().astype(np.float64) - t) / np.abs(t)
        er = np.abs(z[f"{tag}_{col}"].astype(np.float64) - t) / np.abs(t)
        assert np.median(eb) <= max(1.5 * np.median(er), 5e-7), (col, np.median(eb), np.median(er))
        assert np.quantile(eb, 0.99) <= max(2.0 * np.quantile(er, 0.99), 5e-6), (col, eb.max(), er.max())
    relp = np.abs(df["p_wald"].to_numpy() - z[f"{tag}_p_wald"]) / z[f"{tag}_p_wald"]
    assert (relp <= 1e-3).mean() >= 0.99


@pytest.mark.parametrize("n,rotate,grid", [(2000, "auto", False), (2000, "auto", True), (2000, "fp32", False),
                                           (10000, "auto", False), (10000, "fp32", False), (10000, "auto", True)])
def test_pygemma_tier_c_at_config_sizes(n, rotate, grid, monkeypatch):
    """Tier C (eigen=True) at the configs' n: the REAL reference's six columns on 256 SNPs at n = 2 000 and n = 10 000
    (tests/golden/eigen_true_exact_n*.npz, made by make_golden.py eigen_big: float32 ssyevr + sgemm inside the reference) against
    the whole device pipeline — two-stage fp64 eigensolver, fp16x2 genotype rotation (or the fp32-MFMA rotation forced), association.
    X and K regenerate bit for bit from the seed (synth.exact_panel: integer-valued float32 products); Y, W and an fp64 'truth'
    pipeline (host dsyevd + fp64 rotation + oracle) come from the fixture.  The build must not be further from the truth than the
    reference's own float32 pipeline.  Agreement with the reference itself: within 1e-3 in p on >= 99 % of the SNPs at n = 2 000; at
    n = 10 000 the reference's float32 eigendecomposition alone moves its p-values by ~1e-2 from the fp64 truth (measured on this
    fixture), so there the bound per SNP is 1e-3 + 3 x the reference's own distance from the truth."""
    from pygemma import lmm
    from pygemma_amd import synth
    z = np.load(os.path.join(G, f"eigen_true_exact_n{n}.npz"))
    p, c = int(z["p"]), int(z["c"])
    ex = synth.exact_panel(n, p, c, seed=int(z["seed"]))
    X, K = ex["X"], ex["K"]
    del ex
    assert float(K.astype(np.float64).sum()) == float(z["K_sum"])          # the generator's K, bit for bit
    if rotate == "fp32":
        monkeypatch.setenv("PYGEMMA_ROTATE", "fp32")
    df = lmm.pygemma(z["Y"], X, z["W"], K, grid=grid, eigen=True, nproc=1)
    tag = "grid_" if grid else "brent_"
    for col in ["beta", "se_beta", "p_wald"]:
        t = z["truth_" + tag + col].astype(np.float64)
        eb = np.abs(df[col].to_numpy().astype(np.float64) - t) / np.abs(t)
        er = np.abs(z[tag + col].astype(np.float64) - t) / np.abs(t)
        assert np.median(eb) <= max(1.5 * np.median(er), 5e-7), (col, np.median(eb), np.median(er))
        assert np.quantile(eb, 0.99) <= max(2.0 * np.quantile(er, 0.99), 5e-6), (col, eb.max(), er.max())
    pt, pr, pb_ = z["truth_" + tag + "p_wald"].astype(np.float64), z[tag + "p_wald"].astype(np.float64), df["p_wald"].to_numpy()
    relp = np.abs(pb_ - pr) / pr
    ref_err = np.abs(pr - pt) / pt
    bound = 1e-3 + (3.0 * ref_err if n > 2000 else 0.0)
    assert (relp <= bound).mean() >= 0.99, (float((relp <= bound).mean()), float(np.median(relp)), float(np.median(ref_err)))
    assert np.median(np.abs(pb_ - pt) / pt) <= 1e-3, float(np.median(np.abs(pb_ - pt) / pt))     # the build itself stays at the 1e-3 level
    assert (df["lambda"].to_numpy() > 0).all()
    print(f"tier C n={n} rotate={rotate} grid={grid}: median |dp|/p build-vs-truth {np.median(np.abs(pb_ - pt) / pt):.2e}, "
          f"reference-vs-truth {np.median(ref_err):.2e}, build-vs-reference {np.median(relp):.2e}")


def test_pygemma_float64_K_rounded_on_device():
    """K handed over as float64 (numpy's default) gives exactly the results of K.astype(float32) (lmm/lmm.py:127-128)."""
    from pygemma_amd import lmm, synth
    rng = np.random.default_rng(8)
    n, p = 200, 90
    GK = synth.genotypes(rng, n, 2 * n, np.float64)
    K64 = GK @ GK.T / (2 * n)
    X = synth.genotypes(rng, n, p)
    W = np.ones((n, 1), np.float32)
    y = rng.standard_normal((n, 1)).astype(np.float32)
    a = lmm.pygemma(y, X, W, K64)
    b = lmm.pygemma(y, X, W, K64.astype(np.float32))
    for col in ("beta", "se_beta", "tau", "lambda", "F_wald", "p_wald"):
        assert (a[col].to_numpy() == b[col].to_numpy()).all(), col


@pytest.mark.parametrize("kind", ["rank_deficient", "duplicated_samples", "block_diagonal", "scaled_1e-6", "zero"])
def test_pygemma_full_pipeline_degenerate_relatedness_matrices(kind):
    """Tier C on relatedness matrices with degenerate spectra (odd n: the padded eigensolver path): eigenspaces of repeated
    eigenvalues have no unique basis, the statistics do not depend on it — they must agree with an fp64 pipeline (numpy eigh +
    fp64 rotation + the oracle in the reference's order) far inside one standard error (tools/robust_K.py: the same with a report)."""
    from oracle import oracle as O
    from pygemma import lmm
    rng = np.random.default_rng(5)
    n, p, c = 401, 48, 3
    G = rng.binomial(2, 0.3, size=(n, 4 * n)).astype(np.float64)
    G = (G - G.mean(0)) / np.maximum(G.std(0), 1e-9)
    if kind == "rank_deficient":
        K = G[:, : n // 4] @ G[:, : n // 4].T / (n // 4)
    elif kind == "duplicated_samples":
        G[n // 2:] = G[: n - n // 2]
        K = G @ G.T / G.shape[1]
    elif kind == "block_diagonal":
        K = np.zeros((n, n)); s = 0
        while s < n:
            b = min(int(rng.integers(2, 30)), n - s)
            A = rng.standard_normal((b, 3 * b)); K[s:s + b, s:s + b] = A @ A.T / (3 * b); s += b
    elif kind == "scaled_1e-6":
        K = G @ G.T / G.shape[1] * 1e-6
    else:
        K = np.zeros((n, n))
    K32 = K.astype(np.float32)
    K64 = np.tril(K32.astype(np.float64)); K64 = K64 + np.tril(K64, -1).T
    d, U = np.linalg.eigh(K64)
    X = rng.binomial(2, 0.25, size=(n, p)).astype(np.float32)
    W = np.concatenate([np.ones((n, 1)), rng.standard_normal((n, c - 1))], axis=1).astype(np.float32)
    g = U @ (np.sqrt(np.maximum(d, 0) / max(np.abs(d).max(), 1e-300)) * rng.standard_normal(n))
    y = (0.4 * X[:, 0] + 0.7 * g + 0.7 * rng.standard_normal(n)).astype(np.float32).reshape(-1, 1)
    rot = lambda A: (U.T @ A.astype(np.float64)).astype(np.float32)
    for grid in (False, True):
        df = lmm.pygemma(y, X, W, K32, grid=grid)
        tr = O.calculate(np.maximum(d, 0).astype(np.float32), rot(y), rot(W), rot(X), grid=grid, order=0, nthreads=4)
        b, t, se = df["beta"].to_numpy().astype(np.float64), tr["beta"].astype(np.float64), tr["se_beta"].astype(np.float64)
        assert np.isfinite(b).all() and np.isfinite(t).all()
        assert (np.abs(b - t) / se).max() < 1e-3, (kind, grid, (np.abs(b - t) / se).max())
        np.testing.assert_allclose(df["p_wald"].to_numpy(), tr["p_wald"], rtol=2e-3)


@pytest.mark.parametrize("zdt,kdt", [(np.float32, np.float32), (np.float64, np.float32), (np.float32, np.float64)])
def test_design_matrix_Z_on_the_device(zdt, kdt):
    """lmm/lmm.py:124-125: K <- Z K Z' with a NON-SQUARE Z (n = 500 samples, q = 300 random-effect levels) on the device
    (pg_zkzt_dev: two fp64-MFMA products, one rounding): within one float32 rounding of the float64 product — and
    lmm.pygemma(..., Z=Z) is, bit for bit, lmm.pygemma on that matrix."""
    import ctypes as C
    from pygemma_amd import _lib, lmm, synth
    n, q, p, c = 500, 300, 40, 3
    rng = np.random.default_rng(11)
    Kq = synth.panel(q, 4, 1, seed=5)["K"].astype(kdt)
    Z = np.zeros((n, q), zdt)
    Z[np.arange(n), rng.integers(0, q, n)] = 1.0                 # incidence matrix: sample i belongs to level z_i
    Z += (0.05 * rng.standard_normal((n, q))).astype(zdt)        # ... made dense so that every product term counts
    L = _lib.load()
    got = lmm._zkzt(L, Z, Kq)
    exact = Z.astype(np.float64) @ Kq.astype(np.float64) @ Z.astype(np.float64).T
    assert got.dtype == np.float32 and got.shape == (n, n)
    assert np.abs(got.astype(np.float64) - exact).max() <= 2.0 ** -23 * np.abs(exact).max()
    assert (np.abs(got.astype(np.float64) - exact) <= 2.0 ** -23 * np.abs(exact) + 1e-30).mean() >= 0.999
    raw = synth.panel(n, p, c, seed=21)
    a = lmm.pygemma(raw["Y"], raw["X"], raw["W"], Kq, Z=Z)
    b = lmm.pygemma(raw["Y"], raw["X"], raw["W"], got)
    for col in ("beta", "se_beta", "tau", "lambda", "F_wald", "p_wald"):
        assert (a[col].to_numpy() == b[col].to_numpy()).all(), col
    with pytest.raises(ValueError):
        lmm.pygemma(raw["Y"], raw["X"], raw["W"], Kq, Z=Z[:, :-1])
