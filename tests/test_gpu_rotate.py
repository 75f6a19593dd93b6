"""GPU parity of the rotation GEMM (pg_rotate_dev): bit-exact vs the oracle's k-ordered f32 fma chain,
and within f32-GEMM tolerance of float64 U.T @ X (the reference's sgemm is in the same error class)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from pygemma_amd import _lib
    c = _lib.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("n,p", [(64, 32), (257, 130), (300, 128), (1000, 516), (1940, 48)])
def test_rotate_bit_exact_vs_oracle(n, p, ctx):
    from oracle import oracle as O
    from pygemma_amd import ops
    rng = np.random.default_rng(n * 7 + p)
    # asymmetric U (not orthogonal on purpose: a transposed or row/col-swapped kernel cannot pass)
    U = rng.standard_normal((n, n)).astype(np.float32)
    X = rng.standard_normal((n, p)).astype(np.float32)
    got = ops.rotate(U, X, ctx=ctx)
    ref = O.rotate(U, X)
    assert got.shape == ref.shape
    assert (got.view(np.uint32) == ref.view(np.uint32)).all(), np.abs(got - ref).max()
    exact = (U.astype(np.float64).T @ X.astype(np.float64)).T
    scale = np.sqrt(n)
    assert np.abs(got[:, :n] - exact).max() <= 2e-6 * scale * 8
    assert (got[:, n:] == 0).all()


def _geno(rng, n, p, standardise):
    maf = rng.uniform(0.05, 0.5, size=p)
    G = rng.binomial(2, maf, size=(n, p)).astype(np.float64)
    if standardise:
        sd = G.std(axis=0); sd[sd == 0] = 1.0
        G = (G - G.mean(axis=0)) / sd
    return G.astype(np.float32)


def _kernel_env(monkeypatch, kern):
    """the three kernels a genotype block can take: int8 digit planes (shipped), fp16 x 2 on 16 x 16 x 32 (PG_GENO_I8=0: the kernel of
    split-plane blocks, and the r2-r3 genotype kernel), its 32 x 32 x 16 instantiation (PG_GENO_MFMA=32: measured alternative, slower)"""
    monkeypatch.setenv("PG_GENO_I8", "1" if kern == "i8" else "0")
    monkeypatch.setenv("PG_GENO_MFMA", "32" if kern == "f16_32x32" else "16")


@pytest.mark.parametrize("kern", ["i8", "f16", "f16_32x32"])
@pytest.mark.parametrize("n,p,std", [(64, 32, True), (257, 130, True), (300, 128, False), (1000, 516, True), (2000, 200, True)])
def test_rotate_genotype_fast_path(n, p, std, kern, ctx, monkeypatch):
    """genotype path: error vs the fp64 rotation within the fp32-GEMM class (and no worse than the fp32-MFMA path), on each of its kernels."""
    from pygemma_amd import ops
    _kernel_env(monkeypatch, kern)
    rng = np.random.default_rng(n + p)
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    U = Q.astype(np.float32)
    X = _geno(rng, n, p, std)
    X[:, 0] = X[0, 0]                       # a monomorphic column
    got, ok = ops.rotate_geno(U, X, ctx=ctx)
    assert ok
    exact = (U.astype(np.float64).T @ X.astype(np.float64)).T
    f32p = ops.rotate(U, X, ctx=ctx)[:, :n]
    bound = (np.abs(X.astype(np.float64)).T @ np.abs(U.astype(np.float64)))      # sum_i |x_i||u_ik|
    err_g = np.abs(got[:, :n] - exact) / bound
    err_f = np.abs(f32p - exact) / bound
    assert err_g.max() <= 4 * 2.0 ** -24 * np.sqrt(n) and np.median(err_g) <= 2 * max(np.median(err_f), 1e-9)
    assert (got[:, n:] == 0).all()


@pytest.mark.parametrize("kind", ["blocks", "twins", "graded", "identity"])
def test_rotate_int8_planes_with_localised_eigenvectors(kind, ctx, monkeypatch):
    """The int8 kernel holds U as a 24-bit fixed point per EIGENVECTOR (scale = that column's largest entry), so entries far below the
    column's maximum are coarser than in float32.  Eigenvectors of structured K are like that: block-diagonal K (families: each vector
    lives on one block), duplicated samples (twins: (1, -1)/sqrt 2 and zeros), entries graded over many decades, and U = I.  The error
    against the fp64 rotation must stay within the representation's own worst case, and in norm per SNP row no worse than 3x the
    fp32-MFMA kernel (the reference's sgemm class)."""
    from pygemma_amd import ops
    monkeypatch.setenv("PG_GENO_I8", "1")
    rng = np.random.default_rng(11)
    n, p = 900, 300
    if kind == "blocks":
        U = np.zeros((n, n))
        at = 0
        for b in [2, 3, 5, 40, 150, 700]:
            U[at:at + b, at:at + b] = np.linalg.qr(rng.standard_normal((b, b)))[0]; at += b
        U = U[rng.permutation(n)][:, rng.permutation(n)]
    elif kind == "twins":
        U = np.linalg.qr(rng.standard_normal((n, n)))[0]
        U[:, 0] = 0; U[3, 0] = 2 ** -0.5; U[8, 0] = -2 ** -0.5
        U[:, 1] = 1e-5 * rng.standard_normal(n); U[5, 1] = 1.0            # one dominant entry over a floor 5 decades down
    elif kind == "graded":
        U = np.linalg.qr(rng.standard_normal((n, n)))[0] * (10.0 ** rng.uniform(-6, 0, (n, 1)))     # sample rows graded over 6 decades
    else:
        U = np.eye(n)
    U = U.astype(np.float32)
    X = _geno(rng, n, p, kind != "identity")
    got, ok = ops.rotate_geno(U, X, ctx=ctx)
    assert ok == 1
    exact = (U.astype(np.float64).T @ X.astype(np.float64)).T
    f32p = ops.rotate(U, X, ctx=ctx)[:, :n]
    A = np.abs(U.astype(np.float64))
    worst = np.abs(X.astype(np.float64)).sum(0)[:, None] * A.max(0)[None, :]               # sum_i |x_i| max_i |u_ik|
    enc = 16 * 2.0 ** -24 * np.abs(X).max(0).astype(np.float64)[:, None] * A.sum(0)[None, :]   # x = v0 + dx * code holds to 8 ulp of the column's largest value
    err = np.abs(got[:, :n] - exact)
    assert (err <= 2.0 ** -23 * worst + 2.0 ** -23 * np.abs(exact) + enc).all()            # the representation's worst case + the output rounding
    rown = lambda a: np.sqrt((a ** 2).sum(1))
    # measured: 0.3 - 0.6x the fp32-MFMA kernel's error on blocks / twins, 1.3 - 1.5x on the graded rows (6 decades inside every vector)
    assert (rown(err) <= 3 * rown(f32p - exact) + 2.0 ** -24 * rown(exact)).all()
    print(kind, "row-norm error / fp32-MFMA kernel's:", float(np.median(rown(err) / np.maximum(rown(f32p - exact), 1e-300))))
    assert (got[:, n:] == 0).all()
    if kind == "identity":
        assert (got[:, :n] == X.T).all()


def test_rotate_int8_planes_nonfinite_eigenvector(ctx, monkeypatch):
    """A NaN or an Inf inside eigenvector k makes output column k NaN (what 0 * NaN gives the float kernels) and leaves the other
    columns alone: the digit planes are scaled per eigenvector."""
    from pygemma_amd import ops
    monkeypatch.setenv("PG_GENO_I8", "1")
    rng = np.random.default_rng(5)
    n, p = 300, 70
    U = np.linalg.qr(rng.standard_normal((n, n)))[0].astype(np.float32)
    U[17, 3] = np.nan; U[200, 9] = np.inf
    X = _geno(rng, n, p, True)
    got, ok = ops.rotate_geno(U, X, ctx=ctx)
    assert ok == 1
    bad = np.zeros(n, bool); bad[[3, 9]] = True
    assert np.isnan(got[:, :n][:, bad]).all()
    Uc = U.astype(np.float64)[:, ~bad]
    exact = (Uc.T @ X.astype(np.float64)).T
    bound = np.abs(X.astype(np.float64)).T @ np.abs(Uc)
    assert (np.abs(got[:, :n][:, ~bad] - exact) <= 4 * 2.0 ** -24 * np.sqrt(n) * bound).all()


def test_rotate_non_genotype_block_split_path_and_nan_rejection(ctx):
    """A finite block that is not genotype-valued (imputed dosages) goes through the same fp16 GEMM with X split in two
    fp16 planes (is_geno = 2), within the fp32-GEMM error class; a block with a NaN is left to the fp32 kernel."""
    from pygemma_amd import ops
    rng = np.random.default_rng(3)
    n, p = 500, 300
    U = np.linalg.qr(rng.standard_normal((n, n)))[0].astype(np.float32)
    X = _geno(rng, n, p, True)
    X[5, 7] += 0.125; X[9, 7] -= 0.0625     # a fourth AND a fifth value in one column
    X[:, 11] = rng.uniform(0, 2, n)          # a dosage column
    X[:, 12] = rng.standard_normal(n) * 1e-20; X[:, 13] = rng.standard_normal(n) * 1e20   # extreme scales
    got, ok = ops.rotate_geno(U, X, ctx=ctx)
    assert ok == 2
    exact = (U.astype(np.float64).T @ X.astype(np.float64)).T
    f32p = ops.rotate(U, X, ctx=ctx)[:, :n]
    bound = (np.abs(X.astype(np.float64)).T @ np.abs(U.astype(np.float64)))
    err_g = np.abs(got[:, :n] - exact) / bound
    err_f = np.abs(f32p - exact) / bound
    assert err_g.max() <= 4 * 2.0 ** -24 * np.sqrt(n) and np.median(err_g) <= 2 * max(np.median(err_f), 1e-9)
    assert (got[:, n:] == 0).all()
    X = _geno(rng, n, p, True); X[3, 3] = np.nan
    got, ok = ops.rotate_geno(U, X, ctx=ctx)
    assert ok == 0 and got is None


@pytest.mark.parametrize("kern", ["i8", "f16"])
@pytest.mark.parametrize("n,p,std", [(300, 200, False), (1000, 513, True)])
def test_rotate_genotype_with_mean_imputed_missing(n, p, std, kern, ctx, monkeypatch):
    """Columns whose missing calls were imputed with the column mean (one extra value per column, what the reference's
    callers feed: experiments/benchmarks/benchmarks.py:243-244) stay on the genotype path: codes + indicator pass."""
    from pygemma_amd import ops
    _kernel_env(monkeypatch, kern)
    rng = np.random.default_rng(n)
    U = np.linalg.qr(rng.standard_normal((n, n)))[0].astype(np.float32)
    G = rng.binomial(2, rng.uniform(0.05, 0.5, p), size=(n, p)).astype(np.float64)
    miss = rng.random((n, p)) < 0.02
    miss[:, ::7] = False                    # some columns without missing calls
    G[miss] = np.nan
    mu = np.nanmean(G, axis=0)
    G = np.where(np.isnan(G), mu[None, :], G)
    if std:
        G = (G - G.mean(0)) / np.maximum(G.std(0), 1e-9)
    X = G.astype(np.float32)
    got, ok = ops.rotate_geno(U, X, ctx=ctx)
    assert ok
    exact = (U.astype(np.float64).T @ X.astype(np.float64)).T
    f32p = ops.rotate(U, X, ctx=ctx)[:, :n]
    bound = (np.abs(X.astype(np.float64)).T @ np.abs(U.astype(np.float64)))
    err_g = np.abs(got[:, :n] - exact) / bound
    err_f = np.abs(f32p - exact) / bound
    assert err_g.max() <= 4 * 2.0 ** -24 * np.sqrt(n) and np.median(err_g) <= 2 * max(np.median(err_f), 1e-9)
    assert (got[:, n:] == 0).all()


@pytest.mark.parametrize("kern", ["i8", "f16"])
def test_rotate_auto_device_side_path_choice_equals_host_side(kern, ctx, monkeypatch):
    """pg_rotate_auto_dev enqueues every candidate kernel predicated on the detect pass's flags (no host read-back): for each of
    the four kinds of block it must produce exactly what pg_rotate_geno_dev + the caller's fallback produce, and report the path."""
    from pygemma_amd import ops
    _kernel_env(monkeypatch, kern)
    rng = np.random.default_rng(17)
    n, p = 321, 200
    U = np.linalg.qr(rng.standard_normal((n, n)))[0].astype(np.float32)
    geno = _geno(rng, n, p, True)
    imputed = _geno(rng, n, p, False); imputed[rng.random((n, p)) < 0.02] = 0.7310585
    for j in range(p):                                   # one other value per column: the column mean of its called genotypes
        col = imputed[:, j]; col[col == np.float32(0.7310585)] = np.float32(col[col != np.float32(0.7310585)].mean())
    dosage = rng.uniform(0, 2, (n, p)).astype(np.float32)
    nanblk = geno.copy(); nanblk[4, 9] = np.nan
    for X, want in ((geno, 1), (imputed, 1), (dosage, 2), (nanblk, 0)):
        got, path = ops.rotate_auto(U, X, ctx=ctx)
        assert path == want
        ref, ok = ops.rotate_geno(U, X, ctx=ctx)
        assert ok == want
        if want == 0:
            ref = ops.rotate(U, X, ctx=ctx)
            same = (got.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(got) & np.isnan(ref))
            assert same.all() and np.isnan(got[9, :n]).all() and np.isfinite(got[8, :n]).all()
        else:
            assert (got.view(np.uint32) == ref.view(np.uint32)).all()


def test_rotate_auto_extreme_magnitudes(ctx):
    """pg_rotate_auto_dev on blocks at the ends of the float32 range (genotype codes with steps 1e-40 ... 1e20 and a 1e6 offset,
    dosages and normal values scaled per column / per element over 40-70 decades, denormals, values near FLT_MAX, a block holding
    one inf or NaN): float32's own error bound 4 * 2^-24 sqrt(n) sum|x||u| against an fp64 rotation, every finite column finite,
    the non-finite pattern of the exact product kept (tools/adversarial_rotate.py prints the same per case)."""
    from pygemma_amd import ops
    rng = np.random.default_rng(3)
    n, p = 777, 40
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    U = Q.astype(np.float32)
    geno = rng.binomial(2, 0.3, size=(n, p)).astype(np.float64)
    norm = rng.standard_normal((n, p))
    at = (np.arange(n)[:, None] == 5) & (np.arange(p)[None, :] == 7)
    blocks = [(geno * 1e20, 1), (geno * 1e-20, 1), (geno * 1e-40, 1), (geno - 1e6, 1), (geno * (10.0 ** rng.integers(-30, 30, p))[None, :], 1),
              (geno + rng.uniform(-0.3, 0.3, geno.shape), 2), (norm * 1e30, 2), (norm * 1e-30, 2), (norm * 1e-40, 2),
              (norm * (10.0 ** rng.integers(-35, 35, p))[None, :], 2), (norm * (10.0 ** rng.integers(-20, 20, norm.shape)), 2),
              (norm * 3e38 / np.abs(norm).max(), 2), (np.zeros((n, p)), 1), (np.where(at, np.inf, norm), 0), (np.where(at, np.nan, geno), 0)]
    for Xd, want_path in blocks:
        X = Xd.astype(np.float32)
        got, path = ops.rotate_auto(U, X, ctx=ctx)
        assert path == want_path
        got, X64 = got[:, :n].astype(np.float64), X.astype(np.float64)
        with np.errstate(invalid="ignore", over="ignore"):
            exact = (U.astype(np.float64).T @ X64).T
            bound = np.abs(X64).T @ np.abs(U.astype(np.float64))
        fin = np.isfinite(X64).all(0)
        assert np.isfinite(got[fin]).all()
        err = np.abs(got[fin] - exact[fin])
        assert ((err <= 4 * 2.0 ** -24 * np.sqrt(n) * bound[fin]) | (err <= 1.5e-45 * n)).all()
        assert not np.isfinite(got[~fin]).any() or fin.all()       # a column holding inf / NaN comes out non-finite
