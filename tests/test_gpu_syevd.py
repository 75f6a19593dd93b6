"""GPU tests of the fp64 eigensolver pieces and of pg_syevd_dev (SURVEY 8c Tier B: invariants, not
agreement with the reference's float32 LAPACK)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from pygemma_amd import _lib
    c = _lib.Context(0)
    yield c
    c.close()


def _kin(n, seed=0):
    from pygemma_amd import synth
    return synth.panel(n, 4, 1, seed=seed)["K"]


@pytest.mark.parametrize("ta", [0, 1])
@pytest.mark.parametrize("M,N,K", [(128, 128, 8), (200, 130, 77), (513, 260, 128), (64, 1000, 300)])
def test_dgemm_mfma_f64(ta, M, N, K, ctx):
    from pygemma_amd import _lib
    L = _lib.load()
    L.pgx_dgemm_dev.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_double, C.c_void_p, C.c_int64,
                                C.c_void_p, C.c_int64, C.c_double, C.c_void_p, C.c_int64]
    rng = np.random.default_rng(M + N + K + ta)
    A = rng.standard_normal((K, M) if ta else (M, K))
    B = rng.standard_normal((K, N))
    Cm = rng.standard_normal((M, N))
    dA, dB, dC = ctx.to_device(A), ctx.to_device(B), ctx.to_device(Cm)
    _lib.check(L.pgx_dgemm_dev(ctx.handle, ta, M, N, K, -1.5, dA.ptr, A.shape[1], dB.ptr, N, 0.5, dC.ptr, N), "dgemm")
    ctx.sync()
    got = dC.download((M, N), np.float64)
    ref = -1.5 * ((A.T if ta else A) @ B) + 0.5 * Cm
    assert np.abs(got - ref).max() <= 1e-12 * np.sqrt(K) * 10


TA, TB, LOW, SYM, NOSPLIT = 1, 2, 4, 8, 16


def _dgemm_ex(ctx, flags, kxor, M, N, K, alpha, A, lda, B, ldb, beta, Cm, ldc):
    from pygemma_amd import _lib
    L = _lib.load()
    dA, dB, dC = ctx.to_device(A), ctx.to_device(B), ctx.to_device(Cm)
    _lib.check(L.pgx_dgemm_ex_dev(ctx.handle, flags, kxor, M, N, K, alpha, dA.ptr, lda, dB.ptr, ldb, beta, dC.ptr, ldc), "dgemm_ex")
    ctx.sync()
    out = dC.download(Cm.shape, np.float64)
    for b_ in (dA, dB, dC):
        b_.free()
    return out


@pytest.mark.parametrize("ta", [0, 1])
@pytest.mark.parametrize("beta", [0.0, 0.5])
@pytest.mark.parametrize("M,N,K", [(128, 128, 8), (256, 128, 24), (200, 130, 77), (513, 260, 128), (300, 64, 1000), (1000, 1000, 256),
                                   (129, 64, 2051), (640, 192, 3), (2, 2, 5), (4000, 64, 64), (1280, 1280, 39)])
def test_dgemm_ring_kernel_shapes(ta, beta, M, N, K, ctx):
    """The LDS-DMA ring kernel (csrc/dgemm.hpp) on full tiles, ragged edges in M and N, K tails (K % 8 != 0), K shorter than the ring,
    64-wide outputs, with and without the C tile read back (beta), both A layouts, with sub-matrix strides; against NumPy fp64."""
    rng = np.random.default_rng(1000 * M + 10 * N + K + ta)
    lda = ((M if ta else K) + 7) // 2 * 2          # even: 16-byte rows (the ring kernel's precondition; odd strides take the register-staged kernel)
    ldb, ldc = N + 2, N + 4
    A = rng.standard_normal((K if ta else M, lda))
    B = rng.standard_normal((K, ldb))
    Cm = rng.standard_normal((M, ldc))
    got = _dgemm_ex(ctx, TA if ta else 0, 0, M, N, K, -1.5, A, lda, B, ldb, beta, Cm, ldc)
    Aop = A[:, :M].T if ta else A[:, :K]
    ref = Cm.copy()
    ref[:, :N] = -1.5 * (Aop @ B[:, :N]) + beta * Cm[:, :N]
    assert np.abs(got - ref).max() <= 1e-12 * np.sqrt(K) * 10
    assert (got[:, N:] == Cm[:, N:]).all()


@pytest.mark.parametrize("m,pad", [(640, 0), (1000, 64), (2304, 0), (130, 0)])
def test_dgemm_rank_2b_update_on_the_lower_triangle(m, pad, ctx):
    """The band reduction's update A22 -= [V W][W V]' as it is called (csrc/sb2.hip): both operands the SAME k-major array [V W]'
    (128 x ldt), the second read with its k index XOR-ed by 64, lower triangle of C only (tiles above the diagonal untouched)."""
    rng = np.random.default_rng(m)
    mm = m + pad
    ldt = (mm + 255) // 128 * 128
    VWt = np.zeros((128, ldt))
    VWt[:, pad:mm] = rng.standard_normal((128, m))
    Cm = rng.standard_normal((mm, mm + 2))
    got = _dgemm_ex(ctx, TA | LOW, 64, mm, mm, 128, -1.0, VWt, ldt, VWt, ldt, 1.0, Cm, mm + 2)
    V, W = VWt[:64, :mm].T, VWt[64:, :mm].T
    ref = Cm[:, :mm] - (V @ W.T + W @ V.T)
    i, j = np.indices((mm, mm))
    low = (j // 128) <= (i // 128)                       # tiles on or below the diagonal
    assert np.abs(got[:, :mm] - ref)[low].max() <= 1e-12
    assert (got[:, :mm][~low] == Cm[:, :mm][~low]).all()


def test_dgemm_lower_update_of_one_tile_column(ctx):
    """The look-ahead of the band reduction updates the first tile column on its own: C is rectangular (m x 128), lower_only still
    skips the tile above the diagonal (there is none in column 0) — and a rectangular lower C of two tile columns skips tile (0, 1)."""
    rng = np.random.default_rng(3)
    m, ldt = 900, 1152
    VWt = np.zeros((128, ldt)); VWt[:, :m] = rng.standard_normal((128, m))
    V, W = VWt[:64, :m].T, VWt[64:, :m].T
    full = V @ W.T + W @ V.T
    for N in (128, 256):
        Cm = rng.standard_normal((m, N))
        got = _dgemm_ex(ctx, TA | LOW, 64, m, N, 128, -1.0, VWt, ldt, VWt, ldt, 1.0, Cm, N)
        ref = Cm - full[:, :N]
        i, j = np.indices((m, N))
        low = (j // 128) <= (i // 128)
        assert np.abs(got - ref)[low].max() <= 1e-12
        assert (got[~low] == Cm[~low]).all()


@pytest.mark.parametrize("m,N,split", [(640, 64, 0), (1000, 64, 0), (2304, 64, 1), (2304, 64, 0), (777, 128, 0), (1290, 192, 1)])
def test_dgemm_symmetric_A_from_its_lower_triangle(m, N, split, ctx):
    """X = A V with A symmetric and only its lower triangle + the full 128 x 128 diagonal tiles valid (the rest is NaN here and must never
    be read): k-chunks left of / on the diagonal tile m-major, right of it k-major from the transposed position; with and without split K."""
    rng = np.random.default_rng(m + N)
    S = rng.standard_normal((m, m)); S = S + S.T
    i, j = np.indices((m, m))
    A = np.where((j // 128) <= (i // 128), S, np.nan)
    V = rng.standard_normal((m, N + 2))
    Cm = np.zeros((m, N))
    got = _dgemm_ex(ctx, SYM | (0 if split else NOSPLIT), 0, m, N, m, 1.0, A, m, V, N + 2, 0.0, Cm, N)
    assert np.isfinite(got).all()
    assert np.abs(got - S @ V[:, :N]).max() <= 1e-12 * np.sqrt(m) * 10


@pytest.mark.parametrize("n", [5, 64, 65, 130, 300, 777])
def test_sytrd_similarity(n, ctx):
    """T = Q' K Q: the tridiagonal's eigenvalues equal K's (fp64), and Q built from the reflectors is
    orthogonal with Q T Q' = K."""
    import scipy.linalg as sl
    from pygemma_amd import _lib
    L = _lib.load()
    L.pgx_sytrd_dev.argtypes = [C.c_void_p, C.c_int64] + [C.c_void_p] * 5
    K = _kin(n, seed=n)
    dK = ctx.to_device(K)
    dd, de, dt, dV = ctx.alloc(n * 8), ctx.alloc(n * 8), ctx.alloc(n * 8), ctx.alloc(n * n * 8)
    _lib.check(L.pgx_sytrd_dev(ctx.handle, n, dK.ptr, dd.ptr, de.ptr, dt.ptr, dV.ptr), "sytrd")
    d, e, tau = dd.download((n,), np.float64), de.download((n,), np.float64)[: n - 1], dt.download((n,), np.float64)[: n - 1]
    V = dV.download((n, n), np.float64)
    K64 = np.tril(K.astype(np.float64)); K64 = K64 + np.tril(K64, -1).T
    ref = np.linalg.eigvalsh(K64)
    got = sl.eigvalsh_tridiagonal(d, e)
    assert np.abs(got - ref).max() <= 1e-12 * max(1.0, np.abs(ref).max())
    if n <= 300:
        Q = np.eye(n)
        for i in range(n - 1):
            v = V[:, i]
            Q = Q - tau[i] * np.outer(Q @ v, v)
        T = np.diag(d) + np.diag(e, 1) + np.diag(e, -1)
        assert np.abs(Q.T @ Q - np.eye(n)).max() <= 1e-13 * n
        assert np.abs(Q @ T @ Q.T - K64).max() <= 1e-13 * n * np.abs(K64).max()


def _tridiags():
    rng = np.random.default_rng(42)
    out = {}
    for n in (1, 2, 3, 31, 32, 33, 64, 100, 257, 1000):
        out[f"rand{n}"] = (rng.standard_normal(n), rng.standard_normal(max(n - 1, 0)))
    n = 301
    out["toeplitz121"] = (np.full(n, 2.0), np.full(n - 1, -1.0))
    m = 10
    w = np.abs(np.arange(-m, m + 1)).astype(float)
    out["wilkinson21"] = (w, np.ones(2 * m))
    # glued Wilkinson: tight clusters
    d = np.concatenate([w] * 8); e = np.ones(d.size - 1); e[2 * m::2 * m + 1] = 1e-8
    out["glued_wilkinson"] = (d, e[: d.size - 1])
    out["graded"] = (10.0 ** -np.arange(0, 120, dtype=float) ** 0.5, 10.0 ** -np.arange(1, 120, dtype=float) ** 0.5)
    out["zero_offdiag"] = (rng.standard_normal(200), np.where(rng.random(199) < 0.3, 0.0, rng.standard_normal(199)))
    out["identity"] = (np.ones(150), np.zeros(149))
    return out


@pytest.mark.parametrize("batch", ["default", "0"])
@pytest.mark.parametrize("name", list(_tridiags().keys()))
def test_stedc_divide_and_conquer(name, batch, ctx, monkeypatch):
    """T = Z diag(lam) Z' for hard tridiagonals: eigenvalues vs LAPACK (fp64), residual and orthogonality — with the small merges of a
    level in batched launches (default) and every merge on its own (PG_DC_BATCH_MAX=0)."""
    if batch != "default":
        monkeypatch.setenv("PG_DC_BATCH_MAX", batch)
    import scipy.linalg as sl
    from pygemma_amd import _lib
    L = _lib.load()
    L.pgx_stedc_dev.argtypes = [C.c_void_p, C.c_int64] + [C.c_void_p] * 4
    d, e = _tridiags()[name]
    n = d.size
    d = np.ascontiguousarray(d, np.float64); e2 = np.ascontiguousarray(np.concatenate([e, [0.0]]), np.float64)
    ev = np.empty(n)
    dZ = ctx.alloc(n * n * 8)
    _lib.check(L.pgx_stedc_dev(ctx.handle, n, d.ctypes.data, e2.ctypes.data, ev.ctypes.data, dZ.ptr), "stedc")
    Z = dZ.download((n, n), np.float64)
    T = np.diag(d) + (np.diag(e, 1) + np.diag(e, -1) if n > 1 else 0)
    ref = np.linalg.eigvalsh(T) if n > 1 else d.copy()
    scale = max(np.abs(T).max(), 1e-300)
    assert np.all(np.diff(ev) >= 0)
    assert np.abs(ev - ref).max() <= 2e-14 * scale * max(1, np.sqrt(n))
    assert np.abs(Z.T @ Z - np.eye(n)).max() <= 5e-14 * max(1, np.sqrt(n))
    assert np.abs(T @ Z - Z * ev).max() <= 5e-14 * scale * max(1, np.sqrt(n))


@pytest.mark.parametrize("stages", ["1", "2"])
@pytest.mark.parametrize("n", [1, 2, 7, 64, 129, 300, 1000, 1940])
def test_syevd_invariants(n, stages, ctx, monkeypatch):
    """pg_syevd_dev (SURVEY 8c Tier B): ||K - U L U'||_F/||K||_F <= 1e-12 sqrt(n), |U'U - I|_max <= 1e-12,
    eigenvalues vs host LAPACK dsyevd within 1e-12 lambda_max; f32 outputs = clamp + cast of the f64 ones.
    Both reductions: one-stage Householder and two-stage (dense -> band -> tridiagonal; forced here, by default the size decides)."""
    from pygemma_amd import ops
    if stages == "2" and n < 192:
        pytest.skip("the two-stage path starts at n = 192")
    monkeypatch.setenv("PG_SYEVD_STAGES", stages)
    K = _kin(n, seed=100 + n) if n > 2 else np.array([[2.0, 0], [0.5, 1.0]], np.float32)[:n, :n]
    K64 = np.tril(K.astype(np.float64)); K64 = K64 + np.tril(K64, -1).T
    ev32, U32, ev, U = ops.syevd(K, ctx=ctx, want64=True)
    ref = np.linalg.eigvalsh(K64)
    lmax = max(np.abs(ref).max(), 1e-300)
    assert np.abs(ev - ref).max() <= 1e-12 * lmax
    assert np.abs(U.T @ U - np.eye(n)).max() <= 1e-12
    res = np.linalg.norm(K64 - (U * ev) @ U.T) / max(np.linalg.norm(K64), 1e-300)
    assert res <= 1e-12 * np.sqrt(n)
    assert (ev32 == np.maximum(ev, 0).astype(np.float32)).all() and (ev32 >= 0).all()
    assert (U32 == U.astype(np.float32)).all()


def _structured(name, n, rng):
    """Relatedness matrices with the degeneracies real cohorts have (the path's K is whatever the caller computed, lmm.py:151)."""
    if name == "rank_deficient":            # fewer SNPs than samples: n - p exact zero eigenvalues
        G = rng.binomial(2, 0.3, size=(n, n // 4)).astype(np.float64); G -= G.mean(0)
        return G @ G.T / G.shape[1]
    if name == "block_diagonal":            # unrelated families: the tridiagonal form splits
        K = np.zeros((n, n)); s = 0
        while s < n:
            b = min(int(rng.integers(1, 40)), n - s)
            A = rng.standard_normal((b, 3 * b)); K[s:s + b, s:s + b] = A @ A.T / (3 * b); s += b
        return K
    if name == "identity_plus_tiny":
        A = rng.standard_normal((n, n)) * 1e-9
        return np.eye(n) + (A + A.T)
    if name == "rank_one":
        return np.ones((n, n))
    if name == "zero":
        return np.zeros((n, n))
    if name == "duplicated_samples":        # monozygotic twins / sample duplicates: exactly repeated rows and columns
        G = rng.binomial(2, 0.3, size=(n, 2 * n)).astype(np.float64)
        G[n // 2:] = G[: n - n // 2]
        G -= G.mean(0)
        return G @ G.T / G.shape[1]
    if name == "diagonal":
        return np.diag(rng.uniform(0.5, 2.0, n))
    if name == "huge_scale":
        A = rng.standard_normal((n, n)); return (A @ A.T) * 1e30
    if name == "tiny_scale":
        A = rng.standard_normal((n, n)); return (A @ A.T) * 1e-30
    raise KeyError(name)


@pytest.mark.parametrize("name", ["rank_deficient", "block_diagonal", "identity_plus_tiny", "rank_one", "zero", "duplicated_samples",
                                  "diagonal", "huge_scale", "tiny_scale"])
@pytest.mark.parametrize("n,stages", [(130, "1"), (601, "1"), (601, "2"), (1153, "2")])
def test_syevd_structured_matrices(name, n, stages, ctx, monkeypatch):
    """Degenerate spectra through the whole solver (Householder with zero columns, deflation-heavy merges, splits): same Tier-B
    invariants on the float64 outputs; the scales 1e+-30 sit near the ends of the float32 range K arrives in.  Forced onto the
    two-stage path these matrices are the ones whose panels CholeskyQR2 cannot factor (zero / dependent columns): the device
    flag sends them to the one-stage reduction, and what comes back must satisfy the same invariants."""
    from pygemma_amd import ops
    monkeypatch.setenv("PG_SYEVD_STAGES", stages)
    rng = np.random.default_rng(7 * n + len(name))
    K = _structured(name, n, rng).astype(np.float32)
    K64 = np.tril(K.astype(np.float64)); K64 = K64 + np.tril(K64, -1).T
    ev32, U32, ev, U = ops.syevd(K, ctx=ctx, want64=True)
    assert np.isfinite(ev).all() and np.isfinite(U).all()
    ref = np.linalg.eigvalsh(K64)
    lmax = max(np.abs(ref).max(), 1e-300)
    assert np.all(np.diff(ev) >= 0)
    assert np.abs(ev - ref).max() <= 1e-12 * lmax
    assert np.abs(U.T @ U - np.eye(n)).max() <= 1e-12
    res = np.linalg.norm(K64 - (U * ev) @ U.T) / max(np.linalg.norm(K64), 1e-300)
    assert res <= 1e-12 * np.sqrt(n)
    assert (ev32 >= 0).all() and np.isfinite(U32).all()


@pytest.mark.parametrize("stationary", [1, 2, 0])
@pytest.mark.parametrize("n", [192, 300, 777, 1300])
def test_two_stage_pieces(n, stationary, ctx, monkeypatch):
    """The two stages of the tridiagonalisation one at a time (csrc/sb2.hip), against fp64 LAPACK, with both bulge-chasing kernels
    (band rows stationary in LDS, 16 wavefronts per workgroup — the default up to one row block per CU — or 8, two workgroups per CU; rows
    carried through memory):
    stage 1: the band matrix keeps K's spectrum, Q1 (the back-transformation applied to I) is orthogonal and Q1 B Q1' = K, no panel
             needed the fallback on a full-rank K;
    stage 2: the tridiagonal keeps the band's spectrum, Q2 orthogonal, Q2 T Q2' = B, no wait of the bulge-chasing kernel expired."""
    import scipy.linalg as sl
    from pygemma_amd import _lib
    L = _lib.load()
    K = _kin(n, seed=3 * n)
    K64 = np.tril(K.astype(np.float64)); K64 = K64 + np.tril(K64, -1).T
    lam, nrm = np.linalg.eigvalsh(K64), np.abs(K64).max()
    flags = (C.c_int * 4)()
    dK, dA, dZ = ctx.to_device(K), ctx.alloc(n * n * 8), ctx.to_device(np.eye(n))
    _lib.check(L.pgx_sb2_stage1_dev(ctx.handle, n, dK.ptr, dA.ptr, dZ.ptr, flags), "stage 1")
    assert list(flags)[:2] == [0, 0]
    A, Q1 = dA.download((n, n), np.float64), dZ.download((n, n), np.float64)
    i, j = np.indices((n, n))
    Bm = np.where((i >= j) & (i - j <= 64), A, 0.0)
    Bm = Bm + np.tril(Bm, -1).T
    assert np.abs(np.linalg.eigvalsh(Bm) - lam).max() <= 1e-12 * nrm
    assert np.abs(Q1.T @ Q1 - np.eye(n)).max() <= 1e-13
    assert np.abs(Q1 @ Bm @ Q1.T - K64).max() <= 1e-12 * nrm
    dB, dd, de = ctx.to_device(Bm), ctx.alloc(n * 8), ctx.alloc(n * 8)
    dZ.upload(np.eye(n))
    monkeypatch.setenv("PG_BC_STATIONARY", "1" if stationary else "0")
    monkeypatch.setenv("PG_BC_PER_CU", str(max(stationary, 1)))       # 2: the 8-wavefront kernel that runs two workgroups per CU
    _lib.check(L.pgx_sb2_stage2_dev(ctx.handle, n, dB.ptr, dd.ptr, de.ptr, dZ.ptr, flags), "stage 2")
    assert list(flags)[:2] == [0, 0]
    d, e, Q2 = dd.download((n,), np.float64), de.download((n,), np.float64)[: n - 1], dZ.download((n, n), np.float64)
    T = np.diag(d) + np.diag(e, 1) + np.diag(e, -1)
    assert np.abs(sl.eigvalsh_tridiagonal(d, e) - lam).max() <= 1e-12 * nrm
    assert np.abs(Q2.T @ Q2 - np.eye(n)).max() <= 1e-13
    assert np.abs(Q2 @ T @ Q2.T - Bm).max() <= 1e-12 * nrm
    for b in (dK, dA, dZ, dB, dd, de):
        b.free()


@pytest.mark.parametrize("n", [300, 1000])
def test_stage2_back_transformation_one_block_per_trip(n, ctx, monkeypatch):
    """PG_BT2_FOUR=0: the stage-2 reflector blocks applied one per trip of a slab (bt2_apply_kernel) instead of four (bt2_apply4_kernel):
    same Q2 to rounding — both orthogonal, both reproduce the band matrix."""
    import scipy.linalg as sl
    from pygemma_amd import _lib
    L = _lib.load()
    rng = np.random.default_rng(n)
    A = rng.standard_normal((n, n)); A = A + A.T
    i, j = np.indices((n, n)); Bm = np.where(np.abs(i - j) <= 64, A, 0.0)
    nrm = np.abs(Bm).max()
    flags = (C.c_int * 4)()
    dB, dd, de, dZ = ctx.to_device(Bm), ctx.alloc(n * 8), ctx.alloc(n * 8), ctx.to_device(np.eye(n))
    out = {}
    for four in ("1", "0"):
        monkeypatch.setenv("PG_BT2_FOUR", four)
        dZ.upload(np.eye(n))
        _lib.check(L.pgx_sb2_stage2_dev(ctx.handle, n, dB.ptr, dd.ptr, de.ptr, dZ.ptr, flags), "stage 2")
        assert list(flags)[:2] == [0, 0]
        d, e, Q2 = dd.download((n,), np.float64), de.download((n,), np.float64)[: n - 1], dZ.download((n, n), np.float64)
        T = np.diag(d) + np.diag(e, 1) + np.diag(e, -1)
        assert np.abs(Q2.T @ Q2 - np.eye(n)).max() <= 1e-13
        assert np.abs(Q2 @ T @ Q2.T - Bm).max() <= 1e-12 * nrm
        out[four] = Q2
    assert np.abs(out["1"] - out["0"]).max() <= 1e-13
    for b in (dB, dd, de, dZ):
        b.free()


@pytest.mark.parametrize("knob", ["PG_SB2_FUSED", "PG_SB2_CHOL16", "PG_DGEMM_RING", "PG_SB2_LOOKAHEAD"])
@pytest.mark.parametrize("n", [832, 1300])
def test_two_stage_alternative_paths(n, knob, ctx, monkeypatch):
    """The A/B switches of round 4 keep every path alive: the panel chain as separate GEMM launches (PG_SB2_FUSED=0), the one-wavefront
    first-pass Cholesky (PG_SB2_CHOL16=0), the register-staged GEMM kernel everywhere (PG_DGEMM_RING=0), the look-ahead of the band
    reduction on a second stream with the thin next-panel update (PG_SB2_LOOKAHEAD=-1) — same Tier-B invariants as the default path."""
    from pygemma_amd import ops
    monkeypatch.setenv("PG_SYEVD_STAGES", "2")
    monkeypatch.setenv(knob, "-1" if knob == "PG_SB2_LOOKAHEAD" else "0")
    K = _kin(n, seed=5 * n)
    K64 = np.tril(K.astype(np.float64)); K64 = K64 + np.tril(K64, -1).T
    ev32, U32, ev, U = ops.syevd(K, ctx=ctx, want64=True)
    ref = np.linalg.eigvalsh(K64)
    assert np.abs(ev - ref).max() <= 1e-12 * np.abs(ref).max()
    assert np.abs(U.T @ U - np.eye(n)).max() <= 1e-12
    assert np.linalg.norm(K64 - (U * ev) @ U.T) / np.linalg.norm(K64) <= 1e-12 * np.sqrt(n)


def test_stationary_bulge_chasing_wait_expiry_falls_back_to_the_memory_kernel(ctx, monkeypatch):
    """PG_BC_TEST_FAULT=1 removes one workgroup of the stationary bulge-chasing kernel: its neighbours' bounded waits expire, the flag is
    raised (nothing hangs), and pg_syevd_dev repeats stage 2 with the kernel that carries the rows through memory — same answer."""
    n = 900
    K = _kin(n, seed=77)
    K64 = np.tril(K.astype(np.float64)); K64 = K64 + np.tril(K64, -1).T
    monkeypatch.setenv("PG_SYEVD_STAGES", "2")
    monkeypatch.setenv("PG_BC_TEST_FAULT", "1")
    from pygemma_amd import ops
    ev32, U32, ev, U = ops.syevd(K, ctx=ctx, want64=True)
    assert np.abs(U.T @ U - np.eye(n)).max() <= 1e-12
    assert np.linalg.norm(K64 - (U * ev) @ U.T) / np.linalg.norm(K64) <= 1e-12 * np.sqrt(n)
    assert np.abs(ev - np.linalg.eigvalsh(K64)).max() <= 1e-12 * np.abs(ev).max()


def test_two_stage_flags_a_rank_deficient_panel(ctx):
    """A K whose first panel has dependent columns: stage 1 must raise its flag (pg_syevd_dev then takes the one-stage path)."""
    from pygemma_amd import _lib
    L = _lib.load()
    n = 400
    K = np.ones((n, n), np.float32)
    flags = (C.c_int * 4)()
    dK, dA = ctx.to_device(K), ctx.alloc(n * n * 8)
    _lib.check(L.pgx_sb2_stage1_dev(ctx.handle, n, dK.ptr, dA.ptr, None, flags), "stage 1")
    assert flags[0] != 0
    dK.free(); dA.free()


def test_two_stage_solve_on_the_last_device_after_device_zero():
    """Per-device state: the kernels of the two-stage solver that need more than 64 KB of dynamic LDS carry that limit
    as a per-DEVICE function attribute.  One process solves on device 0, then on the LAST visible device: the second solve must launch
    with the attribute set on that device too and satisfy the same invariants.  Skips below two GPUs."""
    from pygemma_amd import _lib, ops
    ndev = _lib.device_count()
    if ndev < 2:
        pytest.skip("needs two visible GPUs")
    n = 832
    K = _kin(n, seed=4242)
    K64 = np.tril(K.astype(np.float64)); K64 = K64 + np.tril(K64, -1).T
    ref = np.linalg.eigvalsh(K64)
    import os
    old = os.environ.get("PG_SYEVD_STAGES")
    os.environ["PG_SYEVD_STAGES"] = "2"
    try:
        for dev in (0, ndev - 1):
            c = _lib.Context(dev)
            try:
                ev32, U32, ev, U = ops.syevd(K, ctx=c, want64=True)
            finally:
                c.close()
            assert np.abs(ev - ref).max() <= 1e-12 * np.abs(ref).max(), f"device {dev}"
            assert np.abs(U.T @ U - np.eye(n)).max() <= 1e-12, f"device {dev}"
            assert np.linalg.norm(K64 - (U * ev) @ U.T) / np.linalg.norm(K64) <= 1e-12 * np.sqrt(n), f"device {dev}"
    finally:
        if old is None:
            os.environ.pop("PG_SYEVD_STAGES", None)
        else:
            os.environ["PG_SYEVD_STAGES"] = old
