"""Deterministic synthetic (n, p, c) association panels (SURVEY.md §8d / BASELINE.md §3).

Genotypes G[i,g] ~ Binomial(2, maf_g), maf_g ~ U(0.05, 0.5), column-standardised;
K = G_K G_K' / p_K from an independent SNP set (full rank, PSD, realistic spectrum);
W = [1 | N(0,1)^(c-1)];  y = 0.2*g_0 + G_K b + e  (h2 = 0.5)  or a pure-noise phenotype.
Pure NumPy: this is input generation, not part of the measured path.
"""
import numpy as np

SEED = 20241115


def genotypes(rng, n, p, dtype=np.float32):
    maf = rng.uniform(0.05, 0.5, size=p)
    G = rng.binomial(2, maf, size=(n, p)).astype(np.float64)
    mu = G.mean(axis=0)
    sd = G.std(axis=0)
    sd[sd == 0] = 1.0
    return ((G - mu) / sd).astype(dtype)


def panel(n, p, c, seed=SEED, null=False, p_k=None, h2=0.5):
    """Returns dict(Y (n,1), X (n,p), W (n,c), K (n,n)) float32 — raw (un-rotated) inputs of lmm.pygemma."""
    rng = np.random.default_rng(seed)
    p_k = p_k or 2 * n
    GK = genotypes(rng, n, p_k, np.float64)
    K = (GK @ GK.T) / p_k
    X = genotypes(rng, n, p)
    W = np.concatenate([np.ones((n, 1)), rng.standard_normal((n, max(c - 1, 0)))], axis=1)[:, :c]
    if null:
        y = rng.standard_normal(n)
    else:
        b = rng.standard_normal(p_k) * np.sqrt(h2 / p_k)
        y = 0.2 * X[:, 0].astype(np.float64) + GK @ b + rng.standard_normal(n) * np.sqrt(1 - h2)
    return {"Y": y.reshape(-1, 1).astype(np.float32), "X": X, "W": W.astype(np.float32), "K": K.astype(np.float32)}


def exact_panel(n, p, c, seed=SEED, p_k=None):
    """Raw inputs of lmm.pygemma that regenerate BIT FOR BIT from a seed on any host (the big Tier-C fixtures store only the seed, Y, W
    and the reference's output columns): X = raw hard calls 0/1/2 as float32; K = C C' / p_k with C = codes - 1 in {-1, 0, 1}
    — every partial sum of the float32 product is an integer below 2^24, so the result does not depend on the BLAS, its threading
    or its summation order; one rounding in the division.  Returns dict(X, K, C) — Y and W come from the fixture (their fp64
    matrix-vector sums are not order-independent)."""
    rng = np.random.default_rng(seed)
    p_k = p_k or 2 * n
    assert p_k < (1 << 24)
    Cm = np.empty((n, p_k), np.float32)
    for s0 in range(0, p_k, 4096):
        e0 = min(p_k, s0 + 4096)
        thr = rng.uniform(0.05, 0.5, e0 - s0)
        u = rng.random((2, n, e0 - s0), dtype=np.float32)
        Cm[:, s0:e0] = (u[0] < thr).astype(np.float32) + (u[1] < thr).astype(np.float32) - 1.0
    K = (Cm @ Cm.T) / np.float32(p_k)
    thr = rng.uniform(0.05, 0.5, p)
    u = rng.random((2, n, p), dtype=np.float32)
    X = (u[0] < thr).astype(np.float32) + (u[1] < thr).astype(np.float32)
    return {"X": np.ascontiguousarray(X), "K": np.ascontiguousarray(K, np.float32), "C": Cm}


def rotated_panel(n, p, c, seed=SEED, null=False, h2=0.5):
    """Eigen-basis inputs (the reference's eigen=False entry, lmm/lmm.py:164-167): d (n,), Xr (n,p), Yr, Wr.
    Rotation done in float64 on the host (input preparation for kernels that start at that boundary)."""
    raw = panel(n, p, c, seed, null, h2=h2)
    d, U = np.linalg.eigh(raw["K"].astype(np.float64))
    d = np.maximum(d, 0.0).astype(np.float32)
    rot = lambda A: (U.T @ A.astype(np.float64)).astype(np.float32)
    return {"d": d, "X": rot(raw["X"]), "Y": rot(raw["Y"]), "W": rot(raw["W"])}


def fast_rotated_panel(n, p, c, seed=SEED, null=False):
    """Cheap large-shape eigen-basis inputs for throughput runs: a synthetic spectrum (bulk + a few
    large eigenvalues, like a GRM's) and Gaussian rotated columns with a planted polygenic component,
    so the Brent/Newton evaluation counts are in the realistic (signal) regime. O(n p)."""
    rng = np.random.default_rng(seed)
    d = np.sort(rng.gamma(2.0, 0.5, size=n)).astype(np.float32)
    d[-5:] *= np.array([3, 5, 8, 12, 40], np.float32)
    d[: max(1, n // 100)] = 0.0
    X = rng.standard_normal((n, p), dtype=np.float32)
    W = rng.standard_normal((n, c), dtype=np.float32)
    if null:
        y = rng.standard_normal(n, dtype=np.float32)
    else:
        lam = 1.0
        y = (rng.standard_normal(n) * np.sqrt(0.5 * (lam * d.astype(np.float64) + 1.0))).astype(np.float32)
        y += 0.05 * X[:, 0]
    return {"d": d, "X": X, "Y": y.reshape(-1, 1), "W": W}


def block_orthogonal(out, seed=0, blk=500):
    """Fills the (n, n) float32 array `out` with a DENSE, orthogonal (to float32 rounding) matrix in O(n^2 blk) flops — an eigenvector
    matrix for tests at sizes where a QR of an n x n Gaussian is out of reach (n = 50 000): U = P1 B1 P2 B2 with B1, B2 block-diagonal
    (random orthogonal blk x blk blocks) and P1, P2 random permutations.  Every entry is a sum of ~blk^2/n products of two entries of
    magnitude blk^-1/2, i.e. of the magnitude n^-1/2 an orthogonal matrix's entries have."""
    n = out.shape[0]
    assert out.shape == (n, n) and n % blk == 0
    rng = np.random.default_rng(seed)
    pool = [np.linalg.qr(rng.standard_normal((blk, blk)))[0].astype(np.float32) for _ in range(8)]
    nb = n // blk
    b1, b2 = rng.integers(0, 8, nb), rng.integers(0, 8, nb)
    s1, s2 = rng.choice([-1.0, 1.0], n).astype(np.float32), rng.choice([-1.0, 1.0], n).astype(np.float32)
    p1, p2 = rng.permutation(n), rng.permutation(n)
    M = np.zeros((blk, n), np.float32)
    for I in range(nb):
        M[:] = 0.0
        src = p2[I * blk:(I + 1) * blk]                      # rows of B2 that P2 brings to the rows of block I
        for t in range(blk):
            J, r = divmod(int(src[t]), blk)
            M[t, J * blk:(J + 1) * blk] = pool[b2[J]][r] * s2[J * blk:(J + 1) * blk]
        R = (pool[b1[I]] * s1[I * blk:(I + 1) * blk, None]) @ M      # signs keep the blocks orthogonal and make them distinct
        out[p1[I * blk:(I + 1) * blk]] = R
    return out


def degenerate_panels(seed=0, n=203, c=3, p=16):
    """(tag, d, W, y, X) with the degeneracies a caller can hand over at the eigen-basis boundary (eigen=False passes eigenvalues
    through unclamped, lmm/lmm.py:196-207 clamps only what it computes itself): SNP columns that are zero / constant / collinear
    with W or y / scaled to the ends of the float32 range / holding NaN or inf, eigenvalues that are zero, huge, tiny, inf, NaN or
    negative, phenotypes and covariates that are zero, scaled, duplicated or non-finite."""
    rng = np.random.default_rng(seed)
    d = np.sort(rng.gamma(0.5, 2.0, n)).astype(np.float32)
    W = np.concatenate([np.ones((n, 1)), rng.standard_normal((n, c - 1))], axis=1).astype(np.float32)
    X = rng.binomial(2, 0.3, size=(n, p)).astype(np.float32) - np.float32(0.6)
    y = (W @ rng.standard_normal(c) + 0.5 * X[:, 0] + rng.standard_normal(n)).astype(np.float32)
    X[:, 0] = 0.0; X[:, 1] = 3.0; X[:, 2] = W[:, 0]; X[:, 3] = W[:, -1]; X[:, 4] = y
    X[:, 5] *= np.float32(1e30); X[:, 6] *= np.float32(1e-30); X[:, 7] *= np.float32(1e-42)
    X[3, 8] = np.nan; X[5, 9] = np.inf; X[5, 10] = -np.inf
    X[:, 11] = 2.0 * W[:, 1] - W[:, 0]
    X[:, 12] = (np.arange(n) == 0)
    with np.errstate(over="ignore"):
        X[:, 13] *= np.float32(3e38)        # overflows to +-inf where |x| > 1
    i = np.arange(n)
    f32 = lambda a: np.ascontiguousarray(a, np.float32)
    yield "plain", d, W, y, X
    yield "d = 0", np.zeros_like(d), W, y, X
    yield "d huge", f32(d * np.float32(1e10)), W, y, X
    yield "d half zero", f32(np.where(i < n // 2, 0, d)), W, y, X
    yield "d tiny", f32(d * np.float32(1e-30)), W, y, X
    yield "d with inf", f32(np.where(i == n - 1, np.inf, d)), W, y, X
    yield "d with NaN", f32(np.where(i == 4, np.nan, d)), W, y, X
    yield "d one negative", f32(np.where(i == 0, -0.5, d)), W, y, X
    yield "d negative small", f32(np.where(i < 3, -1e-4, d)), W, y, X
    yield "y = 0", d, W, np.zeros_like(y), X
    yield "y * 1e20", d, W, f32(y * np.float32(1e20)), X
    yield "y * 1e-20", d, W, f32(y * np.float32(1e-20)), X
    yield "y = w0", d, W, W[:, 0].copy(), X
    yield "y with NaN", d, W, f32(np.where(i == 7, np.nan, y)), X
    yield "W duplicate column", d, f32(np.concatenate([W[:, :2], W[:, 1:2]], axis=1)), y, X
    yield "W zero column", d, f32(np.concatenate([W[:, :2], np.zeros((n, 1))], axis=1)), y, X
    yield "W * 1e20", d, f32(W * np.float32(1e20)), y, X
    yield "W with NaN", d, f32(np.where((i == 9)[:, None] & (np.arange(c) == 1)[None, :], np.nan, W)), y, X


# Rows of degenerate_panels() on which the outcome is cancellation noise (beta ~ 1e29, tau = 1.99e37 ...) and follows the summation
# order inside the reference's BLAS calls, which no restatement reproduces; keyed by the oracle's order (0: the reference's own,
# 1: the kernels').  The NaN pattern is compared on every row; values only off these.
DEGENERATE_SINGULAR_SNPS = {0: (1, 11),          # x constant (collinear with the intercept), x in span(W)
                            1: (1, 2, 3, 4, 11)}  # ... and x = w0, x = w_last, x = y (perfect fit: the residual is rounding noise)
DEGENERATE_SINGULAR_CASES = {0: ("W duplicate column",),
                             1: ("W duplicate column", "y = w0", "y = 0")}   # y in span(W): every statistic is residual noise
