"""Degenerate inputs through pg_assoc (eigen-basis boundary) against the oracle in the kernels' order: every column bit for bit,
NaN where the oracle has NaN.  Nothing may crash, hang or differ.  usage: adversarial_assoc.py [seed]"""
import sys, itertools
import numpy as np
sys.path.insert(0, '/root/repo')
from pygemma_amd import _lib, ops
from oracle import oracle as O
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
ctx = _lib.Context(0)

def base(n, c, p):
    d = np.sort(rng.gamma(0.5, 2.0, n)).astype(np.float32)
    W = np.concatenate([np.ones((n, 1)), rng.standard_normal((n, c - 1))], axis=1).astype(np.float32)
    X = rng.binomial(2, 0.3, size=(n, p)).astype(np.float32) - 0.6
    y = (W @ rng.standard_normal(c) + 0.5 * X[:, 0] + rng.standard_normal(n)).astype(np.float32)
    return d, W, y, X

def x_cases(d, W, y, X):
    n, p = X.shape
    X = X.copy()
    names = []
    def put(j, col, name): X[:, j] = col; names.append((j, name))
    put(0, np.zeros(n), "x = 0"); put(1, np.full(n, 3.0), "x constant")
    put(2, W[:, 0], "x = w0"); put(3, W[:, -1], "x = w_last"); put(4, y, "x = y")
    put(5, X[:, 5] * np.float32(1e30), "x * 1e30"); put(6, X[:, 6] * np.float32(1e-30), "x * 1e-30")
    put(7, X[:, 7] * np.float32(1e-42), "x denormal")
    col = X[:, 8].copy(); col[3] = np.nan; put(8, col, "x one NaN")
    col = X[:, 9].copy(); col[5] = np.inf; put(9, col, "x one +inf")
    col = X[:, 10].copy(); col[5] = -np.inf; put(10, col, "x one -inf")
    put(11, 2.0 * W[:, 1] - W[:, 0] if W.shape[1] > 1 else W[:, 0], "x in span(W)")
    put(12, np.where(np.arange(n) == 0, 1.0, 0.0), "x = e_0")
    put(13, X[:, 13] * np.float32(3e38), "x near f32 max")
    return X, dict(names)

def run(tag, d, W, y, X, names, grid):
    o = O.calculate(d, y, W, X, grid=grid, order=1, nthreads=4)
    try:
        g = ops.assoc(d, W, y, X, grid=grid, ctx=ctx)
    except Exception as ex:
        print(f"EXC  {tag} grid={grid}: {ex!r}"[:200]); return 1
    bad = 0
    for col in ("beta", "se_beta", "tau", "lambda", "F_wald", "p_wald"):
        a, b = np.asarray(g[col]), np.asarray(o[col])
        if col == "p_wald":
            same = np.isclose(a, b, rtol=1e-8, atol=0) | (np.isnan(a) & np.isnan(b))
        else:
            a = a.astype(b.dtype)
            same = (a == b) | (np.isnan(a) & np.isnan(b))
        for j in np.nonzero(~same)[0]:
            print(f"DIFF {tag} grid={grid} snp {j} ({names.get(int(j), 'plain')}): {col} gpu {a[j]!r} oracle {b[j]!r}"); bad += 1
    return bad

bad = tot = 0
n, c, p = 203, 3, 16
for grid in (False, True):
    d, W, y, X0 = base(n, c, p)
    X, names = x_cases(d, W, y, X0)
    cases = {
        "plain": (d, W, y),
        "d = 0": (np.zeros_like(d), W, y),
        "d huge": (d * np.float32(1e10), W, y),
        "d half zero": (np.where(np.arange(n) < n // 2, 0, d).astype(np.float32), W, y),
        "d tiny": (d * np.float32(1e-30), W, y),
        "d with inf": (np.where(np.arange(n) == n - 1, np.inf, d).astype(np.float32), W, y),
        "d with NaN": (np.where(np.arange(n) == 4, np.nan, d).astype(np.float32), W, y),
        "d one negative": (np.where(np.arange(n) == 0, -0.5, d).astype(np.float32), W, y),
        "d negative small": (np.where(np.arange(n) < 3, -1e-4, d).astype(np.float32), W, y),
        "y = 0": (d, W, np.zeros_like(y)),
        "y * 1e20": (d, W, y * np.float32(1e20)),
        "y * 1e-20": (d, W, y * np.float32(1e-20)),
        "y = w0": (d, W, W[:, 0].copy()),
        "y with NaN": (d, W, np.where(np.arange(n) == 7, np.nan, y).astype(np.float32)),
        "W duplicate column": (d, np.concatenate([W[:, :2], W[:, 1:2]], axis=1), y),
        "W zero column": (d, np.concatenate([W[:, :2], np.zeros((n, 1), np.float32)], axis=1), y),
        "W * 1e20": (d, W * np.float32(1e20), y),
        "W with NaN": (d, np.where((np.arange(n) == 9)[:, None] & (np.arange(c) == 1)[None, :], np.nan, W).astype(np.float32), y),
    }
    for tag, (dd, WW, yy) in cases.items():
        b = run(tag, np.ascontiguousarray(dd), np.ascontiguousarray(WW), np.ascontiguousarray(yy), X, names, grid)
        bad += b; tot += 1
        print(f"{'ok  ' if b == 0 else 'BAD '} {tag:22s} grid={grid}", flush=True)
print(f"{tot} degenerate panels x {p} SNPs: {bad} differences")
sys.exit(1 if bad else 0)
