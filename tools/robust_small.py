"""Small / odd shapes through the public entry point: nothing may crash or hang; results must agree with an fp64 pipeline + the oracle."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygemma_amd import lmm, synth
from oracle import oracle as O
bad = 0
for n, p, c in [(4, 1, 1), (5, 2, 1), (7, 3, 2), (17, 5, 3), (63, 7, 1), (64, 1, 5), (65, 9, 2), (129, 33, 4), (200, 1, 10), (333, 257, 1), (1001, 65, 6)]:
    raw = synth.panel(n, p, c, seed=n * 7 + p)
    for kw in ({}, {"grid": True}, {"lrt": True}):
        try:
            df = lmm.pygemma(raw["Y"], raw["X"], raw["W"], raw["K"], **kw)
        except Exception as ex:
            print("EXC", n, p, c, kw, repr(ex)[:150]); bad += 1; continue
        K64 = np.tril(raw["K"].astype(np.float64)); K64 = K64 + np.tril(K64, -1).T
        d, U = np.linalg.eigh(K64)
        rot = lambda A: (U.T @ A.astype(np.float64)).astype(np.float32)
        tr = O.calculate(np.maximum(d, 0).astype(np.float32), rot(raw["Y"]), rot(raw["W"]), rot(raw["X"]), grid=bool(kw.get("grid")), order=0, nthreads=4)
        b, t = df["beta"].to_numpy().astype(np.float64), tr["beta"].astype(np.float64)
        ok = np.isfinite(t)
        rel = np.abs(b[ok] - t[ok]) / np.maximum(np.abs(t[ok]), 1e-6)
        flag = "" if (len(rel) == 0 or np.median(rel) < 5e-3) else "  <-- CHECK"
        if flag: bad += 1
        print(f"n={n:5d} p={p:4d} c={c:2d} {str(kw):16s} rows {len(df)} finite {int(np.isfinite(b).sum())}/{p} median rel dbeta {np.median(rel) if len(rel) else float('nan'):.2e}{flag}", flush=True)
print("problems:", bad)
