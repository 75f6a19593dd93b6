"""Degenerate panels (synth.degenerate_panels) through lmm.pygemma(..., lrt=True, eigen=False) against the oracle's LRT in the
kernels' order: same NaN rows; l_alt / l_null within one float32 ulp, p_lrt within 2e-3 (the tolerance tier of tests/test_gpu_lrt.py)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygemma_amd import lmm, synth
from oracle import oracle as O
bad = 0
for tag, d, W, y, X in synth.degenerate_panels():
    dc = np.maximum(np.float32(0), d)
    try:
        df = lmm.pygemma(y.reshape(-1, 1), X, W, d, eigen=False, lrt=True)
    except Exception as ex:
        print(f"EXC {tag}: {ex!r}"[:200]); bad += 1; continue
    o = O.calculate_lrt(dc, y, W, X, order=1, nthreads=4)
    msg = []
    for col in ("l_alt", "D_lrt", "p_lrt"):
        a, b = df[col].to_numpy().astype(np.float64), np.asarray(o[col], np.float64)
        if (np.isnan(a) != np.isnan(b)).any():
            msg.append(f"{col}: NaN pattern differs at {np.nonzero(np.isnan(a) != np.isnan(b))[0][:6]}")
            continue
        fin = np.isfinite(a) & np.isfinite(b)
        if col == "l_alt" and fin.any():
            ulp = np.spacing(np.abs(b[fin]).astype(np.float32)).astype(np.float64)
            w = (np.abs(a[fin] - b[fin]) / ulp).max()
            if w > 1.0: msg.append(f"l_alt off by {w:.1f} ulp")
        if col == "p_lrt" and fin.any():
            w = np.abs(a[fin] / np.maximum(b[fin], 1e-300) - 1).max()
            if w > 2e-3 and np.abs(a[fin] - b[fin]).max() > 1e-12: msg.append(f"p_lrt rel {w:.1e}")
        if ((np.isinf(a) | np.isinf(b)) & (a != b) & ~np.isnan(a)).any():
            msg.append(f"{col}: inf pattern differs")
    ln_a, ln_b = float(df["l_null"].to_numpy()[0]), float(o["l_null"])
    if not ((np.isnan(ln_a) and np.isnan(ln_b)) or ln_a == ln_b or abs(ln_a - ln_b) <= np.spacing(np.float32(abs(ln_b)))):
        msg.append(f"l_null {ln_a!r} vs {ln_b!r}")
    bad += bool(msg)
    print(f"{'BAD' if msg else 'ok '} {tag:22s} NaN rows {int(np.isnan(df['l_alt'].to_numpy()).sum()):2d}  {'; '.join(msg)}", flush=True)
print("problems:", bad)
