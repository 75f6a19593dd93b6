"""lmm.pygemma end to end from a pinned float32 X with and without the copy of X under the eigensolver (PYGEMMA_PREFETCH_MAX), repeated
in ONE long-lived process (r2 saw one run in four stall 0.15 s in the first allocations after the eigensolver).  usage: probe_prefetch.py"""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygemma_amd import synth, lmm
n, p, c = 10000, 100000, 5
raw = synth.panel(n, 8, c, seed=3)
rng = np.random.default_rng(0)
Xp = lmm.pinned_empty((n, p), np.float32)
for s in range(0, p, 10000):
    Xp[:, s:s + 10000] = rng.binomial(2, 0.3, size=(n, 10000))
ref = None
for mode in (0, int(1e11), 0, int(1e11)):
    lmm._PREFETCH_MAX = mode
    ts = []
    for rep in range(6):
        st = {}
        t = time.time(); df = lmm.pygemma(raw["Y"], Xp, raw["W"], raw["K"], stats=st); ts.append(time.time() - t)
        if ref is None: ref = df
        assert (df["beta"].to_numpy() == ref["beta"].to_numpy()).all()
    print(f"prefetch_max {mode:.0e}: wall " + " ".join(f"{t:.3f}" for t in ts) + f" s; last loop {st['seconds']:.3f} s, prefetched batches {st.get('prefetched_batches', 0)} of {st['batches']}", flush=True)
