"""Where lmm.pygemma's time goes for a PAGEABLE float32 X (4 GB): page-locking in place, the loop, unpinning.  usage: probe_pageable.py"""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygemma_amd import synth, lmm, _lib
n, p, c = 10000, 100000, 5
rng = np.random.default_rng(0)
X = np.empty((n, p), np.float32)
for s in range(0, p, 10000):
    X[:, s:s + 10000] = rng.binomial(2, 0.3, size=(n, 10000))
for rep in range(3):
    t = time.time(); h = _lib.pin(X); t1 = time.time() - t
    t = time.time(); h.close(); t2 = time.time() - t
    print(f"hipHostRegister 4 GB: {t1:.3f} s; hipHostUnregister: {t2:.3f} s", flush=True)
rp = synth.fast_rotated_panel(n, 8, c, seed=1)
for rep in range(2):
    st = {}
    t = time.time(); df = lmm.pygemma(rp["Y"], X, rp["W"], rp["d"], eigen=False, stats=st); dt = time.time() - t
    print(f"lmm.pygemma(eigen=False) pageable X: {dt:.3f} s; stats: " + ", ".join(f"{k}={v:.3f}" if isinstance(v, float) else f"{k}={v}" for k, v in st.items()), flush=True)
Xp = lmm.pinned_empty((n, p), np.float32); Xp[:] = X
for rep in range(2):
    st = {}
    t = time.time(); df = lmm.pygemma(rp["Y"], Xp, rp["W"], rp["d"], eigen=False, stats=st); dt = time.time() - t
    print(f"lmm.pygemma(eigen=False) pinned X:   {dt:.3f} s; loop {st['seconds']:.3f} blocks {st['blocks_s']:.3f}", flush=True)
