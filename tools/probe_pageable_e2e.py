import sys, time, os
import numpy as np
sys.path.insert(0, '/root/repo')
from pygemma_amd import synth, lmm
n, p, c = 10000, 100000, 5
raw = synth.panel(n, 8, c, seed=3)
rng = np.random.default_rng(0)
for rep in range(3):
    X = np.empty((n, p), np.float32)            # a FRESH pageable array each time (first registration of its range)
    for s in range(0, p, 10000):
        X[:, s:s + 10000] = rng.binomial(2, 0.3, size=(n, 10000))
    st = {}
    t = time.time(); df = lmm.pygemma(raw["Y"], X, raw["W"], raw["K"], stats=st); dt = time.time() - t
    print(f"lmm.pygemma(eigen=True) fresh pageable X: {dt:.3f} s; loop {st['seconds']:.3f}; registered_in_place {st['registered_in_place']}", flush=True)
    del X
