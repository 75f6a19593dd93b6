"""Streamed SNP loop of lmm.pygemma (pinned float32 X, eigenpairs cached through eigen=False-like reuse is not possible: K solved once per call)
against the number of workers per GPU and the batch count.  usage: ab_workers.py n p c"""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygemma_amd import synth, lmm
n, p, c = (int(a) for a in sys.argv[1:4])
rng = np.random.default_rng(0)
GK = synth.genotypes(rng, n, n)
K = (GK @ GK.T / n).astype(np.float32)
W = np.concatenate([np.ones((n, 1)), rng.standard_normal((n, c - 1))], axis=1).astype(np.float32)
y = (GK @ (rng.standard_normal(n) * np.sqrt(0.5 / n)).astype(np.float32) + rng.standard_normal(n) * np.sqrt(0.5)).astype(np.float32).reshape(-1, 1)
Xp = lmm.pinned_empty((n, p), np.float32)
for s in range(0, p, 10000):
    Xp[:, s:s + 10000] = rng.binomial(2, 0.3, size=(n, min(10000, p - s)))
ref = None
for workers, count, serial, bmin in ((2, 12, True, 8192), (2, 24, True, 4096), (2, 32, True, 3072), (2, 48, True, 2048), (2, 24, False, 4096), (3, 24, True, 4096)):
    lmm._WORKERS, lmm._BATCH_COUNT, lmm._SERIAL_KERNELS, lmm._BATCH_MIN = workers, count, serial, bmin
    best = None
    for rep in range(2):
        st = {}
        t = time.time(); df = lmm.pygemma(y, Xp, W, K, stats=st); dt = time.time() - t
        if ref is None: ref = df
        assert (df["beta"].to_numpy() == ref["beta"].to_numpy()).all()
        if best is None or st["blocks_s"] < best[0]: best = (st["blocks_s"], dt, st["batches"], st.get("kernel_s"), st.get("dma_s"), st.get("token_s"))
    print(f"workers {workers} batch_count {count} serial {serial} min {bmin}: loop {best[0]*1e3:6.1f} ms = {4.0*n*p/best[0]/1e9:5.1f} GB/s; wall {best[1]:.3f} s; batches {best[2]}; kernel_s {best[3]:.3f} dma_s {best[4]:.3f} token_s {best[5]:.3f} (sums over batches)", flush=True)
