"""Streamed SNP loop of lmm.pygemma (host float32 X, pinned) against the batch size. usage: ab_stream_batch.py n p c sizes..."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygemma_amd import synth, lmm
n, p, c = (int(a) for a in sys.argv[1:4]); sizes = [int(a) for a in sys.argv[4:]]
rng = np.random.default_rng(0)
GK = synth.genotypes(rng, n, n)
K = (GK @ GK.T / n).astype(np.float32)
W = np.concatenate([np.ones((n, 1)), rng.standard_normal((n, c - 1))], axis=1).astype(np.float32)
y = (GK @ (rng.standard_normal(n) * np.sqrt(0.5 / n)).astype(np.float32) + rng.standard_normal(n) * np.sqrt(0.5)).astype(np.float32).reshape(-1, 1)
Xp = lmm.pinned_empty((n, p), np.float32); Xp[:] = rng.binomial(2, 0.3, size=(n, p))
X8 = np.ascontiguousarray(Xp).astype(np.int8)
ref = None
for tag, Xin in (("f32 pinned", Xp), ("int8 pageable", X8)):
    for bs in sizes + sizes[:1]:
        lmm._BATCH_SNPS = bs
        lmm._WORKERS = int(os.environ.get('PG_AB_WORKERS', lmm._WORKERS)); lmm._BATCH_COUNT = int(os.environ.get('PG_AB_COUNT', lmm._BATCH_COUNT)); lmm._SERIAL_KERNELS = bool(int(os.environ.get('PG_AB_SERIAL', '0')))
        if 'PG_AB_PREFETCH' in os.environ: lmm._PREFETCH_MAX = int(float(os.environ['PG_AB_PREFETCH']))
        st = {}
        t = time.time(); df = lmm.pygemma(y, Xin, W, K, stats=st); dt = time.time() - t
        if ref is None: ref = df
        same = bool((df["beta"].to_numpy() == ref["beta"].to_numpy()).all())
        print({k: (round(v, 4) if isinstance(v, float) else v) for k, v in st.items()})
        print(f"{tag:14s} batch {bs:6d}: wall {dt:.3f} s; loop {st.get('seconds', float('nan')):.4f} s; batches {st.get('batches')}; same beta as first run: {same}", flush=True)
