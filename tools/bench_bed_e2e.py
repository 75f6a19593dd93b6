"""End-to-end lmm.pygemma from a packed PLINK .bed image (random genotypes, ~1 % missing calls), one GPU.
usage: bench_bed_e2e.py n p c"""
import sys, time
import numpy as np
sys.path.insert(0, '/root/repo')
from pygemma_amd import synth
from pygemma_amd.bed import PackedBed
from pygemma import lmm
n, p, c = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
rng = np.random.default_rng(0)
t = time.time()
bpr = (n + 3) // 4
# 2-bit codes drawn per call: 00 hom A1 (49 %), 10 het (40 %), 11 hom A2 (10 %), 01 missing (1 %)
lut = np.array([0] * 49 + [2] * 40 + [3] * 10 + [1], np.uint8)
data = np.empty((p, bpr), np.uint8)
for s in range(0, p, 50000):
    e = min(p, s + 50000)
    q = lut[rng.integers(0, 100, size=(e - s, bpr, 4), dtype=np.uint8)]
    data[s:e] = q[:, :, 0] | (q[:, :, 1] << 2) | (q[:, :, 2] << 4) | (q[:, :, 3] << 6)
bed = PackedBed(data, n)
GK = synth.genotypes(rng, n, 2 * n)
K = (GK @ GK.T / (2 * n)).astype(np.float32)
W = np.concatenate([np.ones((n, 1)), rng.standard_normal((n, c - 1))], axis=1).astype(np.float32)
y = (GK @ (rng.standard_normal(2 * n) * np.sqrt(0.5 / (2 * n))).astype(np.float32) + rng.standard_normal(n) * np.sqrt(0.5)).astype(np.float32).reshape(-1, 1)
print(f"inputs: {time.time()-t:.1f} s; packed genotypes {data.nbytes/1e9:.2f} GB (float32 X would be {4.0*n*p/1e9:.1f} GB)", flush=True)
for rep in range(2):
    t = time.time()
    df = lmm.pygemma(y, bed, W, K, verbose=0)
    dt = time.time() - t
    print(f"pygemma(PackedBed n={n}, p={p}, c={c}): {dt:.2f} s wall -> {p/dt:.0f} SNPs/s end-to-end (eigh + upload + decode + rotate + assoc + p)", flush=True)
print(df.describe().loc[["min", "max"]])
