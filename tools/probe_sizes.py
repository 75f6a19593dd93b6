"""pg_syevd_dev called repeatedly in ONE process over a list of sizes (phase times: run with PG_SYEVD_TIMING=1): the n = 1940 outlier
of profiles/r03_other_sizes.txt (second call 0.089 s, dense->band 61 ms).  usage: probe_sizes.py n [n ...]"""
import sys, time
import numpy as np
sys.path.insert(0, '/root/repo')
from pygemma_amd import _lib, synth
L = _lib.load(); ctx = _lib.Context(0)
Ks = {}
for n in [int(a) for a in sys.argv[1:]]:
    if n not in Ks:
        rng = np.random.default_rng(n)
        G = rng.standard_normal((n, 2 * n)).astype(np.float32)
        Ks[n] = (G @ G.T / (2 * n)).astype(np.float32)
    dK = ctx.to_device(Ks[n]); dev = ctx.alloc(n * 4); dU = ctx.alloc(n * n * 4)
    ctx.sync()
    t = time.time()
    _lib.check(L.pg_syevd_dev(ctx.handle, n, dK.ptr, dev.ptr, dU.ptr, None, None), "syevd")
    print(f"== n={n}: {1e3 * (time.time() - t):.1f} ms", file=sys.stderr, flush=True)
    for b in (dK, dev, dU): b.free()
