"""Device-memory and host-RSS drift over repeated calls of the public entry points (hipMemGetInfo queried through libamdhip64
directly — a diagnostic, not part of the product).  usage: leak_check.py [rounds]"""
import ctypes as C, os, sys, resource
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygemma_amd import lmm, synth, _lib, ops
hip = C.CDLL("libamdhip64.so")
def free_bytes():
    f, t = C.c_size_t(), C.c_size_t()
    assert hip.hipMemGetInfo(C.byref(f), C.byref(t)) == 0
    return f.value
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 25
n, p, c = 1000, 3000, 4
raw = synth.panel(n, p, c, seed=1)
X8 = np.clip(np.round(raw["X"]), -128, 127).astype(np.int8)
G = np.random.default_rng(0).binomial(2, 0.3, size=(n, 2 * n)).astype(np.float32)
ctx = _lib.Context(0)
def one():
    lmm.pygemma(raw["Y"], raw["X"], raw["W"], raw["K"])
    lmm.pygemma(raw["Y"], raw["X"], raw["W"], raw["K"], grid=True, lrt=True)
    lmm.pygemma(raw["Y"], X8, raw["W"], raw["K"])
    lmm.kinship(G)
    ops.syevd(raw["K"], ctx=ctx)
    try:
        ops.assoc(np.ones(n, np.float32), np.ones((n, 31), np.float32), raw["Y"], raw["X"][:, :4], ctx=ctx)   # an error path
    except _lib.PgError:
        pass
one(); one()
f0, r0 = free_bytes(), resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
hist = []
for k in range(rounds):
    one()
    hist.append(f0 - free_bytes())
    if k % 5 == 4:
        print(f"round {k + 1}: device bytes held beyond the baseline {hist[-1] / 1e6:.2f} MB; host max RSS +{(resource.getrusage(resource.RUSAGE_SELF).ru_maxrss - r0) / 1e3:.1f} MB", flush=True)
drift = (hist[-1] - hist[len(hist) // 2]) / max(1, len(hist) - len(hist) // 2 - 1)
print(f"device drift per round over the second half: {drift / 1e6:.4f} MB")
sys.exit(1 if drift > 1e6 else 0)
