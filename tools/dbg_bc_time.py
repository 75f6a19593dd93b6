"""Phase times of the bulge-chasing kernel (library built with -DPG_BC_DEBUG): workgroup 1's microseconds in wait / load / compute /
store issue / drain+barrier, summed over its steps. usage: dbg_bc_time.py n"""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygemma_amd import _lib
L = _lib.load(); ctx = _lib.Context(0)
n = int(sys.argv[1])
hb = _lib.pinned_empty((16,), np.int32); hb[:] = -1
L.pgx_sb2_set_debug(C.c_void_p(hb.ctypes.data))
rng = np.random.default_rng(5)
Bm = np.zeros((n, n))
for dgl in range(65):
    v = rng.standard_normal(n - dgl)
    Bm += np.diag(v, -dgl) + (np.diag(v, dgl) if dgl else 0)
dB = ctx.to_device(Bm); dd, de = ctx.alloc(n * 8), ctx.alloc(n * 8)
flags = (C.c_int * 4)()
for rep in range(2):
    t = time.time()
    _lib.check(L.pgx_sb2_stage2_dev(ctx.handle, n, dB.ptr, dd.ptr, de.ptr, None, flags), "stage2")
    dt = time.time() - t
nwg = min(256, n // 128 + 4)
steps_wg1 = sum((n - (s + 1) + 63) // 64 for s in range(1, n - 2, nwg))
us = hb[4:12].astype(float)
print(f"n={n}: hook {dt*1e3:.1f} ms; workgroup 1: {steps_wg1} steps; us total wait/load/update/store/drain/right+reflector/dots/reduce = {us.tolist()}  per step = {(us/steps_wg1).round(2).tolist()}")
