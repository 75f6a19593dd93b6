"""Where one tile of the ring GEMM spends its time: in-kernel stamps of workgroup 300 (PG_DGEMM_TUNE=8 set here).
usage: ring_stamps.py [update|q1|sym]"""
import os, sys, ctypes as C
os.environ["PG_DGEMM_TUNE"] = str(8 | int(os.environ.get("PG_DGEMM_TUNE", "0")))
import numpy as np
sys.path.insert(0, '/root/repo')
from pygemma_amd import _lib
L = _lib.load(); ctx = _lib.Context(0)
L.pgx_ring_stamps.argtypes = [C.c_void_p]
rng = np.random.default_rng(0)
which = sys.argv[1] if len(sys.argv) > 1 else "update"
m = 9984
if which == "update":
    ldt = m + 128
    dA = ctx.to_device(rng.standard_normal((128, ldt))); dC = ctx.to_device(np.zeros((m, m)))
    call = lambda: L.pgx_dgemm_ex_dev(ctx.handle, 1 | 4, 64, m, m, 128, -1.0, dA.ptr, ldt, dA.ptr, ldt, 1.0, dC.ptr, m)
elif which == "q1":
    dA = ctx.to_device(rng.standard_normal((m, 10000))); dB = ctx.to_device(rng.standard_normal((256, 10000))); dC = ctx.to_device(np.zeros((m, 10000)))
    call = lambda: L.pgx_dgemm_ex_dev(ctx.handle, 0, 0, m, 10000, 256, -1.0, dA.ptr, 10000, dB.ptr, 10000, 1.0, dC.ptr, 10000)
else:
    dA = ctx.to_device(rng.standard_normal((m, m))); dB = ctx.to_device(rng.standard_normal((m, 128))); dC = ctx.to_device(np.zeros((m, 64)))
    call = lambda: L.pgx_dgemm_ex_dev(ctx.handle, 8, 0, m, 64, m, 1.0, dA.ptr, m, dB.ptr, 128, 0.0, dC.ptr, 64)
st = (C.c_longlong * 64)(); _lib.check(L.pgx_ring_stamps(st), "stamps")     # makes the buffer
for rep in range(4):
    _lib.check(call(), "dgemm"); ctx.sync()
    st = (C.c_longlong * 64)(); _lib.check(L.pgx_ring_stamps(st), "stamps"); s = list(st)
    t0 = s[0]
    d = lambda a, b: s[b] - s[a]
    print(f"[{which} rep {rep}] cycles: issue prologue {d(0,1)} | first chunk landed {d(1,2)} | chunk 0 -> 1 {d(2,3)} | chunks 1..8 {d(3,4)} (per chunk {d(3,4)/7:.0f}) | "
          f"rest of main loop {d(4,5)} | epilogue {d(5,6)} | total {d(0,6)}")
    print(f"     chunk 8: wait {d(32,33)} | barrier {s[4]-s[33]} | issue DMA {s[34]-s[4]} | prefetch {d(34,35)} | reads + 32 MFMA issued {d(35,36)} | -> chunk 9 done {d(36,37)}")
    print("     epilogue groups: wait+barrier | work+store issue:", " ".join(f"{s[8+2*g]-(s[9+2*(g-1)] if g else s[5])}|{s[9+2*g]-s[8+2*g]}" for g in range(8)))
