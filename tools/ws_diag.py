"""Where the wave-specialised GEMM's consumer and helper spend their time (workgroup 100; PG_DGEMM_TUNE bit 3 set here)."""
import os, sys, ctypes as C
os.environ["PG_DGEMM_TUNE"] = "8"
import numpy as np
sys.path.insert(0, '/root/repo')
from pygemma_amd import _lib
L = _lib.load(); ctx = _lib.Context(0)
L.pgx_ring_stamps.argtypes = [C.c_void_p]
rng = np.random.default_rng(0)
st = (C.c_longlong * 64)(); _lib.check(L.pgx_ring_stamps(st), "stamps")
n = 4096
dA = ctx.to_device(rng.standard_normal((n, n))); dB = ctx.to_device(rng.standard_normal((n, n))); dC = ctx.to_device(np.zeros((n, n)))
for K in (4096, 256):
    for rep in range(2):
        _lib.check(L.pgx_dgemm_ex_dev(ctx.handle, 0, 0, n, n, K, 1.0, dA.ptr, n, dB.ptr, n, 0.0, dC.ptr, n), "dgemm"); ctx.sync()
        _lib.check(L.pgx_ring_stamps(st), "stamps"); s = list(st)
        print(f"K={K}: consumer total {s[41]} cycles for {s[47]} chunks = {s[41]/max(s[47],1):.0f} per chunk; waiting for chunks {s[40]} ({s[46]} sleeps) | "
              f"helper total {s[45]}, pump calls {s[43]}, issues {s[44]} ({s[48]/max(s[44],1):.0f} cycles each), landing waits {s[42]}")
