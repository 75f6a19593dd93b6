"""Diagnosis of the rocprofv3 --pmc abort.  usage: diag_pmc2.py MODE [n]
  blas  : create a context, then ONLY a large multi-threaded host sgemm (no kernels of ours)
  blas1 : the same with OpenBLAS held to one thread
  syevd : pg_syevd_dev at n on a symmetric matrix made without host BLAS
A SIGSEGV prints lib(+offset) frames (tools/segv_trace.c)."""
import ctypes, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pygemma_amd import _lib
mode = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
L = _lib.load(); ctx = _lib.Context(0)
T = ctypes.CDLL(os.path.join(ROOT, "tools", "segv_trace.bin")); T.segv_trace_install()
print("mode", mode, "n", n, "threads", os.cpu_count(), flush=True)
rng = np.random.default_rng(1)
if mode.startswith("blas"):
    if mode == "blas1":
        try:
            from threadpoolctl import threadpool_limits; threadpool_limits(1)
        except Exception as ex: print("threadpoolctl:", ex)
    A = rng.standard_normal((n, 2048), dtype=np.float32)
    t = time.time(); K = A @ A.T; print(f"host sgemm {n}x{n}x2048: {time.time()-t:.2f} s, trace {np.trace(K):.3e}", flush=True)
else:
    A = rng.standard_normal((n, n), dtype=np.float32); K = (A + A.T) * 0.5; del A
    dK = ctx.to_device(K); dev = ctx.alloc(n*4); dU = ctx.alloc(n*n*4)
    for rep in range(2):
        t = time.time()
        _lib.check(L.pg_syevd_dev(ctx.handle, n, dK.ptr, dev.ptr, dU.ptr, None, None), "syevd")
        print(f"syevd n={n}: {time.time()-t:.3f} s (run {rep})", flush=True)
print("done", flush=True)
