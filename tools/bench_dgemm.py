"""fp64 MFMA GEMM rate of the eigensolver's dgemm. usage: bench_dgemm.py M N K [transA] [beta]"""
import sys, time, ctypes as C
import numpy as np
sys.path.insert(0, '/root/repo')
from pygemma_amd import _lib
M, N, K = (int(a) for a in sys.argv[1:4]); ta = int(sys.argv[4]) if len(sys.argv) > 4 else 0; beta = float(sys.argv[5]) if len(sys.argv) > 5 else 0.0
L = _lib.load(); ctx = _lib.Context(0)
L.pgx_dgemm_dev.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_double, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_double, C.c_void_p, C.c_int64]
rng = np.random.default_rng(0)
dA = ctx.to_device(rng.standard_normal((K, M) if ta else (M, K))); dB = ctx.to_device(rng.standard_normal((K, N))); dC = ctx.to_device(np.zeros((M, N)))
lda = M if ta else K
def run():
    _lib.check(L.pgx_dgemm_dev(ctx.handle, ta, M, N, K, 1.0, dA.ptr, lda, dB.ptr, N, beta, dC.ptr, N), "dgemm"); ctx.sync()
run(); ts = []
for _ in range(5):
    t = time.time(); run(); ts.append(time.time() - t)
t = min(ts); print(f"dgemm M={M} N={N} K={K} transA={ta}: {t*1e3:.2f} ms  {2.0*M*N*K/t/1e12:.1f} TFLOP/s (fp64 MFMA peak 78.6)")
