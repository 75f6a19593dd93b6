"""fp64 MFMA GEMM rates at the shapes the eigensolver uses (pgx_dgemm_ex_dev: every mode of csrc/dgemm.hpp).
usage: bench_dgemm_ex.py [reps]   — prints one line per shape: ms per call (reps calls back to back, one sync), useful TF, fraction of 78.6"""
import sys, time, ctypes as C
import numpy as np
sys.path.insert(0, '/root/repo')
from pygemma_amd import _lib
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
only = sys.argv[2] if len(sys.argv) > 2 else ""      # substring filter on the shape name
L = _lib.load(); ctx = _lib.Context(0)
rng = np.random.default_rng(0)
TA, TB, LOW, SYM, NOSPLIT = 1, 2, 4, 8, 16

def bench(name, flags, kxor, M, N, K, beta, lda, ldb, ldc, a_shape, b_shape, c_shape, flops):
    if only and only not in name:
        return
    dA = ctx.to_device(rng.standard_normal(a_shape)); dB = dA if b_shape is None else ctx.to_device(rng.standard_normal(b_shape))
    dC = ctx.to_device(np.zeros(c_shape))
    def run():
        _lib.check(L.pgx_dgemm_ex_dev(ctx.handle, flags, kxor, M, N, K, 1.0e-3, dA.ptr, lda, dB.ptr, ldb, beta, dC.ptr, ldc), name)
    run(); ctx.sync()
    best = 1e9
    for _ in range(3):
        t = time.time()
        for _ in range(reps): run()
        ctx.sync(); best = min(best, (time.time() - t) / reps)
    print(f"{name:58s} {best*1e3:8.3f} ms  {flops/best/1e12:6.1f} TF  {flops/best/78.6e12:5.2f}", flush=True)
    dA.free(); dC.free()
    if dB is not dA: dB.free()

for m in (9984, 7424, 4992, 2560):
    ldt = m + 128
    # rank-128 update on the lower triangle: A = B = [V W]' (128 x ldt, k-major), B read with k ^ 64
    bench(f"update  C({m}x{m}, lower) -= [VW][WV]'  K=128", TA | LOW, 64, m, m, 128, 1.0, ldt, ldt, m, (128, ldt), None, (m, m), 2.0 * (m * m / 2) * 128)
    # X = A22 V, A22 symmetric from its lower triangle
    bench(f"symX    X({m}x64) = A22 V              K={m}", SYM, 0, m, 64, m, 0.0, m, 128, 64, (m, m), (m, 128), (m, 64), 2.0 * m * m * 64)
for m in (9936, 4992):
    bench(f"Q1 W = V'Z   (256 x 10000)             K={m}", TA, 0, 256, 10000, m, 0.0, 10000, 10000, 10000, (m, 10000), (m, 10000), (256, 10000), 2.0 * 256 * 10000 * m)
    bench(f"Q1 Z -= V W2 ({m} x 10000)            K=256", 0, 0, m, 10000, 256, 1.0, 10000, 10000, 10000, (m, 10000), (256, 10000), (m, 10000), 2.0 * 256 * 10000 * m)
for m in (9936, 4992):      # the same product with V handed over transposed (k-major A): is the m-major operand the slower one?
    bench(f"Q1 Z -= Vt' W2 ({m} x 10000), A k-major  K=256", TA, 0, m, 10000, 256, 1.0, m, 10000, 10000, (256, m), (256, 10000), (m, 10000), 2.0 * 256 * 10000 * m)
bench("square 8192^3", 0, 0, 8192, 8192, 8192, 0.0, 8192, 8192, 8192, (8192, 8192), (8192, 8192), (8192, 8192), 2.0 * 8192 ** 3)
bench("square 8192^3, A k-major", TA, 0, 8192, 8192, 8192, 0.0, 8192, 8192, 8192, (8192, 8192), (8192, 8192), (8192, 8192), 2.0 * 8192 ** 3)
bench("D&C merge  Q(10000x5000) U(5000x5000)", 0, 0, 10000, 5000, 5000, 0.0, 5000, 5000, 5000, (10000, 5000), (5000, 5000), (10000, 5000), 2.0 * 10000 * 5000 * 5000)
bench("Gram  P'P (64x64)                      K=9936", TA, 0, 64, 64, 9936, 0.0, 10000, 10000, 64, (9936, 10000), None, (64, 64), 2.0 * 64 * 64 * 9936)
bench("thin  Q = P R^-1 (9936x64)             K=64", 0, 0, 9936, 64, 64, 0.0, 10000, 64, 64, (9936, 10000), (64, 64), (9936, 64), 2.0 * 9936 * 64 * 64)
