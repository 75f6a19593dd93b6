import sys, ctypes as C, numpy as np, scipy.linalg as sl
sys.path.insert(0, '/root/repo')
from pygemma_amd import _lib, ops
L = _lib.load(); ctx = _lib.Context(0)
L.pgx_sytrd_dev.argtypes = [C.c_void_p, C.c_int64] + [C.c_void_p] * 5
L.pgx_stedc_dev.argtypes = [C.c_void_p, C.c_int64] + [C.c_void_p] * 4
n = 601
rng = np.random.default_rng(1)
A = rng.standard_normal((n, n))
for scale in (1.0, 1e-10, 1e-20, 1e-30):
    K = ((A @ A.T) * scale).astype(np.float32)
    K64 = np.tril(K.astype(np.float64)); K64 = K64 + np.tril(K64, -1).T
    ref = np.linalg.eigvalsh(K64); lmax = np.abs(ref).max()
    dK = ctx.to_device(K)
    dd, de, dt, dV = ctx.alloc(n * 8), ctx.alloc(n * 8), ctx.alloc(n * 8), ctx.alloc(n * n * 8)
    _lib.check(L.pgx_sytrd_dev(ctx.handle, n, dK.ptr, dd.ptr, de.ptr, dt.ptr, dV.ptr), "sytrd")
    d, e = dd.download((n,), np.float64), de.download((n,), np.float64)[: n - 1]
    got = sl.eigvalsh_tridiagonal(d, e)
    e2 = np.ascontiguousarray(np.concatenate([e, [0.0]])); ev = np.empty(n); dZ = ctx.alloc(n * n * 8)
    _lib.check(L.pgx_stedc_dev(ctx.handle, n, d.ctypes.data, e2.ctypes.data, ev.ctypes.data, dZ.ptr), "stedc")
    ev32, U32, ev64, U = ops.syevd(K, ctx=ctx, want64=True)
    print(f"scale {scale:g}: sytrd eig err {np.abs(got-ref).max()/lmax:.2e}; stedc vs T err {np.abs(ev-got).max()/lmax:.2e}; syevd err {np.abs(ev64-ref).max()/lmax:.2e}; orth {np.abs(U.T@U-np.eye(n)).max():.2e}")
