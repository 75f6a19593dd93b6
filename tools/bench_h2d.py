"""Host->device paths for one SNP batch (column window of a row-major (n, p) float32 matrix): GB/s of each way to bring it in.
usage: bench_h2d.py [n] [p] [pb]"""
import ctypes as C, sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygemma_amd import _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
p = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
pb = int(sys.argv[3]) if len(sys.argv) > 3 else 32768
L = _lib.load(); ctx = _lib.Context(0)
X = np.ones((n, p), np.float32)
t = time.time(); Xp = _lib.pinned_empty((n, p), np.float32); t_alloc = time.time() - t
t = time.time(); Xp[:] = X; t_fill = time.time() - t
print(f"hipHostMalloc {Xp.nbytes/1e9:.1f} GB: {t_alloc:.3f} s; fill {t_fill:.3f} s", flush=True)
dX = ctx.alloc(n * pb * 4)
nb = n * pb * 4
def timeit(name, fn, reps=3):
    fn(); ctx.sync()
    t = time.time()
    for _ in range(reps): fn()
    ctx.sync(); dt = (time.time() - t) / reps
    print(f"{name:58s} {dt*1e3:8.1f} ms  {nb/dt/1e9:6.1f} GB/s", flush=True)
timeit("2D async, pinned X (pitch p*4 -> pb*4)", lambda: _lib.check(L.pg_memcpy2d_h2d_async(ctx.handle, dX.ptr, pb*4, Xp.ctypes.data, p*4, pb*4, n), "a"))
timeit("2D sync, pageable X (round-1 path)", lambda: _lib.check(L.pg_memcpy2d_h2d(ctx.handle, dX.ptr, pb*4, X.ctypes.data, p*4, pb*4, n), "b"))
stg = C.c_void_p(); t = time.time(); _lib.check(L.pg_host_alloc(ctx.handle, nb, C.byref(stg)), "alloc"); print(f"staging hipHostMalloc {nb/1e9:.2f} GB: {time.time()-t:.3f} s")
for thr in (1, 4, 8, 16, 32):
    timeit(f"pg_stage_rows({thr} thr) pageable -> pinned staging only", lambda: _lib.check(L.pg_stage_rows(stg, pb*4, X.ctypes.data, p*4, pb*4, n, thr), "s"))
timeit("1D async from pinned staging", lambda: _lib.check(L.pg_memcpy_h2d_async(ctx.handle, dX.ptr, stg, nb), "c"))
def both():
    _lib.check(L.pg_stage_rows(stg, pb*4, X.ctypes.data, p*4, pb*4, n, 16), "s"); _lib.check(L.pg_memcpy_h2d_async(ctx.handle, dX.ptr, stg, nb), "c")
timeit("stage(16 thr) + 1D async, serial", both)
t = time.time(); h = _lib.pin(X); print(f"hipHostRegister {X.nbytes/1e9:.1f} GB: {time.time()-t:.3f} s")
timeit("2D async, registered X", lambda: _lib.check(L.pg_memcpy2d_h2d_async(ctx.handle, dX.ptr, pb*4, X.ctypes.data, p*4, pb*4, n), "a"))
# row-chunked 1D copies from pinned X (n separate async copies of pb*4 bytes)
def rows():
    for i in range(0, n):
        L.pg_memcpy_h2d_async(ctx.handle, dX.ptr + i*pb*4, X.ctypes.data + i*p*4, pb*4)
timeit("n x 1D async row copies, registered X", rows, reps=1)
h.close()
