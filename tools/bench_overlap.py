"""Does the matrix-pipe rotation overlap with the VALU-bound association kernel when they run on two streams?
usage: bench_overlap.py [n] [p] [c] [iters]"""
import sys, time, threading, ctypes as C
import numpy as np
sys.path.insert(0, '/root/repo')
from pygemma_amd import _lib, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
p = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
c = int(sys.argv[3]) if len(sys.argv) > 3 else 5
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 10
L = _lib.load()
rng = np.random.default_rng(0)
ldx = (n + 63) // 64 * 64
c1, c2 = _lib.Context(0), _lib.Context(0)
U = np.linalg.qr(rng.standard_normal((n, n)).astype(np.float32))[0].astype(np.float32)
X = synth.genotypes(rng, n, p)
rp = synth.fast_rotated_panel(n, 64, c)
dU, dX = c1.to_device(U), c1.to_device(X)
dprep = c1.alloc(L.pg_geno_prep_bytes(n)); dwork = c1.alloc(L.pg_geno_work_bytes(n, p))
_lib.check(L.pg_geno_prep_dev(c1.handle, n, dU.ptr, n, dprep.ptr), "prep")
dXr1 = c1.alloc(p * ldx * 4); dXr2 = c2.alloc(p * ldx * 4)
dd, dW, dy = c2.to_device(rp["d"]), c2.to_device(rp["W"]), c2.to_device(rp["Y"])
out, F = c2.alloc(p * 16), c2.alloc(p * 16)
flag = C.c_int(0)
def rot():
    _lib.check(L.pg_rotate_geno_dev(c1.handle, n, p, dprep.ptr, dX.ptr, p, dXr1.ptr, ldx, dwork.ptr, C.byref(flag)), "rot")
def assoc():
    _lib.check(L.pg_assoc_dev(c2.handle, n, c, p, dd.ptr, dW.ptr, dy.ptr, dXr2.ptr, ldx, 0, out.ptr, out.ptr + 4*p, out.ptr + 8*p, out.ptr + 12*p, F.ptr, F.ptr + 8*p, None), "assoc")
rot(); c1.sync()
L.pg_memcpy_h2d  # noqa
# give assoc a real rotated block
import ctypes
_lib.check(L.pg_rotate_geno_dev(c1.handle, n, p, dprep.ptr, dX.ptr, p, dXr2.ptr, ldx, dwork.ptr, C.byref(flag)), "rot"); c1.sync()
assoc(); c2.sync()
t = time.time()
for _ in range(iters): rot()
c1.sync(); t_rot = (time.time() - t) / iters
t = time.time()
for _ in range(iters): assoc()
c2.sync(); t_as = (time.time() - t) / iters
def loop(f, ctx):
    for _ in range(iters): f()
    ctx.sync()
t = time.time()
th = [threading.Thread(target=loop, args=(rot, c1)), threading.Thread(target=loop, args=(assoc, c2))]
[x.start() for x in th]; [x.join() for x in th]
t_both = (time.time() - t) / iters
print(f"rotate alone {t_rot*1e3:.2f} ms, assoc alone {t_as*1e3:.2f} ms, sum {1e3*(t_rot+t_as):.2f} ms; both streams concurrently {t_both*1e3:.2f} ms per pair "
      f"-> {p/t_both:.0f} SNPs/s vs {p/(t_rot+t_as):.0f} sequential")
