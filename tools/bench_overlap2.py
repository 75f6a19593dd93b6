"""Rotation of batch k+1 on one stream under the association of batch k on another (two Xr buffers, event-ordered), against the same
batches back to back on one stream.  usage: bench_overlap2.py [n] [B] [nbatch]"""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygemma_amd import _lib, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
NB = int(sys.argv[3]) if len(sys.argv) > 3 else 6
c = 5
L = _lib.load(); A = _lib.Context(0); Bc = _lib.Context(0)
rp = synth.fast_rotated_panel(n, 8, c)
rng = np.random.default_rng(0)
U = np.linalg.qr(rng.standard_normal((n, n)))[0].astype(np.float32)
X = rng.binomial(2, 0.3, size=(n, B)).astype(np.float32)
ldx = (n + 63) // 64 * 64
dU, dX = A.to_device(U), A.to_device(X)
dprep, dwork = A.alloc(L.pg_geno_prep_bytes(n)), A.alloc(L.pg_geno_work_bytes(n, B))
_lib.check(L.pg_geno_prep_dev(A.handle, n, dU.ptr, n, dprep.ptr), "prep")
dXr = [A.alloc(B * ldx * 4), A.alloc(B * ldx * 4)]
dd, dW, dy = A.to_device(rp["d"]), A.to_device(rp["W"]), A.to_device(rp["Y"])
out, F = A.alloc(B * 16), A.alloc(B * 16)
def ev(ctx):
    e = C.c_void_p(); _lib.check(L.pg_event_create(ctx.handle, C.byref(e)), "ev"); return e
rot_done = [ev(A) for _ in range(NB)]; as_done = [ev(Bc) for _ in range(NB)]
def rotate(ctx, k): _lib.check(L.pg_rotate_auto_dev(ctx.handle, n, B, dU.ptr, n, dprep.ptr, dX.ptr, B, dXr[k % 2].ptr, ldx, dwork.ptr, None), "rot")
def assoc(ctx, k): _lib.check(L.pg_assoc_dev(ctx.handle, n, c, B, dd.ptr, dW.ptr, dy.ptr, dXr[k % 2].ptr, ldx, 0, out.ptr, out.ptr + 4*B, out.ptr + 8*B, out.ptr + 12*B, F.ptr, F.ptr + 8*B, None), "assoc")
def serial():
    for k in range(NB): rotate(A, k); assoc(A, k)
    A.sync()
def overlapped():
    for k in range(NB):
        if k >= 2: L.pg_stream_wait_event(A.handle, as_done[k - 2])      # Xr[k%2] free again
        rotate(A, k); L.pg_event_record(A.handle, rot_done[k])
        L.pg_stream_wait_event(Bc.handle, rot_done[k]); assoc(Bc, k); L.pg_event_record(Bc.handle, as_done[k])
    A.sync(); Bc.sync()
for name, fn in (("serial", serial), ("overlapped", overlapped), ("serial", serial), ("overlapped", overlapped)):
    fn(); t = time.time(); fn(); dt = time.time() - t
    print(f"{name:11s} {NB} batches of {B}: {dt*1e3:.2f} ms -> {NB*B/dt/1e6:.3f} M SNPs/s", flush=True)
