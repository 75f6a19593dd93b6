"""Debug run of the bulge-chasing kernel with the heartbeat (library built with -DPG_BC_DEBUG): prints where the kernel is after a few seconds."""
import ctypes as C, os, sys, time, threading
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygemma_amd import _lib
L = _lib.load(); ctx = _lib.Context(0)
n = int(sys.argv[1])
hb = _lib.pinned_empty((16,), np.int32); hb[:] = -1
L.pgx_sb2_set_debug(C.c_void_p(hb.ctypes.data))
rng = np.random.default_rng(5)
A = rng.standard_normal((n, n)); A = A + A.T
i, j = np.indices((n, n)); Bm = np.where(np.abs(i - j) <= 64, A, 0.0)
dB = ctx.to_device(Bm); dd, de = ctx.alloc(n * 8), ctx.alloc(n * 8)
flags = (C.c_int * 4)()
done = []
def work():
    rc = L.pgx_sb2_stage2_dev(ctx.handle, n, dB.ptr, dd.ptr, de.ptr, None, flags)
    done.append(rc)
th = threading.Thread(target=work, daemon=True); th.start()
for t in range(8):
    time.sleep(1.0)
    print(t, "heartbeat (s, k, phase):", hb[:3].tolist(), "done:", done, "flags", list(flags), flush=True)
    if done: break
os._exit(0)
