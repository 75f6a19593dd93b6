"""Phase timing inside rotate_geno_i8_kernel: s_memtime stamps of workgroup 0, stages 8..23, wave 0 (early group) and wave 4 (late group).
Needs the stamped build: tools/build_variant.sh stamps rotate_geno -DPG_GENO_STAMPS, PYGEMMA_HIP_LIB=pygemma_amd/lib_dev/stamps/libpygemma_hip.so
usage: geno_i8_stamps.py [n] [p]       (s_memtime ticks at 100 MHz: 10 ns)"""
import sys, ctypes as C
import numpy as np
sys.path.insert(0, '/root/repo')
from pygemma_amd import _lib, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
p = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
L = _lib.load(); ctx = _lib.Context(0)
rng = np.random.default_rng(0)
U = rng.standard_normal((n, n), dtype=np.float32) / np.sqrt(n)
X = synth.genotypes(rng, n, p)
ldx = (n + 63) // 64 * 64
dU, dX = ctx.to_device(U), ctx.to_device(X); dXr = ctx.alloc(p * ldx * 4)
dprep = ctx.alloc(L.pg_geno_prep_bytes(n)); dwork = ctx.alloc(L.pg_geno_work_bytes(n, p))
_lib.check(L.pg_geno_prep_dev(ctx.handle, n, dU.ptr, n, dprep.ptr), "prep")
ok = C.c_int(0)
for _ in range(3):
    _lib.check(L.pg_rotate_geno_dev(ctx.handle, n, p, dprep.ptr, dX.ptr, p, dXr.ptr, ldx, dwork.ptr, C.byref(ok)), "rot"); ctx.sync()
buf = (C.c_longlong * 192)()
L.pgx_geno_stamps.argtypes = [C.c_void_p]; L.pgx_geno_stamps.restype = C.c_int
assert L.pgx_geno_stamps(buf) == 0
st = np.array(buf, dtype=np.int64).reshape(2, 16, 6)
t0 = st[0, 0, 0]
names = ["mem start", "reads issued (+vm wait, late)", "lgkm landed", "after barrier", "mfma issued", "A landed (early)"]
for g, gn in enumerate(("early wave 0", "late wave 4")):
    print(gn, "— ticks of 10 ns relative to stage 8's start; columns:", names)
    for k in range(16):
        print("  stage %2d: " % (k + 8) + " ".join("%6d" % (st[g, k, j] - t0) for j in range(6)))
    d = np.diff(st[g, :, 0]); print("  stage period: mean %.1f ticks = %.2f us" % (d.mean(), d.mean() / 100))
    print("  mean durations (ticks): mem issue %.1f, lgkm wait %.1f, barrier %.1f, mfma phase %.1f, vm wait %.1f, end barrier -> next %.1f" % (
        (st[g, :, 1] - st[g, :, 0]).mean(), (st[g, :, 2] - st[g, :, 1]).mean(), (st[g, :, 3] - st[g, :, 2]).mean(),
        (st[g, :, 4] - st[g, :, 3]).mean(), (st[g, :, 5] - st[g, :, 4]).mean(), (st[g, 1:, 0] - st[g, :-1, 5]).mean()))
