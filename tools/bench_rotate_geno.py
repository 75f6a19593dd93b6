"""Timing of the fp16-MFMA rotation. usage: bench_rotate_geno.py n p [dosage]   (dosage: continuous X -> the split path)"""
import sys, time, ctypes as C
import numpy as np
sys.path.insert(0, '/root/repo')
from pygemma_amd import _lib, synth
n, p = int(sys.argv[1]), int(sys.argv[2])
L = _lib.load(); ctx = _lib.Context(0)
rng = np.random.default_rng(0)
U = rng.standard_normal((n, n), dtype=np.float32) / np.sqrt(n)
X = synth.genotypes(rng, n, p)
if len(sys.argv) > 3:
    X = (X + rng.uniform(-0.3, 0.3, X.shape).astype(np.float32)).astype(np.float32)   # imputed-dosage-like: not genotype-valued
ldx = (n + 63)//64*64
dU, dX = ctx.to_device(U), ctx.to_device(X); dXr = ctx.alloc(p*ldx*4)
dprep = ctx.alloc(L.pg_geno_prep_bytes(n)); dwork = ctx.alloc(L.pg_geno_work_bytes(n, p))
t = time.time(); _lib.check(L.pg_geno_prep_dev(ctx.handle, n, dU.ptr, n, dprep.ptr), "prep"); ctx.sync(); print(f"prep U: {(time.time()-t)*1e3:.1f} ms")
ok = C.c_int(0)
def run():
    _lib.check(L.pg_rotate_geno_dev(ctx.handle, n, p, dprep.ptr, dX.ptr, p, dXr.ptr, ldx, dwork.ptr, C.byref(ok)), "rot"); ctx.sync()
run(); ts=[]
for _ in range(5):
    t=time.time(); run(); ts.append(time.time()-t)
t=min(ts); print(f"rotate_geno n={n} p={p} ok={ok.value}: {t*1e3:.2f} ms  {p/t:.0f} SNPs/s  ({4*n*n*p*ok.value/t/1e12:.0f} executed fp16 TFLOP/s incl. detect+encode; ok=1 genotype path, 2 split path)")
