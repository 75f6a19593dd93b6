"""Run-to-run determinism at the bench shape: the rotation (fp16x2 genotype path and fp32 path) and the association kernel (Brent,
grid, LRT) repeated on the same resident inputs must return the same bytes every time (no atomics, fixed reduction orders; a race in
an LDS ring or a missing wait would show here).  usage: determinism.py [n] [p] [reps]"""
import sys, zlib, ctypes as C
import numpy as np
sys.path.insert(0, '/root/repo')
from pygemma_amd import _lib, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
p = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 25
c = 5
L = _lib.load(); ctx = _lib.Context(0)
rng = np.random.default_rng(0)
Q, _ = np.linalg.qr(rng.standard_normal((n, n)).astype(np.float32))
U = np.ascontiguousarray(Q, np.float32)
X = rng.binomial(2, rng.uniform(0.05, 0.5, p), size=(n, p)).astype(np.float32)
d = np.sort(rng.gamma(0.5, 2.0, n)).astype(np.float32)
W = np.concatenate([np.ones((n, 1)), rng.standard_normal((n, c - 1))], axis=1).astype(np.float32)
y = (0.05 * X[:, 0] + rng.standard_normal(n)).astype(np.float32)
ldx = (n + 63) // 64 * 64
dU, dX, dd, dW, dy = (ctx.to_device(a) for a in (U, X, d, W, y))
dprep, dwork = ctx.alloc(L.pg_geno_prep_bytes(n)), ctx.alloc(L.pg_geno_work_bytes(n, p))
dXr, dXr32 = ctx.alloc(p * ldx * 4), ctx.alloc(p * ldx * 4)
res = ctx.alloc(p * 64)
_lib.check(L.pg_geno_prep_dev(ctx.handle, n, dU.ptr, n, dprep.ptr), "prep")
def crc(buf, nbytes):
    return zlib.crc32(buf.download((nbytes,), np.uint8).tobytes())
seen = {}
def note(tag, v):
    if tag not in seen: seen[tag] = v
    elif seen[tag] != v:
        print(f"NON-DETERMINISTIC {tag}: {seen[tag]:08x} vs {v:08x}"); sys.exit(1)
r0 = res.ptr
for it in range(reps):
    _lib.check(L.pg_rotate_auto_dev(ctx.handle, n, p, dU.ptr, n, dprep.ptr, dX.ptr, p, dXr.ptr, ldx, dwork.ptr, None), "rot"); ctx.sync()
    note("rotate fp16x2", crc(dXr, p * ldx * 4))
    if it % 5 == 0:
        _lib.check(L.pg_rotate_dev(ctx.handle, n, p, dU.ptr, n, dX.ptr, p, dXr32.ptr, ldx), "rot32"); ctx.sync()
        note("rotate fp32", crc(dXr32, p * ldx * 4))
    for grid in (0, 1):
        _lib.check(L.pg_assoc_dev(ctx.handle, n, c, p, dd.ptr, dW.ptr, dy.ptr, dXr.ptr, ldx, grid, r0 + 16 * p, r0 + 20 * p, r0 + 24 * p, r0 + 28 * p, r0, r0 + 8 * p, None), "assoc"); ctx.sync()
        note(f"assoc grid={grid}", crc(res, 32 * p))
    if it % 5 == 0:
        _lib.check(L.pg_assoc_lrt_dev(ctx.handle, n, c, p, dd.ptr, dW.ptr, dy.ptr, dXr.ptr, ldx, 0, r0 + 16 * p, r0 + 20 * p, r0 + 24 * p, r0 + 28 * p, r0, r0 + 8 * p,
                                      r0 + 32 * p, r0 + 40 * p, r0 + 48 * p, r0 + 56 * p), "lrt"); ctx.sync()
        note("assoc lrt", crc(res, 64 * p))
    if it % 5 == 4: print(f"rep {it + 1}: identical so far ({', '.join(f'{k} {v:08x}' for k, v in seen.items())})", flush=True)
print("deterministic over", reps, "repetitions")
# the eigensolver: host threads share the D&C leaves (disjoint slices), every device reduction has a fixed order
from pygemma_amd import ops
K = synth.panel(3001, 4, 1, seed=3)["K"]
for it in range(6):
    ev32, U32, ev64, U64 = ops.syevd(K, ctx=ctx, want64=True)
    note("syevd eigenvalues", zlib.crc32(ev64.tobytes())); note("syevd eigenvectors", zlib.crc32(U64.tobytes()))
print("syevd n=3001 deterministic over 6 solves:", {k: f"{v:08x}" for k, v in seen.items() if k.startswith("syevd")})
