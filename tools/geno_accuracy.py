"""Error of the genotype rotation path and of the fp32-MFMA rotation against an fp64 rotation.  usage: geno_accuracy.py [n] [p]"""
import sys
import numpy as np
sys.path.insert(0, '/root/repo')
from pygemma_amd import _lib, ops
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
p = int(sys.argv[2]) if len(sys.argv) > 2 else 512
rng = np.random.default_rng(7)
U64 = np.linalg.qr(rng.standard_normal((n, n)))[0]
U = U64.astype(np.float32)
maf = rng.uniform(0.05, 0.5, p)
G = rng.binomial(2, maf, size=(n, p)).astype(np.float64)
X = ((G - G.mean(0)) / np.maximum(G.std(0), 1e-9)).astype(np.float32)
with _lib.Context(0) as ctx:
    got, ok = ops.rotate_geno(U, X, ctx=ctx)
    f32p = ops.rotate(U, X, ctx=ctx)[:, :n]
exact = (U.astype(np.float64).T @ X.astype(np.float64)).T      # rotation with the float32 U the kernels are given
truth = (U64.T @ X.astype(np.float64)).T                         # rotation with the un-rounded U
rms = lambda a: float(np.sqrt(np.mean(a ** 2)))
print(f"n={n} p={p} ok={ok}  rms |Xr|={rms(exact):.3g}")
print(f"  genotype path  vs fp64(U32'X): rms {rms(got[:, :n] - exact):.3g}  max {np.abs(got[:, :n] - exact).max():.3g}")
print(f"  fp32 MFMA path vs fp64(U32'X): rms {rms(f32p - exact):.3g}  max {np.abs(f32p - exact).max():.3g}")
print(f"  float32 rounding of U alone  : rms {rms(exact - truth):.3g}")
