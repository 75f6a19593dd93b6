#!/usr/bin/env python3
"""Timing of the REAL reference (rlangefe/pygemma, built by oracle/build_ref.py into /tmp) on the bench slice, in the BUILD
container (the reference cannot travel to the GPU box) — the fixture BASELINE.md §3.2 / SURVEY §8(d) ask for, with the
oracle timed beside it on the same inputs so that bench.py's cpu_baseline (oracle on the GPU box's cores) can be converted.
Writes profiles/r02_reference_timing.json.   usage: python oracle/build_ref.py && python tools/time_reference.py"""
import contextlib, io, json, os, platform, sys, time, warnings
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.environ.get("PYGEMMA_REF", "/tmp/pygemma_ref"))
warnings.filterwarnings("ignore")
with contextlib.redirect_stdout(io.StringIO()):
    from pygemma import lmm as ref
from oracle import oracle as O
from pygemma_amd import synth
import scipy, scipy.linalg

def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        return fn(*a, **k)

n, c, S = 10000, 5, 64
rp = synth.fast_rotated_panel(n, S, c, seed=10005)
d, Y, W, X = rp["d"], rp["Y"], np.ascontiguousarray(rp["W"]), np.ascontiguousarray(rp["X"])
out = {"where": "build container (no GPU)", "cpu_model": next((l.split(":")[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")), platform.processor()),
       "cores": len(os.sched_getaffinity(0)), "versions": {"numpy": np.__version__, "scipy": scipy.__version__},
       "shape": {"n": n, "c": c, "snps": S, "inputs": "synth.fast_rotated_panel(n, S, c, seed=10005): eigen-basis inputs (eigen=False entry)"}}
res = {}
for grid in (False, True):
    for nproc in (1, 8):
        t = time.time(); df = quiet(ref.pygemma, Y, X, W, d, eigen=False, grid=grid, nproc=nproc); dt = time.time() - t
        res[f"reference_{'grid' if grid else 'brent'}_nproc{nproc}"] = {"seconds": dt, "snps_per_s": S / dt}
    for thr in (1, 8):
        t = time.time(); o = O.calculate(d, Y, W, X, grid=grid, order=0, nthreads=thr); dt = time.time() - t
        res[f"oracle_{'grid' if grid else 'brent'}_threads{thr}"] = {"seconds": dt, "snps_per_s": S / dt}
    same = float((o["beta"].view(np.uint32) == df["beta"].to_numpy().view(np.uint32)).mean())
    res[f"oracle_vs_reference_beta_bit_identical_{'grid' if grid else 'brent'}"] = same
out["per_snp_path"] = res
rng = np.random.default_rng(1)
U = rng.standard_normal((n, n), dtype=np.float32); Xs = rng.standard_normal((n, 2048), dtype=np.float32)
t = time.time(); _ = U.T @ Xs; dt = time.time() - t
out["rotation_sgemm"] = {"what": "U.T @ X (lmm/lmm.py:244), numpy/OpenBLAS float32, 2048 SNPs", "seconds": dt, "snps_per_s": 2048 / dt, "gflops": 2.0 * n * n * 2048 / dt / 1e9}
G = synth.genotypes(rng, n, 2 * n); K = (G @ G.T / (2 * n)).astype(np.float32); del G
t = time.time(); ev, Uk = scipy.linalg.eigh(K); dt = time.time() - t
out["eigh_float32"] = {"what": "scipy.linalg.eigh(K) float32 (lmm/lmm.py:152), n=10000", "seconds": dt}
r = out["per_snp_path"]
out["ratio_oracle8_over_reference8_brent"] = r["oracle_brent_threads8"]["snps_per_s"] / r["reference_brent_nproc8"]["snps_per_s"]
out["ratio_oracle1_over_reference1_brent"] = r["oracle_brent_threads1"]["snps_per_s"] / r["reference_brent_nproc1"]["snps_per_s"]
json.dump(out, open(os.path.join(ROOT, "profiles", "r02_reference_timing.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
