#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X-native pyGEMMA hot path.

Metric (BASELINE.json): SNPs/sec (whole node) at n=10,000, c=5; K-eigendecomposition wall-clock.
Workload (BASELINE.json configs[2], the shape the metric is quoted on): synthetic n=10,000 individuals,
c=5 covariates; one "step" = one batch of B SNPs per GPU through the per-SNP hot path with inputs resident in
HBM: rotation X <- U'X (fp32 MFMA GEMM)  ->  REML lambda search (decade scan + Brent + Newton) ->
beta/se/tau/Wald F -> p-value (all on device) [-> RCCL all-gather of the 32-byte result rows when N > 1].
The one-time eigendecomposition of K is timed separately (`eigh_seconds`), as the metric asks.

    python bench.py --gpus 1 --steps 10 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
           bench.py --gpus N --steps K --warmup W

One process per GPU; SNP batches are independent units (no data-path collective except the result gather):
"scaling": "weak".  torch is imported only when WORLD_SIZE > 1 (rendezvous, barrier, MAX-reduce of the time,
RCCL all-gather); the compute path is the C ABI in include/pygemma_hip.h.
Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from pygemma_amd import _lib, synth  # noqa: E402

F32_MFMA_PEAK_TF = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 FLOP/clk/CU
BF16_MFMA_PEAK_TF = 2500.0 # dense bf16/fp16 MFMA peak (MI355X_MICROARCH.md)
F64_VALU_PEAK_TF = 78.6    # fp64 vector peak (= fp64 matrix peak on MI355X): 128 FLOP/clk/CU


def pmc_traffic(kernel):
    """HBM-side bytes per launch of `kernel` from the committed rocprofv3 PMC summary (profiles/*_summary.json:
    separate --pmc FETCH_SIZE / WRITE_SIZE passes at this bench's shapes, gfx950 x2 fetch correction); None if absent."""
    try:
        import glob
        f = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_summary.json")))[-1]
        return json.load(open(f))["pmc"][kernel]["hbm_bytes_per_launch"]
    except Exception:
        return None


def host_cores():
    """CPU threads this process may actually use (affinity and cgroup quota), for the CPU-baseline leg."""
    n = len(os.sched_getaffinity(0))
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = max(1, min(n, int(float(q) / float(per))))
    except Exception:
        pass
    return n


def log(rank, *a):
    if rank == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


def make_inputs(n, c, B, rank, p_k, null=False):
    """Deterministic synthetic panel: standardised Binomial(2, maf) genotypes; K from an independent
    SNP set; y = 0.2 g0 + G_K b + e (h2 = 0.5).  K/y/W are identical on every rank, the SNP batch is per rank."""
    rng = np.random.default_rng(synth.SEED)
    GK = synth.genotypes(rng, n, p_k)                                   # (n, p_k) float32
    Wm = np.concatenate([np.ones((n, 1)), rng.standard_normal((n, c - 1))], axis=1).astype(np.float32)
    b = (rng.standard_normal(p_k) * np.sqrt(0.5 / p_k)).astype(np.float32)
    rngx = np.random.default_rng(synth.SEED + 1000 + rank)
    X = synth.genotypes(rngx, n, B)                                     # (n, B) float32
    g0 = synth.genotypes(np.random.default_rng(synth.SEED + 1), n, 1)[:, 0]
    y = 0.2 * g0 + GK @ b + rng.standard_normal(n).astype(np.float32) * np.sqrt(0.5)
    if null:                                                            # the second, pure-noise phenotype
        y = rng.standard_normal(n).astype(np.float32)
    return GK, Wm, y.astype(np.float32).reshape(-1, 1), X


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--n", type=int, default=10000)
    ap.add_argument("--c", type=int, default=5)
    ap.add_argument("--batch", type=int, default=16384, help="SNPs per GPU per step")
    ap.add_argument("--grid", type=int, default=0, help="1 = calc_lambda_restricted(grid=True) path")
    ap.add_argument("--cpu-sample", type=int, default=256, help="SNPs of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--null", type=int, default=0, help="1 = pure-noise phenotype (SURVEY 8d second phenotype) instead of the polygenic one")
    ap.add_argument("--fp32-rotate", type=int, default=0, help="1 = force the fp32-MFMA rotation even for genotype-valued X")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        if world == 1 and a.gpus > 1:
            raise SystemExit("bench.py --gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
    n, c, B = a.n, a.c, a.batch
    L = _lib.load()
    if _lib.device_count() < 1:
        raise SystemExit("no GPU visible: the MI355X path has no CPU fallback")

    torch = dist = None
    stream_ptr = None
    if world > 1:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))   # nccl == RCCL on ROCm
        stream_ptr = torch.cuda.current_stream().cuda_stream
    ctx = _lib.Context(local_rank, stream=stream_ptr)
    vp = C.c_void_p

    # ---------------- inputs, resident in HBM before any timed region
    t0 = time.time()
    p_k = 2 * n
    GK, Wm, y, X = make_inputs(n, c, B, rank, p_k, null=bool(a.null))
    log(rank, f"synthetic inputs n={n} c={c} B={B} p_k={p_k}: {time.time() - t0:.1f} s (host)")
    ldx = (n + 63) // 64 * 64
    dGK = ctx.to_device(GK)
    dGt = ctx.alloc(p_k * ldx * 4)
    dK = ctx.alloc(n * n * 4)
    _lib.check(L.pg_transpose_dev(ctx.handle, n, p_k, dGK.ptr, p_k, dGt.ptr, ldx), "transpose")
    _lib.check(L.pg_kinship_dev(ctx.handle, n, p_k, dGt.ptr, ldx, dK.ptr), "kinship")
    ctx.sync()
    dGK.free(); dGt.free()

    # ---------------- H1: eigendecomposition, timed on its own (one-time cost)
    dev, dU = ctx.alloc(n * 4), ctx.alloc(n * n * 4)
    eigh_s = []
    for _ in range(2):
        t = time.time()
        _lib.check(L.pg_syevd_dev(ctx.handle, n, dK.ptr, dev.ptr, dU.ptr, None, None), "pg_syevd_dev")   # synchronous
        eigh_s.append(time.time() - t)
    log(rank, f"syevd n={n}: {min(eigh_s):.3f} s")
    dK.free()

    # rotate [y | W] with the same kernel (lmm.py:245-246), re-lay out as (n x c) / (n)
    q = 1 + c
    dYW = ctx.to_device(np.ascontiguousarray(np.concatenate([y, Wm], axis=1)))
    dYWr = ctx.alloc(q * ldx * 4)
    _lib.check(L.pg_rotate_dev(ctx.handle, n, q, dU.ptr, n, dYW.ptr, q, dYWr.ptr, ldx), "rotate yw")
    ctx.sync()
    YWr = dYWr.download((q, ldx), np.float32)[:, :n]
    dy = ctx.to_device(np.ascontiguousarray(YWr[0]))
    dW = ctx.to_device(np.ascontiguousarray(YWr[1:].T))
    ldX = B
    dX = ctx.to_device(X)
    dXr = ctx.alloc(B * ldx * 4)
    # rotation: genotype fast path (the synthetic panel is standardised hard calls, like every caller's input) unless forced off
    dprep = ctx.alloc(L.pg_geno_prep_bytes(n))
    dwork = ctx.alloc(L.pg_geno_work_bytes(n, B))
    _lib.check(L.pg_geno_prep_dev(ctx.handle, n, dU.ptr, n, dprep.ptr), "pg_geno_prep_dev")
    geno_used = [0, 0]
    if world > 1:
        res_t = torch.empty(32 * B, dtype=torch.uint8, device="cuda")
        all_t = torch.empty(32 * B * world, dtype=torch.uint8, device="cuda")
        res_ptr = res_t.data_ptr()
    else:
        res_buf = ctx.alloc(32 * B)
        res_ptr = res_buf.ptr
    # result row block: [F (B f64) | p (B f64) | beta | se | tau | lambda (B f32 each)]
    pF, pP = res_ptr, res_ptr + 8 * B
    pb, ps, pt, pl = (res_ptr + 16 * B + 4 * B * k for k in range(4))
    dstats = ctx.alloc(16)
    L.pg_memset(ctx.handle, dstats.ptr, 0, 16)

    ev = []
    def new_event():
        e = vp()
        _lib.check(L.pg_event_create(ctx.handle, C.byref(e)), "event")
        ev.append(e)
        return e

    def step(e0=None, e1=None, e2=None):
        if e0: L.pg_event_record(ctx.handle, e0)
        is_geno = C.c_int(0)
        if not a.fp32_rotate:
            _lib.check(L.pg_rotate_geno_dev(ctx.handle, n, B, dprep.ptr, dX.ptr, ldX, dXr.ptr, ldx, dwork.ptr, C.byref(is_geno)),
                       "pg_rotate_geno_dev")
        if not is_geno.value:
            _lib.check(L.pg_rotate_dev(ctx.handle, n, B, dU.ptr, n, dX.ptr, ldX, dXr.ptr, ldx), "pg_rotate_dev")
        geno_used[1 if is_geno.value else 0] += 1
        if e1: L.pg_event_record(ctx.handle, e1)
        _lib.check(L.pg_assoc_dev(ctx.handle, n, c, B, dev.ptr, dW.ptr, dy.ptr, dXr.ptr, ldx, a.grid,
                                  pb, ps, pt, pl, pF, pP, dstats.ptr), "pg_assoc_dev")
        if e2: L.pg_event_record(ctx.handle, e2)
        if world > 1:
            dist.all_gather_into_tensor(all_t, res_t)   # RCCL over xGMI: 32 B per SNP

    def barrier():
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()
        ctx.sync()

    for _ in range(a.warmup):
        step()
    L.pg_memset(ctx.handle, dstats.ptr, 0, 16)
    events = [(new_event(), new_event(), new_event()) for _ in range(a.steps)]
    barrier()
    t = time.perf_counter()
    for k in range(a.steps):
        step(*events[k])
    barrier()
    elapsed = time.perf_counter() - t
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # per-kernel durations from the HIP events recorded on the launch stream
    # one extra, untimed-by-the-metric launch of each rotation kernel alone for the per-kernel roofline legs
    def time_call(fn):
        ea, eb = new_event(), new_event()
        L.pg_event_record(ctx.handle, ea); fn(); L.pg_event_record(ctx.handle, eb)
        m = C.c_float(); L.pg_event_elapsed_ms(ctx.handle, ea, eb, C.byref(m)); return m.value * 1e-3
    t_f32 = time_call(lambda: _lib.check(L.pg_rotate_dev(ctx.handle, n, B, dU.ptr, n, dX.ptr, ldX, dXr.ptr, ldx), "pg_rotate_dev"))
    rot_ms, assoc_ms = [], []
    ms = C.c_float()
    for e0, e1, e2 in events:
        L.pg_event_elapsed_ms(ctx.handle, e0, e1, C.byref(ms)); rot_ms.append(ms.value)
        L.pg_event_elapsed_ms(ctx.handle, e1, e2, C.byref(ms)); assoc_ms.append(ms.value)
    stats = dstats.download((2,), np.uint64).astype(np.float64) / (B * a.steps)

    if rank != 0:
        if world > 1:
            dist.barrier()                   # rank 0 finishes its report, then everybody tears the group down together
            dist.destroy_process_group()
        return

    # sanity on the last batch (never inside the timed region)
    if world > 1:
        host = all_t[: 32 * B].cpu().numpy()
    else:
        host = res_buf.download((32 * B,), np.uint8)
    beta = host[16 * B: 20 * B].view(np.float32)
    pv = host[8 * B: 16 * B].view(np.float64)
    lam = host[28 * B: 32 * B].view(np.float32)
    assert np.isfinite(beta).all() and ((pv >= 0) & (pv <= 1)).all(), "non-finite results"

    value = world * B * a.steps / elapsed
    rot_avg = float(np.mean(rot_ms)) * 1e-3
    assoc_avg = float(np.mean(assoc_ms)) * 1e-3
    rot_flops = 2.0 * n * n * B                       # algorithmic: 2 n^2 per SNP (SURVEY 8d stage R)
    m = c + 2
    # algorithmic fp64 flops of the assoc stage per SNP: decade scan 11 lambdas x 2 powers x m entries x 2n,
    # + per SNP-specific evaluation m(m+1)/2 entries x (2 | 3) powers x 2n
    assoc_flops_snp = 11 * 2 * m * 2.0 * n + (stats[0] * 2 + stats[1] * 3) * (m * (m + 1) / 2) * 2.0 * n
    rl_rotate = ({"kernel": "rotate_geno_kernel (+detect/encode): fp16 MFMA 16x16x32, U split in 2 fp16 planes, fp32 accumulate",
                  "bound": "mfma", "achieved": rot_flops / rot_avg / 1e12, "peak": BF16_MFMA_PEAK_TF / 2.0, "unit": "TFLOP/s",
                  "frac": rot_flops / rot_avg / 1e12 / (BF16_MFMA_PEAK_TF / 2.0),
                  "peak_note": "algorithmic 2n^2 flop/SNP against the dense fp16 MFMA peak (2500 TF) divided by the 2 fp16 passes a "
                               "24-bit U needs; executed fp16 rate = 2x achieved",
                  "traffic": pmc_traffic("rotate_geno_kernel") if (n, B) == (10000, 16384) else None,
                  "algorithmic_bytes": 4.0 * n * B + 4.0 * n * n + 4.0 * ldx * B, "avg_launch_ms": rot_avg * 1e3}
                 if geno_used[1] else
                 {"kernel": "rotate_kernel<4> (fp32 MFMA 32x32x2)", "bound": "mfma", "achieved": rot_flops / rot_avg / 1e12,
                  "peak": F32_MFMA_PEAK_TF, "unit": "TFLOP/s", "frac": rot_flops / rot_avg / 1e12 / F32_MFMA_PEAK_TF,
                  "traffic": pmc_traffic("rotate_kernel") if (n, B) == (10000, 16384) else None,
                  "algorithmic_bytes": 4.0 * n * B + 4.0 * n * n + 4.0 * ldx * B, "avg_launch_ms": rot_avg * 1e3})
    rl_assoc = {"kernel": "assoc_kernel<%d> (+setup, p-values): fp64 VALU FMAs of the Gram passes, wave per SNP" % c, "bound": "mfma",
                "achieved": assoc_flops_snp * B / assoc_avg / 1e12, "peak": F64_VALU_PEAK_TF, "unit": "TFLOP/s",
                "frac": assoc_flops_snp * B / assoc_avg / 1e12 / F64_VALU_PEAK_TF,
                "peak_note": "fp64 vector peak = fp64 matrix peak on MI355X (78.6 TF); only the Gram FMAs are counted (conversions, "
                             "h*w products, reciprocals, reductions and sweeps are not): the VALU issue port is ~82 % busy (profiles/)",
                "traffic": pmc_traffic("assoc_kernel") if (n, B, c) == (10000, 16384, 5) else None,
                "algorithmic_bytes": (4.0 * ldx + 36) * B, "avg_launch_ms": assoc_avg * 1e3,
                "hbm_GBps_algorithmic": (4.0 * n + 36) * B / assoc_avg / 1e9}
    out = {
        "metric": "SNPs/sec (whole node) at n=10,000 c=5; K-eigendecomp wallclock",
        "value": value, "unit": "SNPs/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": 1e3 * elapsed / a.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f16x2->f32 (genotype rotation) | f32 (general rotation) + f64 (Gram/sweeps)", "data": "synthetic",
        "config": {"workload": f"BASELINE configs[2]: synthetic n={n}, c={c}, {B} SNPs/GPU/step: rotate (U'X) + REML "
                               f"{'grid' if a.grid else 'decade-scan+Brent+Newton'} + Wald F + p on device"
                               + ("; RCCL all-gather of result rows" if world > 1 else ""),
                   "n": n, "c": c, "snps_per_gpu_per_step": B, "lambda_path": "grid" if a.grid else "brent",
                   "phenotype": "pure noise" if a.null else "polygenic h2=0.5 + one causal SNP",
                   "parallelism": f"snp-shards x{world}"},
        "eigh_seconds": min(eigh_s),
        "eigh_note": "fp64 Householder tridiagonalisation + divide&conquer + back-transform on device, n=%d, one-time" % n,
        # the dominant kernel of the step = the longer of the two launches
        "roofline": rl_assoc if assoc_avg >= rot_avg else rl_rotate,
        "roofline_rotate": rl_rotate,
        "roofline_assoc": rl_assoc,
        "roofline_fp32_rotate": {"kernel": "rotate_kernel<4> (fp32 MFMA 32x32x2; the path for non-genotype X)", "bound": "mfma",
                                 "achieved": rot_flops / t_f32 / 1e12, "peak": F32_MFMA_PEAK_TF, "unit": "TFLOP/s",
                                 "frac": rot_flops / t_f32 / 1e12 / F32_MFMA_PEAK_TF, "avg_launch_ms": t_f32 * 1e3,
                                 "traffic": pmc_traffic("rotate_kernel") if (n, B) == (10000, 16384) else None},
        "rotation_path": "genotype f16x2" if geno_used[1] else "fp32 MFMA",
        "stage_snps_per_s_per_gpu": {"rotate": B / rot_avg, "assoc": B / assoc_avg},
        "evals_per_snp": {"fast": float(stats[0]), "newton": float(stats[1])},
        "lambda_median": float(np.median(lam)),
    }

    # ---------------- CPU baseline: the oracle ("port"), bounded sample of the SAME workload, rank 0 / N=1 only
    if a.cpu_sample > 0 and world == 1:
        try:
            from oracle import oracle as O
            S = min(a.cpu_sample, B)
            Uh = dU.download((n, n), np.float32)
            dh = dev.download((n,), np.float32)
            Xs = np.ascontiguousarray(X[:, :S])
            nthr = min(O.lib().orc_max_threads(), host_cores())
            t = time.time()
            Xrs = O.rotate(Uh, Xs, ldx=ldx)
            t_rot = time.time() - t
            t = time.time()
            orc = O.calculate(dh, YWr[0], np.ascontiguousarray(YWr[1:].T), np.ascontiguousarray(Xrs[:, :n]), grid=bool(a.grid),
                              order=0, nthreads=nthr, snp_major=True)
            t_as = time.time() - t
            same = float((orc["beta"].view(np.uint32) == beta[:S].view(np.uint32)).mean())
            rel = float(np.max(np.abs(orc["beta"].astype(np.float64) - beta[:S]) / orc["se_beta"].astype(np.float64)))
            out["cpu_baseline"] = {"value": S / (t_rot + t_as), "unit": "SNPs/s", "cores": int(nthr), "kind": "port",
                                   "sample": f"first {S} SNPs of the step batch: oracle rotate {t_rot:.2f} s + calculate {t_as:.2f} s "
                                             f"(OpenMP, {nthr} threads); GPU vs oracle beta: {100 * same:.1f}% rows bit-identical, max |dbeta|/se {rel:.1e} "
                                             f"(the oracle rotates with an fp32 fma chain; bit-identity is expected only with --fp32-rotate 1)"}
        except Exception as ex:   # the baseline is a report, not a dependency of the measurement
            out["cpu_baseline"] = {"value": None, "unit": "SNPs/s", "cores": 0, "kind": "port", "sample": f"failed: {ex}"}
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
