#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X-native pyGEMMA hot path.

Metric (BASELINE.json): SNPs/sec (whole node) at n=10,000, c=5; K-eigendecomposition wall-clock.
Workload (BASELINE.json configs[2], the shape the metric is quoted on): synthetic n=10,000 individuals, c=5 covariates,
p=100,000 SNPs per GPU.  One "step" = ONE PASS OVER ALL p SNPs of the rank, in HBM-resident batches of B SNPs, through the
per-SNP hot path:  rotation X <- U'X  ->  REML lambda search (decade scan + Brent + Newton) -> beta/se/tau/Wald F -> p-value,
all on the device, then ONE RCCL all-gather of the rank's 32-byte result rows when N > 1 (gather once at the end).
The one-time eigendecomposition of K is timed separately (`eigh_seconds`), as the metric asks.

    python bench.py                              # 1 GPU
    python bench.py --gpus N                     # spawns N ranks itself (one process per GPU), or run under any launcher:
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N

One process per GPU; SNP shards are independent (weak scaling: p per GPU fixed).  Rank 0 builds K (syrk on the device), runs
the eigensolver once and broadcasts U, d and the rotated y/W over RCCL; every rank generates its own genotype shard.
No PyTorch anywhere: rendezvous is pygemma_amd/dist.py (stdlib), collectives are the C ABI's pg_comm_* (librccl).
Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

F32_MFMA_PEAK_TF = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 FLOP/clk/CU
F16_MFMA_PEAK_TF = 2500.0  # dense bf16/fp16 MFMA peak (MI355X_MICROARCH.md)
I8_MFMA_PEAK_TOPS = 5000.0  # dense int8 MFMA peak: 2x the bf16 rate per clock (MI355X_MICROARCH.md, MFMA table)
F64_VALU_PEAK_TF = 78.6    # fp64 vector peak (= fp64 matrix peak on MI355X): 128 FLOP/clk/CU


def pmc_traffic(kernel):
    """(HBM-side bytes per launch of `kernel`, source file) from the newest committed rocprofv3 PMC summary (profiles/*_summary.json:
    separate --pmc FETCH_SIZE / WRITE_SIZE passes at this bench's batch shape, gfx950 x2 fetch correction); (None, None) if absent."""
    try:
        import glob
        for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_summary.json")), reverse=True):
            j = json.load(open(f))
            v = j.get("pmc", {}).get(kernel, {}).get("hbm_bytes_per_launch")
            if v is not None:
                sh = j.get("batch_shape", {"n": 10000, "batch": 16384, "c": 5})
                return v, os.path.relpath(f, ROOT), (sh["n"], sh["batch"], sh["c"])
    except Exception:
        pass
    return None, None, None


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


def cpu_model():
    try:
        return next(l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name"))
    except Exception:
        return "unknown"


def log(rank, *a):
    if rank == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


def geno_codes(rng, n, p, chunk=8192):
    """Hard calls 0/1/2 ~ Binomial(2, maf_g), maf_g ~ U(0.05, 0.5) as int8, generated from two uniform byte planes
    (fast: ~1 GB/s): code = [u1 < maf] + [u2 < maf]."""
    out = np.empty((n, p), np.int8)
    for s in range(0, p, chunk):
        e = min(p, s + chunk)
        thr = np.floor(rng.uniform(0.05, 0.5, e - s) * 256.0).astype(np.uint8)
        u = rng.integers(0, 256, size=(2, n, e - s), dtype=np.uint8)
        out[:, s:e] = (u[0] < thr).astype(np.int8) + (u[1] < thr).astype(np.int8)
    return out


def spawn_ranks(a):
    """`python bench.py --gpus N` without a launcher: start N child processes (one per GPU) BEFORE anything touches HIP in this
    process, hand them RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*, pass rank 0's JSON line through, return the worst exit code."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   PYGEMMA_RDZV_KEY=f"bench{os.getpid()}_{os.urandom(8).hex()}")   # per-launch nonce
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    # wait for all; if one rank dies, the others would sit in the communicator's rendezvous for ever: end them (exact PIDs)
    rc = 0
    live = list(procs)
    while live:
        time.sleep(0.2)
        for pr in list(live):
            r = pr.poll()
            if r is None:
                continue
            live.remove(pr)
            if r != 0:
                rc = max(rc, abs(r))
                for other in live:
                    other.terminate()
                t_end = time.time() + 10
                for other in live:
                    try:
                        other.wait(timeout=max(0.1, t_end - time.time()))
                    except subprocess.TimeoutExpired:
                        other.kill()
                live = []
                break
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n", type=int, default=10000)
    ap.add_argument("--c", type=int, default=5)
    ap.add_argument("--snps", type=int, default=100000, help="SNPs per GPU per step (configs[2]: p = 100,000)")
    ap.add_argument("--batch", type=int, default=100000, help="SNPs per HBM-resident batch (the default holds the whole step in one: kernels long enough "
                    "to settle at their own clocks; 16384 costs 3 %%)")
    ap.add_argument("--grid", type=int, default=0, help="1 = calc_lambda_restricted(grid=True) path")
    ap.add_argument("--cpu-sample", type=int, default=1024, help="SNPs of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--null", type=int, default=0, help="1 = pure-noise phenotype (SURVEY 8d second phenotype) instead of the polygenic one")
    ap.add_argument("--weak", type=int, default=0, help="1 = weak-signal phenotype (h2 = 0.02): drives Newton towards its iteration cap")
    ap.add_argument("--fp32-rotate", type=int, default=0, help="1 = force the fp32-MFMA rotation even for genotype-valued X")
    ap.add_argument("--raw-codes", type=int, default=0, help="1 = feed the raw 0/1/2 hard calls as float32 instead of the standardised columns SURVEY 8(d) specifies")
    ap.add_argument("--e2e", type=int, default=1, help="1 = after the timed region also run lmm.pygemma from host X (incl. H2D and eigh)")
    ap.add_argument("--eigh-cache", default="", help="npz path: reuse U, d (and K-derived inputs) from a previous run instead of running "
                                                     "the eigensolver (rocprofv3 --pmc passes on the per-SNP kernels)")
    a = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        raise SystemExit(spawn_ranks(a))        # nothing in this process has touched the GPU

    from pygemma_amd import _lib, dist, synth
    rank, world, local_rank = dist.env_rank()
    if a.gpus != world:
        log(rank, f"note: --gpus {a.gpus} but WORLD_SIZE={world}; using the launcher's world size")
    n, c, P = a.n, a.c, a.snps
    B = max(1, min(a.batch, P))       # a batch is never larger than the step
    L = _lib.load()
    if _lib.device_count() < 1:
        raise SystemExit("no GPU visible: the MI355X path has no CPU fallback")
    ctx = _lib.Context(local_rank)
    # PYGEMMA_BENCH_FORCE_COMM=1: form the (1-rank) communicator and run every collective on a single GPU too (rehearsal of the N > 1 path)
    comm = dist.init(ctx) if (world > 1 or os.environ.get("PYGEMMA_BENCH_FORCE_COMM")) else None
    vp = C.c_void_p
    ldx = (n + 63) // 64 * 64
    q = 1 + c

    # ---------------- K, eigendecomposition, rotated y/W: rank 0 computes, everybody receives (RCCL broadcast)
    dU, dev, dy, dW = ctx.alloc(n * n * 4), ctx.alloc(n * 4), ctx.alloc(n * 4), ctx.alloc(n * c * 4)
    eigh_s, kin_s, Khost, yW = [], None, None, None
    p_k = 2 * n
    t0 = time.time()
    rng = np.random.default_rng(synth.SEED)
    if rank == 0:
        cache = a.eigh_cache and os.path.exists(a.eigh_cache)
        Wm = np.concatenate([np.ones((n, 1)), rng.standard_normal((n, c - 1))], axis=1).astype(np.float32)
        if cache:
            z = np.load(a.eigh_cache)
            assert z["U"].shape == (n, n) and z["yW"].shape == (n, q), "eigh cache is for another shape"
            dU.upload(z["U"]); dev.upload(z["d"]); yW = z["yW"]
            log(rank, f"eigenpairs from {a.eigh_cache} (eigensolver skipped)")
        else:
            GK8 = geno_codes(rng, n, p_k)                                        # raw hard calls; standardised on the device
            b = (rng.standard_normal(p_k) * np.sqrt((0.02 if a.weak else 0.5) / p_k)).astype(np.float32)
            GKf = GK8.astype(np.float32)
            mu, sd = GKf.mean(0), GKf.std(0); sd[sd == 0] = 1
            g0 = geno_codes(np.random.default_rng(synth.SEED + 1), n, 1)[:, 0].astype(np.float32)
            g0 = (g0 - g0.mean()) / max(g0.std(), 1e-6)
            yv = 0.2 * g0 * (0.0 if a.weak else 1.0) + ((GKf - mu) / sd) @ b + rng.standard_normal(n).astype(np.float32) * np.sqrt(0.98 if a.weak else 0.5)
            if a.null:
                yv = rng.standard_normal(n).astype(np.float32)
            yW = np.ascontiguousarray(np.concatenate([yv.reshape(-1, 1).astype(np.float32), Wm], axis=1))
            dG = ctx.to_device(GKf)
            dK = ctx.alloc(n * n * 4)
            ctx.sync(); t = time.time()
            _lib.check(L.pg_kinship_geno_dev(ctx.handle, n, p_k, dG.ptr, p_k, 1, dK.ptr), "pg_kinship_geno_dev")   # N3: standardise + syrk
            ctx.sync(); kin_s = time.time() - t
            dG.free(); del GKf, GK8
            for _ in range(2):
                t = time.time()
                _lib.check(L.pg_syevd_dev(ctx.handle, n, dK.ptr, dev.ptr, dU.ptr, None, None), "pg_syevd_dev")   # synchronous
                eigh_s.append(time.time() - t)
            log(rank, f"kinship syrk n={n} p_k={p_k}: {kin_s * 1e3:.1f} ms; syevd: {min(eigh_s):.3f} s")
            Khost = dK.download((n, n), np.float32) if (a.e2e and world == 1) else None      # for the end-to-end leg (host inputs)
            dK.free()
            if a.eigh_cache:
                np.savez(a.eigh_cache, U=dU.download((n, n), np.float32), d=dev.download((n,), np.float32), yW=yW)
        # rotate [y | W] with the fp32 MFMA kernel (lmm.py:245-246), re-lay out as (n) / (n x c)
        dYW = ctx.to_device(yW)
        dYWr = ctx.alloc(q * ldx * 4)
        _lib.check(L.pg_rotate_dev(ctx.handle, n, q, dU.ptr, n, dYW.ptr, q, dYWr.ptr, ldx), "rotate yw")
        ctx.sync()
        YWr = dYWr.download((q, ldx), np.float32)[:, :n]
        dy.upload(np.ascontiguousarray(YWr[0])); dW.upload(np.ascontiguousarray(YWr[1:].T))
        dYW.free(); dYWr.free()
    t_bc = None
    if comm is not None:
        comm.barrier(); t = time.time()
        for buf, nb in ((dU, n * n * 4), (dev, n * 4), (dy, n * 4), (dW, n * c * 4)):
            comm.broadcast(buf.ptr, nb, 0)
        comm.barrier(); t_bc = time.time() - t
    log(rank, f"K + eigh + rotated y/W on rank 0{'' if t_bc is None else f', RCCL broadcast {t_bc * 1e3:.1f} ms'}: {time.time() - t0:.1f} s")

    # ---------------- this rank's genotype shard, resident in HBM as float32 (n x P) before any timed region
    t0 = time.time()
    X8 = geno_codes(np.random.default_rng(synth.SEED + 1000 + rank), n, P)
    dX8 = ctx.to_device(X8)     # stays: the int8-resident leg of the report rotates straight from it
    dX = ctx.alloc(n * P * 4)
    Xstd = None
    if a.raw_codes:
        _lib.check(L.pg_cast_i8_f32_dev(ctx.handle, n, P, dX8.ptr, 0, P, dX.ptr, P), "pg_cast_i8_f32_dev")   # reference layout (n, P) float32
    else:
        # hard calls standardised per SNP (what the reference's callers feed lmm.pygemma), float32, reference layout (n, P).
        # Each column still holds <= 3 (almost) equally spaced values, so the device's own detection takes the same genotype path.
        Xstd = np.empty((n, P), np.float32)
        for s0 in range(0, P, 8192):
            blk = X8[:, s0:s0 + 8192].astype(np.float32)
            mu, sd = blk.mean(0, dtype=np.float64), blk.std(0, dtype=np.float64)
            sd[sd == 0] = 1.0
            Xstd[:, s0:s0 + 8192] = (blk - mu.astype(np.float32)) / sd.astype(np.float32)
        dX.upload(Xstd)
    ctx.sync()
    log(rank, f"genotype shard n={n} P={P} ({'raw codes' if a.raw_codes else 'standardised'}) generated + resident: {time.time() - t0:.1f} s (host RNG)")
    batches = [(s, min(P, s + B)) for s in range(0, P, B)]
    dXr = ctx.alloc(B * ldx * 4)
    dprep = ctx.alloc(L.pg_geno_prep_bytes(n))
    dwork = ctx.alloc(L.pg_geno_work_bytes(n, B))
    _lib.check(L.pg_geno_prep_dev(ctx.handle, n, dU.ptr, n, dprep.ptr), "pg_geno_prep_dev")
    # result block of the rank: P rows of 32 bytes [F (P f64) | p (P f64) | beta | se | tau | lambda (P f32 each)]
    res = ctx.alloc(32 * P)
    allres = ctx.alloc(32 * P * world) if comm is not None else None
    pF, pP = res.ptr, res.ptr + 8 * P
    pb, ps, pt, pl = (res.ptr + 16 * P + 4 * P * k for k in range(4))
    dstats = ctx.alloc(16)
    dpath = ctx.alloc(4)
    L.pg_memset(ctx.handle, dpath.ptr, 0, 4)

    ev_pool = []
    def new_event():
        e = vp()
        _lib.check(L.pg_event_create(ctx.handle, C.byref(e)), "event")
        ev_pool.append(e)
        return e

    def step(events=None, fp32=False, int8=False, src=None, nsnp=None):
        # src / nsnp: another resident float32 matrix (n, nsnp) in place of the shard (the dosage leg of the report)
        xptr, ldsrc = (dX.ptr, P) if src is None else (src.ptr, nsnp)
        for bi, (s, e) in enumerate(batches if src is None else [(s, min(nsnp, s + B)) for s in range(0, nsnp, B)]):
            pbn = e - s
            if events: L.pg_event_record(ctx.handle, events[bi][0])
            if int8:         # the same genotypes as the int8 matrix a caller may hand over (lmm.pygemma takes it as it is): 4x fewer bytes to scan
                _lib.check(L.pg_rotate_auto_i8_dev(ctx.handle, n, pbn, dprep.ptr, dX8.ptr + s, 0, P, dXr.ptr, ldx, dwork.ptr, None),
                           "pg_rotate_auto_i8_dev")
            elif not fp32:   # path chosen on the device from the block's values (genotype codes -> int8 digit-plane MFMA); no host read-back
                _lib.check(L.pg_rotate_auto_dev(ctx.handle, n, pbn, dU.ptr, n, dprep.ptr, xptr + 4 * s, ldsrc, dXr.ptr, ldx, dwork.ptr,
                                                dpath.ptr), "pg_rotate_auto_dev")
            else:
                _lib.check(L.pg_rotate_dev(ctx.handle, n, pbn, dU.ptr, n, xptr + 4 * s, ldsrc, dXr.ptr, ldx), "pg_rotate_dev")
            if events: L.pg_event_record(ctx.handle, events[bi][1])
            _lib.check(L.pg_assoc_dev(ctx.handle, n, c, pbn, dev.ptr, dW.ptr, dy.ptr, dXr.ptr, ldx, a.grid,
                                      pb + 4 * s, ps + 4 * s, pt + 4 * s, pl + 4 * s, pF + 8 * s, pP + 8 * s, dstats.ptr), "pg_assoc_dev")
            if events: L.pg_event_record(ctx.handle, events[bi][2])
        if comm is not None and src is None:
            dist.gather_result_rows(comm, res.ptr, allres.ptr, P)      # RCCL over xGMI: 32 B per SNP, once per pass

    def barrier():
        if comm is not None:
            comm.barrier()
        ctx.sync()

    for _ in range(a.warmup):
        step(fp32=bool(a.fp32_rotate))
    L.pg_memset(ctx.handle, dstats.ptr, 0, 16)
    events = [[(new_event(), new_event(), new_event()) for _ in batches] for _ in range(a.steps)]
    barrier()
    t = time.perf_counter()
    for k in range(a.steps):
        step(events[k], fp32=bool(a.fp32_rotate))
    barrier()
    elapsed = time.perf_counter() - t
    if comm is not None:
        elapsed = comm.allreduce_max(elapsed)

    # per-kernel durations of the FULL batches from the HIP events recorded on the launch stream
    ms = C.c_float()
    rot_ms, assoc_ms = [], []
    for evs in events:
        for bi, (s, e) in enumerate(batches):
            if e - s != B:
                continue
            L.pg_event_elapsed_ms(ctx.handle, evs[bi][0], evs[bi][1], C.byref(ms)); rot_ms.append(ms.value)
            L.pg_event_elapsed_ms(ctx.handle, evs[bi][1], evs[bi][2], C.byref(ms)); assoc_ms.append(ms.value)
    stats = dstats.download((2,), np.uint64).astype(np.float64) / (P * a.steps)

    if rank != 0:
        if comm is not None:
            comm.barrier()                   # rank 0 finishes its report, then everybody tears the communicator down together
            comm.close()
        return

    # ---------------- report (rank 0), never inside the timed region
    def time_step(fp32):
        ctx.sync(); tt = time.perf_counter(); step(fp32=fp32); ctx.sync(); return time.perf_counter() - tt
    used_path = int(dpath.download((1,), np.int32)[0])                    # before the other path's pass overwrites nothing: fp32 does not write it
    t_other = time_step(not a.fp32_rotate) if comm is None else None      # the other rotation path, one pass
    t_int8 = None
    if comm is None:
        step(int8=True); ctx.sync(); tt = time.perf_counter(); step(int8=True); ctx.sync(); t_int8 = time.perf_counter() - tt
    # dosage leg: finite X that is not genotype-valued (imputed dosages) takes the split-plane path (X itself in two fp16 planes)
    t_dos, Pd, dos_path = None, min(P, 16384), None
    if comm is None:
        rd = np.random.default_rng(synth.SEED + 77)
        Xd = np.clip(X8[:, :Pd].astype(np.float32) + rd.uniform(-0.3, 0.3, (n, Pd)).astype(np.float32), 0.0, 2.0)
        dXd = ctx.to_device(np.ascontiguousarray(Xd))
        step(src=dXd, nsnp=Pd); ctx.sync()
        dos_path = int(dpath.download((1,), np.int32)[0])
        tt = time.perf_counter(); step(src=dXd, nsnp=Pd); ctx.sync(); t_dos = time.perf_counter() - tt
        dXd.free(); del Xd
        step()                 # the shard's own rows back in `res` for the checks below
        ctx.sync()
    # work-per-SNP tail: one more pass with the evaluation trace on
    dtrace = ctx.alloc(4 * B)
    tail = None
    try:
        _lib.check(L.pg_assoc_set_eval_trace(ctx.handle, dtrace.ptr), "trace on")
        e0, e1 = new_event(), new_event()
        s0, s1 = batches[0]
        _lib.check(L.pg_rotate_dev(ctx.handle, n, s1 - s0, dU.ptr, n, dX.ptr, P, dXr.ptr, ldx), "pg_rotate_dev")
        L.pg_event_record(ctx.handle, e0)
        _lib.check(L.pg_assoc_dev(ctx.handle, n, c, s1 - s0, dev.ptr, dW.ptr, dy.ptr, dXr.ptr, ldx, a.grid, pb, ps, pt, pl, pF, pP, None), "pg_assoc_dev")
        L.pg_event_record(ctx.handle, e1)
        L.pg_event_elapsed_ms(ctx.handle, e0, e1, C.byref(ms))
        tr = dtrace.download((s1 - s0,), np.uint32)
        ev = (tr & 0xffff).astype(np.int64) + (tr >> 16).astype(np.int64)
        tail = {"evals_per_snp_median": float(np.median(ev)), "evals_per_snp_p99": float(np.quantile(ev, 0.99)), "evals_per_snp_max": int(ev.max()),
                "newton_max": int((tr >> 16).max()), "ms_per_launch_traced": float(ms.value), "snps": int(s1 - s0),
                "note": "SNP-specific evaluations (Brent + Newton + final logL) per SNP on the first batch; the 11-point decade scan is extra and the same for every SNP"}
    finally:
        L.pg_assoc_set_eval_trace(ctx.handle, None)
    host = (allres if comm is not None else res).download((32 * P,), np.uint8)
    beta = host[16 * P: 20 * P].view(np.float32)
    pv = host[8 * P: 16 * P].view(np.float64)
    lam = host[28 * P: 32 * P].view(np.float32)
    assert np.isfinite(beta).all() and ((pv >= 0) & (pv <= 1)).all(), "non-finite results"

    value = world * P * a.steps / elapsed
    rot_avg, assoc_avg = float(np.mean(rot_ms)) * 1e-3, float(np.mean(assoc_ms)) * 1e-3
    rot_flops = 2.0 * n * n * B                       # algorithmic: 2 n^2 per SNP (SURVEY 8d stage R)
    m = c + 2
    # algorithmic fp64 flops of the assoc stage per SNP: decade scan 11 lambdas x 2 powers x m entries x 2n,
    # + per SNP-specific evaluation m(m+1)/2 entries x (2 | 3) powers x 2n
    assoc_flops_snp = 11 * 2 * m * 2.0 * n + (stats[0] * 2 + stats[1] * 3) * (m * (m + 1) / 2) * 2.0 * n
    # SURVEY 8(d) row S-brent as written: (4 E_b + 6 E_n) n m with m = c + 2 entries per evaluation (E_b counts the 11 scan points too)
    assoc_flops_snp_8d = (4.0 * (11 + stats[0]) + 6.0 * stats[1]) * n * m
    used_geno = (not a.fp32_rotate) and used_path == 1
    geno_i8 = os.environ.get("PG_GENO_I8", "1") != "0"      # the library's default: genotype codes on the int8 pipe
    tr_rot, src_rot, shape_rot = pmc_traffic(("rotate_geno_i8_kernel" if geno_i8 else "rotate_geno_kernel") if used_geno else "rotate_kernel")
    tr_as, src_as, shape_as = pmc_traffic("assoc_kernel")
    same_shape = (n, B, c) == shape_rot == shape_as          # the counters were collected per launch of this batch shape
    rl_rotate = ({"kernel": "rotate_geno_i8_kernel (+detect/encode): int8 MFMA 16x16x64, U as 3 int8 digit planes per eigenvector, exact int32 accumulate",
                  "bound": "mfma", "achieved": rot_flops / rot_avg / 1e12, "peak": I8_MFMA_PEAK_TOPS / 3.0, "unit": "TFLOP/s",
                  "frac": rot_flops / rot_avg / 1e12 / (I8_MFMA_PEAK_TOPS / 3.0),
                  "peak_note": "algorithmic 2n^2 flop/SNP against the dense int8 MFMA peak (5000 TOP/s) divided by the 3 int8 passes a "
                               "24-bit U needs; executed int8 rate = 3x achieved (+ 1.1 % K padding and 1/256 pad rows)"}
                 if used_geno and geno_i8 else
                 {"kernel": "rotate_geno_kernel (+detect/encode): fp16 MFMA 16x16x32, U split in 2 fp16 planes, fp32 accumulate",
                  "bound": "mfma", "achieved": rot_flops / rot_avg / 1e12, "peak": F16_MFMA_PEAK_TF / 2.0, "unit": "TFLOP/s",
                  "frac": rot_flops / rot_avg / 1e12 / (F16_MFMA_PEAK_TF / 2.0),
                  "peak_note": "algorithmic 2n^2 flop/SNP against the dense fp16 MFMA peak (2500 TF) divided by the 2 fp16 passes a "
                               "24-bit U needs; executed fp16 rate = 2x achieved"}
                 if used_geno else
                 {"kernel": "rotate_kernel<4> (fp32 MFMA 32x32x2)", "bound": "mfma", "achieved": rot_flops / rot_avg / 1e12,
                  "peak": F32_MFMA_PEAK_TF, "unit": "TFLOP/s", "frac": rot_flops / rot_avg / 1e12 / F32_MFMA_PEAK_TF})
    rl_rotate.update({"traffic": tr_rot if same_shape else None, "traffic_source": src_rot if same_shape else None,
                      "algorithmic_bytes": 4.0 * n * B + 4.0 * n * n + 4.0 * ldx * B, "avg_launch_ms": rot_avg * 1e3, "units_per_launch": B})
    rl_assoc = {"kernel": "assoc_kernel<%d> (+setup, p-values): fp64 VALU FMAs of the Gram passes, wave per SNP" % c, "bound": "valu",
                "achieved": assoc_flops_snp * B / assoc_avg / 1e12, "peak": F64_VALU_PEAK_TF, "unit": "TFLOP/s",
                "frac": assoc_flops_snp * B / assoc_avg / 1e12 / F64_VALU_PEAK_TF,
                "flop_model": "this engine's own count, NOT SURVEY 8(d)'s: 11-point decade scan x 2 powers x m entries x 2n, plus per SNP-specific "
                              "evaluation ALL m(m+1)/2 = %d Gram entries x (2|3) powers x 2n (lambda is per SNP, so the entries among W and y cannot be "
                              "shared between SNPs); SURVEY 8(d) S-brent as written counts m = %d entries per evaluation: (4 E_b + 6 E_n) n m" % (m * (m + 1) // 2, m),
                "flops_per_snp_survey_8d_literal": assoc_flops_snp_8d,
                "frac_survey_8d_literal": assoc_flops_snp_8d * B / assoc_avg / 1e12 / F64_VALU_PEAK_TF,
                "peak_note": "fp64 vector peak = fp64 matrix peak on MI355X (78.6 TF); only the Gram FMAs are counted (conversions, "
                             "h*w products, reciprocals, reductions and sweeps are not)",
                "traffic": tr_as if same_shape else None, "traffic_source": src_as if same_shape else None,
                "algorithmic_bytes": (4.0 * ldx + 36) * B, "avg_launch_ms": assoc_avg * 1e3, "units_per_launch": B,
                "flops_per_snp": assoc_flops_snp, "hbm_GBps_algorithmic": (4.0 * n + 36) * B / assoc_avg / 1e9}
    rot_label = ("f32 MFMA" if not used_geno else
                 "i8x3->i32 MFMA (genotype codes exact, integer accumulation exact; U as 24-bit fixed point per eigenvector in three int8 planes)" if geno_i8
                 else "f16x2->f32 MFMA (genotype codes exact; U in two fp16 planes)")
    out = {
        "metric": "SNPs/sec (whole node) at n=10,000 c=5; K-eigendecomp wallclock",
        "value": value, "unit": "SNPs/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": 1e3 * elapsed / a.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": f"{rot_label} rotation + f64 Gram/sweeps (f32 rounding points of the reference)", "data": "synthetic",
        "config": {"workload": f"BASELINE configs[2]: synthetic n={n}, c={c}, p={P} SNPs per GPU per step in {len(batches)} HBM-resident batches of "
                               f"<= {B}; X = {'raw 0/1/2 hard calls' if a.raw_codes else 'hard calls standardised per SNP (SURVEY 8d)'}, float32 (n, p): rotate (U'X) + REML {'grid' if a.grid else 'decade-scan+Brent+Newton'} + Wald F + p on device"
                               + ("; one RCCL all-gather of the result rows per step" if comm is not None else ""),
                   "n": n, "c": c, "snps_per_gpu_per_step": P, "batch": B, "lambda_path": "grid" if a.grid else "brent",
                   "phenotype": "pure noise" if a.null else ("weak signal h2=0.02" if a.weak else "polygenic h2=0.5 + one causal SNP"),
                   "parallelism": f"snp-shards x{world}"},
        "eigh_seconds": min(eigh_s) if eigh_s else None,
        "eigh_note": "fp64 eigensolver on device, n=%d, one-time, on rank 0 only; U, d, rotated y/W broadcast over RCCL" % n,
        "kinship_syrk_seconds": kin_s,
        "rccl_broadcast_seconds": t_bc,
        # the dominant kernel of the step = the longer of the two launches
        "roofline": rl_assoc if assoc_avg >= rot_avg else rl_rotate,
        "roofline_rotate": rl_rotate,
        "roofline_assoc": rl_assoc,
        # the metric names the eigendecomposition too: its 10 n^3 / 3 flops (tridiagonalisation 4/3 + back-transformation 2: the count of a
        # one-stage LAPACK dsyevd; the two-stage solver executes more) against the fp64 MFMA peak
        "roofline_eigh": ({"bound": "mfma", "achieved": 10.0 * n ** 3 / 3.0 / min(eigh_s) / 1e12, "peak": 78.6, "unit": "TFLOP/s",
                           "frac": 10.0 * n ** 3 / 3.0 / min(eigh_s) / 78.6e12, "flop_model": "10 n^3 / 3 (dsyevd count) / eigh_seconds",
                           "by_phase": "profiles/r04_syevd_summary.json"} if eigh_s else None),
        "rotation_path": "fp32 MFMA (reference-arithmetic sgemm class)" if not used_geno else ("genotype i8x3" if geno_i8 else "genotype f16x2"),
        "stage_snps_per_s_per_gpu": {"rotate": B / rot_avg, "assoc": B / assoc_avg},
        "evals_per_snp": {"fast": float(stats[0]), "newton": float(stats[1])},
        "lambda_median": float(np.median(lam)),
        "work_tail": tail,
        "cpu_model": cpu_model(),
    }
    if t_other is not None:
        key = "value_genotype_rotate" if a.fp32_rotate else "value_fp32_rotate"
        out[key] = {"value": P / t_other, "unit": "SNPs/s", "ms_per_step": 1e3 * t_other,
                    "note": ("same pass with the genotype rotation" if a.fp32_rotate else
                             "same pass with the fp32-MFMA rotation forced (--fp32-rotate 1): the reference-arithmetic figure, bit-comparable "
                             "to an fp32 fma chain; also the rate for X holding NaN/Inf") + "; one untimed-by-the-metric pass"}

    if t_int8 is not None:
        out["value_int8_X"] = {"value": P / t_int8, "unit": "SNPs/s", "ms_per_step": 1e3 * t_int8,
                               "note": "same pass with X resident as the int8 genotype matrix (a dtype lmm.pygemma accepts as it is, lmm/lmm.py:121-122 "
                                       "casts any dtype): detect/encode scan 1 byte per genotype; one untimed-by-the-metric pass"}
    if t_dos is not None:
        out["value_dosage_split"] = {"value": Pd / t_dos, "unit": "SNPs/s", "ms_per_step": 1e3 * t_dos, "snps": Pd, "rotation_path_code": dos_path,
                                     "note": "same rotate + assoc pass on %d SNPs of non-genotype finite X (hard call + uniform(-0.3, 0.3), clipped to [0, 2]): "
                                             "the device's detection sends the block to the split-plane rotation (X itself in two fp16 planes, 2 passes); "
                                             "one untimed-by-the-metric pass" % Pd}
    # ---------------- end to end through the public entry point, host inputs, H2D and eigh included (N = 1 only)
    if a.e2e and world == 1 and Khost is not None:
        try:
            from pygemma_amd import lmm
            dX_host = Xstd if Xstd is not None else dX.download((n, P), np.float32)
            Xp = lmm.pinned_empty((n, P), np.float32)
            Xp[:] = dX_host
            Yh, Wh = np.ascontiguousarray(yW[:, :1]), np.ascontiguousarray(yW[:, 1:])     # the un-rotated inputs rank 0 started from
            e2e = {}
            for tag, Xin in (("pinned_X", Xp), ("pageable_X", dX_host)):
                st = {}
                tt = time.perf_counter()
                df = lmm.pygemma(Yh, Xin, Wh, Khost, stats=st, verbose=int(os.environ.get("PYGEMMA_BENCH_VERBOSE", "0")))
                dt = time.perf_counter() - tt
                e2e[tag] = {"seconds": dt, "snps_per_s": P / dt, "snp_loop_seconds": st.get("seconds"), "snp_loop_snps_per_s": P / st["seconds"],
                            "host_to_device_GBps_incl_compute": st["bytes_in"] / st["seconds"] / 1e9, "batches": st["batches"],
                            "prefetched_batches": st.get("prefetched_batches", 0),
                            "stages_s": {k: round(float(st[k]), 4) for k in ("dma_s", "kernel_s", "blocks_s", "setup_s", "worker_alloc_s") if k in st}}
                assert np.isfinite(df["beta"].to_numpy()).all()
            e2e["note"] = (f"lmm.pygemma(Y, X, W, K) from host float32 arrays, p={P}: K upload + eigh + rotation + scan + frame; snp_loop = the "
                           "streamed SNP loop alone (H2D DMA of X included). PCIe-inclusive, never `value`.")
            out["e2e"] = e2e
            del Xp, dX_host
        except Exception as ex:
            out["e2e"] = {"error": repr(ex)}

    # ---------------- CPU baseline: bounded sample of the SAME workload on the box's host cores, rank 0 / N=1 only.
    # rotation = numpy @ (OpenBLAS sgemm: what the reference calls, lmm.py:244) when NumPy has a BLAS, else the oracle's fma chain;
    # per-SNP path = the oracle ("port": C + OpenMP restatement, bit-validated against the reference in the build container).
    if a.cpu_sample > 0 and world == 1:
        try:
            from oracle import oracle as O
            S = min(a.cpu_sample, B)
            Uh = dU.download((n, n), np.float32)
            dh = dev.download((n,), np.float32)
            Xs = np.ascontiguousarray(dX.download((n, P), np.float32)[:, :S])
            yr_h, Wr_h = dy.download((n,), np.float32), dW.download((n, c), np.float32)
            nthr = min(O.lib().orc_max_threads(), host_cores())
            tt = time.time()
            Xrs = np.ascontiguousarray((Uh.T @ Xs).T)                 # (S, n) SNP-major
            t_rot = time.time() - tt
            rot_kind = "numpy @ (BLAS sgemm)"
            tt = time.time()
            orc = O.calculate(dh, yr_h, Wr_h, Xrs, grid=bool(a.grid), order=0, nthreads=nthr, snp_major=True)
            t_as = time.time() - tt
            rel = float(np.max(np.abs(orc["beta"].astype(np.float64) - beta[:S]) / orc["se_beta"].astype(np.float64)))
            cb = {"value": S / (t_rot + t_as), "unit": "SNPs/s", "cores": int(nthr), "kind": "port", "cpu_model": cpu_model(),
                  "sample": f"first {S} SNPs of the shard: rotation {rot_kind} {t_rot:.3f} s + oracle calculate {t_as:.2f} s (OpenMP, {nthr} threads); "
                            f"GPU vs CPU beta: max |dbeta|/se {rel:.1e}",
                  "per_snp_path_snps_per_s": S / t_as, "rotation_snps_per_s": S / t_rot}
            try:
                rt = json.load(open(os.path.join(ROOT, "profiles", "r02_reference_timing.json")))
                cb["reference_timing_fixture"] = {
                    "file": "profiles/r02_reference_timing.json",
                    "note": "the REAL reference (Cython) timed in the build container on the same kind of slice (it cannot travel to the GPU box)",
                    "where": rt["where"], "cpu_model": rt["cpu_model"], "cores": rt["cores"],
                    "reference_brent_nproc8_snps_per_s": rt["per_snp_path"]["reference_brent_nproc8"]["snps_per_s"],
                    "oracle_brent_threads8_snps_per_s": rt["per_snp_path"]["oracle_brent_threads8"]["snps_per_s"],
                    "oracle_over_reference": rt["ratio_oracle8_over_reference8_brent"],
                    "reference_eigh_float32_seconds": rt["eigh_float32"]["seconds"]}
            except Exception:
                pass
            out["cpu_baseline"] = cb
        except Exception as ex:   # the baseline is a report, not a dependency of the measurement
            out["cpu_baseline"] = {"value": None, "unit": "SNPs/s", "cores": 0, "kind": "port", "sample": f"failed: {ex!r}"}
    print(json.dumps(out), flush=True)
    if comm is not None:
        comm.barrier()
        comm.close()


if __name__ == "__main__":
    main()
