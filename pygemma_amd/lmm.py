"""lmm.pygemma(Y, X, W, K, snps=...) — drop-in for the reference's entry point (lmm/lmm.py:87-411) with
everything under it running on MI355X through the C ABI (include/pygemma_hip.h):

    eigh(K)                     -> pg_syevd_dev   (fp64 Householder + divide & conquer, lmm.py:152/197)
    U.T @ X, U.T @ Y, U.T @ W   -> pg_rotate_geno_dev for genotype-valued SNP blocks (f16x2 MFMA), else
                                   pg_rotate_dev  (fp32 MFMA GEMM)                      (lmm.py:243-246)
    Pool(nproc).imap(calculate) -> pg_assoc_dev   (wave-per-SNP fused lambda search + Wald test, lmm.py:378-403,461-495)
    stats.f.sf                  -> on device      (lmm.py:482)

Same signature, same casts to float32 (lmm.py:115-128), same output DataFrame schema (columns beta, se_beta,
tau as float32; lambda, F_wald, p_wald as float64; SNPs as object when `snps` is given; lmm.py:403-409).
`nproc` = number of GPUs to spread contiguous SNP blocks over, exactly like SampleIter's ceil(p/nproc)
column blocks (lmm.py:427-434); results come back in block order = SNP order.

Streaming (BASELINE configs 4-5): X never has to fit on a GPU.  Every GPU takes its block in batches; a batch is DMA'd
from PINNED host memory (directly out of X: the caller's own pinned array — `pinned_empty`/`pin` — or X page-locked in place
for the duration of the call; a pinned staging buffer filled by copy threads only where that fails) on one stream while the
previous batch computes on another, and its 32-byte
result rows come back through a pinned buffer.  With several GPUs the eigenvectors travel once, GPU 0 -> all, as one
RCCL broadcast over xGMI (pg_comm_*; no PyTorch).
There is no CPU fallback: without the HIP library or a GPU this raises.
"""
import contextlib
import ctypes as C
import json
import os
import threading
import time
import zlib

import numpy as np
import pandas as pd

from . import _lib
from ._lib import pin, pinned_empty     # noqa: F401  (re-exported: how a caller hands over pinned inputs)
from .bed import PackedBed
from .model import *          # noqa: F401,F403  (precompute_mat, calc_lambda_restricted, newton, the *_overload scalars)
from . import model as _model

__all__ = ["pygemma", "SampleIter", "pinned_empty", "pin", "kinship"] + _model.__all__

_BATCH_BYTES = 6 << 30   # device bytes for one SNP batch of one worker (raw block, rotated block, fp16 planes)
_BATCH_SNPS = 32768      # SNPs per batch at most: the unit of copy/compute overlap and of checkpointing
_WORKERS = 2             # host threads (each with its own stream and buffers) per GPU
_SERIAL_KERNELS = False  # True: the workers of a GPU take turns on its compute units (measured: no gain for float32 X, 8 % slower for int8 X)
_BATCH_MIN = 8192        # ... and at least, while the block is cut into up to _BATCH_COUNT batches: the first batch's DMA and the
_BATCH_COUNT = 12        # last batch's kernels are the part of the loop that does not overlap (measured at p = 100 000: 4 batches
                         # 0.149-0.172 s, 8: 0.116, 12: 0.112, 16: 0.112 — tools/ab_stream_batch.py)
_STAGE_THREADS = 8       # host copy threads per worker for the pageable -> pinned leg
# Device bytes of X per GPU that may be copied in while the eigensolver runs.  OFF by default (0): with the caller's X in
# hipHostMalloc'ed memory (pinned_empty) it takes the copy time out of the loop (p = 400 000 float32: 0.96 -> 0.90 s end to end), but
# inside a long-lived process the first allocations after the eigensolver were seen to stall for ~0.15 s in one run out of four, and
# from memory page-locked in place (hipHostRegister) the run was 0.25 s SLOWER every time (tools/ab_stream_batch.py, bench.py e2e leg
# with PYGEMMA_PREFETCH_MAX=1e11).  Opt-in: PYGEMMA_PREFETCH_MAX=<bytes>; only an X the caller pinned himself is prefetched.
_PREFETCH_MAX = int(float(os.environ.get("PYGEMMA_PREFETCH_MAX", 0)))
_PREFETCH_MARGIN = 8 << 30  # ... and what is left untouched beside the eigensolver's and the workers' buffers


class SampleIter:
    """Contiguous column blocks of ceil(p/nproc) SNPs (lmm/lmm.py:413-436) — here: one block per GPU."""

    def __init__(self, p, nproc):
        self.p, self.nproc = int(p), int(nproc)

    def __iter__(self):
        cols = int(np.ceil(self.p / self.nproc))
        for r in range(self.nproc):
            a, b = r * cols, min((r + 1) * cols, self.p)
            if a < b:
                yield a, b


_T_LOG = [time.time()]


def _log(verbose, msg):
    if verbose > 0:
        now = time.time()
        print(f"[pygemma_amd +{now - _T_LOG[0]:7.3f} s] {msg}", flush=True)
        _T_LOG[0] = now


def _rotate_small(ctx, L, n, dU, A):
    """U.T @ A for a narrow host matrix A (n,q) (Y and W): same MFMA kernel, SNP-major result transposed back."""
    q = A.shape[1]
    ldx = (n + 63) // 64 * 64
    dA = ctx.to_device(np.ascontiguousarray(A, np.float32))
    dO = ctx.alloc(q * ldx * 4)
    _lib.check(L.pg_rotate_dev(ctx.handle, n, q, dU.ptr, n, dA.ptr, q, dO.ptr, ldx), "pg_rotate_dev")
    ctx.sync()
    out = dO.download((q, ldx), np.float32)[:, :n].T.copy()
    dA.free(); dO.free()
    return out


_COLS = ("beta", "se_beta", "tau", "lambda", "F_wald", "p_wald")
_LRT_COLS = ("l_alt", "l_null", "D_lrt", "p_lrt")


def _part_path(ckpt, s, e):
    return os.path.join(ckpt, f"part_{s:012d}_{e:012d}.npz")


def _crc(a):
    return int(zlib.crc32(np.ascontiguousarray(a).view(np.uint8).reshape(-1)))


def _block_fingerprint(X, s, e):
    """Cheap identity of the raw SNP block [s, e): CRC-32 of 64 evenly spaced sample rows (of byte columns for packed
    records).  Stored in every checkpoint part and compared before a part is restored, so that parts of a run on another
    genotype matrix of the same shape are refused instead of being mixed into the result."""
    if isinstance(X, PackedBed):
        rec = X.data[s:e]
        step = max(1, rec.shape[1] // 64)
        return _crc(rec[:, ::step])
    n = X.shape[0]
    rows = np.unique(np.linspace(0, n - 1, min(n, 64)).astype(np.int64))
    return _crc(X[rows, s:e])


class _Staging:
    """Pinned host buffers of one worker: the raw batch on its way in (unless X itself is pinned) and the result rows on
    their way out."""

    def __init__(self, ctx, L, in_bytes, out_bytes):
        self.ctx, self.L = ctx, L
        self.inp = self._alloc(in_bytes) if in_bytes else None
        self.out = self._alloc(out_bytes)

    def _alloc(self, nbytes):
        p = C.c_void_p()
        _lib.check(self.L.pg_host_alloc(self.ctx.handle, int(nbytes), C.byref(p)), "pg_host_alloc")
        return p.value

    def close(self):
        for q in (self.inp, self.out):
            if q:
                self.L.pg_host_free(self.ctx.handle, q)
        self.inp = self.out = None


def _batch_geometry(n, a, b):
    """(SNPs per batch, row pitch in elements of a raw sample-major batch) for the block [a, b) of one GPU."""
    ldx = (n + 63) // 64 * 64
    pb_max = max(256, int(_BATCH_BYTES // (12 * ldx)) // 256 * 256)   # raw block + rotated block + two fp16 planes
    pb_max = min(pb_max, _BATCH_SNPS, b - a)
    nbat = max(-(-(b - a) // pb_max), min(_BATCH_COUNT, -(-(b - a) // _BATCH_MIN)))
    pb_max = min(pb_max, (-(-(b - a) // nbat) + 255) // 256 * 256)
    return pb_max, (pb_max + 15) // 16 * 16


class _View:
    """A range of a device buffer (what a worker needs of a batch: its address)."""

    def __init__(self, ptr):
        self.ptr = ptr


class _Prefetch:
    """X on its way to the devices WHILE the eigensolver runs (eigen=True from K; X a C-contiguous, page-locked host matrix): one
    host thread per GPU DMAs the leading batches of that GPU's block into ONE device buffer (allocated here, before the eigensolver
    starts: no allocation competes with its launches) on a stream of its own — PCIe and the HBM-bound tridiagonalisation do not
    compete — within what hipMemGetInfo leaves after the eigensolver's workspace (~60 n^2 bytes on GPU 0) and the workers' own
    buffers.  The SNP loop then takes those batches instead of issuing the copies: the loop runs at the kernels' rate, and at
    288 GB of HBM a 40 GB float32 X is resident before U is."""

    def __init__(self, L, X, blocks, n, esz, verbose):
        self.L, self.X, self.n, self.esz, self.verbose = L, X, n, esz, verbose
        self.lock = threading.Lock()
        self.ready = [dict() for _ in blocks]
        self.ctxs, self.plans = [], []
        self.bytes = 0
        self.stop = False                        # set when the eigensolver is done: the batch in flight is the last one
        self.threads = []
        try:
            self._plan(blocks, n, esz)
        except BaseException:                    # a context or buffer of GPU g failed: release those of the GPUs before it
            for ctx in self.ctxs:
                ctx.close()
            self.ctxs = []
            raise
        self.threads = [threading.Thread(target=self._run, args=(g,)) for g in range(len(blocks))]
        for th in self.threads:
            th.start()

    def _plan(self, blocks, n, esz):
        for g, (a, b) in enumerate(blocks):
            ctx = _lib.Context(g)
            self.ctxs.append(ctx)
            free, _total = ctx.mem_info()
            reserve = (96 * n * n if g == 0 else 12 * n * n) + 2 * _BATCH_BYTES + _PREFETCH_MARGIN      # GPU 0: the eigensolver's 10 n^2 doubles + its outputs
            budget = min(free - reserve, _PREFETCH_MAX)
            pb_max, ldX = _batch_geometry(n, a, b)
            per = n * ldX * esz
            batches = [(s, min(s + pb_max, b)) for s in range(a, b, pb_max)]
            batches = batches[: max(0, int(budget // per))]
            big = ctx.alloc(per * len(batches)) if batches else None
            self.plans.append((batches, big, per, ldX))

    def _run(self, g):
        try:
            n, esz, p = self.n, self.esz, self.X.shape[1]
            ctx = self.ctxs[g]
            batches, big, per, ldX = self.plans[g]
            done = 0
            for k, (s, e) in enumerate(batches):
                if self.stop:
                    break
                dst = big.ptr + k * per
                _lib.check(self.L.pg_memcpy2d_h2d_async(ctx.handle, dst, ldX * esz, self.X.ctypes.data + esz * s, p * esz, (e - s) * esz, n),
                           "pg_memcpy2d_h2d_async")
                ctx.sync()                       # a batch is offered only once it has landed
                with self.lock:
                    self.ready[g][(s, e)] = _View(dst)
                    self.bytes += n * (e - s) * esz
                done += n * (e - s) * esz
                _log(self.verbose - 1, f"GPU {g}: SNPs [{s},{e}) prefetched")
            _log(self.verbose, f"GPU {g}: {done / 1e9:.2f} GB of X prefetched during the eigendecomposition")
        except Exception as ex:                  # prefetching is an optimisation: on any failure the loop copies as usual
            _log(self.verbose, f"GPU {g}: prefetch stopped ({ex!r})")

    def join(self):
        self.stop = True
        for th in self.threads:
            th.join()

    def take(self, g, s, e):
        with self.lock:
            return self.ready[g].pop((s, e), None)

    def close(self):
        self.join()
        for ctx in self.ctxs:
            ctx.close()                          # frees the buffers (after the loop: hipFree inside it would drain the device)
        self.ctxs = []


def _run_block(device, a, b, n, c, d, Wr, yr, X, dU, comm, grid, eigen, lrt, out, errs, verbose, ckpt=None, stats=None, pre=None):
    """One GPU: SNP columns [a,b) of X through (rotate | transpose) -> assoc, in batches.  Two host threads per GPU, each with
    its own stream, device buffers and pinned staging, take batches from a shared list, so that the host->device DMA of one
    batch overlaps the kernels of the other.  `dU`: GPU 0's resident eigenvectors (device 0) or None; with a communicator the
    other GPUs receive them by RCCL broadcast.  With `ckpt` every finished batch is written to disk (and batches found there,
    with a matching fingerprint of the raw block, are not redone)."""
    try:
        L = _lib.load()
        ctx0 = comm.ctx if comm is not None else _lib.Context(device)
        t_blk = time.time()
        try:
            ldx = (n + 63) // 64 * 64
            dprep = None
            if eigen and comm is not None:
                # U: resident on GPU 0 (the eigensolver's output); the other GPUs get it over xGMI.  The collective comes before
                # anything else of this thread that can fail (its receive buffer was allocated by the caller), so that no GPU is
                # left waiting in the broadcast for a peer that has already given up.
                t0 = time.time()
                _lib.check(L.pg_comm_broadcast_dev(comm.handle, dU.ptr, n * n * 4, 0), "pg_comm_broadcast_dev")
                ctx0.sync()
                _log(verbose, f"GPU {device}: U ({n * n * 4 / 1e9:.2f} GB) broadcast over RCCL in {time.time() - t0:.3f} s")
            dd, dW, dy = ctx0.to_device(d), ctx0.to_device(Wr), ctx0.to_device(yr)
            if eigen:
                # genotype fast path of the rotation (<= 3 equally spaced values per column), fp32 MFMA otherwise
                dprep = ctx0.alloc(L.pg_geno_prep_bytes(n))
                _lib.check(L.pg_geno_prep_dev(ctx0.handle, n, dU.ptr, n, dprep.ptr), "pg_geno_prep_dev")
            ctx0.sync()
            packed = isinstance(X, PackedBed)
            x8 = (not packed) and X.dtype in (np.int8, np.uint8)
            x64 = (not packed) and X.dtype == np.float64
            force_fp32 = os.environ.get("PYGEMMA_ROTATE", "") == "fp32" and not (packed or x8 or x64)
            esz = 1 if x8 else (8 if x64 else 4)
            pb_max, ldX = _batch_geometry(n, a, b)
            bpr = (n + 3) // 4
            p = X.shape[1]
            direct = (not packed) and _lib.is_pinned(X)        # X itself is page-locked: DMA straight out of it
            snp_major = (not packed) and X.flags.f_contiguous and not X.flags.c_contiguous     # (float32 only: pygemma() sees to it)
            nout = 32 + (32 if lrt else 0)                      # result bytes per SNP
            todo = []
            for s in range(a, b, pb_max):
                e = min(s + pb_max, b)
                if ckpt and os.path.exists(_part_path(ckpt, s, e)):
                    with np.load(_part_path(ckpt, s, e)) as z:
                        if "fingerprint" not in z.files or int(z["fingerprint"]) != _block_fingerprint(X, s, e):
                            raise ValueError(f"checkpoint part {_part_path(ckpt, s, e)} was computed from different genotypes "
                                             f"(fingerprint of SNPs [{s},{e}) differs): refusing to mix runs")
                        for col in _COLS + (_LRT_COLS if lrt else ()):
                            out[col][s:e] = z[col]
                    _log(verbose, f"GPU {device}: SNPs [{s},{e}) restored from {ckpt}")
                else:
                    todo.append((s, e))
            lock = threading.Lock()
            gpu = threading.Lock() if _SERIAL_KERNELS else contextlib.nullcontext()   # one batch's kernels at a time; the other worker's DMA runs under them
            if stats is not None:
                stats["setup_s"] = max(stats.get("setup_s", 0.0), time.time() - t_blk)

            def worker():
                try:
                    ctx = _lib.Context(device)
                    stg = None
                    t_w = time.time()
                    try:
                        raw_bytes = pb_max * bpr if packed else n * ldX * esz
                        # one device allocation per worker, cut into its buffers: every hipMalloc / hipFree of a few hundred MB costs
                        # 1 - 2 ms of host time that nothing overlaps (r4: worker set-up 3.8 ms, teardown 9.2 ms with five allocations)
                        sizes = [raw_bytes, pb_max * n * 4 if (snp_major and eigen) else 0, pb_max * ldx * 4, pb_max * nout,
                                 L.pg_geno_work_bytes(n, pb_max) if eigen else 0]
                        offs = np.concatenate([[0], np.cumsum([(int(z) + 255) // 256 * 256 for z in sizes])])
                        arena = ctx.alloc(int(offs[-1]))
                        dX, dT, dXr, dres, dwork = (_View(arena.ptr + int(o)) if z else None for o, z in zip(offs[:-1], sizes))
                        dXf = None       # float32 image of an 8-bit block, only if one does not qualify for the genotype path
                        stg = _Staging(ctx, L, 0 if direct else raw_bytes, pb_max * nout)
                        hres = (C.c_char * (pb_max * nout)).from_address(stg.out)
                        _lib.check(L.pg_assoc_warm(ctx.handle, n, c), "pg_assoc_warm")      # the first batch's kernels then queue without a host stall
                        if stats is not None:
                            with lock:
                                stats["worker_alloc_s"] = max(stats.get("worker_alloc_s", 0.0), time.time() - t_w)
                        while True:
                            with lock:
                                if not todo or errs:
                                    return
                                s, e = todo.pop(0)
                            pb = e - s
                            t_in = time.time()
                            pbuf, dXc = None, dX     # dXc: where this batch's raw block is
                            # ---- the raw block travels to the device on this worker's stream
                            if packed:   # SNP records [s, e) of the .bed image: contiguous bytes
                                rec = X.data[s:e]
                                if rec.flags.c_contiguous:
                                    _lib.check(L.pg_stage_rows(stg.inp, rec.nbytes, rec.ctypes.data, rec.nbytes, rec.nbytes, 1, 1), "pg_stage_rows")
                                else:
                                    C.memmove(stg.inp, np.ascontiguousarray(rec).ctypes.data, pb * bpr)
                                _lib.check(L.pg_memcpy_h2d_async(ctx.handle, dX.ptr, stg.inp, pb * bpr), "pg_memcpy_h2d_async")
                            elif snp_major:    # SNPs [s, e) are pb contiguous rows of n floats
                                src = X.ctypes.data + 4 * n * s
                                dst, dpitch = (dT.ptr, 4 * n) if eigen else (dXr.ptr, 4 * ldx)       # eigen=False: already the layout the scan reads
                                if direct:
                                    _lib.check(L.pg_memcpy2d_h2d_async(ctx.handle, dst, dpitch, src, 4 * n, 4 * n, pb), "pg_memcpy2d_h2d_async")
                                else:
                                    _lib.check(L.pg_stage_rows(stg.inp, 4 * n, src, 4 * n, 4 * n, pb, _STAGE_THREADS), "pg_stage_rows")
                                    _lib.check(L.pg_memcpy2d_h2d_async(ctx.handle, dst, dpitch, stg.inp, 4 * n, 4 * n, pb), "pg_memcpy2d_h2d_async")
                                if eigen:      # (pb x n) -> (n x ldX): the transposition kernel with the roles of n and p exchanged
                                    _lib.check(L.pg_transpose_dev(ctx.handle, pb, n, dT.ptr, n, dX.ptr, ldX), "pg_transpose_dev")
                            elif pre is not None and (pbuf := pre.take(device, s, e)) is not None:
                                dXc = pbuf       # landed on the device while the eigensolver ran
                            else:
                                src = X.ctypes.data + esz * s
                                if direct:
                                    _lib.check(L.pg_memcpy2d_h2d_async(ctx.handle, dX.ptr, ldX * esz, src, p * esz, pb * esz, n),
                                               "pg_memcpy2d_h2d_async")
                                else:    # pageable X: copy threads gather the column window into pinned staging, then one dense DMA
                                    _lib.check(L.pg_stage_rows(stg.inp, ldX * esz, src, p * esz, pb * esz, n, _STAGE_THREADS), "pg_stage_rows")
                                    _lib.check(L.pg_memcpy_h2d_async(ctx.handle, dX.ptr, stg.inp, n * ldX * esz), "pg_memcpy_h2d_async")
                            if _SERIAL_KERNELS:
                                ctx.sync()       # the block has landed; the kernels wait for the token, not for the link
                            t_dma = time.time()
                            with gpu:
                                t_tok = time.time()
                                # ---- rotation: decode + impute + rotate for .bed records; otherwise the path is chosen on the device
                                if packed:
                                    _lib.check(L.pg_rotate_bed_dev(ctx.handle, n, pb, dprep.ptr, dX.ptr, bpr, int(X.count_A1), dXr.ptr, ldx,
                                                                   dwork.ptr), "pg_rotate_bed_dev")
                                elif x64:
                                    is_geno = C.c_int(0)
                                    _lib.check(L.pg_rotate_geno_f64_dev(ctx.handle, n, pb, dprep.ptr, dXc.ptr, ldX, dXr.ptr, ldx, dwork.ptr,
                                                                        C.byref(is_geno)), "pg_rotate_geno_f64_dev")
                                    if not is_geno.value:
                                        dXf = dXf or ctx.alloc(n * ldX * 4)
                                        _lib.check(L.pg_cast_f64_f32_dev(ctx.handle, n, pb, dXc.ptr, ldX, dXf.ptr, ldX), "pg_cast_f64_f32_dev")
                                        _lib.check(L.pg_rotate_dev(ctx.handle, n, pb, dU.ptr, n, dXf.ptr, ldX, dXr.ptr, ldx), "pg_rotate_dev")
                                elif x8:         # 8-bit block (always finite): genotype codes or split planes, chosen on the device
                                    _lib.check(L.pg_rotate_auto_i8_dev(ctx.handle, n, pb, dprep.ptr, dXc.ptr, int(X.dtype == np.uint8), ldX,
                                                                       dXr.ptr, ldx, dwork.ptr, None), "pg_rotate_auto_i8_dev")
                                elif eigen and force_fp32:   # PYGEMMA_ROTATE=fp32: the reference-arithmetic kernel for every block (tests, A/B)
                                    _lib.check(L.pg_rotate_dev(ctx.handle, n, pb, dU.ptr, n, dXc.ptr, ldX, dXr.ptr, ldx), "pg_rotate_dev")
                                elif eigen:      # float32 block: path (genotype fp16x2 / split planes / fp32 MFMA) chosen on the device, no host wait
                                    _lib.check(L.pg_rotate_auto_dev(ctx.handle, n, pb, dU.ptr, n, dprep.ptr, dXc.ptr, ldX, dXr.ptr, ldx,
                                                                    dwork.ptr, None), "pg_rotate_auto_dev")
                                elif not snp_major:
                                    _lib.check(L.pg_transpose_dev(ctx.handle, n, pb, dXc.ptr, ldX, dXr.ptr, ldx), "pg_transpose_dev")
                                # result block: [F | p | beta | se | tau | lambda] (+ [l_alt | l_null | D_lrt | p_lrt] f64 with lrt)
                                r0 = dres.ptr
                                if lrt:
                                    _lib.check(L.pg_assoc_lrt_dev(ctx.handle, n, c, pb, dd.ptr, dW.ptr, dy.ptr, dXr.ptr, ldx, int(grid),
                                                                  r0 + 16 * pb, r0 + 20 * pb, r0 + 24 * pb, r0 + 28 * pb, r0, r0 + 8 * pb,
                                                                  r0 + 32 * pb, r0 + 40 * pb, r0 + 48 * pb, r0 + 56 * pb), "pg_assoc_lrt_dev")
                                else:
                                    _lib.check(L.pg_assoc_dev(ctx.handle, n, c, pb, dd.ptr, dW.ptr, dy.ptr, dXr.ptr, ldx, int(grid),
                                                              r0 + 16 * pb, r0 + 20 * pb, r0 + 24 * pb, r0 + 28 * pb, r0, r0 + 8 * pb, None),
                                               "pg_assoc_dev")
                                _lib.check(L.pg_memcpy_d2h_async(ctx.handle, stg.out, r0, pb * nout), "pg_memcpy_d2h_async")
                                ctx.sync()
                                t_ker = time.time()
                            if pbuf is not None and stats is not None:      # (the buffer is freed with the prefetcher: hipFree would drain the device here)
                                with lock:
                                    stats["prefetched_batches"] = stats.get("prefetched_batches", 0) + 1
                            hb = np.frombuffer(hres, np.uint8, pb * nout)
                            FP = hb[:16 * pb].view(np.float64).reshape(2, pb)
                            res = hb[16 * pb:32 * pb].view(np.float32).reshape(4, pb)
                            out["beta"][s:e], out["se_beta"][s:e], out["tau"][s:e] = res[0], res[1], res[2]
                            out["lambda"][s:e] = res[3].astype(np.float64)
                            out["F_wald"][s:e], out["p_wald"][s:e] = FP[0], FP[1]
                            if lrt:
                                LR = hb[32 * pb:64 * pb].view(np.float64).reshape(4, pb)
                                for k, col in enumerate(_LRT_COLS):
                                    out[col][s:e] = LR[k]
                            if stats is not None:
                                with lock:
                                    stats["batches"] += 1
                                    stats["bytes_in"] += pb * bpr if packed else n * pb * esz
                                    stats["batch_s"] += time.time() - t_in
                                    stats["dma_s"] = stats.get("dma_s", 0.0) + (t_dma - t_in)       # copy-in (waited for only when the kernels are serialised)
                                    stats["token_s"] = stats.get("token_s", 0.0) + (t_tok - t_dma)  # waiting for the other worker's kernels
                                    stats["kernel_s"] = stats.get("kernel_s", 0.0) + (t_ker - t_tok)
                            if ckpt:
                                tmp = _part_path(ckpt, s, e) + ".tmp.npz"
                                np.savez(tmp, fingerprint=np.int64(_block_fingerprint(X, s, e)),
                                         **{col: out[col][s:e] for col in _COLS + (_LRT_COLS if lrt else ())})
                                os.replace(tmp, _part_path(ckpt, s, e))
                            _log(verbose, f"GPU {device}: SNPs [{s},{e}) done")
                    finally:
                        t_td = time.time()
                        if stg is not None:
                            ctx.sync()
                            stg.close()
                        ctx.close()
                        if stats is not None:
                            with lock:
                                stats["worker_teardown_s"] = max(stats.get("worker_teardown_s", 0.0), time.time() - t_td)
                except Exception as ex:  # surfaced by the caller; never swallowed
                    errs.append(ex)

            workers = [threading.Thread(target=worker) for _ in range(min(_WORKERS, len(todo)))]
            for th in workers:
                th.start()
            for th in workers:
                th.join()
        finally:
            t_td = time.time()
            if comm is None:
                ctx0.close()
            if stats is not None:
                stats["block_teardown_s"] = max(stats.get("block_teardown_s", 0.0), time.time() - t_td)
    except Exception as ex:  # surfaced by the caller; never swallowed
        errs.append(ex)


class _Comm:
    """One communicator of a pg_comm_init_all group + the context it is bound to."""

    def __init__(self, handle, ctx):
        self.handle, self.ctx = handle, ctx


def _make_comms(L, ndev):
    ctxs = [_lib.Context(g) for g in range(ndev)]
    hs = (C.c_void_p * ndev)(*[c.handle for c in ctxs])
    outs = (C.c_void_p * ndev)()
    rc = L.pg_comm_init_all(ndev, hs, outs)
    if rc:
        msg = L.pg_last_error().decode()
        for h in outs:
            if h:
                L.pg_comm_destroy(h)
        for c in ctxs:
            c.close()
        raise _lib.PgError(f"pg_comm_init_all({ndev}) failed (code {rc}): {msg}")
    return [_Comm(C.c_void_p(outs[g]), ctxs[g]) for g in range(ndev)]


def kinship(G, standardize=True, device=0):
    """K = Z Z' / p from the (n, p) genotype matrix G on the GPU (the step before the path in the reference's callers:
    experiments/animal_gwas/run_gwas.py:46-56, tests/test_pygemma.py:184-192): columns centred and divided by their
    standard deviation (population, ddof=0; sd == 0 -> 1) when `standardize`, then a lower-triangle syrk on the MFMA pipe.
    Returns the (n, n) float32 matrix (both triangles)."""
    L = _lib.load()
    G = np.ascontiguousarray(G, np.float32)
    n, p = G.shape
    with _lib.Context(device) as ctx:
        dG = ctx.to_device(G)
        dK = ctx.alloc(n * n * 4)
        _lib.check(L.pg_kinship_geno_dev(ctx.handle, n, p, dG.ptr, p, int(bool(standardize)), dK.ptr), "pg_kinship_geno_dev")
        ctx.sync()
        return dK.download((n, n), np.float32)


def _zkzt(L, Z, K):
    """K <- Z K Z' (lmm/lmm.py:124-125) on GPU 0: Z (n, q), K (q, q), float32 or float64 each; returns the float32 (n, n) matrix that
    lmm.py:127-128 would hand to the eigensolver.  The reference does this as host BLAS products (4e12 flops at n = q = 10 000)."""
    if Z.ndim != 2 or K.ndim != 2 or K.shape[0] != K.shape[1] or Z.shape[1] != K.shape[0]:
        raise ValueError(f"Z {Z.shape} and K {K.shape} do not multiply as Z K Z'")
    def dev_ready(a):
        if a.dtype not in (np.float32, np.float64):
            a = a.astype(np.float64)
        return np.ascontiguousarray(a)
    Z, K = dev_ready(Z), dev_ready(K)
    n, q = Z.shape
    with _lib.Context(0) as ctx:
        dZ, dK, dO = ctx.to_device(Z), ctx.to_device(K), ctx.alloc(n * n * 4)
        _lib.check(L.pg_zkzt_dev(ctx.handle, n, q, dZ.ptr, int(Z.dtype == np.float64), q, dK.ptr, int(K.dtype == np.float64), q, dO.ptr, n),
                   "pg_zkzt_dev")
        out = dO.download((n, n), np.float32)
        for b_ in (dZ, dK, dO):
            b_.free()
    return out


def pygemma(Y, X, W, K, Z=None, snps=None, verbose=0, disable_checks=True, de=False, grid=False, eigen=True, nproc=1,
            checkpoint=None, lrt=False, eigenpairs=None, stats=None):
    """Per-SNP LMM association scan (GEMMA-style REML + Wald test) — signature of lmm/lmm.py:87.

    Y (n,1) phenotype; X (n,p) genotypes; W (n,c) covariates; K (n,n) relatedness matrix — or, with
    eigen=False, the (n,) vector of its eigenvalues with X, Y, W already rotated (lmm.py:164-167).
    Returns a pandas.DataFrame with columns beta, se_beta, tau, lambda, F_wald, p_wald[, SNPs].
    Beyond the reference (all keyword-only in spirit, defaults reproduce the reference):
      * X may be a `pygemma_amd.bed.PackedBed` (PLINK .bed kept packed; missing calls mean-imputed on the device, as the
        reference's callers do on the host before calling);
      * `checkpoint` names a directory that receives every finished SNP batch; a rerun with the same inputs picks up from
        what is there (the reference has no restart), a rerun with other inputs is refused;
      * `lrt=True` appends l_alt, l_null, D_lrt, p_lrt — the likelihood-ratio columns the reference sketches and leaves
        commented out (lmm.py:137-141, 277-300), from its own ML functions (lmm.py:22-84, pygemma_model.pyx:1542-1603);
      * `eigenpairs=(eigenVals, U)`: a precomputed eigendecomposition of K (U column j = eigenvector j, as scipy.linalg.eigh
        returns it) — K is ignored, U is streamed to the GPU(s) from host memory (pinned if made with `pinned_empty`) and the
        rotation runs on the device (BASELINE config 5);
      * `stats`: a dict that receives streaming counters (batches, bytes_in, seconds).
    """
    if de:
        # calculate_de is broken upstream (unpacks 4 of SampleIter's 5-tuple, lmm/lmm.py:499 vs :434)
        raise NotImplementedError("de=True is broken in the reference (lmm/lmm.py:499) and is not provided")
    L = _lib.load()
    packed = isinstance(X, PackedBed)                         # extension: a PLINK .bed image instead of the float matrix
    if packed and not eigen:
        raise ValueError("a PackedBed holds raw genotypes: it cannot be used with eigen=False (pre-rotated inputs)")
    if eigenpairs is not None and not eigen:
        raise ValueError("eigenpairs supplies U for the rotation: it goes with eigen=True")
    Y, W = np.asarray(Y), np.asarray(W)
    K = np.asarray(K) if K is not None else None
    if not packed:
        X = np.asarray(X)
    nproc = min(int(nproc), X.shape[1])                      # lmm.py:113
    if Y.dtype != np.float32:
        Y = Y.astype(np.float32).reshape(-1, 1)              # lmm.py:115-116
    if W.dtype != np.float32:
        W = W.astype(np.float32)                             # lmm.py:118-119
    # int8/uint8/float64 X stays as it is up to the device when it is rotated there: the kernels convert each element to
    # float32 as they read it, which is what the cast below does (round to nearest) — no host copy of the matrix
    x8 = (not packed) and eigen and X.dtype in (np.int8, np.uint8, np.float64)
    if not packed and not x8 and X.dtype != np.float32:
        X = X.astype(np.float32)                             # lmm.py:121-122
    if eigenpairs is None:
        if Z is not None:
            K = _zkzt(L, np.asarray(Z), K)                   # lmm.py:124-125, on GPU 0 (two fp64-MFMA products, one rounding to float32)
        k64 = eigen and K.dtype == np.float64                # rounded to float32 on the device instead (same values, no host copy)
        if K.dtype != np.float32 and not k64:
            K = K.astype(np.float32)                         # lmm.py:127-128
    # A Fortran-ordered float32 X is the SNP-major image (p rows of n samples): its batches are contiguous byte ranges that go to the
    # device as they are (and are turned sample-major there when they have to be rotated) — no transposing host copy of the matrix
    snp_major = (not packed) and X.ndim == 2 and X.dtype == np.float32 and X.flags.f_contiguous and not X.flags.c_contiguous
    if not packed and not X.flags.c_contiguous and not snp_major:
        X = np.ascontiguousarray(X)
    n, p = X.shape
    c = W.shape[1]
    if Y.shape[0] != n or W.shape[0] != n:
        raise ValueError(f"shape mismatch: Y {Y.shape}, X {X.shape}, W {W.shape}")
    ngpu = _lib.device_count()
    if ngpu < 1:
        raise _lib.PgError("no MI355X visible: pygemma_amd has no CPU path")
    blocks = list(SampleIter(p, max(1, min(nproc, ngpu))))   # one contiguous block per GPU, none empty
    ndev = len(blocks)

    t0 = time.time()
    ectx = dU0 = None  # the eigensolver's context on GPU 0 and U resident there (reused by GPU 0's SNP loop)
    xpin = pre = None
    pin_thread, pin_box = None, {}    # page-locking of a pageable X beside the eigensolver
    comms = None
    # ONE try/finally owns every resource made from here on (prefetch contexts and threads, the page-lock of X, the communicators,
    # the eigensolver's context): whatever raises in between — a bad K, os.makedirs, the manifest, a worker — they are released
    # (every close() below is idempotent, so the earlier explicit ones stay harmless)
    try:
        if eigen and eigenpairs is None and not packed and not checkpoint and X.flags.c_contiguous and _PREFETCH_MAX > 0:
            # the eigensolver leaves PCIe idle for ~0.6 s at n = 10 000 (52 s at 50 000): X starts moving now
            try:
                if _lib.is_pinned(X):
                    pre = _Prefetch(L, X, blocks, n, X.dtype.itemsize, verbose)
            except _lib.PgError as ex:
                _log(verbose, f"prefetch not started ({ex})")
        if eigen and not packed and X.size and not _lib.is_pinned(X) and (X.flags.c_contiguous or X.flags.f_contiguous) and X.nbytes >= (256 << 20):
            # Page-locking a pageable X in place costs ~0.06 s/GB the first time a range is registered (0.23 s for 4 GB: the kernel pins a million
            # pages; 2 ms on a repeat) — as long as the eigensolver runs anyway, it happens beside it on a thread of its own (r4: lmm.pygemma
            # from a pageable 4 GB X 0.77 -> ~0.5 s)
            def _pin_bg():
                try:
                    pin_box["h"] = _lib.pin(X)
                except _lib.PgError as ex:
                    pin_box["err"] = ex
            pin_thread = threading.Thread(target=_pin_bg, daemon=True)
            pin_thread.start()
        if eigen:
            ectx = _lib.Context(0)
            try:
                if eigenpairs is not None:
                    ev, Uh = eigenpairs
                    Uh = np.asarray(Uh)
                    if Uh.shape != (n, n):
                        raise ValueError(f"eigenpairs: U must be ({n},{n}), got {Uh.shape}")
                    if Uh.dtype != np.float32 or not Uh.flags.c_contiguous:
                        Uh = np.ascontiguousarray(Uh, np.float32)                       # lmm.py:154
                    eigenVals = np.maximum(0.0, np.asarray(ev)).astype(np.float32).reshape(-1)   # lmm.py:157-160
                    if eigenVals.shape[0] != n:
                        raise ValueError(f"eigenpairs: {n} eigenvalues expected, got {eigenVals.shape}")
                    dU0 = ectx.alloc(n * n * 4)
                    if _lib.is_pinned(Uh):     # one DMA straight out of the caller's pinned array
                        _lib.check(L.pg_memcpy_h2d_async(ectx.handle, dU0.ptr, Uh.ctypes.data, Uh.nbytes), "pg_memcpy_h2d_async")
                        ectx.sync()
                    else:                      # pageable: row panels through a pinned double buffer, copy threads ahead of the DMA
                        rows = max(1, min(n, (256 << 20) // (n * 4)))
                        stg = _Staging(ectx, L, rows * n * 4, rows * n * 4)      # .inp / .out used as the two halves
                        try:
                            halves, k = (stg.inp, stg.out), 0
                            evs = [C.c_void_p(), C.c_void_p()]
                            for e_ in evs:
                                _lib.check(L.pg_event_create(ectx.handle, C.byref(e_)), "pg_event_create")
                            used = [False, False]
                            for r0 in range(0, n, rows):
                                r1 = min(n, r0 + rows)
                                if used[k]:
                                    _lib.check(L.pg_event_sync(ectx.handle, evs[k]), "pg_event_sync")
                                _lib.check(L.pg_stage_rows(halves[k], n * 4, Uh.ctypes.data + r0 * n * 4, n * 4, n * 4, r1 - r0, _STAGE_THREADS),
                                           "pg_stage_rows")
                                _lib.check(L.pg_memcpy_h2d_async(ectx.handle, dU0.ptr + r0 * n * 4, halves[k], (r1 - r0) * n * 4), "pg_memcpy_h2d_async")
                                _lib.check(L.pg_event_record(ectx.handle, evs[k]), "pg_event_record")
                                used[k] = True
                                k ^= 1
                            ectx.sync()
                            for e_ in evs:
                                L.pg_event_destroy(ectx.handle, e_)
                        finally:
                            stg.close()
                    _log(verbose, f"Eigenvectors uploaded ({n * n * 4 / 1e9:.2f} GB) - {time.time() - t0:.3f} s")
                else:
                    if K.shape != (n, n):
                        raise ValueError(f"K must be ({n},{n}) when eigen=True, got {K.shape}")
                    if k64:
                        dK64 = ectx.to_device(np.ascontiguousarray(K))
                        dK = ectx.alloc(n * n * 4)
                        _lib.check(L.pg_cast_f64_f32_dev(ectx.handle, n, n, dK64.ptr, n, dK.ptr, n), "pg_cast_f64_f32_dev")   # lmm.py:127-128
                        ectx.sync()
                        dK64.free()
                    else:
                        dK = ectx.to_device(K)
                    dev, dU0 = ectx.alloc(n * 4), ectx.alloc(n * n * 4)
                    _lib.check(L.pg_syevd_dev(ectx.handle, n, dK.ptr, dev.ptr, dU0.ptr, None, None), "pg_syevd_dev")
                    dK.free()
                    eigenVals = dev.download((n,), np.float32)       # ascending, clamped >= 0, float32 (lmm.py:152-160)
                    _log(verbose, f"Eigendecomposition computed - {time.time() - t0:.3f} s")
                assert (eigenVals >= 0).all()                    # lmm.py:162
                t1 = time.time()
                YW = _rotate_small(ectx, L, n, dU0, np.concatenate([Y.reshape(n, -1)[:, :1], W], axis=1))
                Yr, Wr = YW[:, :1], np.ascontiguousarray(YW[:, 1:])
                _log(verbose, f"Left multiplied Y, W by U.T - {time.time() - t1:.3f} s")
            except BaseException:
                if pre is not None:
                    pre.close()
                if xpin is not None:
                    xpin.close()
                ectx.close()
                raise
        else:
            eigenVals = np.maximum(0.0, K).astype(np.float32).reshape(-1)   # lmm.py:166-167
            if eigenVals.shape[0] != n:
                raise ValueError(f"with eigen=False K must hold the {n} eigenvalues, got {K.shape}")
            Yr, Wr = Y.reshape(n, -1)[:, :1], np.ascontiguousarray(W)

        if not disable_checks:
            # lmm.py:253-256 (the reference tests the rotated arrays; a NaN anywhere in a raw column makes that
            # whole rotated column NaN, so testing the inputs raises in exactly the same cases)
            if (not packed and X.dtype.kind == 'f' and np.isnan(X).any()) or np.isnan(Yr).any() or np.isnan(Wr).any():
                if pre is not None:
                    pre.close()
                if xpin is not None:
                    xpin.close()
                if ectx is not None:
                    ectx.close()
                raise ValueError("NaNs present in data")

        _log(verbose, f"Running {p} SNPs with {n} individuals on {ndev} GPU(s)...")
        out = {"beta": np.empty(p, np.float32), "se_beta": np.empty(p, np.float32), "tau": np.empty(p, np.float32),
               "lambda": np.empty(p, np.float64), "F_wald": np.empty(p, np.float64), "p_wald": np.empty(p, np.float64)}
        if lrt:
            for col in _LRT_COLS:
                out[col] = np.empty(p, np.float64)
        errs, threads = [], []
        if stats is not None:
            stats.update({"batches": 0, "bytes_in": 0, "batch_s": 0.0, "gpus": ndev})
        yr1 = np.ascontiguousarray(Yr.reshape(-1), np.float32)
        if checkpoint:
            os.makedirs(checkpoint, exist_ok=True)
            # identity of the run: shapes, options, batch geometry and the SNP-independent inputs themselves (rotated y, W and the
            # eigenvalues, byte for byte); the genotypes are fingerprinted per part (_block_fingerprint)
            key = {"n": int(n), "p": int(p), "c": int(c), "grid": bool(grid), "eigen": bool(eigen), "ndev": int(ndev), "lrt": bool(lrt),
                   "batch_snps": int(_BATCH_SNPS), "batch_bytes": int(_BATCH_BYTES), "batch_min": int(_BATCH_MIN), "batch_count": int(_BATCH_COUNT),
                   "y_crc": _crc(yr1), "w_crc": _crc(Wr), "d_crc": _crc(eigenVals)}
            mf = os.path.join(checkpoint, "manifest.json")
            if os.path.exists(mf):
                with open(mf) as f:
                    if json.load(f) != key:
                        if ectx is not None:
                            ectx.close()
                        raise ValueError(f"checkpoint directory {checkpoint} belongs to a different run (manifest mismatch)")
            else:
                with open(mf, "w") as f:
                    json.dump(key, f)
        t2 = time.time()
        comms = None
        try:
            if pre is not None:
                pre.join()
                if stats is not None:
                    stats["prefetched_bytes"] = int(pre.bytes)
            if pin_thread is not None:
                pin_thread.join()
                pin_thread = None
                xpin = pin_box.pop("h", None)
                if "err" in pin_box:
                    _log(verbose, f"X could not be page-locked in place ({pin_box['err']}); staging through pinned buffers")
            elif not packed and not _lib.is_pinned(X):
                # page-lock the caller's X in place for the duration of the scan (hipHostRegister: 2 ms/GB for a range the kernel has pinned
                # before, ~60 ms/GB the first time: tools/probe_pageable.py) so that every batch is one 2-D DMA straight out of it; if the
                # range cannot be registered (e.g. a read-only file mapping) the workers fall back to copy threads + a pinned staging buffer
                try:
                    xpin = _lib.pin(X)
                except _lib.PgError as ex:
                    _log(verbose, f"X could not be page-locked in place ({ex}); staging through pinned buffers")
            if stats is not None:
                stats["pinned_input"] = bool((not packed) and _lib.is_pinned(X))
                stats["registered_in_place"] = xpin is not None
            dUs = [dU0] + [None] * (ndev - 1)
            if ndev > 1 and eigen:
                comms = _make_comms(L, ndev)      # RCCL communicator over the GPUs of this process: U goes GPU 0 -> all over xGMI
                for g in range(1, ndev):          # receive buffers made here: an allocation failure surfaces before any thread waits in the collective
                    dUs[g] = comms[g].ctx.alloc(n * n * 4)
            for dev_id, (a, b) in enumerate(blocks):
                th = threading.Thread(target=_run_block, args=(dev_id, a, b, n, c, eigenVals, Wr, yr1, X,
                                                               dUs[dev_id], comms[dev_id] if comms else None,
                                                               grid, eigen, lrt, out, errs, verbose, checkpoint, stats, pre))
                th.start()
                threads.append(th)
            for th in threads:
                th.join()
            if stats is not None:
                stats["blocks_s"] = time.time() - t2          # page-locking + every GPU's block; what follows is teardown (frees)
        finally:
            if pre is not None:
                pre.close()
            if xpin is not None:
                xpin.close()
            if comms:
                for cm in comms:
                    if cm.handle:
                        L.pg_comm_destroy(cm.handle)
                        cm.handle = None
                    cm.ctx.close()
            if ectx is not None:
                ectx.close()
    finally:
        if pre is not None:
            pre.close()
        if pin_thread is not None:        # an exception before the loop: the registration may still be under way
            pin_thread.join()
            if pin_box.get("h") is not None:
                pin_box.pop("h").close()
        if xpin is not None:
            xpin.close()
        if comms:
            for cm in comms:
                if cm.handle:
                    L.pg_comm_destroy(cm.handle)
                    cm.handle = None
                cm.ctx.close()
        if ectx is not None:
            ectx.close()
    if errs:
        raise errs[0]
    if stats is not None:
        stats["seconds"] = time.time() - t2
    _log(verbose, f"Finished testing {p} SNPs in {time.time() - t2:.3f} s")
    cols = list(_COLS) + (list(_LRT_COLS) if lrt else [])
    results_df = pd.DataFrame(out, columns=cols)                                                     # lmm.py:403
    if snps is not None:
        results_df["SNPs"] = snps                                                                    # lmm.py:408-409
    return results_df
