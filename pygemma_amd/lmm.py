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
There is no CPU fallback: without the HIP library or a GPU this raises.
"""
import ctypes as C
import json
import os
import threading
import time

import numpy as np
import pandas as pd

from . import _lib
from .bed import PackedBed
from .model import *          # noqa: F401,F403  (precompute_mat, calc_lambda_restricted, newton, the *_overload scalars)
from . import model as _model

__all__ = ["pygemma", "SampleIter"] + _model.__all__

_BATCH_BYTES = 6 << 30   # device bytes for one SNP batch of one worker (raw block, rotated block, fp16 planes)
_BATCH_SNPS = 32768      # SNPs per batch at most: the unit of copy/compute overlap and of checkpointing


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


def _log(verbose, msg):
    if verbose > 0:
        print(f"[pygemma_amd] {msg}", flush=True)


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


def _part_path(ckpt, s, e):
    return os.path.join(ckpt, f"part_{s:012d}_{e:012d}.npz")


def _run_block(device, a, b, n, c, d, Wr, yr, X, U_host, grid, eigen, out, errs, verbose, ckpt=None):
    """One GPU: SNP columns [a,b) of X through (rotate | transpose) -> assoc, in batches.  Two host threads per GPU, each with
    its own stream and buffers, take batches from a shared list, so that the host->device copy of one batch overlaps the
    kernels of the other.  With `ckpt` every finished batch is written to disk (and batches found there are not redone)."""
    try:
        L = _lib.load()
        ctx0 = _lib.Context(device)
        try:
            ldx = (n + 63) // 64 * 64
            dd, dW, dy = ctx0.to_device(d), ctx0.to_device(Wr), ctx0.to_device(yr)
            # U: already resident (GPU 0 keeps the eigensolver's output) or uploaded from the host copy
            dU = (U_host if isinstance(U_host, _lib.DeviceBuffer) else ctx0.to_device(U_host)) if eigen else None
            dprep = None
            if eigen:   # genotype fast path of the rotation (<= 3 equally spaced values per column), fp32 MFMA otherwise
                dprep = ctx0.alloc(L.pg_geno_prep_bytes(n))
                _lib.check(L.pg_geno_prep_dev(ctx0.handle, n, dU.ptr, n, dprep.ptr), "pg_geno_prep_dev")
            ctx0.sync()
            packed = isinstance(X, PackedBed)
            x8 = (not packed) and X.dtype in (np.int8, np.uint8)
            x64 = (not packed) and X.dtype == np.float64
            esz = 1 if x8 else (8 if x64 else 4)
            pb_max = max(256, int(_BATCH_BYTES // (12 * ldx)) // 256 * 256)   # raw block + rotated block + two fp16 planes
            pb_max = min(pb_max, _BATCH_SNPS, b - a)
            ldX = (pb_max + 15) // 16 * 16
            bpr = (n + 3) // 4
            p = X.shape[1]
            todo = []
            for s in range(a, b, pb_max):
                e = min(s + pb_max, b)
                if ckpt and os.path.exists(_part_path(ckpt, s, e)):
                    with np.load(_part_path(ckpt, s, e)) as z:
                        for col in _COLS:
                            out[col][s:e] = z[col]
                    _log(verbose, f"GPU {device}: SNPs [{s},{e}) restored from {ckpt}")
                else:
                    todo.append((s, e))
            lock = threading.Lock()

            def worker():
                try:
                    ctx = _lib.Context(device)
                    try:
                        dX = ctx.alloc(pb_max * bpr if packed else n * ldX * esz)
                        dXf = None       # float32 image of an 8-bit block, only if one does not qualify for the genotype path
                        dXr = ctx.alloc(pb_max * ldx * 4)
                        dout, dF = ctx.alloc(pb_max * 16), ctx.alloc(pb_max * 16)
                        dwork = ctx.alloc(L.pg_geno_work_bytes(n, pb_max)) if eigen else None
                        while True:
                            with lock:
                                if not todo or errs:
                                    return
                                s, e = todo.pop(0)
                            pb = e - s
                            if packed:   # SNP records [s, e) of the .bed image: contiguous bytes; decode + impute + rotate on the device
                                rec = np.ascontiguousarray(X.data[s:e])
                                _lib.check(L.pg_memcpy_h2d(ctx.handle, dX.ptr, rec.ctypes.data, rec.nbytes), "pg_memcpy_h2d")
                                _lib.check(L.pg_rotate_bed_dev(ctx.handle, n, pb, dprep.ptr, dX.ptr, bpr, int(X.count_A1), dXr.ptr, ldx,
                                                               dwork.ptr), "pg_rotate_bed_dev")
                            else:
                                _lib.check(L.pg_memcpy2d_h2d(ctx.handle, dX.ptr, ldX * esz, X.ctypes.data + esz * s, p * esz, pb * esz, n),
                                           "pg_memcpy2d_h2d")
                                if x64:
                                    is_geno = C.c_int(0)
                                    _lib.check(L.pg_rotate_geno_f64_dev(ctx.handle, n, pb, dprep.ptr, dX.ptr, ldX, dXr.ptr, ldx, dwork.ptr,
                                                                        C.byref(is_geno)), "pg_rotate_geno_f64_dev")
                                    if not is_geno.value:
                                        dXf = dXf or ctx.alloc(n * ldX * 4)
                                        _lib.check(L.pg_cast_f64_f32_dev(ctx.handle, n, pb, dX.ptr, ldX, dXf.ptr, ldX), "pg_cast_f64_f32_dev")
                                        _lib.check(L.pg_rotate_dev(ctx.handle, n, pb, dU.ptr, n, dXf.ptr, ldX, dXr.ptr, ldx), "pg_rotate_dev")
                                elif x8:
                                    is_geno = C.c_int(0)
                                    _lib.check(L.pg_rotate_geno_i8_dev(ctx.handle, n, pb, dprep.ptr, dX.ptr, int(X.dtype == np.uint8), ldX,
                                                                       dXr.ptr, ldx, dwork.ptr, C.byref(is_geno)), "pg_rotate_geno_i8_dev")
                                    if not is_geno.value:
                                        dXf = dXf or ctx.alloc(n * ldX * 4)
                                        _lib.check(L.pg_cast_i8_f32_dev(ctx.handle, n, pb, dX.ptr, int(X.dtype == np.uint8), ldX, dXf.ptr, ldX),
                                                   "pg_cast_i8_f32_dev")
                                        _lib.check(L.pg_rotate_dev(ctx.handle, n, pb, dU.ptr, n, dXf.ptr, ldX, dXr.ptr, ldx), "pg_rotate_dev")
                                elif eigen:
                                    is_geno = C.c_int(0)
                                    _lib.check(L.pg_rotate_geno_dev(ctx.handle, n, pb, dprep.ptr, dX.ptr, ldX, dXr.ptr, ldx, dwork.ptr,
                                                                    C.byref(is_geno)), "pg_rotate_geno_dev")
                                    if not is_geno.value:
                                        _lib.check(L.pg_rotate_dev(ctx.handle, n, pb, dU.ptr, n, dX.ptr, ldX, dXr.ptr, ldx), "pg_rotate_dev")
                                else:
                                    _lib.check(L.pg_transpose_dev(ctx.handle, n, pb, dX.ptr, ldX, dXr.ptr, ldx), "pg_transpose_dev")
                            _lib.check(L.pg_assoc_dev(ctx.handle, n, c, pb, dd.ptr, dW.ptr, dy.ptr, dXr.ptr, ldx, int(grid),
                                                      dout.ptr, dout.ptr + 4 * pb, dout.ptr + 8 * pb, dout.ptr + 12 * pb,
                                                      dF.ptr, dF.ptr + 8 * pb, None), "pg_assoc_dev")
                            ctx.sync()
                            res = dout.download((4, pb), np.float32)
                            FP = dF.download((2, pb), np.float64)
                            out["beta"][s:e], out["se_beta"][s:e], out["tau"][s:e] = res[0], res[1], res[2]
                            out["lambda"][s:e] = res[3].astype(np.float64)
                            out["F_wald"][s:e], out["p_wald"][s:e] = FP[0], FP[1]
                            if ckpt:
                                tmp = _part_path(ckpt, s, e) + ".tmp.npz"
                                np.savez(tmp, **{col: out[col][s:e] for col in _COLS})
                                os.replace(tmp, _part_path(ckpt, s, e))
                            _log(verbose, f"GPU {device}: SNPs [{s},{e}) done")
                    finally:
                        ctx.close()
                except Exception as ex:  # surfaced by the caller; never swallowed
                    errs.append(ex)

            workers = [threading.Thread(target=worker) for _ in range(min(2, len(todo)))]
            for th in workers:
                th.start()
            for th in workers:
                th.join()
        finally:
            ctx0.close()
    except Exception as ex:  # surfaced by the caller; never swallowed
        errs.append(ex)


def pygemma(Y, X, W, K, Z=None, snps=None, verbose=0, disable_checks=True, de=False, grid=False, eigen=True, nproc=1,
            checkpoint=None):
    """Per-SNP LMM association scan (GEMMA-style REML + Wald test) — signature of lmm/lmm.py:87.

    Y (n,1) phenotype; X (n,p) genotypes; W (n,c) covariates; K (n,n) relatedness matrix — or, with
    eigen=False, the (n,) vector of its eigenvalues with X, Y, W already rotated (lmm.py:164-167).
    Returns a pandas.DataFrame with columns beta, se_beta, tau, lambda, F_wald, p_wald[, SNPs].
    Beyond the reference: X may be a `pygemma_amd.bed.PackedBed` (PLINK .bed kept packed; missing calls mean-imputed on
    the device, as the reference's callers do on the host before calling); `checkpoint` names a directory that receives
    every finished SNP batch, and a rerun with the same inputs picks up from what is there (the reference has no restart).
    """
    if de:
        # calculate_de is broken upstream (unpacks 4 of SampleIter's 5-tuple, lmm/lmm.py:499 vs :434)
        raise NotImplementedError("de=True is broken in the reference (lmm/lmm.py:499) and is not provided")
    L = _lib.load()
    packed = isinstance(X, PackedBed)                         # extension: a PLINK .bed image instead of the float matrix
    if packed and not eigen:
        raise ValueError("a PackedBed holds raw genotypes: it cannot be used with eigen=False (pre-rotated inputs)")
    Y, W, K = np.asarray(Y), np.asarray(W), np.asarray(K)
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
    if Z is not None:
        K = np.asarray(Z) @ K @ np.asarray(Z).T              # lmm.py:124-125
    k64 = eigen and K.dtype == np.float64                    # rounded to float32 on the device instead (same values, no host copy)
    if K.dtype != np.float32 and not k64:
        K = K.astype(np.float32)                             # lmm.py:127-128
    if not packed:
        X = np.ascontiguousarray(X)
    n, p = X.shape
    c = W.shape[1]
    if Y.shape[0] != n or W.shape[0] != n:
        raise ValueError(f"shape mismatch: Y {Y.shape}, X {X.shape}, W {W.shape}")
    ngpu = _lib.device_count()
    if ngpu < 1:
        raise _lib.PgError("no MI355X visible: pygemma_amd has no CPU path")
    ndev = max(1, min(nproc, ngpu))

    t0 = time.time()
    U_host = None      # host copy of U: only needed to seed GPUs other than 0
    ectx = dU0 = None  # the eigensolver's context on GPU 0 and U resident there (reused by GPU 0's SNP loop)
    if eigen:
        if K.shape != (n, n):
            raise ValueError(f"K must be ({n},{n}) when eigen=True, got {K.shape}")
        ectx = _lib.Context(0)
        try:
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
            assert (eigenVals >= 0).all()                    # lmm.py:162
            _log(verbose, f"Eigendecomposition computed - {time.time() - t0:.3f} s")
            t1 = time.time()
            YW = _rotate_small(ectx, L, n, dU0, np.concatenate([Y.reshape(n, -1)[:, :1], W], axis=1))
            Yr, Wr = YW[:, :1], np.ascontiguousarray(YW[:, 1:])
            if ndev > 1:
                U_host = dU0.download((n, n), np.float32)
            _log(verbose, f"Left multiplied Y, W by U.T - {time.time() - t1:.3f} s")
        except BaseException:
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
            raise ValueError("NaNs present in data")

    _log(verbose, f"Running {p} SNPs with {n} individuals on {ndev} GPU(s)...")
    out = {"beta": np.empty(p, np.float32), "se_beta": np.empty(p, np.float32), "tau": np.empty(p, np.float32),
           "lambda": np.empty(p, np.float64), "F_wald": np.empty(p, np.float64), "p_wald": np.empty(p, np.float64)}
    errs, threads = [], []
    if checkpoint:
        os.makedirs(checkpoint, exist_ok=True)
        key = {"n": int(n), "p": int(p), "c": int(c), "grid": bool(grid), "eigen": bool(eigen), "ndev": int(ndev),
               "batch_snps": int(_BATCH_SNPS), "batch_bytes": int(_BATCH_BYTES),
               "y_sum": float(np.asarray(Yr, np.float64).sum()), "d_sum": float(np.asarray(eigenVals, np.float64).sum())}
        mf = os.path.join(checkpoint, "manifest.json")
        if os.path.exists(mf):
            with open(mf) as f:
                if json.load(f) != key:
                    raise ValueError(f"checkpoint directory {checkpoint} belongs to a different run (manifest mismatch)")
        else:
            with open(mf, "w") as f:
                json.dump(key, f)
    t2 = time.time()
    yr1 = np.ascontiguousarray(Yr.reshape(-1), np.float32)
    try:
        for dev_id, (a, b) in enumerate(SampleIter(p, ndev)):
            th = threading.Thread(target=_run_block, args=(dev_id, a, b, n, c, eigenVals, Wr, yr1, X,
                                                           dU0 if (dev_id == 0 and dU0 is not None) else U_host, grid, eigen,
                                                           out, errs, verbose, checkpoint))
            th.start()
            threads.append(th)
        for th in threads:
            th.join()
    finally:
        if ectx is not None:
            ectx.close()
    if errs:
        raise errs[0]
    _log(verbose, f"Finished testing {p} SNPs in {time.time() - t2:.3f} s")
    results_df = pd.DataFrame(out, columns=["beta", "se_beta", "tau", "lambda", "F_wald", "p_wald"])   # lmm.py:403
    if snps is not None:
        results_df["SNPs"] = snps                                                                    # lmm.py:408-409
    return results_df
