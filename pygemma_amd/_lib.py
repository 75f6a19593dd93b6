"""ctypes binding of libpygemma_hip.so (C ABI: include/pygemma_hip.h).

The MI355X path has NO CPU fallback: if the shared library is missing or no GPU is visible the
calls raise.  The checker (the CPU oracle) is test infrastructure and is never imported from here.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PYGEMMA_HIP_LIB") or os.path.join(_HERE, "lib", "libpygemma_hip.so")   # env: A/B builds
_lib = None

SYMBOLS = [
    "pg_last_error", "pg_version", "pg_device_count", "pg_ctx_create", "pg_ctx_create_on_stream",
    "pg_ctx_destroy", "pg_ctx_sync", "pg_ctx_device", "pg_mem_info", "pg_malloc", "pg_free", "pg_memcpy_h2d", "pg_memcpy_d2h",
    "pg_memset", "pg_memcpy2d_h2d", "pg_event_create", "pg_event_destroy", "pg_event_record", "pg_event_elapsed_ms",
    "pg_kinship_dev", "pg_geno_prep_bytes", "pg_geno_work_bytes", "pg_geno_prep_dev", "pg_rotate_geno_dev", "pg_assoc_dev", "pg_assoc", "pg_fdist_sf_dev", "pg_transpose_dev", "pg_rotate_dev", "pg_syevd_dev",
    "pg_precompute_mat_dev", "pg_newton_dev", "pg_reml_scalars_dev", "pg_ml_scalars_dev", "pg_rotate_bed_dev", "pg_rotate_geno_i8_dev", "pg_cast_i8_f32_dev", "pg_assoc_multi", "pg_rotate_geno_f64_dev", "pg_cast_f64_f32_dev",
    "pg_host_alloc", "pg_host_free", "pg_host_register", "pg_host_unregister", "pg_memcpy_h2d_async", "pg_memcpy_d2h_async",
    "pg_memcpy_d2d_async", "pg_memcpy2d_h2d_async", "pg_stage_rows", "pg_event_sync", "pg_stream_wait_event",
    "pg_comm_unique_id", "pg_comm_init_rank", "pg_comm_init_all", "pg_comm_destroy", "pg_comm_size", "pg_comm_rank",
    "pg_comm_broadcast_dev", "pg_comm_allgather_dev", "pg_comm_allreduce_f64_dev", "pg_comm_barrier", "pg_comm_group_start",
    "pg_comm_group_end", "pgx_dgemm_dev", "pgx_sytrd_dev", "pgx_stedc_dev", "pgx_sb2_stage1_dev", "pgx_sb2_stage2_dev", "pgx_sb2_set_debug", "pg_kinship_geno_dev", "pg_assoc_lrt_dev", "pg_rotate_auto_dev", "pg_assoc_set_eval_trace", "pg_assoc_warm", "pg_rotate_auto_i8_dev",
    "pg_zkzt_dev", "pgx_dgemm_ex_dev", "pgx_ring_stamps",
]


class PgError(RuntimeError):
    pass


def load():
    """Load the HIP library (build it first with __graft_entry__.build() / make -C pygemma_amd/csrc)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PgError(f"{LIB_PATH} not found: build the HIP extension (python -c 'import __graft_entry__ as g; g.build()'). "
                      "pygemma_amd has no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    vp, i64, i32, sz = C.c_void_p, C.c_int64, C.c_int, C.c_size_t
    L.pg_last_error.restype = C.c_char_p
    L.pg_version.restype = C.c_char_p
    L.pg_device_count.restype = i32
    L.pg_ctx_create.argtypes = [i32, C.POINTER(vp)]
    L.pg_ctx_create_on_stream.argtypes = [i32, vp, C.POINTER(vp)]
    L.pg_ctx_destroy.argtypes = [vp]
    L.pg_ctx_destroy.restype = None
    L.pg_ctx_sync.argtypes = [vp]
    L.pg_ctx_device.argtypes = [vp]
    L.pg_mem_info.argtypes = [vp, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    L.pg_malloc.argtypes = [vp, sz, C.POINTER(vp)]
    L.pg_free.argtypes = [vp, vp]
    L.pg_memcpy_h2d.argtypes = [vp, vp, vp, sz]
    L.pg_memcpy_d2h.argtypes = [vp, vp, vp, sz]
    L.pg_memset.argtypes = [vp, vp, i32, sz]
    L.pg_assoc_dev.argtypes = [vp, i64, i32, i64, vp, vp, vp, vp, i64, i32, vp, vp, vp, vp, vp, vp, vp]
    L.pg_assoc.argtypes = [vp, i64, i32, i64, vp, vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp]
    L.pg_assoc_lrt_dev.argtypes = [vp, i64, i32, i64, vp, vp, vp, vp, i64, i32] + [vp] * 10
    L.pg_assoc_lrt_dev.restype = i32
    L.pg_fdist_sf_dev.argtypes = [vp, i64, vp, C.c_double, vp]
    L.pg_assoc_multi.argtypes = [i32, i64, i32, i64, vp, vp, vp, vp, i32, vp, vp, vp, vp, vp, vp]
    L.pg_assoc_multi.restype = i32
    L.pg_transpose_dev.argtypes = [vp, i64, i64, vp, i64, vp, i64]
    L.pg_rotate_dev.argtypes = [vp, i64, i64, vp, i64, vp, i64, vp, i64]
    L.pg_kinship_dev.argtypes = [vp, i64, i64, vp, i64, vp]
    L.pg_kinship_geno_dev.argtypes = [vp, i64, i64, vp, i64, i32, vp]
    L.pg_kinship_geno_dev.restype = i32
    L.pg_geno_prep_bytes.argtypes = [i64]
    L.pg_geno_prep_bytes.restype = sz
    L.pg_geno_work_bytes.argtypes = [i64, i64]
    L.pg_geno_work_bytes.restype = sz
    L.pg_geno_prep_dev.argtypes = [vp, i64, vp, i64, vp]
    L.pg_geno_prep_dev.restype = i32
    L.pg_rotate_geno_dev.argtypes = [vp, i64, i64, vp, vp, i64, vp, i64, vp, C.POINTER(i32)]
    L.pg_rotate_geno_dev.restype = i32
    L.pg_assoc_set_eval_trace.argtypes = [vp, vp]
    L.pg_assoc_set_eval_trace.restype = i32
    L.pg_assoc_warm.argtypes = [vp, i64, i32]
    L.pg_assoc_warm.restype = i32
    L.pg_rotate_auto_i8_dev.argtypes = [vp, i64, i64, vp, vp, i32, i64, vp, i64, vp, vp]
    L.pg_rotate_auto_i8_dev.restype = i32
    L.pg_rotate_auto_dev.argtypes = [vp, i64, i64, vp, i64, vp, vp, i64, vp, i64, vp, vp]
    L.pg_rotate_auto_dev.restype = i32
    L.pg_rotate_bed_dev.argtypes = [vp, i64, i64, vp, vp, i64, i32, vp, i64, vp]
    L.pg_rotate_bed_dev.restype = i32
    L.pg_rotate_geno_i8_dev.argtypes = [vp, i64, i64, vp, vp, i32, i64, vp, i64, vp, C.POINTER(i32)]
    L.pg_rotate_geno_i8_dev.restype = i32
    L.pg_cast_i8_f32_dev.argtypes = [vp, i64, i64, vp, i32, i64, vp, i64]
    L.pg_cast_i8_f32_dev.restype = i32
    L.pg_rotate_geno_f64_dev.argtypes = [vp, i64, i64, vp, vp, i64, vp, i64, vp, C.POINTER(i32)]
    L.pg_rotate_geno_f64_dev.restype = i32
    L.pg_cast_f64_f32_dev.argtypes = [vp, i64, i64, vp, i64, vp, i64]
    L.pg_cast_f64_f32_dev.restype = i32
    L.pg_memcpy2d_h2d.argtypes = [vp, vp, sz, vp, sz, sz, sz]
    L.pg_event_create.argtypes = [vp, C.POINTER(vp)]
    L.pg_event_destroy.argtypes = [vp, vp]
    L.pg_event_record.argtypes = [vp, vp]
    L.pg_event_elapsed_ms.argtypes = [vp, vp, vp, C.POINTER(C.c_float)]
    L.pg_syevd_dev.argtypes = [vp, i64, vp, vp, vp, vp, vp]
    f32 = C.c_float
    L.pg_precompute_mat_dev.argtypes = [vp, i64, i32, f32, vp, vp, vp, i32, vp, vp, vp, vp, vp]
    L.pg_newton_dev.argtypes = [vp, i64, i32, f32, f32, f32, vp, vp, vp, vp]
    L.pg_reml_scalars_dev.argtypes = [vp, i64, i32, vp, vp]
    L.pg_ml_scalars_dev.argtypes = [vp, i64, vp, vp]
    L.pg_ml_scalars_dev.restype = i32
    L.pg_host_alloc.argtypes = [vp, sz, C.POINTER(vp)]
    L.pg_host_free.argtypes = [vp, vp]
    L.pg_host_register.argtypes = [vp, vp, sz]
    L.pg_host_unregister.argtypes = [vp, vp]
    L.pg_memcpy_h2d_async.argtypes = [vp, vp, vp, sz]
    L.pg_memcpy_d2h_async.argtypes = [vp, vp, vp, sz]
    L.pg_memcpy_d2d_async.argtypes = [vp, vp, vp, sz]
    L.pg_memcpy2d_h2d_async.argtypes = [vp, vp, sz, vp, sz, sz, sz]
    L.pg_stage_rows.argtypes = [vp, sz, vp, sz, sz, sz, i32]
    L.pg_event_sync.argtypes = [vp, vp]
    L.pg_stream_wait_event.argtypes = [vp, vp]
    L.pg_comm_unique_id.argtypes = [vp]
    L.pg_comm_init_rank.argtypes = [vp, i32, i32, vp, C.POINTER(vp)]
    L.pg_comm_init_all.argtypes = [i32, C.POINTER(vp), C.POINTER(vp)]
    L.pg_comm_destroy.argtypes = [vp]
    L.pg_comm_size.argtypes = [vp]
    L.pg_comm_rank.argtypes = [vp]
    L.pg_comm_broadcast_dev.argtypes = [vp, vp, sz, i32]
    L.pg_comm_allgather_dev.argtypes = [vp, vp, vp, sz]
    L.pg_comm_allreduce_f64_dev.argtypes = [vp, vp, sz, i32]
    L.pg_comm_barrier.argtypes = [vp]
    L.pgx_dgemm_dev.argtypes = [vp, i32, i64, i64, i64, C.c_double, vp, i64, vp, i64, C.c_double, vp, i64]
    L.pgx_dgemm_ex_dev.argtypes = [vp, i32, i32, i64, i64, i64, C.c_double, vp, i64, vp, i64, C.c_double, vp, i64]
    L.pgx_dgemm_ex_dev.restype = i32
    L.pg_zkzt_dev.argtypes = [vp, i64, i64, vp, i32, i64, vp, i32, i64, vp, i64]
    L.pg_zkzt_dev.restype = i32
    L.pgx_sytrd_dev.argtypes = [vp, i64, vp, vp, vp, vp, vp]
    L.pgx_stedc_dev.argtypes = [vp, i64, vp, vp, vp, vp]
    L.pgx_sb2_stage1_dev.argtypes = [vp, i64, vp, vp, vp, vp]
    L.pgx_sb2_stage2_dev.argtypes = [vp, i64, vp, vp, vp, vp, vp]
    L.pgx_sb2_set_debug.argtypes = [vp]
    for name in ("pg_host_alloc", "pg_host_free", "pg_host_register", "pg_host_unregister", "pg_memcpy_h2d_async", "pg_memcpy_d2h_async",
                 "pg_memcpy_d2d_async", "pg_memcpy2d_h2d_async", "pg_stage_rows", "pg_event_sync", "pg_stream_wait_event", "pg_comm_unique_id",
                 "pg_comm_init_rank", "pg_comm_init_all", "pg_comm_destroy", "pg_comm_size", "pg_comm_rank", "pg_comm_broadcast_dev",
                 "pg_comm_allgather_dev", "pg_comm_allreduce_f64_dev", "pg_comm_barrier", "pg_comm_group_start", "pg_comm_group_end",
                 "pgx_dgemm_dev", "pgx_sytrd_dev", "pgx_stedc_dev", "pgx_sb2_stage1_dev", "pgx_sb2_stage2_dev", "pgx_sb2_set_debug"):
        getattr(L, name).restype = i32
    for name in ("pg_ctx_create", "pg_ctx_create_on_stream", "pg_ctx_sync", "pg_ctx_device", "pg_malloc", "pg_free",
                 "pg_memcpy_h2d", "pg_memcpy_d2h", "pg_memset", "pg_assoc_dev", "pg_assoc", "pg_fdist_sf_dev",
                 "pg_transpose_dev", "pg_rotate_dev", "pg_syevd_dev", "pg_memcpy2d_h2d", "pg_event_create", "pg_event_destroy",
                 "pg_event_record", "pg_event_elapsed_ms", "pg_kinship_dev", "pg_precompute_mat_dev", "pg_newton_dev",
                 "pg_reml_scalars_dev"):
        getattr(L, name).restype = i32
    _lib = L
    return L


def check(rc, what=""):
    if rc != 0:
        raise PgError(f"{what} failed (code {rc}): {load().pg_last_error().decode()}")


class DeviceBuffer:
    """hipMalloc'd buffer owned by a Context (freed with it or by .free())."""

    def __init__(self, ctx, nbytes):
        self.ctx, self.nbytes = ctx, int(nbytes)
        p = C.c_void_p()
        check(load().pg_malloc(ctx.handle, self.nbytes, C.byref(p)), "pg_malloc")
        self.ptr = p.value

    def free(self):
        if self.ptr:
            load().pg_free(self.ctx.handle, self.ptr)
            self.ptr = None

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        check(load().pg_memcpy_h2d(self.ctx.handle, self.ptr, arr.ctypes.data, arr.nbytes), "pg_memcpy_h2d")
        return self

    def download(self, shape, dtype, offset=0):
        out = np.empty(shape, dtype)
        check(load().pg_memcpy_d2h(self.ctx.handle, out.ctypes.data, self.ptr + offset, out.nbytes), "pg_memcpy_d2h")
        return out


class Context:
    """One per (process, GPU): wraps pg_ctx (device id + stream + scratch)."""

    def __init__(self, device=0, stream=None):
        L = load()
        h = C.c_void_p()
        if stream is None:
            check(L.pg_ctx_create(int(device), C.byref(h)), "pg_ctx_create")
        else:
            check(L.pg_ctx_create_on_stream(int(device), C.c_void_p(stream), C.byref(h)), "pg_ctx_create_on_stream")
        self.handle = h
        self.device = device
        self._bufs = []

    def alloc(self, nbytes):
        b = DeviceBuffer(self, nbytes)
        self._bufs.append(b)
        return b

    def to_device(self, arr):
        arr = np.ascontiguousarray(arr)
        return self.alloc(max(arr.nbytes, 4)).upload(arr)

    def sync(self):
        check(load().pg_ctx_sync(self.handle), "pg_ctx_sync")

    def mem_info(self):
        """(free, total) bytes of this context's device."""
        f, t = C.c_size_t(), C.c_size_t()
        check(load().pg_mem_info(self.handle, C.byref(f), C.byref(t)), "pg_mem_info")
        return int(f.value), int(t.value)

    def close(self):
        if self.handle:
            for b in self._bufs:
                b.free()
            self._bufs = []
            load().pg_ctx_destroy(self.handle)
            self.handle = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def device_count():
    return load().pg_device_count()


# ---- pinned host memory (S1: streaming SNP batches / eigenvectors from the host) ---------------------------------------
_pinned = {}     # base address -> nbytes of every live pinned range this process made (pinned_empty / pin)


class _PinnedBlock:
    """Owner of one hipHostMalloc'd range; freed when the last NumPy view of it goes away."""

    def __init__(self, nbytes, device=0):
        import weakref
        self.ctx = Context(device)
        p = C.c_void_p()
        check(load().pg_host_alloc(self.ctx.handle, max(int(nbytes), 1), C.byref(p)), "pg_host_alloc")
        self.ptr, self.nbytes = p.value, int(nbytes)
        _pinned[self.ptr] = self.nbytes
        self._fin = weakref.finalize(self, _PinnedBlock._release, self.ctx, self.ptr)

    @staticmethod
    def _release(ctx, ptr):
        _pinned.pop(ptr, None)
        try:
            load().pg_host_free(ctx.handle, ptr)
            ctx.close()
        except Exception:
            pass


def pinned_empty(shape, dtype=np.float32, device=0):
    """np.empty in page-locked host memory (hipHostMalloc, portable): the array to np.fromfile()/copy a genotype or
    eigenvector matrix into so that lmm.pygemma streams it by DMA without a staging copy."""
    dtype = np.dtype(dtype)
    nbytes = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
    blk = _PinnedBlock(nbytes, device)
    buf = (C.c_char * max(nbytes, 1)).from_address(blk.ptr)
    arr = np.frombuffer(buf, dtype=dtype, count=nbytes // dtype.itemsize).reshape(shape)
    arr.flags.writeable = True
    # arr.base -> memoryview -> buf: tie the block's life to the buffer object the array is based on
    buf._pg_block = blk
    return arr


def pin(arr, device=0):
    """Page-lock an existing contiguous (C- or Fortran-ordered) NumPy array in place (hipHostRegister).  Returns a handle whose
    .close() unpins; pinning costs ~5 ms/GB (tools/bench_h2d.py)."""
    assert arr.flags.c_contiguous or arr.flags.f_contiguous
    ctx = Context(device)
    check(load().pg_host_register(ctx.handle, arr.ctypes.data, arr.nbytes), "pg_host_register")
    _pinned[arr.ctypes.data] = arr.nbytes

    class _Pin:
        def close(self_inner):
            if _pinned.pop(arr.ctypes.data, None) is not None:
                load().pg_host_unregister(ctx.handle, arr.ctypes.data)
                ctx.close()
    return _Pin()


def is_pinned(arr):
    """True when the array's bytes lie inside a range pinned through this module."""
    try:
        from numpy.lib.array_utils import byte_bounds
    except ImportError:          # NumPy 1.x
        byte_bounds = np.byte_bounds
    lo, hi = byte_bounds(arr)
    for base, nb in list(_pinned.items()):
        if base <= lo and hi <= base + nb:
            return True
    return False
