"""One process per GPU: SNP-range sharding identical to the reference's SampleIter (lmm/lmm.py:427-434), the RCCL
communicator of the C ABI (pg_comm_*: librccl over xGMI, no PyTorch) with a standard-library rendezvous, and the gather of
the 32-byte per-SNP result rows in rank order = SNP order (the reference's ordered concatenation, lmm/lmm.py:393,401).

    comm = dist.init()                       # RANK / WORLD_SIZE / LOCAL_RANK / MASTER_PORT from the launcher's environment
    comm.broadcast(dU.ptr, nbytes, root=0)   # eigenvectors from the rank that ran the eigensolver
    comm.allgather(res.ptr, all.ptr, nbytes) # padded row blocks
    comm.close()

Any launcher that sets those variables works (torch.distributed.run, srun, mpirun wrappers, bench.py's own spawner).
The torch.distributed helpers at the bottom serve the CPU (gloo) tests of the sharding + row packing logic only.
"""
import ctypes as C
import os
import time

import numpy as np


def shard_range(p, rank, world):
    """Columns [a, b) of rank `rank`: contiguous blocks of ceil(p/world), like SampleIter."""
    cols = int(np.ceil(p / world))
    a = min(rank * cols, p)
    return a, min(a + cols, p)


# ---- rendezvous: the 128-byte RCCL id travels rank 0 -> others through a file on the node --------------------------------
def _rdzv_path():
    d = os.environ.get("PYGEMMA_RDZV_DIR") or ("/dev/shm" if os.path.isdir("/dev/shm") else "/tmp")
    key = "_".join(str(x) for x in (os.environ.get("MASTER_PORT", "0"), os.environ.get("TORCHELASTIC_RUN_ID", "none"),
                                    os.environ.get("TORCHELASTIC_RESTART_COUNT", "0"),
                                    os.environ.get("PYGEMMA_RDZV_KEY") or os.getppid()))     # ranks of one launch share a parent
    return os.path.join(d, f"pygemma_rdzv_{key}.id")


def exchange_id(rank, world, make_id, timeout=300.0, path=None):
    """Rank 0 calls make_id() -> bytes and publishes it (write + atomic rename); the other ranks wait for the file.
    Returns the id on every rank.  Single node by construction (the path's job is one node's GPUs)."""
    if world == 1:
        return make_id()
    path = path or _rdzv_path()
    if rank == 0:
        uid = make_id()
        tmp = f"{path}.{os.getpid()}.tmp"
        with open(tmp, "wb") as f:
            f.write(uid)
        os.replace(tmp, path)
        return uid
    t0 = time.time()
    while True:
        try:
            with open(path, "rb") as f:
                uid = f.read()
            if uid:
                return uid
        except FileNotFoundError:
            pass
        if time.time() - t0 > timeout:
            raise TimeoutError(f"rank {rank}: no RCCL id at {path} after {timeout:.0f} s (is rank 0 running?)")
        time.sleep(0.01)


def retire_id(rank, path=None):
    """After the communicator exists (its creation synchronises all ranks) rank 0 removes the file."""
    if rank == 0:
        try:
            os.unlink(path or _rdzv_path())
        except FileNotFoundError:
            pass


class Communicator:
    """pg_comm of this rank, bound to `ctx` (collectives are enqueued on ctx's stream)."""

    def __init__(self, ctx, rank, world, uid):
        from . import _lib
        self._lib, self.L = _lib, _lib.load()
        self.ctx, self.rank, self.world = ctx, int(rank), int(world)
        h = C.c_void_p()
        buf = (C.c_char * 128).from_buffer_copy(uid[:128].ljust(128, b"\0"))
        _lib.check(self.L.pg_comm_init_rank(ctx.handle, self.world, self.rank, buf, C.byref(h)), "pg_comm_init_rank")
        self.handle = h
        self._scalar = ctx.alloc(8)

    def broadcast(self, ptr, nbytes, root=0):
        self._lib.check(self.L.pg_comm_broadcast_dev(self.handle, ptr, int(nbytes), int(root)), "pg_comm_broadcast_dev")

    def allgather(self, send_ptr, recv_ptr, nbytes_per_rank):
        self._lib.check(self.L.pg_comm_allgather_dev(self.handle, send_ptr, recv_ptr, int(nbytes_per_rank)), "pg_comm_allgather_dev")

    def allreduce_max(self, value):
        """MAX of a Python float over the ranks (bench.py's max-over-ranks time)."""
        self._scalar.upload(np.array([value], np.float64))
        self._lib.check(self.L.pg_comm_allreduce_f64_dev(self.handle, self._scalar.ptr, 1, 1), "pg_comm_allreduce_f64_dev")
        self.ctx.sync()
        return float(self._scalar.download((1,), np.float64)[0])

    def barrier(self):
        self._lib.check(self.L.pg_comm_barrier(self.handle), "pg_comm_barrier")

    def close(self):
        if self.handle:
            self.L.pg_comm_destroy(self.handle)
            self.handle = None


def env_rank():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", os.environ.get("RANK", "0")))


def init(ctx=None):
    """Communicator over the launcher's ranks (RANK, WORLD_SIZE, LOCAL_RANK); `ctx` defaults to a new context on GPU LOCAL_RANK."""
    from . import _lib
    rank, world, local = env_rank()
    ctx = ctx or _lib.Context(local)
    L = _lib.load()

    def make_id():
        buf = (C.c_char * 128)()
        _lib.check(L.pg_comm_unique_id(buf), "pg_comm_unique_id")
        return bytes(buf.raw)
    uid = exchange_id(rank, world, make_id)
    comm = Communicator(ctx, rank, world, uid)
    retire_id(rank)
    return comm


def gather_result_rows(comm, res_ptr, all_ptr, cols):
    """All-gather of every rank's padded block of `cols` 32-byte result rows ([F | p | beta | se | tau | lambda], the layout
    pg_assoc_multi uses): afterwards all_ptr holds world x cols rows in rank order = SNP order."""
    comm.allgather(res_ptr, all_ptr, 32 * int(cols))


def unpack_block(host_bytes, cols, count):
    """One rank's block (cols rows, `count` of them real) -> dict of the six columns."""
    b = np.frombuffer(host_bytes, np.uint8, 32 * cols)
    F = b[:8 * cols].view(np.float64)[:count]
    pv = b[8 * cols:16 * cols].view(np.float64)[:count]
    f4 = b[16 * cols:32 * cols].view(np.float32).reshape(4, cols)[:, :count]
    return {"beta": f4[0].copy(), "se_beta": f4[1].copy(), "tau": f4[2].copy(), "lambda": f4[3].astype(np.float64),
            "F_wald": F.copy(), "p_wald": pv.copy()}


# ---- row packing over torch.distributed (gloo on CPU): the tests' stand-in transport for the same sharding logic -----------
def pack_rows(res):
    """dict of per-SNP columns -> (p_local, 8) float32 rows carrying the bits of
    [beta, se, tau, lambda(f32), F_wald (f64 = 2 words), p_wald (f64 = 2 words)] = 32 B per SNP."""
    p = len(res["beta"])
    rows = np.empty((p, 8), np.float32)
    rows[:, 0], rows[:, 1], rows[:, 2] = res["beta"], res["se_beta"], res["tau"]
    rows[:, 3] = np.asarray(res["lambda"], np.float64).astype(np.float32)
    rows[:, 4:6] = np.ascontiguousarray(res["F_wald"], np.float64).view(np.float32).reshape(p, 2)
    rows[:, 6:8] = np.ascontiguousarray(res["p_wald"], np.float64).view(np.float32).reshape(p, 2)
    return rows


def unpack_rows(rows):
    rows = np.ascontiguousarray(rows, np.float32)
    return {"beta": rows[:, 0].copy(), "se_beta": rows[:, 1].copy(), "tau": rows[:, 2].copy(),
            "lambda": rows[:, 3].astype(np.float64),
            "F_wald": np.ascontiguousarray(rows[:, 4:6]).view(np.float64).reshape(-1),
            "p_wald": np.ascontiguousarray(rows[:, 6:8]).view(np.float64).reshape(-1)}


def gather_rows(rows_local, p, device=None):
    """All ranks contribute their (p_local, 8) block; every rank gets the (p, 8) table in SNP order.
    Blocks are padded to ceil(p/world) rows so one all_gather moves everything (a single small message per rank)."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    cols = int(np.ceil(p / world))
    buf = torch.zeros((cols, 8), dtype=torch.float32, device=device)
    t = torch.as_tensor(rows_local, dtype=torch.float32, device=device)
    buf[: t.shape[0]] = t
    out = torch.empty((world * cols, 8), dtype=torch.float32, device=device)
    dist.all_gather_into_tensor(out, buf)
    full = out.cpu().numpy()
    keep = np.concatenate([np.arange(r * cols, r * cols + (shard_range(p, r, world)[1] - shard_range(p, r, world)[0]))
                           for r in range(world)])
    return full[keep]
