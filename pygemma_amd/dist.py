"""One process per GPU: SNP-range sharding identical to the reference's SampleIter (lmm/lmm.py:427-434), the RCCL
communicator of the C ABI (pg_comm_*: librccl over xGMI, no PyTorch) with a standard-library rendezvous, and the gather of
the 32-byte per-SNP result rows in rank order = SNP order (the reference's ordered concatenation, lmm/lmm.py:393,401).

    comm = dist.init()                       # RANK / WORLD_SIZE / LOCAL_RANK / MASTER_PORT from the launcher's environment
    comm.broadcast(dU.ptr, nbytes, root=0)   # eigenvectors from the rank that ran the eigensolver
    comm.allgather(res.ptr, all.ptr, nbytes) # padded row blocks
    comm.close()

Any launcher that sets those variables works (torch.distributed.run, srun, mpirun wrappers, bench.py's own spawner).
Nothing here imports torch: the world-size-2 CPU test (tests/test_dist_gloo.py) moves the same 32-byte row blocks as bytes
over gloo with a transport helper of its own.
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
ID_BYTES = 128


def _launch_epoch():
    """Start time (Unix seconds) of the process all ranks of one launch share as parent, from /proc: an id file older than
    that was left by an earlier launch that died between publishing and retiring it (same port, recycled pid)."""
    try:
        with open(f"/proc/{os.getppid()}/stat") as f:
            ticks = int(f.read().rsplit(")", 1)[1].split()[19])          # field 22: starttime, in clock ticks since boot
        with open("/proc/stat") as f:
            btime = next(int(line.split()[1]) for line in f if line.startswith("btime"))
        return btime + ticks / os.sysconf("SC_CLK_TCK")
    except (OSError, ValueError, StopIteration, IndexError):
        return 0.0


def _rdzv_path():
    """<dir>/pygemma_rdzv_<uid>/<key>.id — a directory of this user's own (mode 0700).  The key is the launcher's nonce
    (PYGEMMA_RDZV_KEY: bench.py's spawner draws a fresh one per launch) or, under launchers that give none, the port, the elastic run
    id / restart count and the shared parent's pid."""
    base = os.environ.get("PYGEMMA_RDZV_DIR") or ("/dev/shm" if os.path.isdir("/dev/shm") else "/tmp")
    d = os.path.join(base, f"pygemma_rdzv_{os.getuid()}")
    os.makedirs(d, mode=0o700, exist_ok=True)
    st = os.lstat(d)
    if not os.path.isdir(d) or os.path.islink(d) or st.st_uid != os.getuid() or (st.st_mode & 0o077):
        raise PermissionError(f"rendezvous directory {d} is not a private directory of this user")
    key = "_".join(str(x) for x in (os.environ.get("MASTER_PORT", "0"), os.environ.get("TORCHELASTIC_RUN_ID", "none"),
                                    os.environ.get("TORCHELASTIC_RESTART_COUNT", "0"),
                                    os.environ.get("PYGEMMA_RDZV_KEY") or os.getppid()))     # ranks of one launch share a parent
    return os.path.join(d, f"{key}.id")


def exchange_id(rank, world, make_id, timeout=300.0, path=None):
    """Rank 0 calls make_id() -> bytes and publishes it (exclusive create, mode 0600, atomic rename); the other ranks wait for a
    file of exactly ID_BYTES bytes, owned by this user and not older than the launch.  Returns the id on every rank.
    Single node by construction (the path's job is one node's GPUs)."""
    if world == 1:
        return make_id()
    path = path or _rdzv_path()
    if rank == 0:
        uid = make_id()
        if len(uid) != ID_BYTES:
            raise ValueError(f"RCCL id of {len(uid)} bytes (expected {ID_BYTES})")
        tmp = f"{path}.{os.getpid()}.tmp"
        fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL | os.O_NOFOLLOW, 0o600)
        with os.fdopen(fd, "wb") as f:
            f.write(uid)
        os.replace(tmp, path)          # replaces a stale file of the same key, if any
        return uid
    t0, epoch = time.time(), _launch_epoch()
    while True:
        try:
            fd = os.open(path, os.O_RDONLY | os.O_NOFOLLOW)
            with os.fdopen(fd, "rb") as f:
                st = os.fstat(f.fileno())
                uid = f.read(ID_BYTES + 1)
            if len(uid) == ID_BYTES and st.st_uid == os.getuid() and st.st_mtime >= epoch - 1.0:
                return uid
        except FileNotFoundError:
            pass
        if time.time() - t0 > timeout:
            raise TimeoutError(f"rank {rank}: no fresh {ID_BYTES}-byte RCCL id at {path} after {timeout:.0f} s (is rank 0 running?)")
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
        if len(uid) != ID_BYTES:
            raise ValueError(f"RCCL id of {len(uid)} bytes (expected {ID_BYTES})")
        buf = (C.c_char * ID_BYTES).from_buffer_copy(uid)
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


def pack_block(res, cols):
    """Inverse of unpack_block: the six columns of `count` <= cols SNPs -> the 32 * cols bytes of one rank's padded block
    (what the device writes; used where a block is assembled on the host: tests, tools)."""
    count = len(res["beta"])
    b = np.zeros(32 * cols, np.uint8)
    b[:8 * cols].view(np.float64)[:count] = res["F_wald"]
    b[8 * cols:16 * cols].view(np.float64)[:count] = res["p_wald"]
    f4 = b[16 * cols:32 * cols].view(np.float32).reshape(4, cols)
    for k, name in enumerate(("beta", "se_beta", "tau", "lambda")):
        f4[k, :count] = np.asarray(res[name]).astype(np.float32)
    return b.tobytes()


def unpack_gathered(host_bytes, p, world):
    """The all-gathered buffer (world blocks of ceil(p/world) rows, rank order) -> the six columns of all p SNPs in SNP order."""
    cols = int(np.ceil(p / world))
    parts = []
    for r in range(world):
        a, b = shard_range(p, r, world)
        parts.append(unpack_block(host_bytes[32 * cols * r:32 * cols * (r + 1)], cols, b - a))
    return {k: np.concatenate([q[k] for q in parts]) for k in parts[0]}
