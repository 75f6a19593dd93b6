"""world_size-2 gloo test (CPU) of the N>1 path's host logic, on the product's own block format: SampleIter-style sharding
(dist.shard_range), one rank's padded block of 32-byte result rows laid out [F | p | beta | se | tau | lambda] exactly as
pg_assoc_multi / bench.py hand it to pg_comm_allgather_dev, an all-gather of those bytes in rank order, and
dist.unpack_block / dist.unpack_gathered on the result.  gloo moves the bytes (uint8) where RCCL does on the GPUs; the transport
helper lives here, not in the package (which imports no torch).  Each rank computes its shard with the oracle (the checker stands
in for the GPU kernels: what is under test is the sharding + exchange + unpacking, not the arithmetic)."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys, numpy as np
    sys.path.insert(0, %r)
    import torch
    import torch.distributed as dist
    from oracle import oracle as O
    from pygemma_amd import dist as pgd
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    z = np.load(os.path.join(%r, "tests", "golden", "panel_weak_n300_c3.npz"))
    p = 37                                   # ragged: ceil(37/2) = 19 + 18
    X = np.ascontiguousarray(z["X"][:, :p])
    a, b = pgd.shard_range(p, rank, world)
    cols = int(np.ceil(p / world))
    res = O.calculate(z["d"], z["Y"], z["W"], np.ascontiguousarray(X[:, a:b]), grid=False, order=1, nthreads=1)
    block = pgd.pack_block(res, cols)                                    # 32 * cols bytes, the device's layout
    assert len(block) == 32 * cols
    mine = torch.frombuffer(bytearray(block), dtype=torch.uint8)
    everything = torch.empty(world * 32 * cols, dtype=torch.uint8)
    dist.all_gather_into_tensor(everything, mine)                        # what pg_comm_allgather_dev does over xGMI
    if rank == 0:
        open(sys.argv[1], "wb").write(everything.numpy().tobytes())
    dist.barrier()
    dist.destroy_process_group()
""")


def test_two_rank_gloo_shard_and_gather(tmp_path):
    from oracle import oracle as O
    from pygemma_amd import dist as pgd
    assert [pgd.shard_range(37, r, 2) for r in range(2)] == [(0, 19), (19, 37)]
    assert [pgd.shard_range(5, r, 8) for r in range(8)][:6] == [(0, 1), (1, 2), (2, 3), (3, 4), (4, 5), (5, 5)]
    script = tmp_path / "worker.py"
    script.write_text(WORKER % (ROOT, ROOT))
    out = tmp_path / "rows.bin"
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, OMP_NUM_THREADS="1")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                           "--master-addr", "127.0.0.1", "--master-port", str(port), str(script), str(out)],
                          env=env, timeout=600)
    raw = open(out, "rb").read()
    assert len(raw) == 2 * 32 * 19
    got = pgd.unpack_gathered(raw, 37, 2)
    # the second rank's block alone, through unpack_block (18 real rows of 19)
    blk1 = pgd.unpack_block(raw[32 * 19:], 19, 18)
    assert (blk1["beta"].view(np.uint32) == got["beta"][19:].view(np.uint32)).all()
    z = np.load(os.path.join(ROOT, "tests", "golden", "panel_weak_n300_c3.npz"))
    ref = O.calculate(z["d"], z["Y"], z["W"], np.ascontiguousarray(z["X"][:, :37]), grid=False, order=1, nthreads=2)
    for col in ("beta", "se_beta", "tau", "F_wald", "p_wald"):
        a, b = np.ascontiguousarray(got[col]), np.ascontiguousarray(ref[col].astype(got[col].dtype))
        assert a.shape == (37,) and (a.view(np.uint8) == b.view(np.uint8)).all(), col
    assert (got["lambda"] == ref["lambda"]).all()
    # and the reference's own rows for those SNPs
    assert (got["beta"].view(np.uint32) == z["brent_beta"][:37].view(np.uint32)).all()


def test_block_layout_round_trip_and_padding():
    """pack_block is the inverse of unpack_block; the padding rows of a short block never reach the output."""
    from pygemma_amd import dist as pgd
    rng = np.random.default_rng(3)
    res = {"beta": rng.standard_normal(5).astype(np.float32), "se_beta": rng.random(5).astype(np.float32),
           "tau": rng.random(5).astype(np.float32), "lambda": rng.random(5).astype(np.float32).astype(np.float64),
           "F_wald": rng.random(5), "p_wald": rng.random(5)}
    raw = pgd.pack_block(res, 8)
    assert len(raw) == 256
    # layout [F | p | beta | se | tau | lambda]: F of SNP 0 is the first 8 bytes, beta of SNP 0 sits at byte 16 * cols
    assert np.frombuffer(raw, np.float64, 1)[0] == res["F_wald"][0]
    assert np.frombuffer(raw, np.float32, 1, offset=16 * 8)[0] == res["beta"][0]
    back = pgd.unpack_block(raw, 8, 5)
    for k in res:
        assert (np.asarray(back[k]) == np.asarray(res[k])).all(), k


RDZV_WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %r)
    from pygemma_amd import dist
    rank, world, local = dist.env_rank()
    uid = dist.exchange_id(rank, world, lambda: bytes(range(128)), timeout=60)
    assert uid == bytes(range(128)) and len(uid) == 128, uid
    a, b = dist.shard_range(1001, rank, world)
    open(sys.argv[1] + f".{rank}", "w").write(f"{a} {b}")
""")


def test_rccl_id_rendezvous_two_processes_without_torch(tmp_path):
    """The stdlib rendezvous of pygemma_amd/dist.py (what bench.py and any launcher use to hand the 128-byte RCCL id from rank 0
    to the others): two processes with the launcher's environment variables, no torch, no GPU."""
    script = tmp_path / "rdzv.py"
    script.write_text(RDZV_WORKER % ROOT)
    out = str(tmp_path / "done")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    # a stale id of the same key, left by a launch that died before retiring it (older than this launch's parent): rank 1, started
    # first, must not take it — nor a short file
    key = f"test{os.getpid()}"
    d = tmp_path / f"pygemma_rdzv_{os.getuid()}"
    d.mkdir(mode=0o700)
    stale = d / f"{port}_none_0_{key}.id"
    stale.write_bytes(b"\xff" * 128)
    os.utime(stale, (1.0e9, 1.0e9))
    for r in (1, 0):          # rank 1 first: it has to wait for rank 0's file
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   PYGEMMA_RDZV_KEY=key, PYGEMMA_RDZV_DIR=str(tmp_path))
        for k in ("TORCHELASTIC_RUN_ID", "TORCHELASTIC_RESTART_COUNT"):
            env.pop(k, None)
        procs.append(subprocess.Popen([sys.executable, str(script), out], env=env))
        if r == 1:
            import time
            time.sleep(1.0)   # rank 1 is polling with the stale file in place
    assert [p.wait(timeout=120) for p in procs] == [0, 0]
    assert open(out + ".0").read() == "0 501" and open(out + ".1").read() == "501 1001"
    from pygemma_amd import dist
    got = dist.unpack_block(np.arange(32 * 4, dtype=np.uint8).tobytes(), 4, 3)
    assert got["beta"].shape == (3,) and got["F_wald"].dtype == np.float64
