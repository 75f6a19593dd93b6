"""world_size-2 gloo test (CPU) of the N>1 path's host logic: SampleIter-style sharding, 32-byte result rows,
all-gather back into SNP order.  Each rank computes its shard with the oracle (the checker stands in for the GPU
kernels here: what is under test is the sharding + exchange, not the arithmetic)."""
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
    import torch.distributed as dist
    from oracle import oracle as O
    from pygemma_amd import dist as pgd
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    z = np.load(os.path.join(%r, "tests", "golden", "panel_weak_n300_c3.npz"))
    p = 37                                   # ragged: ceil(37/2) = 19 + 18
    X = np.ascontiguousarray(z["X"][:, :p])
    a, b = pgd.shard_range(p, rank, world)
    res = O.calculate(z["d"], z["Y"], z["W"], np.ascontiguousarray(X[:, a:b]), grid=False, order=1, nthreads=1)
    full = pgd.gather_rows(pgd.pack_rows(res), p)
    if rank == 0:
        np.save(sys.argv[1], full)
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
    out = tmp_path / "rows.npy"
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, OMP_NUM_THREADS="1")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                           "--master-addr", "127.0.0.1", "--master-port", str(port), str(script), str(out)],
                          env=env, timeout=600)
    rows = np.load(out)
    got = pgd.unpack_rows(rows)
    z = np.load(os.path.join(ROOT, "tests", "golden", "panel_weak_n300_c3.npz"))
    ref = O.calculate(z["d"], z["Y"], z["W"], np.ascontiguousarray(z["X"][:, :37]), grid=False, order=1, nthreads=2)
    for col in ("beta", "se_beta", "tau", "F_wald", "p_wald"):
        a, b = np.ascontiguousarray(got[col]), np.ascontiguousarray(ref[col].astype(got[col].dtype))
        assert (a.view(np.uint8) == b.view(np.uint8)).all(), col
    assert (got["lambda"] == ref["lambda"]).all()
    # and the reference's own rows for those SNPs
    assert (got["beta"].view(np.uint32) == z["brent_beta"][:37].view(np.uint32)).all()


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
    for r in (1, 0):          # rank 1 first: it has to wait for rank 0's file
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   PYGEMMA_RDZV_KEY=f"test{os.getpid()}", PYGEMMA_RDZV_DIR=str(tmp_path))
        procs.append(subprocess.Popen([sys.executable, str(script), out], env=env))
    assert [p.wait(timeout=120) for p in procs] == [0, 0]
    assert open(out + ".0").read() == "0 501" and open(out + ".1").read() == "501 1001"
    from pygemma_amd import dist
    got = dist.unpack_block(np.arange(32 * 4, dtype=np.uint8).tobytes(), 4, 3)
    assert got["beta"].shape == (3,) and got["F_wald"].dtype == np.float64
