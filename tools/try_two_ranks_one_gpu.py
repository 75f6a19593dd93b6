"""Does this RCCL accept two ranks on ONE device (a 2-rank rehearsal on a 1-GPU box)?  Spawns two processes, both on GPU 0, and
tries to form the communicator + one broadcast + one all-gather.  Exit code 0 = it works; otherwise prints RCCL's refusal."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if "RANK" not in os.environ:
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533",
                   PYGEMMA_RDZV_KEY=f"try{os.getpid()}", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)], env=env))
    t0 = time.time(); rcs = [None, None]
    while time.time() - t0 < 90 and any(rc is None for rc in rcs):
        rcs = [p.poll() for p in procs]; time.sleep(0.2)
    for p in procs:
        if p.poll() is None: p.terminate()
    print("exit codes:", rcs); sys.exit(0 if rcs == [0, 0] else 1)
sys.path.insert(0, ROOT)
import numpy as np
from pygemma_amd import _lib, dist
rank = int(os.environ["RANK"])
ctx = _lib.Context(0)
try:
    comm = dist.init(ctx)
except Exception as ex:
    print(f"rank {rank}: communicator refused: {ex}", flush=True); sys.exit(3)
buf = ctx.to_device(np.full(1024, rank + 1, np.float32))
comm.broadcast(buf.ptr, 4096, root=0); ctx.sync()
got = buf.download((1024,), np.float32)
allb = ctx.alloc(2 * 4096); mine = ctx.to_device(np.full(1024, 10 + rank, np.float32))
comm.allgather(mine.ptr, allb.ptr, 4096); ctx.sync()
g = allb.download((2048,), np.float32)
ok = (got == 1).all() and (g[:1024] == 10).all() and (g[1024:] == 11).all()
print(f"rank {rank}: broadcast + all-gather over two ranks on one GPU: {'ok' if ok else 'WRONG'}", flush=True)
comm.barrier(); comm.close()
sys.exit(0 if ok else 4)
