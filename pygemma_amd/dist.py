"""Multi-rank helpers: SNP-range sharding identical to the reference's SampleIter (lmm/lmm.py:427-434) and the
gather of per-SNP result rows.  Backend-agnostic torch.distributed ("nccl" = RCCL over xGMI on MI355X, "gloo" on CPU
for the tests).  torch is imported lazily: the single-GPU path never needs it."""
import numpy as np


def shard_range(p, rank, world):
    """Columns [a, b) of rank `rank`: contiguous blocks of ceil(p/world), like SampleIter."""
    cols = int(np.ceil(p / world))
    a = min(rank * cols, p)
    return a, min(a + cols, p)


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
