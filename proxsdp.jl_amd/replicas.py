"""Multi-GPU plumbing for the paths that do not shard.

Every benchmarked BASELINE config has ONE PSD block, and a single block does not
shard: N GPUs run N independent solves ("replicas only"),
one process per GPU, no data-path collective.  `torch.distributed` (backend
"nccl" = RCCL on the GPU box, "gloo" in the CPU tests) is used only for the
barrier that brackets the timed region and for the MAX / SUM reductions of the
per-rank timings and unit counts."""
import os


def rank_info():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def replica_seed(base_seed, rank):
    """Each replica solves its own instance of the same family."""
    return int(base_seed) + int(rank)


def assign_blocks(n_blocks, world):
    """Block -> rank map for block-diagonal models (one block per GPU when
    n_blocks == world; round-robin otherwise).  Used by the block-sharded path
    (next row of SURVEY.md section 8e) and by its tests."""
    return [b % world for b in range(n_blocks)]


def init(backend, rank, world, device=None):
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    kw = {}
    if device is not None:
        kw["device_id"] = device
    dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return dist


def aggregate(dist, steps_local, seconds_local, device="cpu"):
    """value = (sum over ranks of the units processed) / (max over ranks of the time)."""
    import torch
    if dist is None:
        return float(steps_local), float(seconds_local)
    t = torch.tensor([float(seconds_local)], dtype=torch.float64, device=device)
    s = torch.tensor([float(steps_local)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(s, op=dist.ReduceOp.SUM)
    return float(s.item()), float(t.item())
