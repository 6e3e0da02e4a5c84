"""N > 1 path of bench.py on CPU: two gloo ranks, barrier + MAX-over-ranks timing
and SUM of units, as the replicas-only multi-GPU mode does with RCCL."""
import multiprocessing as mp
import os

import pytest

from proxsdp_jl_amd import replicas


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import numpy as np
    from proxsdp_jl_amd import problems
    dist = replicas.init("gloo", rank, world)
    r, lr, w = replicas.rank_info()
    pr = problems.maxcut(40, seed=replicas.replica_seed(5, r))      # each replica its own instance
    dist.barrier()
    steps, secs = replicas.aggregate(dist, steps_local=100 + r, seconds_local=1.0 + 0.5 * r)
    dist.barrier()
    q.put((r, w, steps, secs, float(np.abs(pr.c).sum())))
    dist.destroy_process_group()


def test_two_rank_replica_aggregation():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 400)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, w0, s0, t0, c0), (r1, w1, s1, t1, c1) = out
    assert (r0, r1) == (0, 1) and w0 == w1 == 2
    assert s0 == s1 == 201.0                  # SUM of units
    assert t0 == t1 == 1.5                    # MAX of times
    assert c0 != c1                           # different instances per replica


def test_block_assignment():
    assert replicas.assign_blocks(8, 8) == list(range(8))
    assert replicas.assign_blocks(8, 2) == [0, 1, 0, 1, 0, 1, 0, 1]
    assert replicas.assign_blocks(3, 1) == [0, 0, 0]
