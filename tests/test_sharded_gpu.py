"""Block-sharded solve: two ranks, one PSD block each, scalars
exchanged through torch.distributed (gloo here; RCCL in production) -- must reproduce the
single-process solve of the whole block-diagonal model.  Both ranks share the one GPU of
the test box."""
import multiprocessing as mp
import os

import numpy as np
import pytest

from proxsdp_jl_amd import binding as B
from proxsdp_jl_amd import problems as P
from proxsdp_jl_amd.optimizer import Optimizer

pytestmark = pytest.mark.gpu


def _model():
    return P.block_diag_problems([P.maxcut(120, seed=1), P.maxcut(150, seed=2)], name="two-maxcut")


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from proxsdp_jl_amd import replicas, sharded
    dist = replicas.init("gloo", rank, world)
    opt, sol, maps = sharded.solve_sharded(_model(), dist, rank, world, device_id=0, max_iter=300)
    q.put((rank, sol.status, int(sol.iter), sol.objval, sol.dual_objval, sol.gap, int(sol.final_rank),
           maps["vars"], sol.primal, sol.trace[:, [1, 2, 7, 11]]))
    dist.destroy_process_group()


def test_two_shards_reproduce_the_single_process_solve():
    assert B.device_count() > 0
    pr = _model()
    opt = Optimizer(max_iter=300, support_path=1)
    ref = opt.optimize(pr, trace_capacity=300)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 300)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted((q.get(timeout=600) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    x = np.zeros(pr.n)
    for (rank, status, it, obj, dobj, gap, frank, vars_, primal, tr) in out:
        assert status == ref.status and it == ref.iter
        assert abs(obj - ref.objval) <= 1e-9 * (1 + abs(ref.objval))
        assert abs(dobj - ref.dual_objval) <= 1e-9 * (1 + abs(ref.dual_objval))
        assert frank == ref.final_rank
        assert np.array_equal(tr[:, 3], ref.trace[:, 11])                       # same linesearch trials
        assert np.allclose(tr[:, :3], ref.trace[:, [1, 2, 7]], rtol=1e-9, atol=1e-12)
        x[vars_] = primal
    assert np.allclose(x, ref.primal, rtol=0, atol=1e-9)
