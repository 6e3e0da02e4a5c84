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


def _coupled_model():
    """the two Max-Cut blocks plus ONE equality row that touches both (X1[0,0] + 2 X2[0,0] = 3, consistent
    with diag = 1) and one coupling inequality (X1[1,1] - X2[1,1] <= 0.5): rows no shard owns alone"""
    import scipy.sparse as sp
    pr = _model()
    n1 = P.maxcut(120, seed=1).n
    row = sp.csr_matrix(([1.0, 2.0], ([0, 0], [0, n1])), shape=(1, pr.n))
    g = sp.csr_matrix(([1.0, -1.0], ([0, 0], [2, n1 + 2])), shape=(1, pr.n))        # (1,1) is entry 2 of a triangle
    return P.Problem(n=pr.n, A=sp.vstack([pr.A, row]).tocsc(), b=np.append(pr.b, 3.0),
                     G=sp.vstack([pr.G, g]).tocsc(), h=np.append(pr.h, 0.5), c=pr.c, psd=pr.psd, name="two-maxcut-coupled")


def _worker(rank, world, port, q, coupled=False, backend="gloo"):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from proxsdp_jl_amd import replicas, sharded
    dev = None
    if backend == "nccl":
        import torch
        torch.cuda.set_device(rank)
        dev = torch.device("cuda", rank)
    dist = replicas.init(backend, rank, world, device=dev)
    model = _coupled_model() if coupled else _model()
    opt, sol, maps = sharded.solve_sharded(model, dist, rank, world, device_id=rank if backend == "nccl" else 0,
                                           collective_device=dev, max_iter=300)
    q.put((rank, sol.status, int(sol.iter), sol.objval, sol.dual_objval, sol.gap, int(sol.final_rank),
           maps["vars"], sol.primal, sol.trace[:, [1, 2, 7, 11]]))
    dist.destroy_process_group()


def _ngpu():
    import torch
    return torch.cuda.device_count()


@pytest.mark.parametrize("coupled,backend", [(False, "gloo"), (True, "gloo"), (True, "nccl")],
                         ids=["block-diagonal-gloo", "coupling-rows-gloo", "coupling-rows-rccl"])
def test_two_shards_reproduce_the_single_process_solve(coupled, backend):
    """gloo: both ranks share the one GPU of the test box (coupling rows all-reduced through host
    memory); rccl: one GPU per rank, the coupling rows of M x all-reduced on the library's DEVICE
    buffer over xGMI -- runs only where >= 2 GPUs are visible."""
    assert B.device_count() > 0
    if backend == "nccl" and _ngpu() < 2:
        pytest.skip("needs 2 GPUs (RCCL over xGMI); the gloo variant covers the same path on one GPU")
    pr = _coupled_model() if coupled else _model()
    opt = Optimizer(max_iter=300, support_path=1)
    ref = opt.optimize(pr, trace_capacity=300)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 300) + (17 if coupled else 0) + (29 if backend == "nccl" else 0)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, coupled, backend)) for r in range(2)]
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


def _devptr_worker(port, q):
    os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    from proxsdp_jl_amd import replicas, sharded
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist = replicas.init("nccl", 0, 1, device=dev)
    t = torch.arange(1000, dtype=torch.float64, device=dev) * 0.5
    sharded.make_reduce_vec(dist, dev)(t.data_ptr(), t.numel(), True)        # RCCL all-reduce on a RAW device pointer
    red = sharded.make_reduce(dist, dev, 1)
    sums, maxs = np.array([1.5, 2.5]), np.array([-3.0])
    red(sums, maxs)
    q.put((t.cpu().numpy(), sums, maxs))
    dist.destroy_process_group()


def test_rccl_collectives_on_raw_device_pointers():
    """the device-buffer form of proxsdp_problem.reduce_vec_fn (RCCL all-reduce on a raw device pointer
    wrapped through __cuda_array_interface__) and the single all-gather scalar reduce, on a 1-rank
    RCCL group: validates the plumbing the 2-GPU variant above needs, on the one GPU of the box."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_devptr_worker, args=(29950 + (os.getpid() % 40), q))
    p.start()
    t, sums, maxs = q.get(timeout=300)
    p.join(timeout=60)
    assert p.exitcode == 0
    assert np.array_equal(t, np.arange(1000) * 0.5)
    assert sums.tolist() == [1.5, 2.5] and maxs.tolist() == [-3.0]
