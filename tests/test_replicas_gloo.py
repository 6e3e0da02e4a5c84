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


def _run_bench(extra_env, *argv):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra_env)
    return subprocess.run([sys.executable, os.path.join(root, "bench.py"), *argv], env=env, cwd=root,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)


def test_bench_gpus_flag_spawns_the_ranks_itself():
    """`python bench.py --gpus 2` with no launcher in the environment must reach two ranks (the flag was
    parsed and ignored).  Dry run = the rank plumbing without the solve (no device here), gloo backend."""
    import json
    r = _run_bench({"PROXSDP_BENCH_DRYRUN": "1", "PROXSDP_BENCH_BACKEND": "gloo"}, "--gpus", "2", "--steps", "7")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                      # rank 0 prints ONE line
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["dry_run"] is True
    assert line["units_all_ranks"] == 14.0                # SUM over ranks
    assert line["max_seconds"] == 2.0                     # MAX over ranks (rank r reports 1 + r)


def test_bench_refuses_a_world_size_that_is_not_gpus():
    r = _run_bench({"PROXSDP_BENCH_DRYRUN": "1", "RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1"}, "--gpus", "2")
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)
    r = _run_bench({"PROXSDP_BENCH_DRYRUN": "1"}, "--gpus", "1", "--steps", "3")
    assert r.returncode == 0 and '"n_gpus": 1' in r.stdout


def test_block_assignment():
    assert replicas.assign_blocks(8, 8) == list(range(8))
    assert replicas.assign_blocks(8, 2) == [0, 1, 0, 1, 0, 1, 0, 1]
    assert replicas.assign_blocks(3, 1) == [0, 0, 0]


# ----------------------------------------------------------------- block-sharded path (host side)
def test_split_block_diagonal():
    import numpy as np
    from proxsdp_jl_amd import problems, sharded
    a, b = problems.mimo(3, seed=1), problems.maxcut(5, seed=2)
    pr = problems.block_diag_problems([a, b])
    s0, m0 = sharded.split_block_diagonal(pr, [0, 1], 0)
    s1, m1 = sharded.split_block_diagonal(pr, [0, 1], 1)
    assert s0.n == a.n and s1.n == b.n
    assert (s0.A != a.A).nnz == 0 and (s0.G != a.G).nnz == 0 and np.array_equal(s0.c, a.c)
    assert (s1.A != b.A).nnz == 0 and np.array_equal(s1.b, b.b) and s1.G.shape[0] == 0
    assert np.array_equal(np.sort(np.concatenate([m0["vars"], m1["vars"]])), np.arange(pr.n))
    assert m0["coupling"] is None and m1["coupling"] is None
    # a row coupling the two blocks: rejected on request, otherwise carried by BOTH shards (each with
    # its own columns), listed in `coupling`, owned by the lowest shard that touches it
    import scipy.sparse as sp
    cp = problems.Problem(n=pr.n, A=sp.vstack([pr.A, sp.csr_matrix(([1.0, 2.0], ([0, 0], [0, pr.n - 1])), shape=(1, pr.n))]).tocsc(),
                          b=np.append(pr.b, 7.0), G=pr.G, h=pr.h, c=pr.c, psd=pr.psd)
    with pytest.raises(ValueError):
        sharded.split_block_diagonal(cp, [0, 1], 0, allow_coupling=False)
    c0, k0 = sharded.split_block_diagonal(cp, [0, 1], 0)
    c1, k1 = sharded.split_block_diagonal(cp, [0, 1], 1)
    assert c0.A.shape[0] == a.A.shape[0] + 1 and c1.A.shape[0] == b.A.shape[0] + 1
    assert np.array_equal(k0["coupling"]["rows"], [a.A.shape[0]]) and np.array_equal(k1["coupling"]["rows"], [b.A.shape[0]])
    assert k0["coupling"]["owned"].tolist() == [1] and k1["coupling"]["owned"].tolist() == [0]
    assert c0.b[-1] == 7.0 and c1.b[-1] == 7.0
    assert c0.A[-1].nnz == 1 and c0.A[-1, 0] == 1.0 and c1.A[-1].nnz == 1 and c1.A[-1, c1.n - 1] == 2.0


def _reduce_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import numpy as np
    from proxsdp_jl_amd import sharded
    dist = replicas.init("gloo", rank, world)
    red = sharded.make_reduce(dist)
    sums = np.array([1.0 + rank, 10.0]); maxs = np.array([float(rank), -1.0 - rank, 5.0])
    red(sums, maxs)
    # the coupling-row all-reduce on a host buffer (gloo path of proxsdp_problem.reduce_vec_fn)
    redv = sharded.make_reduce_vec(dist)
    buf = np.array([1.0 + rank, 2.0, -3.0 * rank])
    redv(buf.ctypes.data, len(buf), False)
    q.put((rank, sums.tolist(), maxs.tolist(), buf.tolist()))
    dist.destroy_process_group()


def test_sharded_reduce_two_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + (os.getpid() % 200)
    procs = [ctx.Process(target=_reduce_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, sums, maxs, buf in out:
        assert sums == [3.0, 20.0] and maxs == [1.0, -1.0, 5.0]
        assert buf == [3.0, 4.0, -3.0]


def _fail_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from proxsdp_jl_amd import problems, sharded
    dist = replicas.init("gloo", rank, world)
    pr = problems.maxcut(5, seed=2)                     # ONE block for two ranks
    try:
        sharded.solve_sharded(pr, dist, rank, world)
        q.put((rank, "no error"))
    except ValueError as e:
        q.put((rank, "ValueError"))
    except RuntimeError as e:
        q.put((rank, "RuntimeError"))
    dist.destroy_process_group()


def test_sharded_solve_fails_on_every_rank_together():
    """more ranks than PSD blocks: every rank raises before any solve collective (no deadlock)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29900 + (os.getpid() % 90)
    procs = [ctx.Process(target=_fail_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(o[1] in ("ValueError", "RuntimeError") for o in out)
