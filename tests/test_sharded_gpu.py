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


def _worker(rank, world, port, q, coupled=False, backend="gloo", extra=None):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from proxsdp_jl_amd import replicas, sharded
    dev = None
    native = backend == "native"          # the library's own RCCL collectives; torch.distributed (gloo) only ships the id
    if backend == "nccl":
        import torch
        torch.cuda.set_device(rank)
        dev = torch.device("cuda", rank)
    dist = replicas.init("gloo" if native else backend, rank, world, device=dev)
    model = _coupled_model() if coupled else _model()
    comm = sharded.make_native_comm(dist, rank, world, device_id=rank) if native else None
    opt, sol, maps = sharded.solve_sharded(model, dist, rank, world, device_id=rank if backend != "gloo" else 0,
                                           collective_device=dev, native_comm=comm, max_iter=300, **(extra or {}))
    if native:
        assert sol.stats["rccl_reductions"] >= sol.iter, "native RCCL path not taken"
        B.rccl_comm_destroy(comm)
    q.put((rank, sol.status, int(sol.iter), sol.objval, sol.dual_objval, sol.gap, int(sol.final_rank),
           maps["vars"], sol.primal, sol.trace[:, [1, 2, 7, 11]]))
    dist.destroy_process_group()


def _ngpu():
    import torch
    return torch.cuda.device_count()


@pytest.mark.parametrize("coupled,backend", [(False, "gloo"), (True, "gloo"), (True, "nccl"), (True, "native")],
                         ids=["block-diagonal-gloo", "coupling-rows-gloo", "coupling-rows-rccl", "coupling-rows-rccl-native"])
def test_two_shards_reproduce_the_single_process_solve(coupled, backend):
    """gloo: both ranks share the one GPU of the test box (coupling rows all-reduced through host
    memory); rccl: one GPU per rank, the coupling rows of M x all-reduced on the library's DEVICE
    buffer over xGMI -- runs only where >= 2 GPUs are visible."""
    assert B.device_count() > 0
    if backend in ("nccl", "native") and _ngpu() < 2:
        pytest.skip("needs 2 GPUs (RCCL over xGMI); the gloo variant covers the same path on one GPU, "
                    "test_native_rccl_collectives_one_rank the library's own RCCL calls")
    pr = _coupled_model() if coupled else _model()
    opt = Optimizer(max_iter=300, support_path=1)
    ref = opt.optimize(pr, trace_capacity=300)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 300) + (17 if coupled else 0) + (29 if backend == "nccl" else 0) + (43 if backend == "native" else 0)
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


def test_two_shards_with_check_dual_feas_stop_where_the_single_process_solve_stops():
    """Round 6: `check_dual_feas` (pdhg.jl:154-173: the stop rule also asks for dual feasibility, tested every
    check_dual_feas_freq iterations) inside a block-sharded solve -- every shard tests its own columns and the largest value over
    the shards is the model's.  Two gloo ranks sharing the GPU against the single-process solve of the coupled model: same stop
    iteration (later than without the option), same status, objectives and trace."""
    assert B.device_count() > 0
    pr = _coupled_model()
    kw = dict(check_dual_feas=1, check_dual_feas_freq=25, tol_feasibility_dual=1e-3)
    ref = Optimizer(max_iter=300, support_path=1, **kw).optimize(pr, trace_capacity=300)
    plain = Optimizer(max_iter=300, support_path=1).optimize(pr)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 300) + 71
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, True, "gloo", kw)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted((q.get(timeout=600) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    print("stop with / without check_dual_feas:", ref.iter, plain.iter, "status", ref.status)
    for (rank, status, it, obj, dobj, gap, frank, vars_, primal, tr) in out:
        assert status == ref.status and it == ref.iter
        assert abs(obj - ref.objval) <= 1e-9 * (1 + abs(ref.objval))
        assert np.array_equal(tr[:, 3], ref.trace[:len(tr), 11])
        assert np.allclose(tr[:, :3], ref.trace[:len(tr)][:, [1, 2, 7]], rtol=1e-9, atol=1e-12)


def test_two_shards_without_linesearch_reproduce_the_single_process_solve():
    """Round 6: `line_search_flag = false` (dual_step!, pdhg.jl:584-609, fixed steps adapted by the residual
    balance) inside a block-sharded solve.  The sharded loop is built on the support path, which now serves that option with ONE
    candidate taken as it is (no in-place-norm revert).  Checked three ways: the support path against the dense vector passes in one
    process (same iterations, traces to 1e-10), both against the oracle, and two gloo ranks against the single-process solve."""
    import oracle
    assert B.device_count() > 0
    pr = _coupled_model()
    kw = dict(line_search_flag=0)
    dense = Optimizer(max_iter=300, support_path=0, **kw).optimize(pr, trace_capacity=300)
    ref = Optimizer(max_iter=300, support_path=1, **kw).optimize(pr, trace_capacity=300)
    assert ref.iter == dense.iter and ref.status == dense.status
    assert np.allclose(ref.trace[:, [1, 2, 3, 4, 7]], dense.trace[:, [1, 2, 3, 4, 7]], rtol=1e-10, atol=1e-12)
    o = oracle.Options(); o.max_iter = 300; o.line_search_flag = False
    orc = oracle.solve(pr, o, trace=True)
    m = min(len(orc.trace), len(ref.trace))
    exp = np.array([[t["prim_obj"], t["dual_obj"], t["primal_step"]] for t in orc.trace[:m]])
    assert int(orc.iter) == int(ref.iter)
    assert np.allclose(ref.trace[:m][:, [1, 2, 7]], exp, rtol=1e-7, atol=1e-9)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 300) + 97
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, True, "gloo", kw)) for r in range(2)]
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
        assert np.all(tr[:, 3] == 1)                                              # one "trial" per iteration
        assert np.allclose(tr[:, :3], ref.trace[:len(tr)][:, [1, 2, 7]], rtol=1e-9, atol=1e-12)
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


def _native_worker(q):
    """world = 1: the communicator is created and used by the library alone (no torch.distributed at all)"""
    from proxsdp_jl_amd import sharded
    assert B.rccl_available(), "librccl could not be loaded by the library"
    comm = sharded.make_native_comm(None, 0, 1, device_id=0)
    pr = _coupled_model()
    sub, maps = sharded.split_block_diagonal(pr, [0, 0], 0)
    assert maps["coupling"] is None                                   # one shard: nothing couples SHARDS ...
    # ... so mark the two cross-block rows (the last equality and the last inequality) as coupling rows by hand:
    # with one rank the all-reduce is the identity, but the gather -> ncclAllReduce -> scatter path runs
    p_, m_ = sub.A.shape[0], sub.G.shape[0]
    coupling = dict(rows=np.array([p_ - 1, p_ + m_ - 1], dtype=np.int64), owned=np.array([1, 1], dtype=np.int32))
    sol = Optimizer(max_iter=300).optimize(sub, coupling=coupling, nccl_comm=comm, trace_capacity=300)
    B.rccl_comm_destroy(comm)
    q.put((sol.status, int(sol.iter), sol.objval, sol.dual_objval, sol.primal, sol.trace[:, [1, 2, 7, 11]],
           int(sol.stats["rccl_reductions"])))


def test_native_rccl_collectives_one_rank():
    """proxsdp_problem.nccl_comm: the library loads librccl itself (dlopen), creates a 1-rank
    communicator through its helper entry points and issues ncclAllGather (packed scalar record, once per
    iteration) and ncclAllReduce (coupling rows of M x) on its OWN stream -- no Python callback.  A 1-rank
    sharded solve must reproduce the plain solve of the same model bit for bit."""
    assert B.device_count() > 0
    pr = _coupled_model()
    ref = Optimizer(max_iter=300, support_path=1).optimize(pr, trace_capacity=300)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_native_worker, args=(q,))
    p.start()
    status, it, obj, dobj, primal, tr, nred = q.get(timeout=600)
    p.join(timeout=60)
    assert p.exitcode == 0
    assert status == ref.status and it == ref.iter
    assert nred >= 2 * it, nred                                      # one scalar all-gather + one coupling all-reduce per iteration
    assert obj == ref.objval and dobj == ref.dual_objval
    assert np.array_equal(tr, ref.trace[:, [1, 2, 7, 11]])
    assert np.array_equal(primal, ref.primal)


def _failing_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PROXSDP_HIP_FAULT_INJECTION="1")
    from proxsdp_jl_amd import replicas, sharded
    dist = replicas.init("gloo", rank, world)
    kw = dict(max_iter=300)
    if rank == 1:
        kw["debug_fail_iteration"] = 7            # this shard's projection throws in iteration 7
    try:
        sharded.solve_sharded(_coupled_model(), dist, rank, world, device_id=0, **kw)
        q.put((rank, "returned"))
    except B.ProxSDPHipError as e:
        q.put((rank, str(e)))
    dist.destroy_process_group()


def test_a_failing_shard_makes_every_shard_abort_instead_of_hanging():
    """if one shard throws inside its loop the others used to wait in the next collective for ever.
    Now the failing shard keeps its exception, still joins the iteration's collectives (coupling all-reduce, scalar
    record) with a flag in the record, and every shard aborts right after that reduce: the failing one with its own
    error, the others with "another shard ... failed".  Fault injected through options.debug_fail_iteration."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29870 + (os.getpid() % 100)
    procs = [ctx.Process(target=_failing_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = dict(q.get(timeout=300) for _ in procs)              # a hang shows up as queue.Empty here
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert "injected projection failure" in out[1], out
    assert "another shard" in out[0], out


def _mimo_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from proxsdp_jl_amd import replicas, sharded
    dist = replicas.init("gloo", rank, world)
    model = P.block_diag_problems([P.mimo(512, seed=s_) for s_ in range(8)], name="mimo-x8")
    opt, sol, maps = sharded.solve_sharded(model, dist, rank, world, device_id=0, max_iter=400)
    q.put((rank, sol.status, int(sol.iter), sol.objval, sol.dual_objval, int(sol.final_rank), maps["vars"], sol.primal,
           sol.trace[:, [1, 2, 3, 4, 7, 11]], int(sol.stats["batched_block_steps"]), int(sol.stats["lanczos_matvecs"]),
           int(sol.stats["lanczos_restarts"])))
    dist.destroy_process_group()


def test_config4_shape_two_shards_of_four_blocks_reproduce_the_single_process_solve():
    """BASELINE config 4 at its REAL shape through the sharded path: MIMO n = 512 x 8 blocks (PSD side
    513 each, 2.1 M box rows) split 4 + 4 over two ranks (gloo; both share the one GPU of the test box -- the only
    multi-rank evidence obtainable without the 8-GPU node).  Each shard runs the BATCHED multi-block Lanczos driver on
    its four blocks (one launch per step, grid.z = block); together they must reproduce the single-process solve of the
    whole model: the oracle's 76 iterations (tests/golden/solve_mimo_n512_x8.json), same linesearch trials in every
    iteration.  MIMO's iterates have repeated eigenvalues at the truncation rank: the ulp-level
    difference between 'sum over eight blocks' and 'sum over two shards of four' is amplified along the trajectory, exactly
    as between library and oracle in test_config4_... -- so the trace must agree to 1e-9 over the first 20 iterations and to
    1e-7 over all 76 (measured: 1.0e-10 / 9.3e-9), with IDENTICAL mat-vec and restart totals (107 628 / 9262)."""
    import json
    from conftest import GOLDEN
    gold = json.loads((GOLDEN / "solve_mimo_n512_x8.json").read_text())
    pr = P.block_diag_problems([P.mimo(512, seed=s_) for s_ in range(8)], name="mimo-x8")
    ref = Optimizer(max_iter=400, support_path=1).optimize(pr, trace_capacity=400)
    assert ref.status == 1 and ref.iter == gold["iter"] == 76
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29400 + (os.getpid() % 150)
    procs = [ctx.Process(target=_mimo_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted((q.get(timeout=900) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    x = np.zeros(pr.n)
    mv = rs = 0
    for (rank, status, it, obj, dobj, frank, vars_, primal, tr, bsteps, mv_r, rs_r) in out:
        assert status == 1 and it == 76
        assert bsteps > 0, "the shard did not take the batched multi-block driver"
        assert np.array_equal(tr[:, 5], ref.trace[:, 11])                       # linesearch trials, every iteration
        R = ref.trace[:, [1, 2, 3, 4, 7]]
        rel = np.abs(tr[:, :5] - R) / np.maximum(1.0, np.abs(R))
        print("rank", rank, "max relative trace difference: first 20 iterations %.2e, all %d: %.2e" % (rel[:20].max(), len(R), rel.max()))
        assert rel[:20].max() <= 1e-9 and rel.max() <= 1e-7            # measured: 1.0e-10 / 9.3e-9
        assert abs(obj - ref.objval) <= 1e-6 * (1 + abs(ref.objval)) and frank == ref.final_rank
        x[vars_] = primal
        mv += mv_r
        rs += rs_r
    print("mat-vecs", mv, ref.stats["lanczos_matvecs"], "restarts", rs, ref.stats["lanczos_restarts"])
    assert mv == ref.stats["lanczos_matvecs"] and rs == ref.stats["lanczos_restarts"]      # measured: 107628 / 9262 on both
    assert np.allclose(x, ref.primal, rtol=0, atol=1e-5 * max(1.0, np.abs(ref.primal).max()))
    assert abs(ref.objval - gold["objval"]) <= 1e-4 * (1 + abs(gold["objval"]))


def _abort_semantics_worker(q):
    """world = 1, native RCCL: an ARGUMENT error raised before the first collective must leave the caller's communicator
    alone (it used to be aborted, and the header told the caller to destroy an already-released handle)"""
    from proxsdp_jl_amd import sharded
    comm = sharded.make_native_comm(None, 0, 1, device_id=0)
    pr = _coupled_model()
    sub, maps = sharded.split_block_diagonal(pr, [0, 0], 0)
    p_, m_ = sub.A.shape[0], sub.G.shape[0]
    coupling = dict(rows=np.array([p_ - 1, p_ + m_ - 1], dtype=np.int64), owned=np.array([1, 1], dtype=np.int32))
    out = {}
    try:
        Optimizer(max_iter=50, max_linsearch_steps=0).optimize(sub, coupling=coupling, nccl_comm=comm)
        out["first"] = "returned"
    except B.ProxSDPHipError as e:
        out["first"] = (e.code, str(e))
    # the communicator is still alive: the same handle serves a valid solve, then is destroyed normally
    sol = Optimizer(max_iter=50).optimize(sub, coupling=coupling, nccl_comm=comm, trace_capacity=50)
    out["second"] = (int(sol.iter), int(sol.stats["rccl_reductions"]))
    B.rccl_comm_destroy(comm)
    out["destroyed"] = True
    q.put(out)


def test_argument_error_before_any_collective_leaves_the_communicator_usable():
    """PROXSDP_E_COMM_ABORTED is returned only when the library really called ncclCommAbort (a failure after its first
    collective was enqueued, or a bounded wait that expired); an invalid option is PROXSDP_E_INVALID and the handle stays
    valid: solve again on it, destroy it."""
    assert B.device_count() > 0
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_abort_semantics_worker, args=(q,))
    p.start()
    out = q.get(timeout=600)
    p.join(timeout=60)
    assert p.exitcode == 0
    assert out["first"][0] == -1 and "max_linsearch_steps" in out["first"][1] and "aborted" not in out["first"][1], out
    assert out["second"][0] == 50 and out["second"][1] >= 100 and out["destroyed"]


def _abort_after_collective_worker(q):
    """world = 1, native RCCL, fault injected in iteration 7: the failure happens AFTER collectives of this solve were enqueued,
    so the library aborts the communicator on the way out and says so"""
    os.environ["PROXSDP_HIP_FAULT_INJECTION"] = "1"
    from proxsdp_jl_amd import sharded
    comm = sharded.make_native_comm(None, 0, 1, device_id=0)
    pr = _coupled_model()
    sub, maps = sharded.split_block_diagonal(pr, [0, 0], 0)
    p_, m_ = sub.A.shape[0], sub.G.shape[0]
    coupling = dict(rows=np.array([p_ - 1, p_ + m_ - 1], dtype=np.int64), owned=np.array([1, 1], dtype=np.int32))
    try:
        Optimizer(max_iter=50, debug_fail_iteration=7).optimize(sub, coupling=coupling, nccl_comm=comm)
        q.put(("returned", ""))
    except B.ProxSDPHipError as e:
        q.put((e.code, str(e)))
    # (no rccl_comm_destroy: PROXSDP_E_COMM_ABORTED means the handle is already released)


def test_failure_after_a_collective_aborts_the_communicator_and_reports_it():
    """the caller must be able to tell that ncclCommAbort ran -- PROXSDP_E_COMM_ABORTED (-6) and a note in the
    error text -- because an aborted communicator is already released and must not be destroyed or reused."""
    assert B.device_count() > 0
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_abort_after_collective_worker, args=(q,))
    p.start()
    code, msg = q.get(timeout=600)
    p.join(timeout=60)
    assert p.exitcode == 0
    assert code == -6, (code, msg)
    assert "injected projection failure" in msg and "aborted" in msg
