"""The state seam: `oracle.chambolle_pock(resume=..., capture_iteration=...)` and `proxsdp_hip_solve_ex`
(include/proxsdp_hip.h proxsdp_state) -- the solver state at an iteration boundary can be written out and a solve
continued from it, on both sides, so that LATE windows of long solves (Max-Cut n = 4000 beyond iteration 6500: the
implicit full_eig! regime of /root/reference/src/prox_operators.jl:46-59,111-126) are compared with the oracle
without the oracle having to run the hours before them."""
import json

import numpy as np
import pytest

import oracle
from proxsdp_jl_amd import binding as B
from proxsdp_jl_amd import problems as P
from proxsdp_jl_amd.optimizer import Optimizer

from conftest import GOLDEN
from helpers import compact_state, expand_state, load_compact_state, save_compact_state


def _trace_equal(a, b):
    assert len(a) == len(b), (len(a), len(b))
    for x, y in zip(a, b):
        for k in x:
            assert x[k] == y[k], (x["iter"], k, x[k], y[k])


def test_oracle_resume_reproduces_the_uninterrupted_solve_bit_for_bit():
    """Max-Cut n = 110 with max_target_rank_krylov_eigs = 3: 2 -> 3 after iteration 1024, 3 -> 4 (into full_eig!)
    later; state captured at 1000, resumed: same trace bits, same Result."""
    pr = P.maxcut(110, seed=0)
    o = oracle.Options()
    o.max_target_rank_krylov_eigs = 3
    full = oracle.solve(pr, o, trace=True, capture_iteration=1000)
    st = full.state
    assert st["iteration"] == 1000 and st["target_rank"][0] == 2
    tr = [t["target_rank"][0] for t in full.trace]
    assert max(tr) >= 4 and full.stats["full_eigs"] > 0, "window does not reach the full_eig! regime"
    res = oracle.solve(pr, o, trace=True, resume=st)
    _trace_equal([t for t in full.trace if t["iter"] > 1000], res.trace)
    assert res.iter == full.iter and res.status == full.status and res.objval == full.objval
    assert np.array_equal(res.primal, full.primal) and np.array_equal(res.dual_eq, full.dual_eq)
    # the compact fixture form (eigen-factors of x, sparse M'y) is a faithful container
    c = compact_state(st, pr.psd_sides())
    e = expand_state(c)
    assert np.abs(e["x"] - st["x"]).max() <= 1e-14 * np.abs(st["x"]).max() and np.array_equal(e["Mty"], st["Mty"])
    assert len(c["x_factors"][0][0]) <= 6


def test_compact_state_file_round_trip(tmp_path):
    pr = P.maxcut(40, seed=1)
    o = oracle.Options()
    o.max_iter = 60
    st = oracle.solve(pr, o, capture_iteration=50).state
    c = compact_state(st, pr.psd_sides())
    save_compact_state(tmp_path / "s.npz", c)
    d = load_compact_state(tmp_path / "s.npz")
    e0, e1 = expand_state(c), expand_state(d)
    for k in e0:
        assert np.array_equal(np.asarray(e0[k]), np.asarray(e1[k])), k


# ----------------------------------------------------------------- GPU
gpu = pytest.mark.gpu


def _lib_trace(sol):
    return {int(r[0]): r for r in sol.trace}


@gpu
@pytest.mark.parametrize("support_path", [0, 1])
def test_library_capture_and_resume_continue_the_same_solve(support_path):
    """proxsdp_hip_solve_ex: capture after iteration 700, resume from it -- the continuation follows the uninterrupted
    solve (dense vector passes: bit for bit; support path: the first projection after a resume reads the packed iterate
    instead of the factored one, i.e. rounding-level differences)."""
    pr = P.maxcut(150, seed=0)
    kw = dict(max_iter=1100, support_path=support_path)
    full = Optimizer(**kw).optimize(pr, trace_capacity=1100, capture_iteration=700)
    st = full.state
    assert st is not None and st["iteration"] == 700
    res = Optimizer(**kw).optimize(pr, trace_capacity=1100, resume=st)
    a, b = _lib_trace(full), _lib_trace(res)
    assert sorted(b) == list(range(701, 1101))
    for k in sorted(b):
        if support_path == 0:
            assert np.array_equal(a[k][1:12], b[k][1:12]), k
        else:
            assert np.allclose(a[k][1:10], b[k][1:10], rtol=1e-9, atol=1e-11) and a[k][10] == b[k][10] and a[k][11] == b[k][11], k
    assert res.iter == full.iter == 1100
    assert abs(res.objval - full.objval) <= 1e-9 * abs(full.objval)


@gpu
def test_library_state_continued_by_the_oracle_and_back():
    """Max-Cut n = 110, max_target_rank_krylov_eigs = 3 (rank updates after 1024 and later, then the implicit
    full_eig! regime): the LIBRARY's state at iteration 1000 is continued by the oracle (LAPACK in the loop) and by the
    library itself -- same rank schedule, same linesearch trials, traces to 1e-8; and the ORACLE's state at 1000 is
    continued by the library the same way."""
    pr = P.maxcut(110, seed=0)
    kw = dict(max_target_rank_krylov_eigs=3)
    o = oracle.Options()
    o.max_target_rank_krylov_eigs = 3
    lib_full = Optimizer(**kw).optimize(pr, trace_capacity=4000, capture_iteration=1000)
    ora_full = oracle.solve(pr, o, trace=True, capture_iteration=1000)
    assert lib_full.iter == ora_full.iter
    for state, who in ((lib_full.state, "library"), (ora_full.state, "oracle")):
        ora = oracle.solve(pr, o, trace=True, resume=state)
        lib = Optimizer(**kw).optimize(pr, trace_capacity=4000, resume=state)
        assert lib.iter == ora.iter == lib_full.iter, who
        lt = _lib_trace(lib)
        for t in ora.trace:
            r = lt[t["iter"]]
            assert r[10] == t["target_rank"][0] and r[11] == t["trials"], (who, t["iter"])
            for col, key in ((1, "prim_obj"), (2, "dual_obj"), (3, "gap"), (4, "feas"), (7, "primal_step"), (8, "beta")):
                assert abs(r[col] - t[key]) <= 1e-8 * max(1.0, abs(t[key])), (who, t["iter"], key, r[col], t[key])
        assert abs(lib.objval - ora.objval) <= 1e-8 * abs(ora.objval)
        assert lib.stats["full_eigs"] > 0
    # the two states themselves agree (same algorithm, different arithmetic order)
    a, b = lib_full.state, ora_full.state
    assert np.abs(a["x"] - b["x"]).max() <= 1e-8 * np.abs(b["x"]).max()
    assert np.abs(a["y"] - b["y"]).max() <= 1e-8 * max(1.0, np.abs(b["y"]).max())
    for k in ("rank_update", "update_cont", "ada_count"):
        assert a[k] == b[k], k
    assert np.allclose(a["hist"], b["hist"], rtol=1e-7, atol=1e-10)


@gpu
def test_state_seam_argument_errors():
    pr = P.maxcut(40, seed=1)
    st = oracle.solve(pr, oracle.Options(), capture_iteration=30).state if False else None
    o = oracle.Options()
    o.max_iter = 40
    st = oracle.solve(pr, o, capture_iteration=30).state
    bad = dict(st)
    bad["hist"] = st["hist"][:, :100]
    with pytest.raises(ValueError):
        Optimizer(max_iter=50).optimize(pr, resume=bad)
    with pytest.raises(B.ProxSDPHipError, match="hist_len"):
        Optimizer(max_iter=50, convergence_window=100).optimize(pr, resume=st)
    sol = Optimizer(max_iter=20).optimize(pr, capture_iteration=30)       # never reached
    assert sol.state is None and sol.iter == 20
    sol = Optimizer(max_iter=30).optimize(pr, capture_iteration=30)       # the last iteration
    assert sol.state is not None and sol.state["iteration"] == 30
