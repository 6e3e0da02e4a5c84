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
    o = oracle.Options()
    o.max_iter = 40
    st = oracle.solve(pr, o, capture_iteration=30).state
    bad = dict(st)
    bad["hist"] = st["hist"][:, :100]
    with pytest.raises(ValueError):
        Optimizer(max_iter=50).optimize(pr, resume=bad)
    with pytest.raises(ValueError):                                       # the binding sizes hist from the options
        Optimizer(max_iter=50, convergence_window=100).optimize(pr, resume=st)
    # straight through the C ABI: a state whose hist_len does not match the options is PROXSDP_E_INVALID
    import ctypes as C
    S, arr = B._state_struct(pr.n, pr.A.shape[0] + pr.G.shape[0], 1, 200, state=st)
    S.hist_len = 100
    M = B._Marshalled(pr)
    R = B.Result()
    opt = B.default_options()
    rc = B.lib().proxsdp_hip_solve_ex(C.byref(M.P), C.byref(opt), C.byref(R), C.byref(S), None)
    assert rc == -1 and b"hist_len" in B.lib().proxsdp_hip_last_error()
    sol = Optimizer(max_iter=20).optimize(pr, capture_iteration=30)       # never reached
    assert sol.state is None and sol.iter == 20
    sol = Optimizer(max_iter=30).optimize(pr, capture_iteration=30)       # the last iteration
    assert sol.state is not None and sol.state["iteration"] == 30


# ----------------------------------------------------------------- the metric instance's late windows
def _late():
    return json.load(open(GOLDEN / "trace_maxcut_n4000_late.json"))["windows"]


TRACE_KEYS = ((1, "prim_obj"), (2, "dual_obj"), (3, "gap"), (4, "feas"), (5, "prim_res"), (6, "dual_res"), (7, "primal_step"),
              (8, "beta"), (9, "theta"))


def _compare_window(rows, lt, tol, what):
    """library trace rows (dict iteration -> row) against the oracle's golden rows: same target ranks and linesearch
    trials, the same Lanczos mat-vec count wherever the reference runs KrylovKit (target rank <= 16), values to tol"""
    worst = 0.0
    for t in rows:
        r = lt[t["iter"]]
        assert int(r[10]) == t["target_rank"], (what, t["iter"], "target_rank", r[10], t["target_rank"])
        assert int(r[11]) == t["trials"], (what, t["iter"], "linesearch trials", r[11], t["trials"])
        if t["target_rank"] <= 16:
            assert int(r[13]) == t["matvecs"], (what, t["iter"], "Lanczos mat-vecs", r[13], t["matvecs"])
        for col, key in TRACE_KEYS:
            d = abs(r[col] - t[key]) / max(1.0, abs(t[key]))
            worst = max(worst, d)
            assert d <= tol, (what, t["iter"], key, r[col], t[key])
    return worst


@gpu
def test_metric_instance_into_the_full_eig_regime_against_lapack_in_the_loop():
    """Max-Cut n = 4000, seed 0, reference default options.  The oracle was resumed (tests/golden/make_golden_late_n4000.py)
    from the library's state 12 iterations before the 16 -> 17 rank update and run through it into the implicit
    full_eig! regime with LAPACK dsyevr in the loop (prox_operators.jl:46-59,111-126; pdhg.jl:267-283) -- the regime
    that is 65 % of the default solve's time and that the library serves with its OWN algorithm (positive-part
    Lanczos + per-call certificate).  The library, resumed from the same state, must follow: same
    rank schedule, same linesearch trials, same mat-vec counts while KrylovKit is in charge, trace to 1e-8, and the
    same current_rank at the end of the window; every Lanczos-served full_eig! certified."""
    W = _late()["kU"]
    rows = W["rows"]
    st = expand_state(load_compact_state(GOLDEN / "state_maxcut_n4000_kU.npz"))
    k0, k1 = int(st["iteration"]), rows[-1]["iter"]
    assert W["resumed_from"] == k0 and rows[0]["iter"] == k0 + 1
    ranks = [t["target_rank"] for t in rows]
    assert ranks[0] == 16 and ranks[-1] >= 17 and W["full_eigs"] >= 30          # the window does cross into full_eig!
    pr = P.maxcut(4000, seed=0)
    sol = Optimizer(max_iter=k1).optimize(pr, trace_capacity=k1 - k0, resume=st, capture_iteration=k1)
    assert sol.iter == k1
    worst = _compare_window(rows, _lib_trace(sol), 1e-8, "resumed")
    print("worst relative trace difference over %d iterations: %.2e" % (len(rows), worst))
    assert int(sol.state["current_rank"][0]) == rows[-1]["current_rank"]
    assert int(sol.state["target_rank"][0]) == rows[-1]["target_rank"]
    s = sol.stats
    n_full = sum(1 for t in rows if t["target_rank"] > 16)
    assert s["full_eigs"] == n_full == W["full_eigs"]
    served = s["full_eigs_lanczos"]
    assert served >= n_full - 2, "the library's own engine did not serve the regime"      # (first call: dense engine, by design)
    assert s["full_eigs_lanczos_certified"] >= served and s["full_eigs_lanczos_cert_failed"] == 0


@gpu
def test_metric_instance_steady_window_1000_against_the_oracle():
    """the steady Krylov-phase window of SURVEY section 8d (iterations 1001-1060 from the saved state at 1000):
    identical mat-vec counts in every iteration, trace to 1e-9"""
    W = _late()["k1000"]
    rows = W["rows"]
    st = expand_state(load_compact_state(GOLDEN / "state_maxcut_n4000_k1000.npz"))
    k0, k1 = int(st["iteration"]), rows[-1]["iter"]
    pr = P.maxcut(4000, seed=0)
    sol = Optimizer(max_iter=k1).optimize(pr, trace_capacity=k1 - k0, resume=st)
    worst = _compare_window(rows, _lib_trace(sol), 1e-9, "resumed")
    print("worst relative trace difference over %d iterations: %.2e" % (len(rows), worst))


@gpu
def test_metric_instance_from_iteration_one_reaches_the_saved_states_and_follows_the_oracle_windows():
    """the same two windows WITHOUT the seam on the library's side: one uninterrupted default-options solve from
    iteration 1 passes through the committed states (x, y to 1e-9: the states were written by the round-5 build; a
    later build whose reduction orders differ stays within that) and its trace follows the oracle's windows."""
    L = _late()
    pr = P.maxcut(4000, seed=0)
    kU = L["kU"]["resumed_from"]
    k_end = L["kU"]["rows"][-1]["iter"]
    sol = Optimizer(max_iter=k_end).optimize(pr, trace_capacity=k_end, capture_iteration=kU)
    lt = _lib_trace(sol)
    st = expand_state(load_compact_state(GOLDEN / "state_maxcut_n4000_kU.npz"))
    got = sol.state
    assert got["iteration"] == kU and int(got["target_rank"][0]) == int(st["target_rank"][0]) == 16
    sx = np.abs(st["x"]).max()
    assert np.abs(got["x"] - st["x"]).max() <= 1e-9 * sx
    assert np.abs(got["y"] - st["y"]).max() <= 1e-9 * max(1.0, np.abs(st["y"]).max())
    for k in ("rank_update", "update_cont", "ada_count"):
        assert got[k] == st[k], k
    for tag, tol in (("k1000", 1e-8), ("kU", 1e-7)):
        worst = _compare_window(L[tag]["rows"], lt, tol, tag + " from iteration 1")
        print(tag, "worst relative trace difference: %.2e" % worst)


@gpu
def test_metric_instance_last_iterations_and_the_stop_against_the_oracle():
    """The END of the default-options solve: the library's state 31 iterations before its stop (iteration 8620 of 8651),
    continued by the oracle with LAPACK in the loop until ITS stop rule fires (pdhg.jl:248-253), and by the library from
    the same state: both stop OPTIMAL at the same iteration with the same objective; the rows in between agree to 1e-8."""
    W = _late()["kEnd"]
    rows, fin = W["rows"], W["final"]
    st = expand_state(load_compact_state(GOLDEN / "state_maxcut_n4000_kEnd.npz"))
    k0 = int(st["iteration"])
    assert fin["status"] == 1 and rows[-1]["iter"] == fin["iterations"]
    pr = P.maxcut(4000, seed=0)
    opt = Optimizer()
    sol = opt.optimize(pr, trace_capacity=len(rows) + 50, resume=st)
    print("library: stop at", sol.iter, "objective", opt.objective_value(), "| oracle: stop at", fin["iterations"], "objective", fin["objval"])
    assert sol.status == 1 and sol.iter == fin["iterations"] == 8651
    assert abs(opt.objective_value() - fin["objval"]) <= 1e-9 * abs(fin["objval"])
    assert abs(sol.gap - fin["gap"]) <= 1e-8 and sol.final_rank == fin["final_rank"]
    worst = _compare_window(rows, _lib_trace(sol), 1e-8, "to the stop")
    print("worst relative trace difference over the last %d iterations: %.2e" % (len(rows), worst))
    assert sol.stats["full_eigs_lanczos_cert_failed"] == 0


@gpu
@pytest.mark.parametrize("case", ["mixed_cones", "randsdp_dense", "multi_block_small"])
def test_state_seam_on_other_model_classes(case):
    """the seam beyond single-block Max-Cut: a model with PSD + SOC + free variables in shuffled user order (kat_problems.mixed_cones),
    a dense-A randSDP (proxsdp_problem.M_dense, the dense vector path) and a multi-block SDPLIB model with batched small blocks
    (truss1: six 2 x 2 blocks + a scalar).  Library capture -> library resume: the continuation is bit-identical; library
    capture -> ORACLE resume: the oracle follows the library's trace (1e-8)."""
    from kat_problems import mixed_cones
    if case == "mixed_cones":
        pr, iters, cap = mixed_cones(0), 300, 120
    elif case == "randsdp_dense":
        pr, iters, cap = P.randsdp(40, 30, seed=3, dense=True), 200, 80
    else:
        pr, iters, cap = P.sdplib_blocks(GOLDEN / "sdplib" / "truss1.dat-s"), 300, 150
    full = Optimizer(max_iter=iters).optimize(pr, trace_capacity=iters, capture_iteration=cap)
    st = full.state
    assert st is not None and st["iteration"] == cap and full.iter > cap
    res = Optimizer(max_iter=iters).optimize(pr, trace_capacity=iters, resume=st)
    a, b = _lib_trace(full), _lib_trace(res)
    assert sorted(b) == list(range(cap + 1, full.iter + 1)) and res.iter == full.iter and res.status == full.status
    for k in sorted(b):
        assert np.array_equal(a[k][1:12], b[k][1:12]), (case, k)
    assert np.array_equal(res.primal, full.primal)
    if case == "randsdp_dense":
        return                                   # (the oracle has no dense-A entry: sparse twin below)
    o = oracle.Options()
    o.max_iter = iters
    ora = oracle.solve(pr, o, trace=True, resume=st)
    assert ora.iter == full.iter and ora.status == full.status
    for t in ora.trace:
        r = a[t["iter"]]
        assert int(r[11]) == t["trials"], (case, t["iter"])
        for col, key in ((1, "prim_obj"), (2, "dual_obj"), (3, "gap"), (4, "feas"), (7, "primal_step")):
            assert abs(r[col] - t[key]) <= 1e-8 * max(1.0, abs(t[key])), (case, t["iter"], key, r[col], t[key])


@gpu
def test_the_bench_window_itself_follows_the_oracle():
    """bench.py times the solve pinned at target rank 63 (krylovdim 127) after 200 settle iterations.  The library's state INSIDE
    that window (after iteration 250; tools/gen/gpurun_capture_headline_window.py) was continued by the oracle for 20 iterations
    (tests/golden/make_golden_headline_window.py: 72 s of CPU): the library, resumed from the same state with bench.py's options,
    takes the same 127 mat-vecs in every iteration (KrylovKit's count: one full cycle, no restart), the same linesearch trials,
    and its trace agrees to 1e-9 -- the window the metric is measured on is a parity-checked window."""
    G = json.load(open(GOLDEN / "trace_maxcut_n4000_rank63_window.json"))
    rows = G["rows"]
    st = expand_state(load_compact_state(GOLDEN / "state_maxcut_n4000_rank63_k250.npz"))
    k0, k1 = int(st["iteration"]), rows[-1]["iter"]
    assert k0 == 250 and all(r["matvecs"] == 127 and r["target_rank"] == 63 for r in rows)
    pr = P.maxcut(4000, seed=0)
    sol = Optimizer(initial_target_rank=63, max_target_rank_krylov_eigs=64, max_iter=k1).optimize(pr, trace_capacity=k1 - k0, resume=st)
    lt = _lib_trace(sol)
    worst = 0.0
    for t in rows:
        r = lt[t["iter"]]
        assert int(r[10]) == 63 and int(r[11]) == t["trials"] and int(r[13]) == t["matvecs"] == 127, t["iter"]
        for col, key in TRACE_KEYS:
            d = abs(r[col] - t[key]) / max(1.0, abs(t[key]))
            worst = max(worst, d)
            assert d <= 1e-9, (t["iter"], key, r[col], t[key])
    print("worst relative trace difference over the bench window's %d iterations: %.2e" % (len(rows), worst))
    # the first projection after a resume reads the packed iterate; from the second on the operator form is back
    assert sol.stats["fop_projections"] >= len(rows) - 1
