"""Generates the two LARGE-instance fixtures from the CPU oracle (minutes of CPU each; run once
in the build container, results committed as numbers only):

  trace_maxcut_n4000.json   first 30 PDHG iterations of the metric's instance (Max-Cut ER n=4000,
                            seed 0, reference default options): the per-iteration trace columns
  solve_maxcut_n1000.json   BASELINE config 2 (Max-Cut ER n=1000, seed 0) solved to
                            tol_gap = tol_feasibility = 1e-4 with reference default options:
                            status, iterations, objective, dual objective, gap, rank schedule

  trace_maxcut_n4000_rank63.json  the HEADLINE regime of bench.py: the same instance started at target rank 63
                            (initial_target_rank = 63, max_target_rank_krylov_eigs = 64: krylovdim 127), first
                            TRACE4000R63_ITERS (12) iterations: trace columns + Lanczos mat-vecs per iteration
  solve_maxcut_n2000.json   Max-Cut ER n=2000, seed 0, solved to tol 1e-4 with reference default options
                            (hours of CPU: past target rank 16 every iteration is a LAPACK full_eig!)

  solve_maxG51_full_eig.json  BASELINE config 5: SDPLIB maxG51 with full_eig_decomp = true solved to tol 1e-4 (round 3)

Run from the repo root:  python tests/golden/make_golden_large.py [trace4000] [solve1000] [trace4000r63] [solve2000] [solvemaxg51]
The inputs are regenerated from the seed by the tests through the same generator
(proxsdp_jl_amd.problems.maxcut)."""
import json
import os
import pathlib
import sys
import time

ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import oracle  # noqa: E402
from oracle import Options  # noqa: E402
from proxsdp_jl_amd import problems as P  # noqa: E402

OUT = pathlib.Path(__file__).resolve().parent


def rows_of(r):
    return [[t["iter"], t["prim_obj"], t["dual_obj"], t["gap"], t["feas"], t["prim_res"], t["dual_res"],
             t["primal_step"], t["beta"], t["theta"], t["target_rank"][0], t["trials"]] for t in r.trace]


def trace4000():
    pr = P.maxcut(4000, seed=0)
    o = Options()
    o.max_iter = int(os.environ.get("TRACE4000_ITERS", "30"))
    mv = []

    def cb(it, xin, xout, p, arc):
        mv.append(int(arc[0].matvecs))
    t0 = time.time()
    r = oracle.solve(pr, o, trace=True, proj_callback=cb)
    per_iter = [mv[0]] + [mv[i] - mv[i - 1] for i in range(1, len(mv))]
    (OUT / "trace_maxcut_n4000.json").write_text(json.dumps(dict(
        n=4000, seed=0, status=r.status, iter=r.iter, objval=r.objval, rows=rows_of(r), matvecs=per_iter,
        wall_s=time.time() - t0)))
    print("trace4000", r.status, r.iter, r.objval, per_iter, time.time() - t0)


def solve1000():
    pr = P.maxcut(1000, seed=0)
    o = Options()
    o.time_limit = 36000.0
    t0 = time.time()
    r = oracle.solve(pr, o, trace=True)
    sched = []
    for t in r.trace:                       # (iteration, target_rank) at every change
        if not sched or sched[-1][1] != t["target_rank"][0]:
            sched.append([t["iter"], t["target_rank"][0]])
    (OUT / "solve_maxcut_n1000.json").write_text(json.dumps(dict(
        n=1000, seed=0, tol=1e-4, status=r.status, iter=r.iter, objval=r.objval, dual_objval=r.dual_objval,
        gap=r.gap, final_rank=int(r.final_rank), full_eigs=int(r.stats["full_eigs"]),
        rank_schedule=sched, wall_s=time.time() - t0)))
    print("solve1000", r.status, r.iter, r.objval, r.dual_objval, r.gap, time.time() - t0)


def trace4000r63():
    pr = P.maxcut(4000, seed=0)
    o = Options()
    o.max_iter = int(os.environ.get("TRACE4000R63_ITERS", "12"))
    o.initial_target_rank = 63
    o.max_target_rank_krylov_eigs = 64
    mv, rs = [], []

    def cb(it, xin, xout, p, arc):
        mv.append(int(arc[0].matvecs))
        rs.append(int(getattr(arc[0], "restarts", 0)))
    t0 = time.time()
    r = oracle.solve(pr, o, trace=True, proj_callback=cb)
    per_iter = [mv[0]] + [mv[i] - mv[i - 1] for i in range(1, len(mv))]
    (OUT / "trace_maxcut_n4000_rank63.json").write_text(json.dumps(dict(
        n=4000, seed=0, initial_target_rank=63, max_target_rank_krylov_eigs=64, status=r.status, iter=r.iter,
        objval=r.objval, rows=rows_of(r), current_rank=[t["current_rank"][0] for t in r.trace], matvecs=per_iter,
        wall_s=time.time() - t0)))
    print("trace4000r63", r.status, r.iter, r.objval, per_iter, time.time() - t0)


def solve2000():
    pr = P.maxcut(2000, seed=0)
    o = Options()
    o.time_limit = 12 * 3600.0
    t0 = time.time()
    r = oracle.solve(pr, o, trace=True)
    sched = []
    for t in r.trace:
        if not sched or sched[-1][1] != t["target_rank"][0]:
            sched.append([t["iter"], t["target_rank"][0]])
    (OUT / "solve_maxcut_n2000.json").write_text(json.dumps(dict(
        n=2000, seed=0, tol=1e-4, status=r.status, iter=r.iter, objval=r.objval, dual_objval=r.dual_objval,
        gap=r.gap, final_rank=int(r.final_rank), full_eigs=int(r.stats["full_eigs"]),
        rank_schedule=sched, wall_s=time.time() - t0)))
    print("solve2000", r.status, r.iter, r.objval, r.dual_objval, r.gap, time.time() - t0)


def solve_maxg51():
    """BASELINE config 5 by the oracle: SDPLIB maxG51 (n = 1000) with full_eig_decomp = true (every projection is LAPACK's
    full_eig!), tol 1e-4: status, iterations, objective, dual objective, gap, final rank + every 50th trace row."""
    pr = P.sdplib(OUT / "sdplib" / "maxG51.dat-s")
    o = Options()
    o.full_eig_decomp = True
    o.time_limit = 6 * 3600.0
    t0 = time.time()
    r = oracle.solve(pr, o, trace=True)
    rows = rows_of(r)
    (OUT / "solve_maxG51_full_eig.json").write_text(json.dumps(dict(
        instance="maxG51", full_eig_decomp=True, tol=1e-4, status=r.status, iter=r.iter, objval=r.objval,
        dual_objval=r.dual_objval, gap=r.gap, final_rank=int(r.final_rank), full_eigs=int(r.stats["full_eigs"]),
        rows_every_50=rows[49::50], last_row=rows[-1], wall_s=time.time() - t0)))
    print("solve_maxg51", r.status, r.iter, r.objval, r.dual_objval, r.gap, time.time() - t0)


if __name__ == "__main__":
    which = sys.argv[1:] or ["trace4000", "solve1000"]
    if "solvemaxg51" in which:
        solve_maxg51()
    if "trace4000" in which:
        trace4000()
    if "solve1000" in which:
        solve1000()
    if "trace4000r63" in which:
        trace4000r63()
    if "solve2000" in which:
        solve2000()
