"""Generates the committed golden fixtures from the CPU oracle (oracle/), which is
itself pinned by the reference's known-answer tests (tests/test_oracle_kat.py).
Run from the repo root:  python tests/golden/make_golden.py

Fixtures (SURVEY.md section 8c "golden vectors"):
  psd_projection.npz   packed in -> packed out, rank, min_eig for planted spectra,
                       n in {3, 7, 101, 257}, Lanczos and full-eig paths
  traces.json          per-iteration traces (first iterations) of Max-Cut README n=4,
                       SDPLIB mcp124-1 and Max-Cut ER n=200
  kat_results.json     final Result of every reference KAT
The fixtures contain numbers only (inputs are regenerated from seeds by the
tests through the same generators)."""
import json
import math
import pathlib
import sys

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import oracle  # noqa: E402
from oracle import Options, eig as oeig, pdhg as opdhg  # noqa: E402
from proxsdp_jl_amd import problems as P  # noqa: E402
from kat_problems import KATS  # noqa: E402

OUT = pathlib.Path(__file__).resolve().parent


from helpers import planted_packed, oracle_project, PROJ_CASES, smat  # noqa: E402


def main():
    proj = {}
    for name, n, seed, top, tr, full in PROJ_CASES:
        x = planted_packed(n, seed, top)
        y, rank, mineig, arc = oracle_project(x, n, tr, full)
        proj[name + "__out"] = y
        proj[name + "__meta"] = np.array([n, seed, tr, int(full), rank, mineig, arc.matvecs, arc.converged_eigs])
        print(name, "rank", rank, "min_eig", mineig, "matvecs", arc.matvecs)
    np.savez_compressed(OUT / "psd_projection.npz", **proj)

    traces = {}
    cases = [("maxcut_readme_n4", P.maxcut_readme(), 200),
             ("sdplib_mcp124-1", P.sdplib(OUT / "sdplib" / "mcp124-1.dat-s"), 120),
             ("maxcut_er_n200_s0", P.maxcut(200, seed=0), 120)]
    for name, pr, iters in cases:
        o = Options()
        o.max_iter = iters
        # iterations whose projection input has lambda_r == lambda_{r+1} (to 1e-8 |X|): there the
        # reference's rank-r truncation (prox_operators.jl:99-106) is defined only up to a rotation
        # inside the eigenspace, so two correct eigensolvers may continue on different trajectories
        n_side = pr.psd_sides()[0]
        degenerate, restarts, prev = [], [], [0]

        def cb(it, xin, xout, p, arc_list, n_side=n_side, degenerate=degenerate, restarts=restarts, prev=prev):
            if n_side < 2:
                return
            rs = arc_list[0].restarts - prev[0]
            prev[0] = arc_list[0].restarts
            if rs > 0:
                restarts.append(it)
            tr = int(p.target_rank[0])
            if tr < n_side:
                X = smat(xin[:n_side * (n_side + 1) // 2], n_side)
                w = np.linalg.eigvalsh(X)[::-1]
                if w[tr - 1] > 0.0 and abs(w[tr - 1] - w[tr]) <= 1e-8 * np.linalg.norm(X):
                    degenerate.append(it)
        r = oracle.solve(pr, o, trace=True, proj_callback=cb)
        print(name, "restart iterations", restarts[:10], "degenerate iterations", degenerate[:10])
        traces[name] = dict(status=r.status, iter=r.iter, objval=r.objval, restart_iters=restarts,
                            degenerate_iters=degenerate, rows=[
            [t["iter"], t["prim_obj"], t["dual_obj"], t["gap"], t["feas"], t["prim_res"], t["dual_res"],
             t["primal_step"], t["beta"], t["theta"], t["target_rank"][0], t["trials"]] for t in r.trace])
        print(name, r.status, r.iter, r.objval)
    (OUT / "traces.json").write_text(json.dumps(traces))

    kat = {}
    for name, (build, expected, atol, xexp) in KATS.items():
        o = Options()
        o.tol_gap = o.tol_feasibility = 1e-6
        r = oracle.solve(build(), o)
        kat[name] = dict(status=r.status, objval=r.objval, dual_objval=r.dual_objval, iter=r.iter,
                         final_rank=r.final_rank, gap=r.gap, primal=list(map(float, r.primal)),
                         dual_eq=list(map(float, r.dual_eq)), dual_in=list(map(float, r.dual_in)),
                         slack_eq=list(map(float, r.slack_eq)), slack_in=list(map(float, r.slack_in)),
                         dual_cone=list(map(float, r.dual_cone)),
                         primal_feasible=bool(r.primal_feasible_user_tol),
                         dual_feasible=bool(r.dual_feasible_user_tol), expected=expected)
        print(name, r.status, r.objval, r.iter)
    (OUT / "kat_results.json").write_text(json.dumps(kat, indent=1))


if __name__ == "__main__":
    main()
