"""SDPLIB gpp500-1 and mcp500-1 (side 501 / 500: BASELINE config 5's second instance and its Max-Cut sibling) solved to tol 1e-4 by
the CPU oracle with reference default options (Krylov path; minutes of CPU each).  Writes tests/golden/solve_sdplib500.json
(asserted by test_sdplib_500_instances_solved_to_tolerance_against_the_oracle_solves)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle
from oracle import Options
from proxsdp_jl_amd import problems as P
out = {}
for name in (sys.argv[1:] or ["gpp500-1", "mcp500-1"]):
    pr = P.sdplib(os.path.join(ROOT, "tests", "golden", "sdplib", name + ".dat-s"))
    o = Options(); o.time_limit = 4 * 3600.0
    t0 = time.time()
    r = oracle.solve(pr, o, trace=True)
    sched = []
    for t in r.trace:
        if not sched or sched[-1][1] != t["target_rank"][0]:
            sched.append([t["iter"], t["target_rank"][0]])
    out[name] = dict(status=r.status, iter=r.iter, objval=r.objval, dual_objval=r.dual_objval, gap=r.gap, final_rank=int(r.final_rank),
                     matvecs=int(r.stats["lanczos_matvecs"]), rank_schedule=sched, wall_s=time.time() - t0)
    print(name, out[name], flush=True)
    json.dump(out, open(os.path.join(ROOT, "tests", "golden", "solve_sdplib500.json"), "w"))
