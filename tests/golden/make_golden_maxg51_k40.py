"""SDPLIB maxG51 on the Krylov path with ONE reference option changed -- eigsolver_min_lanczos = 40 (options.jl: 25) -- solved to
tol 1e-4 by the CPU oracle (~25 min of CPU).  With the default 25 every projection from iteration 4013 on runs into KrylovKit's
100-restart limit (make_golden_maxg51_default.py); a Krylov space of 40 converges them, and the solve ends OPTIMAL.
Writes tests/golden/solve_maxG51_krylov40.json (asserted by test_maxG51_krylov_path_with_min_lanczos_40_takes_the_oracles_iterations)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle
from oracle import Options
from proxsdp_jl_amd import problems as P
pr = P.sdplib(os.path.join(ROOT, "tests", "golden", "sdplib", "maxG51.dat-s"))
o = Options(); o.eigsolver_min_lanczos = 40; o.time_limit = 4 * 3600.0
mv = []
t0 = time.time()
r = oracle.solve(pr, o, trace=True, proj_callback=lambda it, xi, xo, p_, arc: mv.append(sum(int(a.matvecs) for a in arc)))
sched = []
for t in r.trace:
    if not sched or sched[-1][1] != t["target_rank"][0]:
        sched.append([t["iter"], t["target_rank"][0]])
rows = [[t["iter"], t["prim_obj"], t["dual_obj"], t["gap"], t["feas"], t["prim_res"], t["dual_res"], t["primal_step"], t["beta"], t["theta"],
         t["target_rank"][0], t["trials"]] for t in r.trace]
json.dump(dict(status=r.status, iter=r.iter, objval=r.objval, dual_objval=r.dual_objval, gap=r.gap, final_rank=int(r.final_rank),
               matvecs=int(mv[-1]), rank_schedule=sched, rows_every_50=rows[49::50], wall_s=time.time() - t0),
          open(os.path.join(ROOT, "tests", "golden", "solve_maxG51_krylov40.json"), "w"))
print("maxG51 k40", r.status, r.iter, r.objval, mv[-1], time.time() - t0)
