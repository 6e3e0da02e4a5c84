"""SDPLIB maxG51 with REFERENCE DEFAULT options (Krylov path): the CPU oracle's first 4150 PDHG iterations (12 min of CPU) --
Lanczos mat-vecs of every iteration, the rank schedule and every 25th trace row.  At iteration 4013 the target rank goes
8 -> 9 and from then on EVERY projection runs KrylovKit's 100 restarts (krylovdim = max(2 nev + 1, 25) = 25, 719 mat-vecs, 8 of
9 pairs converged): the reference's algorithm at its own defaults, which is why this instance does not reach tol 1e-4 in
minutes on any engine.  Writes tests/golden/trace_maxG51_default.json
(asserted by test_maxG51_default_options_follows_the_oracle_into_the_100_restart_regime)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle
from oracle import Options
from proxsdp_jl_amd import problems as P
pr = P.sdplib(os.path.join(ROOT, "tests", "golden", "sdplib", "maxG51.dat-s"))
o = Options(); o.max_iter = int(os.environ.get("ITERS", "4150"))
mv = []
t0 = time.time()
r = oracle.solve(pr, o, trace=True, proj_callback=lambda it, xi, xo, p_, arc: mv.append(sum(int(a.matvecs) for a in arc)))
rows = [[t["iter"], t["prim_obj"], t["dual_obj"], t["gap"], t["feas"], t["prim_res"], t["dual_res"], t["primal_step"], t["beta"], t["theta"],
         t["target_rank"][0], t["trials"]] for t in r.trace]
per = [mv[0]] + [mv[i] - mv[i - 1] for i in range(1, len(mv))]
sched = []
for t in rows:
    if not sched or sched[-1][1] != t[10]:
        sched.append([t[0], t[10]])
json.dump(dict(iter=r.iter, status=r.status, matvecs=per, rank_schedule=sched, rows_every_25=rows[24::25], wall_s=time.time() - t0),
          open(os.path.join(ROOT, "tests", "golden", "trace_maxG51_default.json"), "w"))
print("maxG51 default", r.status, r.iter, sum(per), time.time() - t0)
