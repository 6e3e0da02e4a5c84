"""BASELINE config 3 at its CPU-comparable size (randSDP n = 500, m = 1000, seed 0): the first 200 PDHG
iterations by the CPU oracle with reference default options -- trace columns and Lanczos mat-vecs per iteration.
Writes tests/golden/trace_randsdp_n500_m1000.json (asserted by test_randsdp_config3_cpu_comparable_size_matches_oracle_trace)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle
from oracle import Options
from proxsdp_jl_amd import problems as P
pr = P.randsdp(500, 1000, seed=0)
o = Options(); o.max_iter = int(os.environ.get("ITERS", "200"))
mv = []
t0 = time.time()
r = oracle.solve(pr, o, trace=True, proj_callback=lambda it, xi, xo, p_, arc: mv.append(sum(int(a.matvecs) for a in arc)))
rows = [[t["iter"], t["prim_obj"], t["dual_obj"], t["gap"], t["feas"], t["prim_res"], t["dual_res"], t["primal_step"], t["beta"], t["theta"],
         t["target_rank"][0], t["trials"]] for t in r.trace]
per = [mv[0]] + [mv[i] - mv[i - 1] for i in range(1, len(mv))]
json.dump(dict(n=500, m=1000, seed=0, status=r.status, iter=r.iter, rows=rows, matvecs=per, wall_s=time.time() - t0),
          open(os.path.join(ROOT, "tests", "golden", "trace_randsdp_n500_m1000.json"), "w"))
print("randsdp500", r.status, r.iter, r.objval, sum(per), time.time() - t0)
