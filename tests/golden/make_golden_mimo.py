"""BASELINE config 4 solved by the CPU oracle: MIMO n = 512 x 8 blocks, reference default options, tol 1e-4 (~35 s).
Writes tests/golden/solve_mimo_n512_x8.json (asserted by test_config4_mimo_8x512_solved_to_tolerance_against_the_oracle_solve)."""
import sys, time, json
import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import oracle
from oracle import Options
from proxsdp_jl_amd import problems as P
model = P.block_diag_problems([P.mimo(512, seed=s) for s in range(8)], name="mimo-x8")
o = Options(); o.time_limit = 7200.0
t0 = time.time()
r = oracle.solve(model, o, trace=True)
rows = [[t["iter"], t["prim_obj"], t["dual_obj"], t["gap"], t["feas"], t["prim_res"], t["dual_res"], t["primal_step"], t["beta"], t["theta"], t["target_rank"][0], t["trials"]] for t in r.trace]
json.dump(dict(config="MIMO n=512 x 8 blocks (seeds 0..7), reference default options, tol 1e-4", status=r.status, iter=r.iter, objval=r.objval,
               dual_objval=r.dual_objval, gap=r.gap, final_rank=int(r.final_rank), rows=rows, matvecs=int(r.stats["lanczos_matvecs"]),
               wall_s=time.time() - t0), open(os.path.join(ROOT, 'tests', 'golden', 'solve_mimo_n512_x8.json'), 'w'))
print("mimo8", r.status, r.iter, r.objval, r.dual_objval, r.gap, r.stats["lanczos_matvecs"], time.time() - t0)
