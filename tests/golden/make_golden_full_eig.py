"""More SDPLIB instances solved to tol 1e-4 by the CPU oracle with full_eig_decomp = true (every projection is LAPACK's
full_eig!): the regime of BASELINE config 5 on maxG11 (n = 800), mcp250-1 and mcp500-1
(gpp124-2 was started too and stopped after 10 min of CPU: the gpp family needs > 100 000 iterations in this regime).
Writes tests/golden/solve_sdplib_full_eig.json (asserted by test_sdplib_full_eig_solves_take_the_oracles_iterations)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle
from oracle import Options
from proxsdp_jl_amd import problems as P
out = {}
path = os.path.join(ROOT, "tests", "golden", "solve_sdplib_full_eig.json")
for name in (sys.argv[1:] or ["mcp250-1", "mcp500-1", "maxG11"]):
    pr = P.sdplib(os.path.join(ROOT, "tests", "golden", "sdplib", name + ".dat-s"))
    o = Options(); o.full_eig_decomp = True; o.time_limit = 3 * 3600.0
    t0 = time.time()
    r = oracle.solve(pr, o, trace=True)
    rows = [[t["iter"], t["prim_obj"], t["dual_obj"], t["gap"], t["feas"], t["prim_res"], t["dual_res"], t["primal_step"], t["beta"], t["theta"],
             t["target_rank"][0], t["trials"]] for t in r.trace]
    out[name] = dict(status=r.status, iter=r.iter, objval=r.objval, dual_objval=r.dual_objval, gap=r.gap, final_rank=int(r.final_rank),
                     rows_every_50=rows[49::50], last_row=rows[-1], wall_s=time.time() - t0)
    print(name, r.status, r.iter, r.objval, time.time() - t0, flush=True)
    json.dump(out, open(path, "w"))
