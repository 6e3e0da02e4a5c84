"""SDPLIB mcp500-1 / gpp500-1 with reference default options: the CPU oracle's first ITERS (default 400) PDHG iterations --
trace columns and Lanczos mat-vecs per iteration -- so that the GPU test can say WHERE the library's trajectory leaves the
oracle's instead of only comparing end states of two chaotic 5000-iteration solves.
Writes tests/golden/trace_sdplib500.json (asserted by test_sdplib_500_instances_follow_the_oracle_trace)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle
from oracle import Options
from proxsdp_jl_amd import problems as P
dst = os.path.join(ROOT, "tests", "golden", "trace_sdplib500.json")
out = json.load(open(dst)) if os.path.exists(dst) else {}
for name in (sys.argv[1:] or ["mcp500-1", "gpp500-1"]):
    pr = P.sdplib(os.path.join(ROOT, "tests", "golden", "sdplib", name + ".dat-s"))
    o = Options(); o.max_iter = int(os.environ.get("ITERS", "400"))
    mv = []
    t0 = time.time()
    r = oracle.solve(pr, o, trace=True, proj_callback=lambda it, xi, xo, p_, arc: mv.append(sum(int(a.matvecs) for a in arc)))
    rows = [[t["iter"], t["prim_obj"], t["dual_obj"], t["gap"], t["feas"], t["prim_res"], t["dual_res"], t["primal_step"], t["beta"], t["theta"],
             t["target_rank"][0], t["trials"]] for t in r.trace]
    per = [mv[0]] + [mv[i] - mv[i - 1] for i in range(1, len(mv))]
    out[name] = dict(status=r.status, iter=r.iter, rows=rows, matvecs=per, wall_s=time.time() - t0)
    print(name, r.status, r.iter, sum(per), time.time() - t0, flush=True)
    json.dump(out, open(dst, "w"))
