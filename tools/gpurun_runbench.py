"""The reference's own benchmark script (test/runbench.jl: RANDSDP 5 x 5, SENSORLOC 100..400, SDPLIB gpp / mcp 124..500, MIMO 100 /
500 / 1000; reference default options, time limit 300 s) through the library, one row per instance in the columns of its log
(class, prob_ref, time, obj, rank, lin_feas, sdp_feas) plus status and iterations.  Data the reference draws from Julia's RNG is
drawn from NumPy (own seeds); the SDPLIB files are the reference's.  gpurun -- python tools/gpurun_runbench.py [out.md]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from proxsdp_jl_amd import problems as P
from proxsdp_jl_amd.optimizer import Optimizer, TERMINATION

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
data = os.path.join(root, "tests", "golden", "sdplib_all")
if not os.path.isdir(data):
    data = os.path.join(root, "tests", "golden", "sdplib")
jobs = [("RANDSDP", "1", lambda: P.randsdp(5, 5, seed=1))]
jobs += [("SENSORLOC", str(n), (lambda n=n: P.sensorloc(n, seed=0))) for n in (100, 200, 300, 400)]
for fam, cls in (("gpp", "SDPLIB_gp"), ("mcp", "SDPLIB_mc")):
    for size in ("124", "250", "500"):
        for k in "1234":
            f = os.path.join(data, f"{fam}{size}-{k}.dat-s")
            if os.path.exists(f):
                jobs.append((cls, os.path.basename(f), (lambda f=f: P.sdplib(f))))
jobs += [("MIMO", str(n), (lambda n=n: P.mimo(n, seed=0))) for n in (100, 500, 1000)]
lines = ["| class | prob_ref | status | iterations | time s | obj | rank | lin_feas | sdp_feas (lambda_min) |", "|---|---|---|---|---|---|---|---|---|"]
Optimizer(max_iter=5).optimize(P.maxcut(120, seed=0))          # (first-call set-up of the process, outside every row)
Optimizer(max_iter=5).optimize(P.randsdp(5, 5, seed=7))         # (... incl. rocSOLVER's small-size code objects: the exit path's first dsyevd
                                                                #  of a block of side 5 took 7.3 s on a fresh box, profiles/r06_runbench.md)
Optimizer(max_iter=3, full_eig_decomp=1, full_eig_sign=0).optimize(P.maxcut(500, seed=0))   # (... and its dsyevd at side 500: the exit path of
                                                                # mcp500-1 falls back to it -- 3 to 17 s of library loading on a fresh box,
                                                                # counted in that row's time in the first calls of round 6)
t_all = time.time()
for cls, ref, build in jobs:
    pr = build()
    s = Optimizer(time_limit=300.0).optimize(pr)
    x = s.primal
    lin = 0.0
    if pr.p: lin = max(lin, float(np.abs(pr.A @ x - pr.b).max()))
    if pr.m: lin = max(lin, float(np.maximum(pr.G @ x - pr.h, 0.0).max()))
    sdp = min(float(np.linalg.eigvalsh(P.unpack_psd(x[v], side)).min()) for v, side in zip(pr.psd, pr.psd_sides()))
    obj = pr.user_objective(s.objval) if hasattr(pr, "user_objective") else s.objval
    line = f"| {cls} | {ref} | {TERMINATION[s.status]} | {s.iter} | {s.time:.2f} | {s.objval:.6g} | {s.final_rank} | {lin:.2e} | {sdp:.2e} |"
    print(line, flush=True)
    lines.append(line)
lines.append("")
lines.append("Sum of the solve times: %.1f s (wall clock of the loop incl. model building: %.1f s)" % (sum(float(l.split("|")[5]) for l in lines[2:-1]), time.time() - t_all))
print(lines[-1])
out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "gpurun_out", "r06_runbench.md")
os.makedirs(os.path.dirname(out), exist_ok=True)
open(out, "w").write("\n".join(lines) + "\n")
