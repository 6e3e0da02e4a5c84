"""SDPLIB maxG51 / maxG32 with REFERENCE DEFAULT options (VERDICT r3 task 4): where the time goes.
   python tools/gpurun_maxg51_default.py [name] [time_limit] [k=v,...]"""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pathlib import Path
from proxsdp_jl_amd import problems as P
from proxsdp_jl_amd.optimizer import Optimizer
name = sys.argv[1] if len(sys.argv) > 1 else "maxG51"
tl = float(sys.argv[2]) if len(sys.argv) > 2 else 400.0
extra = {}
for kv in filter(None, (sys.argv[3] if len(sys.argv) > 3 else "").split(",")):
    k, v = kv.split("="); extra[k] = float(v)
pr = P.sdplib(Path(__file__).resolve().parent.parent / "tests" / "golden" / "sdplib" / f"{name}.dat-s")
o = Optimizer(time_limit=tl, **extra)
s = o.optimize(pr, trace_capacity=60000)
tr = np.asarray(s.trace)
st = s.stats
out = dict(name=name, options=extra, status=o.termination_status(), iterations=int(s.iter), time_s=s.time, objective=float(s.objval), gap=float(s.gap),
           matvecs=int(st["lanczos_matvecs"]), restarts=int(st["lanczos_restarts"]), host_eig_s=st["host_eig_time"], full_eigs=int(st["full_eigs"]),
           full_eigs_lanczos=int(st["full_eigs_lanczos"]), full_eigs_sign=int(st["full_eigs_sign"]), sign_engine=int(st["sign_engine_projections"]),
           sign_rejected=int(st["sign_engine_rejected"]), fop=int(st["fop_projections"]), t_psd=st["t_psd"], t_linesearch=st["t_linesearch"], loop=st["loop_time"])
sched = []
for r in sorted(set(int(v) for v in tr[:, 10])):
    m = tr[:, 10] == r; idx = np.nonzero(m)[0]
    t_in = float(tr[idx[-1], 12] - (tr[idx[0] - 1, 12] if idx[0] > 0 else 0.0))
    sched.append(dict(target_rank=r, first_iter=int(idx[0]) + 1, iterations=int(m.sum()), wall_s=round(t_in, 3), mv_per_it=round(float(tr[m, 13].mean()), 1), ms_per_it=round(1e3 * t_in / int(m.sum()), 3)))
out["rank_schedule"] = sched
os.makedirs("gpurun_out", exist_ok=True)
tag = name + ("_" + "_".join(f"{k}{v:g}" for k, v in extra.items()) if extra else "")
json.dump(out, open(f"gpurun_out/default_{tag}.json", "w"), indent=1)
print(json.dumps(out, indent=1))
