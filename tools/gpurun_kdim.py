"""Round 3: Krylov dimension of the Lanczos-served full_eig! (options.full_eig_lanczos_kdim10) now that the K x K
eigensolve costs a third (host_eig_merge): default-options solve of the metric instance."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from proxsdp_jl_amd import problems as P
from proxsdp_jl_amd.optimizer import Optimizer
pr = P.maxcut(4000, seed=0)
out = {}
for kd in (20, 25, 30, 40, 50):
    o = Optimizer(time_limit=200.0, full_eig_lanczos_kdim10=kd)
    s = o.optimize(pr)
    out[kd] = dict(status=s.status, iter=int(s.iter), obj=s.objval, time=s.time, matvecs=int(s.stats["lanczos_matvecs"]),
                   restarts=int(s.stats["lanczos_restarts"]), host_eig_s=s.stats["host_eig_time"], by_lanczos=int(s.stats["full_eigs_lanczos"]),
                   checks=int(s.stats["full_eigs_lanczos_checks"]), full_eigs=int(s.stats["full_eigs"]))
    print(kd, out[kd], flush=True)
json.dump(out, open("gpurun_out/kdim_r3.json", "w"), indent=1)
