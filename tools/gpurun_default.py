"""Time-to-tol of the metric's instance with REFERENCE DEFAULT options (full_eig! regime after
target_rank 17) with and without full_eig_lanczos.  gpurun helper."""
import sys, time, json
sys.path.insert(0, ".")
from proxsdp_jl_amd import problems
from proxsdp_jl_amd.optimizer import Optimizer
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
pr = problems.maxcut(n, seed=0)
for kw in (dict(), dict(max_target_rank_krylov_eigs=64)):
    o = Optimizer(time_limit=200.0, **kw)
    s = o.optimize(pr)
    st = s.stats
    print(json.dumps(dict(kw=kw, status=o.termination_status(), time=s.time, iter=int(s.iter), obj=o.objective_value(), gap=s.gap,
                          full_eigs=int(st["full_eigs"]), by_lanczos=int(st["full_eigs_lanczos"]), matvecs=int(st["lanczos_matvecs"]),
                          restarts=int(st["lanczos_restarts"]), host_eig_s=st["host_eig_time"], rank=int(s.final_rank))))
