"""time-to-tol of the metric's instance (rank-64 knob) under library-only knobs.  gpurun helper.
usage: python tools/gpurun_ttt.py '{"lanczos_cycle_kernel": 1}' '{"lanczos_warm_start": 1}' ..."""
import sys, json
sys.path.insert(0, ".")
from proxsdp_jl_amd import problems
from proxsdp_jl_amd.optimizer import Optimizer
pr = problems.maxcut(4000, seed=0)
for arg in sys.argv[1:] or ["{}"]:
    kw = json.loads(arg)
    o = Optimizer(time_limit=200.0, max_target_rank_krylov_eigs=64, **kw)
    s = o.optimize(pr)
    st = s.stats
    print(json.dumps(dict(kw=kw, status=o.termination_status(), time=s.time, loop=st["loop_time"], iter=int(s.iter),
                          it_per_s=s.iter / st["loop_time"], obj=o.objective_value(), gap=s.gap, matvecs=int(st["lanczos_matvecs"]),
                          restarts=int(st["lanczos_restarts"]), host_eig_s=st["host_eig_time"], cycles=int(st["cycle_launches"]),
                          rank=int(s.final_rank), t_psd=st["t_psd"], t_ls=st["t_linesearch"])))
