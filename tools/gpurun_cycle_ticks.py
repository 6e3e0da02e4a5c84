"""Per-phase tick breakdown of the persistent cycle kernel (PROXSDP_HIP_DEBUG_CYCLE) against the step kernels,
Max-Cut n = 4000, Krylov phase (default options) and the rank-31 regime.  gpurun -- python tools/gpurun_cycle_ticks.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PROXSDP_HIP_DEBUG_CYCLE"] = "1"
from proxsdp_jl_amd import problems as P
from proxsdp_jl_amd.optimizer import Optimizer
pr = P.maxcut(4000, seed=0)
for kw in (dict(max_iter=600), dict(max_iter=300, initial_target_rank=12, max_target_rank_krylov_eigs=13),
           dict(max_iter=200, initial_target_rank=31, max_target_rank_krylov_eigs=32)):
    for cyc in (0, 1, int(sys.argv[1]) if len(sys.argv) > 1 else 1):
        s = Optimizer(lanczos_cycle_kernel=cyc, **kw).optimize(pr)
        st = s.stats
        print(kw, "cycle", cyc, "iter", s.iter, "loop_time %.3f" % st["loop_time"], "matvecs", st["lanczos_matvecs"],
              "cycle_steps", st["cycle_steps"], "cycle_launches", st["cycle_launches"],
              "us/matvec %.2f" % (1e6 * st["loop_time"] / max(st["lanczos_matvecs"], 1)), flush=True)
