"""time-to-tol run with the stats that split the Lanczos time (host K x K eigensolves = t_primal)."""
import sys, time; sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from proxsdp_jl_amd import problems as P
from proxsdp_jl_amd.optimizer import Optimizer
pr = P.maxcut(4000, seed=0)
o = Optimizer(tol_gap=1e-4, tol_feasibility=1e-4, max_target_rank_krylov_eigs=64)
t = time.time(); s = o.optimize(pr); dt = time.time() - t
st = s.stats
print("status", s.status, "iter", s.iter, "wall %.2f s loop %.2f s" % (dt, st["loop_time"]))
print("matvecs", st["lanczos_matvecs"], "restarts", st["lanczos_restarts"], "calls", st["lanczos_calls"])
print("host eig total %.3f s = %.1f us per eigensolve (%d eigensolves)" % (
    st["t_primal"], 1e6 * st["t_primal"] / (st["lanczos_restarts"] + st["lanczos_calls"]), st["lanczos_restarts"] + st["lanczos_calls"]))
print("t_psd %.2f t_linesearch %.2f t_residual %.2f" % (st["t_psd"], st["t_linesearch"], st["t_residual"]))
