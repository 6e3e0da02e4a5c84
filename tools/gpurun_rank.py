"""Max-Cut n=4000 started at target rank sqrt(n) (220 iterations): the window bench.py reports as rank_sqrt_n."""
import sys; sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from proxsdp_jl_amd import problems as P
from proxsdp_jl_amd.optimizer import Optimizer
r0 = int(sys.argv[1]) if len(sys.argv) > 1 else 63
it = int(sys.argv[2]) if len(sys.argv) > 2 else 120
s = Optimizer(max_iter=it, initial_target_rank=r0, max_target_rank_krylov_eigs=64).optimize(P.maxcut(4000, seed=0), trace_capacity=it)
st = s.stats
print("iters", s.iter, "matvecs/iter", st["lanczos_matvecs"] / s.iter, "restarts/iter", st["lanczos_restarts"] / s.iter,
      "host eig ms/iter", 1e3 * st["t_primal"] / s.iter, "loop ms/iter", 1e3 * st["loop_time"] / s.iter)
