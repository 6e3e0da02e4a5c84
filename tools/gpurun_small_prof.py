"""Three one-block models in the small-block regime (sides 22, 33, 60), 2000 iterations each, for rocprofv3 --kernel-trace --stats."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from proxsdp_jl_amd import problems as P
from proxsdp_jl_amd.optimizer import Optimizer
for pr in (P.mimo(21, seed=0), P.mimo(32, seed=0), P.maxcut(60, seed=0)):
    s = Optimizer(max_iter=2000, tol_gap=1e-8, tol_feasibility=1e-8, min_iter=10**9).optimize(pr)
    print(pr.name, s.iter, s.stats["loop_time"])
