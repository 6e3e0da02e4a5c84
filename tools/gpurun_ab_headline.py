import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from proxsdp_jl_amd import binding
if sys.argv[1] != "current":
    binding.LIB_PATH = binding.pathlib.Path(os.path.abspath(sys.argv[1]))
from proxsdp_jl_amd import problems as P
from proxsdp_jl_amd.optimizer import Optimizer
pr = P.maxcut(4000, seed=0)
out = []
for it in (300, 600):
    s = Optimizer(max_iter=it, initial_target_rank=63, max_target_rank_krylov_eigs=64).optimize(pr)
    out.append((it, s.stats["loop_time"]))
rate = (out[1][0] - out[0][0]) / (out[1][1] - out[0][1])
s = Optimizer(max_iter=3000).optimize(pr)
print(sys.argv[1][-12:], "rank-63 window (iterations 301-600): %.1f it/s;" % rate, "default options, 3000 iterations: %.3f s" % s.stats["loop_time"], flush=True)
