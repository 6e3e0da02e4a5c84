# torch-free driver for PMC passes: one short dense-path solve (k_primal_update = calibration
# kernel with known traffic: reads 3*8*Nx, writes 8*Nx with 8-byte-per-lane accesses) and
# isolated symv launches.
import sys; sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import numpy as np
from proxsdp_jl_amd import binding as B, problems as P
from proxsdp_jl_amd.optimizer import Optimizer
n = 4000
pr = P.maxcut(n, seed=0)
o = Optimizer(max_iter=6, support_path=0)
s = o.optimize(pr)
print("solve ok", s.iter, s.stats["lanczos_matvecs"])
