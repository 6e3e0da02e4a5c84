"""default-options solve + headline window on a given build: python tools/gpurun_t2t_ab.py <lib.so> <tag>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from proxsdp_jl_amd import binding
binding.LIB_PATH = binding.pathlib.Path(os.path.abspath(sys.argv[1]))
from proxsdp_jl_amd import problems as P
from proxsdp_jl_amd.optimizer import Optimizer
pr = P.maxcut(4000, seed=0)
s = Optimizer(max_iter=260, initial_target_rank=63, max_target_rank_krylov_eigs=64).optimize(pr, trace_capacity=260)
tr = s.trace
print(sys.argv[2], "headline it/s 201-260: %.1f" % (60 / (tr[259, 12] - tr[199, 12])), "obj %.12f" % tr[259, 1], flush=True)
o = Optimizer(time_limit=200.0)
s = o.optimize(pr)
print(sys.argv[2], "t2t", o.termination_status(), int(s.iter), "%.3f s" % s.time, "obj %.12f" % s.objval, int(s.stats["lanczos_matvecs"]), flush=True)
