"""MIMO n = 512 x 8 on two builds of the library in the same session: python tools/gpurun_mimo_ab.py <lib.so> <tag>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from proxsdp_jl_amd import binding
binding.LIB_PATH = binding.pathlib.Path(os.path.abspath(sys.argv[1]))
from proxsdp_jl_amd import problems as P
from proxsdp_jl_amd.optimizer import Optimizer
model = P.block_diag_problems([P.mimo(512, seed=s) for s in range(8)], name="mimo-x8")
for rep in range(3):
    s = Optimizer(max_iter=60).optimize(model, trace_capacity=60)
    tr = s.trace
    print(sys.argv[2], "it/s over iterations 11-60: %.1f" % (50 / (tr[59, 12] - tr[9, 12])), "obj %.12e" % tr[59, 1], int(s.stats["lanczos_matvecs"]), flush=True)
