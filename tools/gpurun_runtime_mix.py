"""Do torch's bundled HIP runtime and the system runtime behind libproxsdp_hip.so coexist in one process?"""
import sys; sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import numpy as np
order = sys.argv[1]
from proxsdp_jl_amd import problems as P
from proxsdp_jl_amd.optimizer import Optimizer
def mine():
    s = Optimizer(max_iter=50).optimize(P.maxcut(150, seed=0))
    print("  lib solve ok, iter", s.iter, flush=True)
def tor():
    import torch
    print("  torch.cuda.is_available", torch.cuda.is_available(), flush=True)
    t = torch.ones(1000, device="cuda:0", dtype=torch.float64)
    print("  torch sum", float(t.sum()), flush=True)
    return t
if order == "torch_first":
    t = tor(); mine(); print("  again", float((t * 2).sum()))
    pr = P.randsdp_device(40, 30, seed=1)
    s = Optimizer(max_iter=100).optimize(pr)
    print("  dense-on-device solve ok", s.iter, s.objval)
else:
    mine(); tor()
