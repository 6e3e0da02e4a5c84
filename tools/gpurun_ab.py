"""A/B of two builds of the library on the same solves (traces to files): python tools/gpurun_ab.py <lib.so> <tag>"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from proxsdp_jl_amd import binding
binding.LIB_PATH = binding.pathlib.Path(os.path.abspath(sys.argv[1]))
from proxsdp_jl_amd import problems as P
from proxsdp_jl_amd.optimizer import Optimizer
tag = sys.argv[2]
os.makedirs("gpurun_out/ab", exist_ok=True)
g = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "sdplib")
cases = {"maxcut2000": (P.maxcut(2000, seed=0), dict(max_iter=2700)),
         "mcp500-1": (P.sdplib(os.path.join(g, "mcp500-1.dat-s")), dict(max_iter=2500)),
         "maxcut4000r63": (P.maxcut(4000, seed=0), dict(max_iter=60, initial_target_rank=63, max_target_rank_krylov_eigs=64))}
for name, (pr, kw) in cases.items():
    s = Optimizer(**kw).optimize(pr, trace_capacity=kw["max_iter"])
    np.save(f"gpurun_out/ab/{tag}_{name}.npy", np.asarray(s.trace))
    print(tag, name, s.iter, s.objval, int(s.stats["lanczos_matvecs"]), flush=True)
