"""Shape sweep of the dense-A kernels at the BASELINE randSDP size (PROXSDP_DMV / PROXSDP_DMT)."""
import os, sys; sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import torch
torch.cuda.init()
from proxsdp_jl_amd import problems as P
from proxsdp_jl_amd.optimizer import Optimizer
pr = P.randsdp_device(2000, 4000, seed=0)
def run(dmv, dmt):
    os.environ["PROXSDP_DMV"] = str(dmv); os.environ["PROXSDP_DMT"] = str(dmt)
    s = Optimizer(max_iter=14).optimize(pr)
    st = s.stats
    print("DMV %d DMT %d: %.3f ms/pass over %d passes  (%.0f GB/s)  obj %.9e" % (
        dmv, dmt, st["dense_ms"] / st["dense_passes"], st["dense_passes"],
        8.0 * 4000 * 2001000 / (st["dense_ms"] / st["dense_passes"] * 1e-3) / 1e9, s.objval), flush=True)
run(0, 0)  # (the sweep switches were removed after tuning; see pdhg_loop.hip.hpp dense_mv/dense_mtv)
