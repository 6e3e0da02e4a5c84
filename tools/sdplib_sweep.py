"""SDPLIB Max-Cut family through the library at tol 1e-4 against the literature optima
(tests/golden/sdplib/README.md).  usage: sdplib_sweep.py [--engine] [name ...]
--engine: psd_sign_engine = 1 (sign-function projection where it is the cheaper way to the same projection)"""
import sys, time; sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from pathlib import Path
import numpy as np
from proxsdp_jl_amd import problems as P
from proxsdp_jl_amd.optimizer import Optimizer
LIT = {"mcp124-1": 141.99, "mcp250-1": 317.26, "mcp500-1": 598.15, "maxG11": 629.16, "maxG32": 1567.64,
       "maxG51": 4003.81, "maxG55": 9999.21}
g = Path(__file__).resolve().parent.parent / "tests" / "golden" / "sdplib"
ENGINE = "--engine" in sys.argv
names = [a for a in sys.argv[1:] if not a.startswith("--")]
for name in (names or LIT):
    pr = P.sdplib(g / f"{name}.dat-s")
    o = Optimizer(tol_gap=1e-4, tol_feasibility=1e-4, max_target_rank_krylov_eigs=64, time_limit=150.0,
                  psd_sign_engine=1 if ENGINE else 0)
    t = time.time(); s = o.optimize(pr); dt = time.time() - t
    side = pr.psd_sides()[0]
    print("%-9s n=%5d status %d iter %6d time %6.2f s obj %.4f lit %.4f rel %.2e rank %d matvecs %d full_eigs %d fop %d sign_engine %d" % (
        name, side, s.status, s.iter, dt, s.objval, LIT[name], abs(abs(s.objval) - LIT[name]) / LIT[name],
        s.final_rank, s.stats["lanczos_matvecs"], s.stats["full_eigs"], s.stats["fop_projections"], s.stats["sign_engine_projections"]), flush=True)
