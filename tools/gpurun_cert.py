"""Independent optimality certificate for a solve: primal feasibility + PSD, dual cone PSD, gap --
computed on the host with LAPACK from the returned arrays (not from the solver's own flags)."""
import sys, time; sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from pathlib import Path
import numpy as np
from proxsdp_jl_amd import problems as P
from proxsdp_jl_amd.optimizer import Optimizer
name = sys.argv[1] if len(sys.argv) > 1 else "maxG55"
g = Path(__file__).resolve().parent.parent / "tests" / "golden" / "sdplib"
pr = P.maxcut(int(name[2:]), seed=0) if name.startswith("er") else P.sdplib(g / f"{name}.dat-s")
o = Optimizer(tol_gap=1e-4, tol_feasibility=1e-4, max_target_rank_krylov_eigs=64, time_limit=300.0)
t = time.time(); s = o.optimize(pr); dt = time.time() - t
n = pr.psd_sides()[0]
print(name, "status", s.status, "iter", s.iter, "time %.1f" % dt, "obj", s.objval, "dual obj", s.dual_objval,
      "dual_feasible flag", s.dual_feasible_user_tol, "dual_feasibility", s.dual_feasibility, flush=True)
X = P.unpack_psd(s.primal, n)
Z = P.unpack_psd(s.dual_cone, n)
print("primal: max|diag-1| %.2e" % np.abs(np.diag(X) - 1).max(), " c'x %.4f" % float(pr.c @ s.primal), flush=True)
wx = np.linalg.eigvalsh(X); wz = np.linalg.eigvalsh(Z)
print("lambda_min(X) %.3e  rank(X>1e-6) %d   lambda_min(Z) %.3e  lambda_max(Z) %.3e" % (wx[0], int((wx > 1e-6).sum()), wz[0], wz[-1]))
print("complementarity <X,Z> %.4e   b'y %.4f" % (float(np.sum(X * Z)), float(pr.b @ s.dual_eq)))
