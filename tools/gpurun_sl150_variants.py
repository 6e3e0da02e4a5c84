"""SENSORLOC n = 150 under variants that leave the mathematics alone and change the rounding (other Lanczos start vectors: the same eigenpairs to
krylovkit_tol; a tighter krylovkit_tol; one more Krylov vector):
iteration counts to OPTIMAL and where the traces part from the default build's."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from proxsdp_jl_amd import problems as P
from proxsdp_jl_amd.optimizer import Optimizer
pr = P.sensorloc(150, seed=0)
base = None
rng = np.random.default_rng(7)
for kw in (dict(), dict(resid=1), dict(resid=2), dict(resid=3), dict(krylovkit_tol=1e-13), dict(eigsolver_min_lanczos=26)):
    kw = dict(kw)
    resid = kw.pop("resid", None)
    er = None
    if resid is not None:                                   # another Lanczos start vector: the same eigenpairs to krylovkit_tol, other rounding
        er = rng.standard_normal(152); er /= np.linalg.norm(er)
    s = Optimizer(**kw).optimize(pr, trace_capacity=6000, eig_resid=er)
    kw = dict(kw, start_vector=resid)
    t = np.asarray(s.trace)
    if base is None: base = t
    m = min(len(t), len(base))
    d = np.abs(t[:m, 4] - base[:m, 4]) / np.maximum(np.abs(base[:m, 4]), 1e-300)
    first = {thr: (int(np.nonzero(d > thr)[0][0]) if (d > thr).any() else None) for thr in (1e-10, 1e-6, 1e-3)}
    print(kw, "status", s.status, "iterations", s.iter, "matvecs", s.stats["lanczos_matvecs"], "feasibility parts from the default at", first, flush=True)
