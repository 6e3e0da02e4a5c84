"""certificate search on PSD models (infeasible / unbounded SDP, Lanczos-sized block): support path vs dense path"""
import sys, json, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from kat_problems import infeasible_sdp, unbounded_sdp
from proxsdp_jl_amd.optimizer import Optimizer
for build in (infeasible_sdp, unbounded_sdp):
    for sp in (0, 1):
        o = Optimizer(support_path=sp, time_limit=60.0)
        t = time.time(); s = o.optimize(build(), trace_capacity=200000)
        print(json.dumps(dict(model=build.__name__, support_path=sp, status=int(s.status), string=s.status_string, iter=int(s.iter),
                              cert=bool(s.certificate_found), obj=float(s.objval), wall=time.time() - t, fop=int(s.stats["fop_projections"]),
                              matvecs=int(s.stats["lanczos_matvecs"]))))
