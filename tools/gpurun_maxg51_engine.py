"""maxG51 with reference default options: Krylov path (541 mat-vecs per projection late in the solve, packed operator
because of the hub rows) vs psd_sign_engine = 1."""
import sys, time, json
sys.path.insert(0, ".")
from pathlib import Path
from proxsdp_jl_amd import problems as P
from proxsdp_jl_amd.optimizer import Optimizer
pr = P.sdplib(Path("tests/golden/sdplib") / "maxG51.dat-s")
out = {}
for eng in (1,):
    o = Optimizer(tol_gap=1e-4, tol_feasibility=1e-4, time_limit=200.0, psd_sign_engine=eng)
    t = time.time(); s = o.optimize(pr); dt = time.time() - t
    out[eng] = dict(status=int(s.status), iterations=int(s.iter), time_s=dt, objective=float(s.objval), final_rank=int(s.final_rank),
                    matvecs=int(s.stats["lanczos_matvecs"]), sign_engine=int(s.stats["sign_engine_projections"]),
                    rejected=int(s.stats["sign_engine_rejected"]), full_eigs=int(s.stats["full_eigs"]))
    print(eng, out[eng], flush=True)
json.dump(out, open("gpurun_out/maxg51_engine.json", "w"), indent=1)
