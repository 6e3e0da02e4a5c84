"""BASELINE config 5 solved to tolerance: SDPLIB maxG51 / gpp500-1 with full_eig_decomp = true (every projection is
full_eig!), tol 1e-4, sign-function projection (default: shortened, tested schedule; sign_start_row = 0: full table) vs rocSOLVER (full_eig_sign = 0, time-limited), and the
default-options solve (Krylov path) beside it."""
import sys, time, json
sys.path.insert(0, ".")
from pathlib import Path
from proxsdp_jl_amd import problems as P
from proxsdp_jl_amd.optimizer import Optimizer
LIT = {"maxG51": 4003.81, "gpp500-1": 25.3205}
g = Path("tests/golden/sdplib")
out = {}
for name in (sys.argv[1:] or LIT):
    pr = P.sdplib(g / f"{name}.dat-s")
    for label, kw in (("full_eig_sign", dict(full_eig_decomp=1)),
                      ("full_eig_sign_full_table", dict(full_eig_decomp=1, sign_start_row=0)),
                      ("full_eig_rocsolver", dict(full_eig_decomp=1, full_eig_sign=0)),
                      ("default_options", dict())):
        o = Optimizer(tol_gap=1e-4, tol_feasibility=1e-4, time_limit=120.0, **kw)
        t = time.time(); s = o.optimize(pr); dt = time.time() - t
        r = dict(status=int(s.status), iterations=int(s.iter), time_s=dt, objective=float(s.objval), literature=LIT[name],
                 rel_to_literature=abs(abs(s.objval) - LIT[name]) / LIT[name], final_rank=int(s.final_rank),
                 full_eigs=int(s.stats["full_eigs"]), full_eigs_sign=int(s.stats["full_eigs_sign"]),
                 full_eigs_lanczos=int(s.stats["full_eigs_lanczos"]), lanczos_matvecs=int(s.stats["lanczos_matvecs"]),
                 sign_products=int(s.stats["sign_products"]), sign_short_pass=int(s.stats["sign_short_pass"]),
                 sign_short_fail=int(s.stats["sign_short_fail"]),
                 it_per_s=s.iter / dt)
        out[f"{name}:{label}"] = r
        print(name, label, r, flush=True)
json.dump(out, open("gpurun_out/cfg5.json", "w"), indent=1)
