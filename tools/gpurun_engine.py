"""psd_sign_engine = 1 (cost-based stand-in of the sign-function projection on the Krylov branch) against the
default engine: same iterates?  same iteration counts?  time."""
import sys, time, json
sys.path.insert(0, ".")
from pathlib import Path
import numpy as np
from proxsdp_jl_amd import problems as P
from proxsdp_jl_amd.optimizer import Optimizer
g = Path("tests/golden/sdplib")
cases = [("maxG51", lambda: P.sdplib(g / "maxG51.dat-s"), dict(max_iter=1500)),
         ("gpp500-1", lambda: P.sdplib(g / "gpp500-1.dat-s"), dict()),
         ("mcp500-1", lambda: P.sdplib(g / "mcp500-1.dat-s"), dict()),
         ("mimo512", lambda: P.mimo(512, seed=0), dict()),
         ("maxcut1000", lambda: P.maxcut(1000, seed=0), dict(max_iter=1500))]
want = sys.argv[1:]
out = {}
for name, mk, kw in cases:
    if want and name not in want:
        continue
    pr = mk()
    sols = {}
    for eng in (0, 1):
        o = Optimizer(tol_gap=1e-4, tol_feasibility=1e-4, time_limit=150.0, psd_sign_engine=eng, **kw)
        t = time.time(); s = o.optimize(pr, trace_capacity=kw.get("max_iter", 20000)); dt = time.time() - t
        sols[eng] = s
        out[f"{name}:{eng}"] = dict(status=int(s.status), iterations=int(s.iter), time_s=dt, objective=float(s.objval),
                                    final_rank=int(s.final_rank), matvecs=int(s.stats["lanczos_matvecs"]),
                                    sign_engine=int(s.stats["sign_engine_projections"]),
                                    rejected=int(s.stats["sign_engine_rejected"]), checks=int(s.stats["sign_engine_checks"]),
                                    mismatches=int(s.stats["sign_engine_mismatches"]), it_per_s=s.iter / dt)
        print(name, eng, out[f"{name}:{eng}"], flush=True)
    a, b = sols[0], sols[1]
    m = min(len(a.trace), len(b.trace))
    sc = np.abs(a.trace[:m, 1:5]).max(axis=0) + 1e-300
    d = np.abs(a.trace[:m, 1:5] - b.trace[:m, 1:5]) / sc
    first_bad = int(np.argmax(d.max(axis=1) > 1e-6)) if (d.max(axis=1) > 1e-6).any() else -1
    print(name, "trace rel diff: max over first 200 %.2e, all %.2e, first > 1e-6 at %d; linesearch equal %s; rank cols equal %s" % (
        d[:200].max(), d.max(), first_bad, np.array_equal(a.trace[:m, 11], b.trace[:m, 11]),
        np.array_equal(a.trace[:m, 8], b.trace[:m, 8])), flush=True)
    out[f"{name}:trace"] = dict(max_rel_first200=float(d[:200].max()), max_rel=float(d.max()), first_gt_1e6=first_bad)
json.dump(out, open("gpurun_out/engine.json", "w"), indent=1)
