"""The SENSORLOC set of the reference's benchmark (test/runbench.jl: n = 100, 200, 300, 400; one (n + 2) x (n + 2) PSD block,
~n^2/20 + n^2/10 equality rows) through the library with reference default options, and the CPU oracle beside it where it
finishes.  gpurun -- python tools/gpurun_sensorloc.py [--oracle-upto N]"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from proxsdp_jl_amd import problems as P
from proxsdp_jl_amd.optimizer import Optimizer
upto = int(sys.argv[sys.argv.index("--oracle-upto") + 1]) if "--oracle-upto" in sys.argv else 100
rows = []
for n in (50, 100, 200, 300, 400):
    pr = P.sensorloc(n, seed=0)
    s = Optimizer(time_limit=300.0).optimize(pr)
    st = s.stats
    X = P.unpack_psd(s.primal, n + 2)
    row = dict(n=n, side=n + 2, rows=int(pr.p), status=int(s.status), iterations=int(s.iter), time_s=round(s.time, 3),
               loop_s=round(st["loop_time"], 3), it_per_s=round(s.iter / max(st["loop_time"], 1e-9), 1),
               position_error=float(np.abs(X[:2, 2:] - pr.x_true).max()), lambda_min=float(np.linalg.eigvalsh(X).min()),
               final_rank=int(s.final_rank), lanczos_matvecs=int(st["lanczos_matvecs"]), full_eigs=int(st["full_eigs"]),
               full_eigs_lanczos=int(st["full_eigs_lanczos"]), full_eigs_sign=int(st["full_eigs_sign"]))
    if n <= upto:
        import oracle
        o = oracle.Options(); o.time_limit = 600.0
        t0 = time.time(); r = oracle.solve(pr, o); dt = time.time() - t0
        row.update(oracle_status=int(r.status), oracle_iterations=int(r.iter), oracle_time_s=round(dt, 2),
                   oracle_it_per_s=round(r.iter / max(dt, 1e-9), 1))
    print(json.dumps(row), flush=True)
    rows.append(row)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(rows, open("gpurun_out/r05_sensorloc.json", "w"), indent=1)
