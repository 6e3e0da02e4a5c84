"""Oracle solves of the SENSORLOC family (problems.sensorloc, seed 0, reference default options) at sizes where the CPU oracle
takes minutes: iteration counts / mat-vec totals for tests/test_gpu_parity.py::test_sensorloc_larger_sizes_take_the_committed_oracle_counts.
usage: python tests/golden/make_golden_sensorloc.py 150 200 300   (n = 150: 1 min, 200: 5-10 min, 300: ~1 h on 8 cores)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle
from proxsdp_jl_amd import problems as P
out_path = os.path.join(ROOT, "tests", "golden", "sensorloc_oracle.json")
out = json.load(open(out_path)) if os.path.exists(out_path) else {}
for n in [int(a) for a in sys.argv[1:]]:
    pr = P.sensorloc(n, seed=0)
    o = oracle.Options(); o.time_limit = 4 * 3600.0
    t0 = time.time(); r = oracle.solve(pr, o)
    out[str(n)] = dict(status=int(r.status), iterations=int(r.iter), objval=float(r.objval),
                       lanczos_matvecs=int(r.stats.get("lanczos_matvecs", 0)), oracle_seconds=round(time.time() - t0, 1))
    json.dump(out, open(out_path, "w"), indent=1)
    print(n, out[str(n)], flush=True)
