"""Round 6: the one-workgroup cycle kernel (lanczos_cycle_kernel = 2 / auto) against the step kernels (0) on the medium-block
instances of the reference's benchmark script: bit-identity of the traces and us per PDHG iteration."""
import json, os, sys, time
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
from proxsdp_jl_amd import problems as P
from proxsdp_jl_amd.optimizer import Optimizer
data = os.path.join(root, "tests", "golden", "sdplib_all")
out = os.path.join(root, "gpurun_out", "r06"); os.makedirs(out, exist_ok=True)
cases = [("maxcut150", lambda: P.maxcut(150, seed=2)), ("maxcut300", lambda: P.maxcut(300, seed=3)), ("maxcut500", lambda: P.maxcut(500, seed=4)),
         ("sensorloc100", lambda: P.sensorloc(100, seed=0)), ("sensorloc200", lambda: P.sensorloc(200, seed=0)),
         ("sensorloc300", lambda: P.sensorloc(300, seed=0)), ("sensorloc400", lambda: P.sensorloc(400, seed=0)),
         ("mcp124-1", lambda: P.sdplib(os.path.join(data, "mcp124-1.dat-s"))), ("mcp250-1", lambda: P.sdplib(os.path.join(data, "mcp250-1.dat-s"))),
         ("mcp500-1", lambda: P.sdplib(os.path.join(data, "mcp500-1.dat-s"))),
         ("gpp124-1", lambda: P.sdplib(os.path.join(data, "gpp124-1.dat-s"))), ("gpp250-1", lambda: P.sdplib(os.path.join(data, "gpp250-1.dat-s"))),
         ("gpp500-1", lambda: P.sdplib(os.path.join(data, "gpp500-1.dat-s")))]
only = sys.argv[1:]
rows = []
Optimizer(max_iter=20).optimize(P.maxcut(120, seed=1))        # first-call set-up of the process
for name, mk in cases:
    if only and name not in only:
        continue
    pr = mk()
    res = {}
    for knob in (0, 2, -1):
        s = Optimizer(time_limit=120.0, lanczos_cycle_kernel=knob).optimize(pr, trace_capacity=20000)
        res[knob] = s
        st = s.stats
        rows.append(dict(instance=name, knob=knob, status=int(s.status), iterations=int(s.iter), loop_s=st["loop_time"],
                         us_per_iter=1e6 * st["loop_time"] / max(1, s.iter), matvecs=int(st["lanczos_matvecs"]), restarts=int(st["lanczos_restarts"]),
                         cycle_launches=int(st["cycle_launches"]), t_psd=st["t_psd"], objval=float(s.objval)))
        print(rows[-1], flush=True)
    a, b = res[0], res[2]
    m = min(len(a.trace), len(b.trace))
    cols = [c for c in range(a.trace.shape[1]) if c != 12]
    ta, tb = a.trace[:m][:, cols], b.trace[:m][:, cols]
    same = bool(a.iter == b.iter and np.array_equal(ta, tb) and np.array_equal(a.primal, b.primal))
    first = -1
    if not same:
        d = np.any(ta != tb, axis=1)
        first = int(np.argmax(d)) if d.any() else m
    rows.append(dict(instance=name, bit_identical=same, first_different_row=first))
    print(rows[-1], flush=True)
json.dump(rows, open(os.path.join(out, "block1.json"), "w"), indent=1)
