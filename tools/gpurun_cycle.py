"""persistent Lanczos cycle kernel vs the step kernels: same traces, timing.  gpurun helper."""
import sys, time, json
import numpy as np
sys.path.insert(0, ".")
from proxsdp_jl_amd import problems
from proxsdp_jl_amd.optimizer import Optimizer
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 300
extra = json.loads(sys.argv[3]) if len(sys.argv) > 3 else {}
pr = problems.maxcut(n, seed=0)
res = {}
for cy in ((0, 1) if len(sys.argv) < 5 else (int(sys.argv[4]) if sys.argv[4].isdigit() else 1,)):
    o = Optimizer(max_iter=iters, lanczos_cycle_kernel=cy, support_path=1, profile_symv_every=16, **extra)
    s = o.optimize(pr, trace_capacity=iters)
    st = s.stats
    t = s.trace
    res[cy] = t
    w0 = min(20, iters // 4)
    dt = t[-1, 12] - t[w0 - 1, 12]
    print(json.dumps(dict(cycle=cy, iters=int(s.iter), it_per_s=(len(t) - w0) / dt, matvecs=int(st["lanczos_matvecs"]),
                          restarts=int(st["lanczos_restarts"]), cycle_launches=int(st["cycle_launches"]),
                          cycle_steps=int(st["cycle_steps"]), cycle_us_per_step=1e3 * st["cycle_ms"] / max(1, st["cycle_steps"]),
                          fop=int(st["fop_projections"]), obj=float(t[-1, 1]), host_eig_s=st["host_eig_time"])))
if 0 not in res or 1 not in res: sys.exit(0)
a, b = res[0], res[1]
m = min(len(a), len(b))
for col, nm in ((1, "prim_obj"), (2, "dual_obj"), (7, "step"), (13, "matvecs"), (11, "trials")):
    d = np.abs(a[:m, col] - b[:m, col]) / (1e-300 + np.abs(a[:m, col]).max())
    print(nm, "max rel diff", d.max(), "first >1e-9 at", int(np.argmax(d > 1e-9)) if (d > 1e-9).any() else -1)
