"""Medium PSD sides (101 .. 500: the Krylov path with few workgroups): step kernels against the persistent one-XCD cycle kernel
(`lanczos_cycle_kernel` = 0 / 1), reference default options.  One row per instance: iterations, loop seconds, us per iteration,
cycle launches.  gpurun -- python tools/gpurun_cycle_medium.py [out.md]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from proxsdp_jl_amd import problems as P
from proxsdp_jl_amd.optimizer import Optimizer

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
data = os.path.join(root, "tests", "golden", "sdplib_all")
if not os.path.isdir(data):
    data = os.path.join(root, "tests", "golden", "sdplib")
jobs = [("maxcut150", lambda: P.maxcut(150, seed=0)), ("maxcut300", lambda: P.maxcut(300, seed=0)), ("maxcut500", lambda: P.maxcut(500, seed=0)),
        ("sensorloc100", lambda: P.sensorloc(100, seed=0)), ("sensorloc200", lambda: P.sensorloc(200, seed=0))]
for f in ("mcp124-1", "mcp250-1", "mcp500-1", "gpp124-1", "gpp250-1"):
    p = os.path.join(data, f + ".dat-s")
    if os.path.exists(p):
        jobs.append((f, (lambda p=p: P.sdplib(p))))
Optimizer(max_iter=5).optimize(P.maxcut(120, seed=0))
lines = ["| instance | cycle kernel | status | iterations | loop s | us / iteration | mat-vecs | cycle launches | cycle steps | fop projections | cycle ms |", "|---|---|---|---|---|---|---|---|---|---|---|"]
for name, build in jobs:
    pr = build()
    for knob in (0, 1):
        s = Optimizer(time_limit=120.0, lanczos_cycle_kernel=knob).optimize(pr)
        st = s.stats
        loop = st.get("loop_time", s.time)
        line = f"| {name} | {knob} | {s.status} | {s.iter} | {loop:.3f} | {1e6 * loop / max(s.iter, 1):.0f} | {st.get('lanczos_matvecs')} | {st.get('cycle_launches')} | {st.get('cycle_steps')} | {st.get('fop_projections')} | {st.get('cycle_ms', 0):.0f} |"
        print(line, flush=True)
        lines.append(line)
out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "gpurun_out", "r05_cycle_medium.md")
os.makedirs(os.path.dirname(out), exist_ok=True)
open(out, "w").write("\n".join(lines) + "\n")
