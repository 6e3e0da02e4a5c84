"""Where a PDHG iteration of the runbench's heavy instances goes: section times of the library's stats block."""
import json, os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
from proxsdp_jl_amd import problems as P
from proxsdp_jl_amd.optimizer import Optimizer
data = os.path.join(root, "tests", "golden", "sdplib_all")
Optimizer(max_iter=20).optimize(P.maxcut(120, seed=1))
for name, mk, it in (("sensorloc400", lambda: P.sensorloc(400, seed=0), 1500), ("sensorloc200", lambda: P.sensorloc(200, seed=0), 1500),
                     ("gpp500-1", lambda: P.sdplib(os.path.join(data, "gpp500-1.dat-s")), 1500),
                     ("mcp500-1", lambda: P.sdplib(os.path.join(data, "mcp500-1.dat-s")), 1500)):
    pr = mk()
    s = Optimizer(max_iter=it, lanczos_cycle_kernel=0).optimize(pr)
    st = s.stats
    keys = ["loop_time", "t_primal", "t_psd", "t_linesearch", "t_residual", "lanczos_matvecs", "lanczos_restarts", "linesearch_trials", "host_eig_time", "dense_passes", "fop_projections"]
    print(name, "iters", s.iter, "n,m:", getattr(pr, "n", None), {k: (round(st[k], 4) if isinstance(st[k], float) else st[k]) for k in keys if k in st}, flush=True)
    print("   us/iter", 1e6 * st["loop_time"] / s.iter, "psd", 1e6 * st["t_psd"] / s.iter, "linesearch", 1e6 * st["t_linesearch"] / s.iter, flush=True)
