"""Run one named instance for a number of iterations (profiling target): python run_one.py <name> <iters> [option=value ...]"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
from proxsdp_jl_amd import problems as P
from proxsdp_jl_amd.optimizer import Optimizer
data = os.path.join(root, "tests", "golden", "sdplib_all")
name, iters = sys.argv[1], int(sys.argv[2])
kw = {}
for a in sys.argv[3:]:
    k, v = a.split("=")
    kw[k] = float(v) if "." in v or "e" in v else int(v)
if name.startswith("sensorloc"): pr = P.sensorloc(int(name[9:]), seed=0)
elif name.startswith("maxcut"): pr = P.maxcut(int(name[6:]), seed=2)
elif name.startswith("mimo"): pr = P.mimo(int(name[4:]), seed=0)
elif name.startswith("blocks:"): pr = P.sdplib_blocks(os.path.join(data, name[7:] + ".dat-s"))      # the file's block structure kept
else: pr = P.sdplib(os.path.join(data, name + ".dat-s"))
s = Optimizer(max_iter=iters, **kw).optimize(pr)
st = s.stats
print(name, "status", s.status, "iters", s.iter, "loop_s %.4f" % st["loop_time"], "us/iter %.1f" % (1e6 * st["loop_time"] / max(1, s.iter)),
      "psd %.1f" % (1e6 * st["t_psd"] / max(1, s.iter)), "linesearch %.1f" % (1e6 * st["t_linesearch"] / max(1, s.iter)),
      "matvecs", st["lanczos_matvecs"], "restarts", st["lanczos_restarts"], "obj", s.objval)
