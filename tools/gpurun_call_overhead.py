import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import kat_problems as K
from proxsdp_jl_amd import problems as P
from proxsdp_jl_amd.optimizer import Optimizer
for name, pr in (("sdp_wiki", K.sdp_wiki(False)), ("simple_lp", K.simple_lp()), ("maxcut150", P.maxcut(150, seed=0)), ("maxcut1000", P.maxcut(1000, seed=0))):
    for rep in range(3):
        t0 = time.perf_counter()
        s = Optimizer(max_iter=200).optimize(pr)
        wall = time.perf_counter() - t0
        st = s.stats
        print(f"{name} call {rep}: wall {wall*1e3:.1f} ms, Result.time {s.time*1e3:.1f} ms, init {st['init_time']*1e3:.1f} loop {st['loop_time']*1e3:.1f} exit {st['exit_time']*1e3:.1f} iters {s.iter}", flush=True)
