import sys, time
sys.path.insert(0, ".")
from proxsdp_jl_amd import problems
from proxsdp_jl_amd.optimizer import Optimizer
t=time.time(); pr = problems.maxcut(4000, seed=0); print("generate %.2f s" % (time.time()-t))
for rep in range(3):
    t=time.time(); s = Optimizer(max_iter=2).optimize(pr); w=time.time()-t
    print("rep %d: wall %.3f s  res.time %.3f  init %.3f  loop %.3f  exit %.3f  exit_matvecs %d" % (rep, w, s.time, s.stats["init_time"], s.stats["loop_time"], s.stats["exit_time"], s.stats["exit_matvecs"]))
