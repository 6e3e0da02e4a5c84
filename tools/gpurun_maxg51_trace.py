import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pathlib import Path
from proxsdp_jl_amd import problems as P
from proxsdp_jl_amd.optimizer import Optimizer
pr = P.sdplib(Path(__file__).resolve().parent.parent / "tests" / "golden" / "sdplib" / "maxG51.dat-s")
it = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
s = Optimizer(max_iter=it).optimize(pr, trace_capacity=it)
tr = np.asarray(s.trace)
np.save("gpurun_out/maxg51_trace.npy", tr)
for i in list(range(3990, 4040, 5)) + list(range(4100, it, 400)):
    if i < len(tr): print(i + 1, "po %.6f do %.6f gap %.2e feas %.2e pres %.2e dres %.2e rank %d trials %d mv %d" % (tr[i,1], tr[i,2], tr[i,3], tr[i,4], tr[i,5], tr[i,6], tr[i,10], tr[i,11], tr[i,13]))
print(s.stats["lanczos_restarts"], s.stats["krylov_fallbacks"], s.final_rank)
a, b = 4100, min(len(tr), it) - 1
print("ms/iteration over iterations %d..%d: %.3f" % (a, b, 1e3 * (tr[b, 12] - tr[a, 12]) / (b - a)))
