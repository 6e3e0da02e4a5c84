"""Persistent cycle kernel at G <= 2 (side <= 256): per-iteration trace against the step kernels.  gpurun -- python tools/gpurun_cycle_small_debug.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from proxsdp_jl_amd import problems as P
from proxsdp_jl_amd.optimizer import Optimizer
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for name in ("mcp250-1", "mcp124-1"):
    pr = P.sdplib(os.path.join(root, "tests", "golden", "sdplib", name + ".dat-s"))
    tr = []
    for knob in (0, 1):
        s = Optimizer(max_iter=4000, lanczos_cycle_kernel=knob).optimize(pr, trace_capacity=4000)
        tr.append(np.array(s.trace)[:s.iter])
        print(name, knob, s.iter, s.stats["lanczos_matvecs"], s.stats["lanczos_restarts"], s.stats["cycle_launches"], s.stats["krylov_fallbacks"], s.stats["full_eigs"])
    a, b = tr
    n = min(len(a), len(b))
    first = next((it for it in range(n) if a[it, 13] != b[it, 13]), None)
    print("first iteration with different mat-vec counts:", None if first is None else first + 1)
    for w0 in range(0, n, 250):
        sl = slice(w0, min(w0 + 250, n))
        print("iterations %4d..%4d  mat-vecs/iteration step %.1f cycle %.1f   target rank step %.1f cycle %.1f   max mat-vecs %d / %d" % (
            w0 + 1, sl.stop, a[sl, 13].mean(), b[sl, 13].mean(), a[sl, 10].mean(), b[sl, 10].mean(), a[sl, 13].max(), b[sl, 13].max()))
