"""Register-resident cycle kernel (lanczos_cycle_kernel = 2) against the step kernels (0): bit-for-bit traces, timing, and the
kernel's own per-phase ticks (PROXSDP_HIP_DEBUG_CYCLE=1).  gpurun -- python tools/gpurun_cycle2.py [quick]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from proxsdp_jl_amd import problems as P
from proxsdp_jl_amd.optimizer import Optimizer

def run(pr, cyc, **kw):
    cap = kw.get("max_iter", 100)
    s = Optimizer(lanczos_cycle_kernel=cyc, **kw).optimize(pr, trace_capacity=cap)
    return s

def compare(name, pr, **kw):
    a = run(pr, 0, **kw)
    b = run(pr, 2, **kw)
    ta, tb = np.asarray(a.trace), np.asarray(b.trace)
    m = min(len(ta), len(tb))
    same = a.iter == b.iter and np.array_equal(ta[:m], tb[:m])
    first_bad = -1
    if not same:
        for r in range(m):
            if not np.array_equal(ta[r], tb[r]):
                first_bad = r; break
    sa, sb = a.stats, b.stats
    print(f"{name}: iter {a.iter}/{b.iter} bit-identical traces {same} first differing row {first_bad} "
          f"max |d obj| {np.abs(ta[:m,1]-tb[:m,1]).max():.3e} matvecs {sa['lanczos_matvecs']}/{sb['lanczos_matvecs']} "
          f"restarts {sa['lanczos_restarts']}/{sb['lanczos_restarts']} cycle_steps {sb['cycle_steps']} launches {sb['cycle_launches']} "
          f"loop {sa['loop_time']:.3f}/{sb['loop_time']:.3f} s  us/matvec {1e6*sa['loop_time']/max(sa['lanczos_matvecs'],1):.2f}/"
          f"{1e6*sb['loop_time']/max(sb['lanczos_matvecs'],1):.2f}", flush=True)
    if first_bad >= 0:
        cols = [c for c in range(ta.shape[1]) if not np.array_equal(ta[:m, c], tb[:m, c])]
        print("   differing columns:", cols, "first rows:", [int(np.nonzero(ta[:m, c] != tb[:m, c])[0][0]) for c in cols])
        for c in cols[:4]:
            r = int(np.nonzero(ta[:m, c] != tb[:m, c])[0][0])
            print(f"   col {c} row {r}: {ta[r, c]!r} vs {tb[r, c]!r}")
    return same

quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
ok = True
ok &= compare("maxcut700 r2..16", P.maxcut(700, seed=5), max_iter=160, support_path=1, max_target_rank_krylov_eigs=16, initial_target_rank=2)
ok &= compare("maxcut1500 r30..40", P.maxcut(1500, seed=5), max_iter=160, support_path=1, max_target_rank_krylov_eigs=40, initial_target_rank=30)
if not quick:
    pr = P.maxcut(4000, seed=0)
    ok &= compare("maxcut4000 default 600", pr, max_iter=600)
    ok &= compare("maxcut4000 rank63", pr, max_iter=260, initial_target_rank=63, max_target_rank_krylov_eigs=64)
    ok &= compare("maxcut4000 rank31", pr, max_iter=200, initial_target_rank=31, max_target_rank_krylov_eigs=32)
print("ALL BIT-IDENTICAL" if ok else "DIFFERENCES FOUND")
