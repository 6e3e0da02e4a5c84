import sys, os
sys.path.insert(0, os.getcwd())
from proxsdp_jl_amd import problems as P
from proxsdp_jl_amd.optimizer import Optimizer
for n in (2, 3, 4, 5, 6, 8, 10, 12, 16):
    pr = P.mimo(n - 1, seed=0) if n > 2 else P.mimo(1, seed=0)
    side = pr.psd_sides()[0]
    row = []
    for sbb in (1, 2):
        s = Optimizer(small_block_batch=sbb, max_iter=3000, tol_gap=1e-8, tol_feasibility=1e-8, min_iter=10**9).optimize(pr)
        row.append(1e6 * s.stats["loop_time"] / s.iter)
    print(f"side {side}: Jacobi {row[0]:.1f} us/iter, sign-in-LDS {row[1]:.1f} us/iter", flush=True)
