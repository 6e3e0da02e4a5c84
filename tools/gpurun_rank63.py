"""the bench headline window (pinned target rank 63) only: it/s and the host-eigensolve share.  gpurun helper."""
import sys, json
sys.path.insert(0, ".")
from proxsdp_jl_amd import problems
from proxsdp_jl_amd.optimizer import Optimizer
pr = problems.maxcut(4000, seed=0)
W, K = 20, 300
o = Optimizer(max_iter=W + K, initial_target_rank=63, max_target_rank_krylov_eigs=64)
s = o.optimize(pr, trace_capacity=W + K)
t = s.trace
dt = t[-1, 12] - t[W - 1, 12]
print(json.dumps(dict(it_per_s=K / dt, ms_per_step=1e3 * dt / K, host_eig_ms_per_step=1e3 * s.stats["host_eig_time"] / s.iter,
                      host_eigs=int(s.stats["host_eigs"]), matvecs_per_step=float(t[W:, 13].mean()), obj=float(t[-1, 1]))))
