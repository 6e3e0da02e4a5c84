import sys, os
sys.path.insert(0, os.getcwd())
from proxsdp_jl_amd import problems as P
from proxsdp_jl_amd.optimizer import Optimizer
pr = P.maxcut(4000, seed=0)
for it in (200, 260):
    s = Optimizer(max_iter=it, initial_target_rank=63, max_target_rank_krylov_eigs=64).optimize(pr, trace_capacity=it)
    st = s.stats
    print(it, "host_eig_time", st["host_eig_time"], "overlap", st["host_eig_overlap_time"], "merges", st["host_eig_merges"], "host_eigs", st["host_eigs"], "restarts", st["lanczos_restarts"], "calls", st["lanczos_calls"], "t_psd", st["t_psd"], "loop", st["loop_time"], flush=True)
