import os, sys
sys.path.insert(0, os.getcwd())
from proxsdp_jl_amd import problems as P
from proxsdp_jl_amd.optimizer import Optimizer
pr = P.maxcut(4000, seed=0)
for it in (260, 460):
    s = Optimizer(max_iter=it, initial_target_rank=63, max_target_rank_krylov_eigs=64).optimize(pr)
    st = s.stats
    print("iters", s.iter, "loop %.3f" % st["loop_time"], "host_eig_time %.4f" % st["host_eig_time"], "overlap %.4f" % st["host_eig_overlap_time"], "t_psd %.3f ls %.3f res %.3f" % (st["t_psd"], st["t_linesearch"], st["t_residual"]), flush=True)
