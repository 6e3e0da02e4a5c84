import sys, json; sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import numpy as np
from proxsdp_jl_amd import problems as P
from proxsdp_jl_amd.optimizer import Optimizer
for nb in (1, 4):
    model = P.block_diag_problems([P.mimo(512, seed=s) for s in range(nb)])
    o = Optimizer()
    s = o.optimize(model, trace_capacity=500)
    st = s.stats
    print(nb, "blocks: status", s.status, "iters", s.iter, "loop %.3f s"%st["loop_time"], "psd %.3f"%st["t_psd"], "ls %.3f"%st["t_linesearch"], "res %.3f"%st["t_residual"],
          "mv", st["lanczos_matvecs"], "restarts", st["lanczos_restarts"], "calls", st["lanczos_calls"], "full", st["full_eigs"], "fallbacks", st["krylov_fallbacks"], "host_eig %.3f"%st["t_primal"], "rank", s.final_rank)
    print("   mv/iter first 10:", s.trace[:10,13], "trials", s.trace[:10,11])
