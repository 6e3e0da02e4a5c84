"""PMC driver for the Lanczos kernels: a short Max-Cut solve (torch-free).
usage: gpurun_pmc4.py <lanczos_operator> <max_iter> <n> <support_path>"""
import sys; sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from proxsdp_jl_amd import problems as P
from proxsdp_jl_amd.optimizer import Optimizer
a = [int(x) for x in sys.argv[1:]] + [1, 4, 4000, -1][len(sys.argv) - 1:]
s = Optimizer(max_iter=a[1], lanczos_operator=a[0], support_path=a[3]).optimize(P.maxcut(a[2], seed=0))
print("ok", s.iter, s.stats["lanczos_matvecs"], s.stats["fop_projections"])
