"""PMC driver for the dense-A kernels: a 6-iteration solve of randSDP-shaped n=400, m=1500 (A = 1500 x 80200
doubles = 962 MB, host pointer so no torch in the process)."""
import sys; sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import numpy as np, scipy.sparse as sp
from proxsdp_jl_amd import problems as P
from proxsdp_jl_amd.optimizer import Optimizer
n, m = 400, 1500
N = n * (n + 1) // 2
rng = np.random.default_rng(0)
M = rng.standard_normal((m, N))
pr = P.Problem(n=N, A=sp.csc_matrix((m, N)), b=rng.standard_normal(m), G=sp.csc_matrix((0, N)), h=np.zeros(0),
               c=rng.standard_normal(N), psd=[np.arange(N, dtype=np.int64)], M_dense=M)
s = Optimizer(max_iter=6).optimize(pr)
print("ok", s.iter, s.stats["dense_passes"], "bytes per pass", 8 * m * N)
