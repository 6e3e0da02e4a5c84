"""Isolated sign-function projections for the PMC passes: one projection each at n = 501, 1000, 2000, 4000
(57 products each: k_sym_gemm32 up to side 3072, k_sym_gemm above and for the final product)."""
import sys
import numpy as np
sys.path.insert(0, ".")
from proxsdp_jl_amd import binding as B
sys.path.insert(0, "tools")
from gpurun_sign import svec

for n in [int(a) for a in sys.argv[1:]] or [501, 1000, 2000, 4000]:
    rng = np.random.default_rng(n)
    M = rng.standard_normal((n, n)); A = (M + M.T) / 2
    out, ms, rk, npr = B.full_eig_kernel(svec(A), n, sign=1, repeat=1)
    print(n, ms, rk, npr)
