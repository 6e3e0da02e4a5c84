"""round-2 PMC driver (torch-free): isolated launches of the HBM-bound kernels at n = 4000 --
k_symv_packed x6, then the reconstruction kernels: scalar and MFMA at r = 63 and r = 2000."""
import sys; sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import numpy as np
from proxsdp_jl_amd import binding as B
n = 4000; rng = np.random.default_rng(0)
x = rng.standard_normal(n * (n + 1) // 2); v = rng.standard_normal(n)
y, ms = B.symv_packed(x, n, v, repeat=5); print("symv ok", ms)
for r in (63, 2000):
    Z = rng.standard_normal((n, r)) / np.sqrt(n); lam = np.ones(r)
    for mf in (0, 1):
        out, ms = B.reconstruct(Z, lam, n, repeat=2, mfma=mf); print("recon r", r, "mfma", mf, ms)
