"""reconstruction kernels at n = 4000: scalar-FMA (LDS-staged) vs fp64 MFMA SYRK, r in {2 .. n/2}.  gpurun helper."""
import sys, json
import numpy as np
sys.path.insert(0, ".")
from proxsdp_jl_amd import binding as B
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
N = n * (n + 1) // 2
rng = np.random.default_rng(0)
rows = []
for r in (2, 5, 8, 12, 16, 26, 32, 63, 128, 500, n // 2):
    Z = rng.standard_normal((n, r)) / np.sqrt(n)
    lam = rng.uniform(0.5, 2.0, r)
    rep = 20 if r <= 128 else 5
    _, t0 = B.reconstruct(Z, lam, n, repeat=rep, mfma=0)
    _, t1 = B.reconstruct(Z, lam, n, repeat=rep, mfma=1)
    flops = 2.0 * N * r
    rows.append(dict(r=r, scalar_us=1e3 * t0, mfma_us=1e3 * t1, write_GBs_mfma=8.0 * N / (t1 * 1e-3) / 1e9,
                     tflops_scalar=flops / (t0 * 1e-3) / 1e12, tflops_mfma=flops / (t1 * 1e-3) / 1e12))
    print(json.dumps(rows[-1]))
