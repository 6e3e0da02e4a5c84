"""full_eig! beyond side 4096 (maxG55 / maxG60 sizes): sign-function projection (auto window since round 4) vs rocSOLVER dsyevd,
time per projection on device-resident data; agreement of the two projections."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from proxsdp_jl_amd import binding as B
out = {}
for n in [int(a) for a in (sys.argv[1:] or ["4096", "5000", "7000"])]:
    rng = np.random.default_rng(n)
    Z = rng.standard_normal((n, 60)); M = rng.standard_normal((n, 200)) * 0.2
    X = Z @ Z.T - M @ M.T
    i, j = np.triu_indices(n)
    # packed svec (column-major upper triangle, off-diagonals x sqrt 2)
    Xs = np.where(np.arange(n)[:, None] == np.arange(n)[None, :], X, X * np.sqrt(2.0))
    packed = Xs.T[np.tril_indices(n)]
    a, ms_auto, rk_a, prod = B.full_eig_kernel(packed, n, sign=-1, repeat=3)
    b, ms_dsy, rk_b, _ = B.full_eig_kernel(packed, n, sign=0, repeat=1)
    ld = 64 * ((n + 63) // 64); t64 = ld // 64
    flops = prod * (t64 * (t64 + 1) // 2) * 2.0 * 64 * 64 * ld
    out[n] = dict(sign_ms=ms_auto, products=int(prod), dsyevd_ms=ms_dsy, rank_sign=int(rk_a), rank_dsyevd=int(rk_b),
                  max_abs_diff_over_scale=float(np.abs(a - b).max() / np.abs(b).max()), executed_TFLOPs=flops / (ms_auto * 1e-3) / 1e12,
                  frac_of_fp64_mfma_peak=flops / (ms_auto * 1e-3) / 1e12 / 78.6)
    print(n, out[n], flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/r04_sign_large.json", "w"), indent=1)
