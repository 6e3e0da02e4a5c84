"""full_eig! by the sign-function projection vs rocSOLVER dsyevd vs LAPACK (NumPy): accuracy and time.
Run on the GPU box: python tools/gpurun_sign.py [sizes...]"""
import sys, json, time
import numpy as np
sys.path.insert(0, ".")
import proxsdp_jl_amd as px
from proxsdp_jl_amd import binding as B


def svec(A):
    n = A.shape[0]
    iu = np.triu_indices(n)
    # packed upper triangle, column-major (gj*(gj+1)/2 + gi), off-diagonals * sqrt 2
    out = np.zeros(n * (n + 1) // 2)
    for j in range(n):
        out[j * (j + 1) // 2: j * (j + 1) // 2 + j + 1] = A[: j + 1, j] * np.sqrt(2.0)
        out[j * (j + 1) // 2 + j] = A[j, j]
    return out


def spectra(n, rng):
    Qm, _ = np.linalg.qr(rng.standard_normal((n, n)))
    cases = {}
    lam = rng.standard_normal(n) * 3
    cases["gauss"] = lam
    lam = -np.abs(rng.standard_normal(n)); k = max(2, n // 6); lam[:k] = np.abs(rng.standard_normal(k)) * 20 + 1
    cases["lowrank_pos"] = lam
    lam = rng.standard_normal(n); lam[: n // 4] = 0.0
    cases["zeros"] = lam
    lam = rng.standard_normal(n); lam[:5] = 36.83154802; lam[5:10] = -2.5; lam[10:14] = [1e-12, -1e-12, 3e-9, -3e-9]
    cases["degenerate_tiny"] = lam
    return Qm, cases


res = {}
sizes = [int(a) for a in sys.argv[1:]] or [100, 501, 1000, 2000]
for n in sizes:
    rng = np.random.default_rng(n)
    Qm, cases = spectra(n, rng)
    for name, lam in cases.items():
        A = (Qm * lam) @ Qm.T
        A = 0.5 * (A + A.T)
        w, V = np.linalg.eigh(A)
        ref = (V * np.maximum(w, 0)) @ V.T
        xp = svec(A)
        o1, ms1, r1, p1 = B.full_eig_kernel(xp, n, sign=1, repeat=5)
        o0, ms0, r0, _ = B.full_eig_kernel(xp, n, sign=0, repeat=3)
        rp = svec(ref)
        sc = np.abs(w).max()
        e1 = np.abs(o1 - rp).max() / sc
        e0 = np.abs(o0 - rp).max() / sc
        res[f"{n}:{name}"] = dict(n=n, sign_ms=ms1, dsyevd_ms=ms0, err_sign=e1, err_dsyevd=e0, npos=int((w > 0).sum()),
                                  rank_sign=r1, rank_dsyevd=r0, products=p1,
                                  tflops=p1 * (2.0 * (64 * ((n + 63) // 64)) ** 3) / 2 / (ms1 * 1e-3) / 1e12)
        print(n, name, res[f"{n}:{name}"], flush=True)
json.dump(res, open("gpurun_out/sign.json", "w"), indent=1)
