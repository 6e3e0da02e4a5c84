"""Prototype (NumPy, CPU): warm-started block Chebyshev-filtered subspace iteration for the positive part of the
projection input in the implicit full_eig! regime -- how many BLOCK operator applications does a projection need when it
starts from the previous projection's positive Ritz basis?    Inputs: tools/proto/dump_phase2_inputs.py."""
import sys, glob, math
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from helpers import smat

its = sorted(int(f.split("_")[-1][:-4]) for f in glob.glob("/tmp/phase2_xin_*.npy"))
n = 4000
Z = {it: smat(np.load(f"/tmp/phase2_xin_{it}.npy"), n) for it in its}
print("iterations", its)
ex = {}
for it in its:
    w, Q = np.linalg.eigh(Z[it])
    ex[it] = (w, Q)
    pos = w[w > 0]
    print(it, "npos", len(pos), "smallest positives", pos[:4], "largest", pos[-1], "most negative", w[0],
          "top of the negative part", w[w <= 0][-5:])


def chefsi(A, V0, lo, cut, tol, max_apply=400, deg=8, label=""):
    """block CheFSI: damp [lo, cut]; Rayleigh-Ritz after every `deg` applications"""
    n, wd = V0.shape
    V, _ = np.linalg.qr(V0)
    applies = 0
    hist = []
    while applies < max_apply:
        e = (cut - lo) / 2.0
        c = (cut + lo) / 2.0
        # scaled Chebyshev (Zhou-Saad): sigma recursion keeps the wanted end at O(1)
        top = np.max(np.einsum("ij,ij->j", V, A @ V)); applies_rr = 1   # (the R-R of the previous round gives this for free)
        sigma1 = e / (top - c) if top > c + e else 0.5
        sigma = sigma1
        Y = (A @ V - c * V) * (sigma1 / e); applies += 1
        Vm = V
        for k in range(2, deg + 1):
            sn = 1.0 / (2.0 / sigma1 - sigma)
            Yn = (A @ Y - c * Y) * (2.0 * sn / e) - (sigma * sn) * Vm; applies += 1
            Vm, Y, sigma = Y, Yn, sn
        V, _ = np.linalg.qr(Y)
        AV = A @ V; applies += 1
        H = V.T @ AV
        th, S = np.linalg.eigh((H + H.T) / 2)
        V = V @ S; AV = AV @ S
        R = AV - V * th
        rn = np.linalg.norm(R, axis=0)
        posm = th > 0
        hist.append((applies, int(posm.sum()), float(rn[posm].max()) if posm.any() else 0.0, float(th[~posm].max()) if (~posm).any() else None))
        # new cut: largest non-positive Ritz value (or a bit below the smallest positive)
        if (~posm).any():
            cut = max(th[~posm].max(), lo + 1e-3 * (top - lo))
        if posm.any() and rn[posm].max() <= tol and (~posm).any():
            break
    return th, V, rn, hist, applies


for a, b in zip(its[:-1], its[1:]):
    w0, Q0 = ex[a]
    w1, Q1 = ex[b]
    F = Q0[:, w0 > 0]                         # the previous projection's positive eigenvectors (exact stand-in for its Ritz basis)
    r = F.shape[1]
    npos1 = int((w1 > 0).sum())
    scale = max(abs(w1[0]), abs(w1[-1]))
    sub = np.linalg.norm(Q1[:, w1 > 0] - F @ (F.T @ Q1[:, w1 > 0]), axis=0)
    print(f"\n{a}->{b}: prev positives {r}, new positives {npos1}, scale {scale:.3e}; distance of the new positive eigvecs from span(F): max {sub.max():.2e}")
    rng = np.random.default_rng(0)
    for g in (4, 8, 16):
        for deg in (4, 8, 12):
            G = rng.standard_normal((n, g))
            V0 = np.concatenate([F, G], axis=1)
            lo = w1[0] * 1.01                     # (a Lanczos bound in the real thing)
            cut = 0.0
            th, V, rn, hist, applies = chefsi(Z[b], V0, lo, cut, tol=1e-11 * scale, deg=deg)
            npos = int((th > 0).sum())
            ok = npos == npos1 and np.allclose(np.sort(th[th > 0]), np.sort(w1[w1 > 0]), rtol=0, atol=1e-9 * scale)
            print(f"  guards {g:2d} degree {deg:2d}: {applies:3d} block applications, positives {npos} (exact {npos1}) ok={ok}; history {[(h[0], h[1], '%.1e' % h[2]) for h in hist]}")
