"""Scheme B of the one-sync prototype: the late alpha is applied AT THE GATHER.
L_{k+1} reads two global vectors -- the published g_k and the basis column v_k it was formed against -- and applies the operator to
    w = g_k - h[k] v_k            (h[k] = alpha_k - sigma_k, exact, from the records)
so that A v_{k+1} = (A w - V_{k-1}-terms by the Lanczos relation) / beta_k: no identity built on a previously COMPUTED A v is
reused, the error of a step stays local (factor |g|/beta on eps |A|), nothing compounds."""
import os, sys, json, time
import numpy as np, scipy.linalg
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
from oracle import eig as oeig


def onesync_b(matvec, x0, howmany, krylovdim, maxiter, tol, predictor="last", sigma_table=None, log=None, lagnorm=True, f32_small=False):
    n = x0.shape[0]
    V = np.zeros((n, krylovdim + 1))
    T = np.zeros((krylovdim + 1, krylovdim + 1))
    V[:, 0] = x0 / np.linalg.norm(x0)
    numops = 0; numiter = 1; keep = 0; cyc = 0; converged = 0
    sig_used = {}; last_alpha = 0.0; amp_log = []

    def predict(cycle, k):
        if predictor == "zero":
            return 0.0
        if predictor == "table" and sigma_table is not None and (cycle, k) in sigma_table:
            return sigma_table[(cycle, k)]
        return last_alpha

    while True:
        kcur = keep if cyc > 0 else 0
        tcol = T[:kcur, kcur].copy()
        sig = predict(cyc, kcur)
        Av = matvec(V[:, kcur]); numops += 1
        g = Av - sig * V[:, kcur] - V[:, :kcur] @ tcol
        while True:
            Kfull = kcur + 1
            h = V[:, :Kfull].T @ g
            gg = float(g @ g)
            hk = h[kcur]
            alpha = sig + hk
            last_alpha = alpha
            sig_used[(cyc, kcur)] = alpha
            beta = np.sqrt(max(gg - float(h @ h), 0.0))
            T[kcur, kcur] = alpha
            amp_log.append((cyc, kcur, abs(hk) / max(beta, 1e-300)))
            vnew = (g - V[:, :Kfull] @ h) / beta
            if lagnorm:
                # lagged normalisation: |vnew| is measured in the launch that forms it and the scale is folded into beta / the stored column by the
                # NEXT launch; here simply applied (the prototype checks the numerics of the recurrence, not the bookkeeping)
                nu = np.linalg.norm(vnew)
                vnew = vnew / nu
                beta = beta * nu
            if Kfull == krylovdim or beta <= tol:
                V[:, Kfull] = vnew
                break
            # operator on w = g - hk v_k : both global.  (A w) = beta A v_new + A V_{k-1} h[:k-1]
            w = g - hk * V[:, kcur]
            Aw = matvec(w); numops += 1
            corr = 0.0
            if kcur > 0:
                corr = V[:, :Kfull] @ (T[:Kfull, :kcur] @ h[:kcur])     # rounding-level coefficients x known relation
            Avnew = (Aw - corr) / beta
            V[:, Kfull] = vnew
            T[Kfull, kcur] = T[kcur, Kfull] = beta
            sig = predict(cyc, Kfull)
            g = Avnew - sig * vnew - beta * V[:, kcur]
            kcur = Kfull
        K = Kfull
        Dasc, Uasc = scipy.linalg.eigh(T[:K, :K])
        D = Dasc[::-1].copy(); U = Uasc[:, ::-1].copy()
        f = beta * U[K - 1, :]
        converged = 0
        while converged < K and abs(f[converged]) <= tol:
            converged += 1
        if log is not None:
            Vk = V[:, :K + 1]
            log.append(dict(cycle=cyc, K=K, converged=converged, orth=float(np.abs(Vk.T @ Vk - np.eye(K + 1)).max())))
        if converged >= howmany or numiter == maxiter or beta <= tol:
            break
        keepn = (3 * krylovdim + 2 * converged) // 5
        V[:, :keepn] = V[:, :K] @ U[:, :keepn]
        V[:, keepn] = V[:, K]
        T[:, :] = 0.0
        T[np.arange(keepn), np.arange(keepn)] = D[:keepn]
        T[keepn, :keepn] = f[:keepn]; T[:keepn, keepn] = f[:keepn]
        keep = keepn; cyc += 1; numiter += 1
    if converged > howmany:
        howmany = converged
    vals = D[:howmany].copy()
    vecs = V[:, :K] @ U[:, :howmany]
    return vals, vecs, converged, numiter, numops, dict(V=V[:, :K + 1].copy(), T=T[:K, :K].copy(), beta=beta, sig=sig_used, amp=amp_log)


def compare(tag, X, nev, K, x0, table=None, tol=1e-12, preds=("zero", "last", "table")):
    mv = lambda v: X @ v
    rv, rvec, rconv, rit, rops = oeig.krylovkit_eigsolve(mv, x0, nev, K, 100, tol, False)
    Rr = np.linalg.norm(X @ rvec - rvec * rv, axis=0)
    nrm = np.abs(np.linalg.eigvalsh(X)[[0, -1]]).max() if X.shape[0] <= 1500 else max(abs(rv[0]), 1.0)
    print(f"[{tag}] nev={nev} K={K} |A|~{nrm:.1f} eps|A|={nrm*1.1e-16:.1e}   oracle: conv={rconv} restarts={rit-1} matvecs={rops} max res={Rr[:nev].max():.2e}")
    out = dict(oracle=dict(converged=int(rconv), restarts=int(rit - 1), matvecs=int(rops), res=float(Rr[:nev].max())))
    for pred in preds:
        if pred == "table" and table is None:
            continue
        log = []
        vals, vecs, conv, it, ops, ex = onesync_b(mv, x0, nev, K, 100, tol, predictor=pred, sigma_table=table, log=log)
        res = np.linalg.norm(X @ vecs - vecs * vals, axis=0)
        k = min(len(vals), len(rv), nev)
        amp = np.array([a[2] for a in ex["amp"]])
        orth = max(l["orth"] for l in log)
        print(f"   B/{pred:5s}: conv={conv} restarts={it-1} ops={ops} dval={np.abs(vals[:k]-rv[:k]).max():.2e} res={res[:k].max():.2e} orth(max over cycles)={orth:.2e} "
              f"|delta|/beta max={amp.max():.1f} med={np.median(amp):.2f}")
        out[pred] = dict(converged=int(conv), restarts=int(it - 1), ops=int(ops), dval=float(np.abs(vals[:k]-rv[:k]).max()), res=float(res[:k].max()),
                         orth=float(orth), amp_max=float(amp.max()), amp_med=float(np.median(amp)), sig=ex["sig"])
    return out


if __name__ == "__main__":
    n = 4000
    x0 = oeig.start_vector(n, 1234, 3)
    prev = {}
    for tag, it, nev, K in (("head", 251, 63, 127), ("head", 252, 63, 127), ("head", 253, 63, 127), ("kry", 1001, 5, 25), ("kry", 1002, 5, 25), ("kry", 1003, 5, 25)):
        X = np.load(f"/tmp/os/X_{tag}_{it}.npy")
        r = compare(f"{tag}_{it}", X, nev, K, x0, table=prev.get(tag))
        prev[tag] = r["last"]["sig"]
