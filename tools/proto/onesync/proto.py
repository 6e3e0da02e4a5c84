"""NumPy prototype of a ONE-SYNC-PER-STEP Lanczos (predicted shift, posterior correction) against the oracle's KrylovKit restatement.

Model of the device schedule: launch L_k reads the PUBLISHED vector g_k (global, written by L_{k-1}) and the REDUCED records of
dots taken in L_{k-1} (h = V_k' g_k, |g_k|^2).  Row-local in L_k: g_k, V_k, A g_k (gather on the global g_k).  Everything L_k
forms is a combination of those:
    g_k     = A v_k - sigma_k v_k - V_{k-1} tcol_k          (sigma_k: a PREDICTION of alpha_k; tcol_k: known coupling column)
    alpha_k = sigma_k + h[k]
    beta_k  = sqrt(|g|^2 - |h|^2)                            (Pythagorean)
    v_{k+1} = (g_k - V_k h) / beta_k
    A v_{k+1} = (A g_k - h[k] (g_k + sigma_k v_k + V_{k-1} tcol_k) - V_k T[:, :k-1] h[:k-1]) / beta_k     (Lanczos relation for j < k)
    g_{k+1} = A v_{k+1} - sigma_{k+1} v_{k+1} - beta_k v_k   -> published;  dots V_{k+1}' g_{k+1}, |g_{k+1}|^2 -> records
"""
import os, sys, json, time
import numpy as np, scipy.linalg
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
from oracle import eig as oeig


def onesync_eigsolve(matvec, x0, howmany, krylovdim, maxiter, tol, predictor="last", sigma_table=None, log=None, second_pass=False):
    n = x0.shape[0]
    V = np.zeros((n, krylovdim + 1))
    T = np.zeros((krylovdim + 1, krylovdim + 1))
    v = x0 / np.linalg.norm(x0)
    V[:, 0] = v
    numops = 0
    numiter = 1
    K = 0              # number of CLOSED basis vectors whose alpha is known; V has K+1 columns while g is pending
    keep = 0
    converged = 0
    cyc = 0
    sig_used = {}
    last_alpha = 0.0
    amp_log = []

    def predict(cycle, k):
        if predictor == "zero":
            return 0.0
        if predictor == "table" and sigma_table is not None and (cycle, k) in sigma_table:
            return sigma_table[(cycle, k)]
        return last_alpha

    while True:
        # ---- "first" launch of a cycle: v_{k} (column kcur) is global
        kcur = keep if cyc > 0 else 0          # index of the newest basis column
        tcol = T[:kcur, kcur].copy()           # coupling (zeros at start; f after a restart)
        sig = predict(cyc, kcur)
        Av = matvec(V[:, kcur]); numops += 1
        g = Av - sig * V[:, kcur] - V[:, :kcur] @ tcol
        while True:
            # ---- launch boundary: reduce records of g against V[:, :kcur+1]
            h = V[:, :kcur + 1].T @ g
            gg = float(g @ g)
            alpha = sig + h[kcur]
            last_alpha = alpha
            sig_used[(cyc, kcur)] = alpha
            b2 = gg - float(h @ h)
            beta = np.sqrt(max(b2, 0.0))
            T[kcur, kcur] = alpha
            amp_log.append((cyc, kcur, abs(h[kcur]) / max(beta, 1e-300), np.sqrt(gg) / max(beta, 1e-300)))
            Kfull = kcur + 1
            vnew = (g - V[:, :Kfull] @ h) / beta
            if second_pass:                     # (diagnostic only: an immediate second pass, not available on the one-sync schedule)
                c = V[:, :Kfull].T @ vnew
                vnew = vnew - V[:, :Kfull] @ c
                vnew /= np.linalg.norm(vnew)
            if Kfull == krylovdim or beta <= tol:
                V[:, Kfull] = vnew
                break
            # ---- continue: A v_{new} from A g by linearity
            Ag = matvec(g); numops += 1
            hk = h[kcur]
            corr = hk * (g + sig * V[:, kcur] + V[:, :kcur] @ tcol)
            if kcur > 0:
                corr = corr + V[:, :Kfull] @ (T[:Kfull, :kcur] @ h[:kcur])
            Avnew = (Ag - corr) / beta
            V[:, Kfull] = vnew
            T[Kfull, kcur] = T[kcur, Kfull] = beta
            sig_n = predict(cyc, Kfull)
            g = Avnew - sig_n * vnew - beta * V[:, kcur]
            tcol = np.zeros(Kfull); tcol[kcur] = beta
            sig = sig_n
            kcur = Kfull
        K = Kfull
        # ---- host: eigensolve of T[:K,:K], convergence
        Dasc, Uasc = scipy.linalg.eigh(T[:K, :K])
        D = Dasc[::-1].copy(); U = Uasc[:, ::-1].copy()
        f = beta * U[K - 1, :]
        converged = 0
        while converged < K and abs(f[converged]) <= tol:
            converged += 1
        if log is not None:
            Vk = V[:, :K + 1]
            orth = np.abs(Vk.T @ Vk - np.eye(K + 1)).max()
            log.append(dict(cycle=cyc, K=K, converged=converged, orth=orth, f=np.abs(f[:howmany + 3]).tolist(), beta=beta))
        if converged >= howmany or numiter == maxiter or beta <= tol:
            break
        keepn = (3 * krylovdim + 2 * converged) // 5
        V[:, :keepn] = V[:, :K] @ U[:, :keepn]
        V[:, keepn] = V[:, K]
        T[:, :] = 0.0
        T[np.arange(keepn), np.arange(keepn)] = D[:keepn]
        T[keepn, :keepn] = f[:keepn]; T[:keepn, keepn] = f[:keepn]
        keep = keepn
        cyc += 1
        numiter += 1
    if converged > howmany:
        howmany = converged
    vals = D[:howmany].copy()
    vecs = V[:, :K] @ U[:, :howmany]
    return vals, vecs, converged, numiter, numops, dict(V=V[:, :K + 1].copy(), T=T[:K, :K].copy(), beta=beta, sig=sig_used, amp=amp_log)


def report(tag, X, nev, K, x0, predictors, tol=1e-12, maxiter=100):
    mv = lambda v: X @ v
    t0 = time.time()
    rv, rvec, rconv, rit, rops = oeig.krylovkit_eigsolve(mv, x0, nev, K, maxiter, tol, False)
    nrm = np.abs(rv).max()
    print(f"[{tag}] n={X.shape[0]} nev={nev} K={K} |A|~{nrm:.1f} eps|A|={nrm*1.1e-16:.2e}")
    print(f"  oracle      : converged={rconv} restarts={rit-1} matvecs={rops}  ({time.time()-t0:.1f}s)")
    out = {}
    table = None
    for pred in predictors:
        log = []
        name = pred
        kw = {}
        if pred == "table":
            kw = dict(sigma_table=table)
        vals, vecs, conv, it, ops, ex = onesync_eigsolve(mv, x0, nev, K, maxiter, tol, predictor=pred, log=log, **kw)
        if table is None:
            table = ex["sig"]
        # true residuals of the returned pairs
        R = X @ vecs - vecs * vals
        res = np.linalg.norm(R, axis=0)
        k = min(len(vals), len(rv))
        dv = np.abs(vals[:k] - rv[:k]).max()
        amp = np.array([a[2] for a in ex["amp"]]); amp2 = np.array([a[3] for a in ex["amp"]])
        Vk = ex["V"]
        orth = np.abs(Vk.T @ Vk - np.eye(Vk.shape[1])).max()
        print(f"  onesync/{name:5s}: converged={conv} restarts={it-1} launches~matvecs={ops}  max|dval|={dv:.2e} max true res={res[:nev].max():.2e} "
              f"orth(last cycle)={orth:.2e}  |delta|/beta max={amp.max():.1f} med={np.median(amp):.2f}  |g|/beta max={amp2.max():.1f}")
        for l in log[:6]:
            print(f"      cycle {l['cycle']}: K={l['K']} conv={l['converged']} orth={l['orth']:.2e}")
        out[name] = dict(converged=conv, restarts=it - 1, ops=ops, dval=dv, res=float(res[:nev].max()), orth=float(orth), amp_max=float(amp.max()))
    return dict(oracle=dict(converged=rconv, restarts=rit - 1, matvecs=rops), onesync=out)


if __name__ == "__main__":
    n = 4000
    x0 = oeig.start_vector(n, 1234, 3)
    res = {}
    for tag, it, nev, K in (("head", 251, 63, 127), ("head", 252, 63, 127), ("kry", 1001, 5, 25), ("kry", 1002, 5, 25)):
        X = np.load(f"/tmp/os/X_{tag}_{it}.npy")
        res[f"{tag}_{it}"] = report(f"{tag}_{it}", X, nev, K, x0, ["zero", "last", "table"])
    json.dump(res, open("/tmp/os/proto_results.json", "w"), indent=1)
