"""Table for profiles/r06_onesync_prototype.md: the two one-boundary-per-step Lanczos schemes (proto.py: correction through the
Lanczos relation; proto_b.py: the late alpha applied at the gather) under three predictors of alpha, against the oracle's
KrylovKit restatement, on projection inputs of the metric instance (capture_inputs.py -> /tmp/os)."""
import sys, os, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import proto, proto_b
from oracle import eig as oeig
np.seterr(all="ignore")
n = 4000
x0 = oeig.start_vector(n, 1234, 3)
rows = []
def one(tag, X, nev, K, table):
    mv = lambda v: X @ v
    rv, rvec, rconv, rit, rops = oeig.krylovkit_eigsolve(mv, x0, nev, K, 100, 1e-12, False)
    rres = np.linalg.norm(X @ rvec - rvec * rv, axis=0)[:nev].max()
    rows.append(f"| {tag} | KrylovKit restatement (two syncs per step) | - | {rconv} | {rit-1} | {rops} | - | {rres:.1e} | - | - |")
    out = {}
    for scheme, fn in (("A: relation-corrected", proto.onesync_eigsolve), ("B: alpha at the gather", proto_b.onesync_b)):
        for pred in ("zero", "last", "table"):
            if pred == "table" and table is None:
                continue
            log = []
            try:
                vals, vecs, conv, it, ops, ex = fn(mv, x0, nev, K, 100, 1e-12, predictor=pred, sigma_table=table, log=log)
                ok = np.all(np.isfinite(vals))
            except Exception:
                ok = False
            if not ok:
                rows.append(f"| {tag} | {scheme} | {pred} | breakdown (beta -> 0 / NaN) | | | | | | |")
                continue
            k = min(len(vals), len(rv), nev)
            res = np.linalg.norm(X @ vecs - vecs * vals, axis=0)[:k].max()
            amp = np.array([a[2] for a in ex["amp"]])
            orth = max(l["orth"] for l in log)
            rows.append(f"| {tag} | {scheme} | {pred} | {conv} | {it-1} | {ops} | {np.abs(vals[:k]-rv[:k]).max():.1e} | {res:.1e} | {orth:.1e} | {amp.max():.1f} / {np.median(amp):.2f} |")
            out[(scheme, pred)] = ex["sig"]
            print(rows[-1], flush=True)
    return out.get(("B: alpha at the gather", "last"))
prev = {}
for tag, it, nev, K in (("headline window, iteration 251 (rank 63, K 127)", ("head", 251), 63, 127), ("headline window, iteration 252", ("head", 252), 63, 127),
                        ("default options, iteration 1001 (rank 5, K 25)", ("kry", 1001), 5, 25), ("default options, iteration 1002", ("kry", 1002), 5, 25)):
    X = np.load(f"/tmp/os/X_{it[0]}_{it[1]}.npy")
    # the table predictor: alpha_k measured on the PREVIOUS PDHG iteration's projection (exact there: fixed point of the scheme)
    tab = prev.get(it[0])
    prev[it[0]] = None
    r = one(tag, X, nev, K, tab)
    # exact alphas of THIS matrix for the next iteration's table: a correct run (the oracle's T diagonal is what a converged table holds)
    mv = lambda v: X @ v
    t = None
    for _ in range(4):
        try:
            vals, vecs, conv, itn, ops, ex = proto_b.onesync_b(mv, x0, nev, K, 100, 1e-12, predictor="table" if t else "last", sigma_table=t, log=[])
            t = ex["sig"]
        except Exception:
            break
    prev[it[0]] = t
print("\n".join(rows))
open("/tmp/os/onesync_table.md", "w").write("\n".join(rows) + "\n")
