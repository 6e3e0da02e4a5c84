"""Solver-state fixtures (include/proxsdp_hip.h `proxsdp_state`, the state seam of `proxsdp_hip_solve_ex`) in a compact form,
and the packed svec <-> symmetric matrix maps they need.  Lives in the package (round 6) because `bench.py`'s
steady-window CPU leg loads a committed state too; `tests/helpers.py` re-exports these names."""
import math

import numpy as np


def tri_indices(n):
    jj = np.repeat(np.arange(n), np.arange(1, n + 1))
    ii = np.concatenate([np.arange(j + 1) for j in range(n)]) if n else np.zeros(0, int)
    return ii, jj


def smat(packed, n):
    """packed svec (off-diagonals x sqrt 2) -> full symmetric matrix
    (psd_vec_to_square, /root/reference/src/prox_operators.jl:1-16, plus mirror)."""
    ii, jj = tri_indices(n)
    X = np.zeros((n, n))
    vals = np.where(ii == jj, packed, packed / math.sqrt(2.0))
    X[ii, jj] = vals
    X[jj, ii] = vals
    return X


def svec(X):
    n = X.shape[0]
    ii, jj = tri_indices(n)
    return np.where(ii == jj, X[ii, jj], X[ii, jj] * math.sqrt(2.0))


def compact_state(state, sides, rel=1e-13):
    """A solver state (include/proxsdp_hip.h proxsdp_state, oracle.pdhg.export_state) as a SMALL fixture: every PSD
    block of x is stored by its eigen-factors (the iterate is a projection: rank <= a few dozen), M'y by its
    non-zeros.  `sides`: the PSD block sides, in solver order.  expand_state() rebuilds the dense vectors; the
    round trip is exact to ~1e-16 |x| (checked by the generator), which is all a fixture needs -- both sides of a
    parity test start from the SAME expanded state."""
    x = np.asarray(state["x"], float)
    out = {k: v for k, v in state.items() if k not in ("x", "Mty")}
    off = 0
    fac = []
    for n in sides:
        N = n * (n + 1) // 2
        w, Q = np.linalg.eigh(smat(x[off:off + N], n))
        keep = np.abs(w) > rel * max(1e-300, np.abs(w).max())
        fac.append((w[keep].copy(), Q[:, keep].copy()))
        off += N
    out["x_factors"] = fac
    out["x_tail"] = x[off:].copy()
    nz = np.flatnonzero(np.asarray(state["Mty"]))
    out["Mty_idx"] = nz.astype(np.int64)
    out["Mty_val"] = np.asarray(state["Mty"], float)[nz].copy()
    out["n"] = len(x)
    out["sides"] = np.asarray(sides, np.int64)
    return out


def expand_state(c):
    x = np.zeros(int(c["n"]))
    off = 0
    for n, (w, Q) in zip(c["sides"], c["x_factors"]):
        n = int(n)
        N = n * (n + 1) // 2
        x[off:off + N] = svec((Q * w) @ Q.T)
        off += N
    x[off:] = c["x_tail"]
    Mty = np.zeros(int(c["n"]))
    Mty[c["Mty_idx"]] = c["Mty_val"]
    out = {k: v for k, v in c.items() if k not in ("x_factors", "x_tail", "Mty_idx", "Mty_val", "n", "sides")}
    out["x"], out["Mty"] = x, Mty
    return out


def save_compact_state(path, c):
    flat = {k: v for k, v in c.items() if k != "x_factors"}
    for b, (w, Q) in enumerate(c["x_factors"]):
        flat[f"xw{b}"], flat[f"xQ{b}"] = w, Q
    np.savez_compressed(path, **flat)


def load_compact_state(path):
    z = np.load(path)
    c = {}
    for k in z.files:
        if k.startswith("xw") or k.startswith("xQ"):
            continue
        v = z[k]
        c[k] = v if v.ndim else v.item()
    c["x_factors"] = [(z[f"xw{b}"], z[f"xQ{b}"]) for b in range(len(c["sides"]))]
    return c
