"""Shared test helpers: planted-spectrum matrices in the solver's packed svec
form, and single-block projection through the CPU oracle."""
import math

import numpy as np

from oracle import Options, eig as oeig, pdhg as opdhg
from proxsdp_jl_amd.state_io import (tri_indices, smat, svec, compact_state, expand_state,          # noqa: F401  (re-exported)
                                     save_compact_state, load_compact_state)


def planted_packed(n, seed, top, bulk=(-3.0, 0.5)):
    """Random symmetric matrix with a planted spectrum, in the solver's packed
    svec form (off-diagonals x sqrt 2)."""
    rng = np.random.default_rng(seed)
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    k = min(len(top), n)
    lam = np.concatenate([np.asarray(top[:k], float), rng.uniform(bulk[0], bulk[1], n - k)])
    X = (Q * lam) @ Q.T
    X = (X + X.T) / 2
    jj = np.repeat(np.arange(n), np.arange(1, n + 1))
    ii = np.concatenate([np.arange(j + 1) for j in range(n)])
    return np.where(ii == jj, X[ii, jj], X[ii, jj] * math.sqrt(2.0))


def oracle_project(packed, n, target_rank, full):
    """psd_projection! of one block through the oracle."""
    opt = Options()
    opt.full_eig_decomp = bool(full)
    opt.min_size_krylov_eigs = 0
    opt.max_target_rank_krylov_eigs = max(16, target_rank)
    cones = opdhg.ConicSets([opdhg.SDPSet(np.arange(len(packed)), len(packed), n)], [])

    class A:
        pass
    aff = A()
    aff.n, aff.p, aff.m = len(packed), 0, 0
    a = opdhg.Aux(aff, cones)
    opdhg._setup_blocks(a, cones)
    p = opdhg.Params()
    p.iter = 1
    p.target_rank = np.array([target_rank])
    p.current_rank = np.array([0])
    p.min_eig = np.zeros(1)
    p.stats = {"full_eigs": 0, "krylov_fallbacks": 0}
    arc = [oeig.EigSolverAlloc(n, opt)]
    v = packed.copy()
    opdhg.psd_projection(v, a, cones, opt, p, arc, 1)
    return v, int(p.current_rank[0]), float(p.min_eig[0]), arc[0]


PROJ_CASES = [  # name, n, seed, top eigenvalues, target_rank, full
    ("n3_full", 3, 1, [2.0, 0.5], 2, True),
    ("n7_full", 7, 2, [5.0, 3.0, 1.0], 2, True),
    ("n7_lanczos", 7, 2, [5.0, 3.0, 1.0], 2, False),
    ("n101_lanczos_r2", 101, 3, [40.0, 25.0, 9.0, 4.0], 2, False),
    ("n101_lanczos_r6", 101, 3, [40.0, 25.0, 9.0, 4.0], 6, False),
    ("n101_full", 101, 3, [40.0, 25.0, 9.0, 4.0], 2, True),
    ("n257_lanczos_r4", 257, 4, [90.0, 60.0, 33.0, 12.0, 5.0], 4, False),
    ("n257_full", 257, 4, [90.0, 60.0, 33.0, 12.0, 5.0], 4, True),
]


def capture_restart_projections(pr, iters, count, all_after_first=False, any_iter=False, **optkw):
    """Run the oracle for `iters` PDHG iterations on a single-PSD-block problem and capture the
    projection input / output of the first `count` iterations whose Lanczos needed a thick restart
    (through the oracle's proj_callback test hook).  all_after_first: capture every iteration from
    the first restart on, restart or not."""
    import oracle
    n = pr.psd_sides()[0]
    N = n * (n + 1) // 2
    caps = []
    prev = [0, 0]

    def cb(it, xin, xout, p, arc_list):
        arc = arc_list[0]
        mv, rs = arc.matvecs - prev[0], arc.restarts - prev[1]
        prev[0], prev[1] = arc.matvecs, arc.restarts
        if (rs > 0 or any_iter or (all_after_first and caps)) and len(caps) < count and arc.converged:
            tr = int(p.target_rank[0])
            k = min(tr, arc.converged_eigs)
            vals = np.array(arc.vals[:k])                     # (ARPACK path: ascending, all nev used)
            pos = vals > 0.0
            w = np.linalg.eigvalsh(smat(xin[:N], n))[::-1]
            caps.append(dict(n=n, iter=it, target_rank=tr, rank=int(p.current_rank[0]), min_eig=float(p.min_eig[0]),
                             matvecs=int(mv), restarts=int(rs), converged_eigs=int(arc.converged_eigs),
                             x_in=xin[:N].copy(), x_out=xout[:N].copy(), vals=vals[pos].copy(),
                             vecs=np.array(arc.vecs[:, :k][:, pos]), top=w[:tr + 3].copy()))

    o = Options()
    o.max_iter = iters
    for k_, v_ in optkw.items():
        setattr(o, k_, v_)
    oracle.solve(pr, o, proj_callback=cb)
    return caps


def check_truncated_projection(x_in, n, target_rank, out, ref_out, tol_rel=1e-9, gap_rel=1e-8):
    """The reference's Lanczos projection keeps the top `target_rank` eigenpairs (positive part,
    prox_operators.jl:99-106).  Two correct eigensolvers agree on it to ~tol/gap; when the input has
    lambda_r == lambda_{r+1} (to gap_rel*|X|) the truncation is only defined up to a rotation inside
    that eigenspace.  Returns "tight" when `out` matches `ref_out` to tol_rel*|X|; otherwise asserts,
    with LAPACK on the input, that (i) the cluster really is degenerate and (ii) `out` equals the
    well-separated part exactly plus lambda_cluster times an orthogonal projector of the right rank
    INSIDE the degenerate eigenspace -- and returns "degenerate"."""
    X = smat(x_in, n)
    nx = np.linalg.norm(X)
    if np.linalg.norm(out - ref_out) <= tol_rel * nx:
        return "tight"
    w, Q = np.linalg.eigh(X)
    w, Q = w[::-1], Q[:, ::-1]
    r = target_rank
    lam = w[r - 1]
    assert abs(w[r - 1] - w[r]) <= gap_rel * nx, ("projections differ on a non-degenerate input",
                                                 np.linalg.norm(out - ref_out) / nx, w[:r + 3])
    cl = np.where(np.abs(w - lam) <= 10 * gap_rel * nx)[0]          # the degenerate cluster
    lo = cl.min()                                                  # eigenpairs above it are well separated
    take = r - lo                                                  # how many vectors of the cluster survive
    sep = (Q[:, :lo] * np.maximum(w[:lo], 0.0)) @ Q[:, :lo].T
    R = smat(out, n) - sep
    if lam <= 0.0:
        assert np.linalg.norm(R) <= 1e-7 * nx
        return "degenerate"
    Pc = Q[:, cl] @ Q[:, cl].T
    assert np.linalg.norm(R - Pc @ R @ Pc) <= 1e-7 * nx, "remainder leaves the degenerate eigenspace"
    ev = np.linalg.eigvalsh(R)[::-1] / lam
    assert np.allclose(ev[:take], 1.0, atol=1e-6) and np.abs(ev[take:]).max() <= 1e-6, ev[:take + 2]
    return "degenerate"


# ----------------------------------------------------------------- solver-state fixtures (proxsdp_state / oracle.export_state)


