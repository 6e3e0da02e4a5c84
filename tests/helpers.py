"""Shared test helpers: planted-spectrum matrices in the solver's packed svec
form, and single-block projection through the CPU oracle."""
import math

import numpy as np

from oracle import Options, eig as oeig, pdhg as opdhg


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


