"""Standard-form problem container and instance generators (host side).

The reference reaches its solver through MathOptInterface: `_optimize!`
(/root/reference/src/MOI_wrapper.jl:220-293) turns a JuMP/MOI model into
`AffineSets(n, p, m, A, G, b, h, c)` + `ConicSets(sdpcone, socone)` and calls
`chambolle_pock` (:310).  There is no Julia in this image, so this module builds
that same standard form directly, following what MOI produces for the
reference's own instance builders:

  * Max-Cut            README.md:62-86 (n=4 W verbatim) and Erdos-Renyi graphs
  * randSDP            test/base_randsdp.jl:4-23 + test/moi_randsdp.jl
  * MIMO               test/base_mimo.jl:3-17   + test/moi_mimo.jl
  * SDPLIB (.dat-s)    test/base_sdplib.jl:1-45 + test/moi_sdplib.jl

Julia's MersenneTwister streams are not reproducible outside Julia, so the
generators own their seeds (numpy.random.default_rng); the same arrays are fed
to the CPU oracle and to the HIP library.

Conventions: a PSD variable of side n is the MOI
`PositiveSemidefiniteConeTriangle`: n(n+1)/2 scalar variables, upper triangle
column by column, variable of entry (i<=j) at j(j+1)/2+i (0-based).  A term
`sum_ij F_ij X_ij` over the full square puts F_ii on a diagonal variable and
2*F_ij on an off-diagonal one.
"""
from dataclasses import dataclass, field

import numpy as np
import scipy.sparse as sp


def sympackedlen(n):
    """sympackedlen (MOI_wrapper.jl:218)."""
    return n * (n + 1) // 2


def tri_index(i, j):
    """0-based position of entry (i<=j) in the column-major upper triangle."""
    i, j = np.minimum(i, j), np.maximum(i, j)
    return j * (j + 1) // 2 + i


@dataclass
class Problem:
    """What `_optimize!` hands to chambolle_pock, plus the objective fix-up data
    of MOI_wrapper.jl:252-255,336-337."""
    n: int                       # number of scalar variables
    A: sp.csc_matrix             # p x n   (Zeros rows:        A x = b)
    b: np.ndarray
    G: sp.csc_matrix             # m x n   (Nonpositives rows: G x <= h)
    h: np.ndarray
    c: np.ndarray                # minimisation objective (already sign-flipped for MAX)
    psd: list = field(default_factory=list)    # list of int64 index arrays (0-based), triangle order
    soc: list = field(default_factory=list)    # list of int64 index arrays (0-based)
    max_sense: bool = False
    objective_constant: float = 0.0
    name: str = ""
    # optional dense A (row-major p x n): a C-contiguous float64 numpy array (host) or a torch
    # tensor living on the solve's GPU (proxsdp_problem.M_dense); the sparse A then only
    # carries the shape (no stored entries), G stays sparse
    M_dense: object = None

    @property
    def p(self):
        return self.A.shape[0]

    @property
    def m(self):
        return self.G.shape[0]

    def psd_sides(self):
        out = []
        for v in self.psd:
            L = len(v)
            s = int((np.sqrt(8 * L + 1) - 1) // 2)
            assert s * (s + 1) // 2 == L, "not a triangular number"
            out.append(s)
        return out

    def user_objective(self, objval):
        """sol.objval = obj_sign*sol.objval + constant (MOI_wrapper.jl:336)."""
        return (-1.0 if self.max_sense else 1.0) * objval + self.objective_constant


def _empty(ncols):
    return sp.csc_matrix((0, ncols), dtype=float)


def _sym_coeff_vector(F):
    """Coefficients on the triangle variables of sum_ij F_ij X_ij, F symmetric
    (dense ndarray or scipy sparse)."""
    n = F.shape[0]
    N = sympackedlen(n)
    out = np.zeros(N)
    if sp.issparse(F):
        F = sp.triu(sp.coo_matrix(F))
        i, j, v = F.row.astype(np.int64), F.col.astype(np.int64), F.data
        np.add.at(out, tri_index(i, j), np.where(i == j, v, 2.0 * v))
    else:
        jj = np.repeat(np.arange(n), np.arange(1, n + 1))
        ii = np.concatenate([np.arange(j + 1) for j in range(n)])
        out = np.where(ii == jj, F[ii, jj], 2.0 * F[ii, jj]).astype(float)
    return out


# --------------------------------------------------------------------------- Max-Cut
README_W = np.array([[18.0, -5.0, -7.0, -6.0],
                     [-5.0, 6.0, 0.0, -1.0],
                     [-7.0, 0.0, 8.0, -1.0],
                     [-6.0, -1.0, -1.0, 8.0]])


def maxcut_from_laplacian(L, name="maxcut"):
    """max 0.25*<L,X>, diag(X)=1, X PSD   (README.md:75-81)."""
    n = L.shape[0]
    N = sympackedlen(n)
    c = -0.25 * _sym_coeff_vector(L)          # MAX sense -> minimise -obj
    diag = tri_index(np.arange(n), np.arange(n))
    A = sp.csc_matrix((np.ones(n), (np.arange(n), diag)), shape=(n, N))
    return Problem(n=N, A=A, b=np.ones(n), G=_empty(N), h=np.zeros(0), c=c,
                   psd=[np.arange(N, dtype=np.int64)], max_sense=True, name=name)


def maxcut_readme():
    """The README plumbing instance, W verbatim (README.md:66-72)."""
    return maxcut_from_laplacian(README_W, name="maxcut-readme-n4")


def erdos_renyi_laplacian(n, seed, avg_degree=12.0):
    """Unit-weight G(n, p) with p = avg_degree/(n-1) (about the density of
    SDPLIB maxG51); returns the sparse graph Laplacian."""
    rng = np.random.default_rng(seed)
    p = min(1.0, avg_degree / max(n - 1, 1))
    rows, cols = [], []
    for j in range(1, n):
        hit = np.nonzero(rng.random(j) < p)[0]
        rows.append(hit)
        cols.append(np.full(hit.shape, j))
    i = np.concatenate(rows) if rows else np.zeros(0, int)
    j = np.concatenate(cols) if cols else np.zeros(0, int)
    W = sp.coo_matrix((np.ones(len(i)), (i, j)), shape=(n, n))
    W = (W + W.T).tocsr()
    deg = np.asarray(W.sum(axis=1)).ravel()
    return (sp.diags(deg) - W).tocsr()


def maxcut(n, seed=0, avg_degree=12.0):
    return maxcut_from_laplacian(erdos_renyi_laplacian(n, seed, avg_degree),
                                 name=f"maxcut-er-n{n}-s{seed}")


# --------------------------------------------------------------------------- randSDP
def randsdp(n, m, seed=0, varbounds=True, dense=False):
    """test/base_randsdp.jl:4-23 with test/moi_randsdp.jl's model: m dense
    equality constraints <A_k,X> = b_k, optional -10 <= X[k] <= 10 on the first n
    scalar variables, min <C,X>.  dense=True hands the m x N coefficient matrix over as
    `M_dense` (row-major) instead of a CSC with every entry stored."""
    rng = np.random.default_rng(seed)
    R = rng.random((n, n))
    C = R @ R.T
    Gm = rng.standard_normal((n, n))
    Xbar = Gm @ Gm.T
    N = sympackedlen(n)
    rowsA = np.zeros((m, N))
    b = np.zeros(m)
    for k in range(m):
        Rk = rng.random((n, n))
        Ak = Rk @ Rk.T
        rowsA[k] = _sym_coeff_vector(Ak)
        b[k] = float(np.sum(Ak * Xbar))
    A = sp.csc_matrix((m, N)) if dense else sp.csc_matrix(rowsA)
    if varbounds:
        # for k in 1:n: (-X[k] <= 10) then (X[k] <= 10), interleaved (moi_randsdp.jl:33-45)
        r = np.arange(2 * n)
        cidx = np.repeat(np.arange(n), 2)
        vals = np.tile([-1.0, 1.0], n)
        G = sp.csc_matrix((vals, (r, cidx)), shape=(2 * n, N))
        h = np.full(2 * n, 10.0)
    else:
        G, h = _empty(N), np.zeros(0)
    return Problem(n=N, A=A, b=b, G=G, h=h, c=_sym_coeff_vector(C),
                   psd=[np.arange(N, dtype=np.int64)], name=f"randsdp-n{n}-m{m}-s{seed}",
                   M_dense=rowsA if dense else None)


def randsdp_device(n, m, seed=0, varbounds=True, device="cuda:0"):
    """The same model generated ON THE GPU (torch RNG, so not the numpy instance of
    `randsdp`): at the BASELINE size n=2000, m=4000 the coefficient matrix is
    4000 x 2 001 000 doubles = 64 GB, which is only ever materialised in HBM.
    Returns a Problem whose `M_dense` is a torch tensor on `device`."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    f64 = torch.float64
    N = sympackedlen(n)
    jj = torch.repeat_interleave(torch.arange(n, device=device), torch.arange(1, n + 1, device=device))
    ii = torch.arange(N, device=device) - jj * (jj + 1) // 2
    w = torch.where(ii == jj, 1.0, 2.0).to(f64)          # 2x on off-diagonal triangle variables
    R = torch.rand((n, n), dtype=f64, device=device, generator=g)
    C = R @ R.T
    Gm = torch.randn((n, n), dtype=f64, device=device, generator=g)
    Xbar = Gm @ Gm.T
    M = torch.empty((m, N), dtype=f64, device=device)
    b = torch.empty(m, dtype=f64, device=device)
    for k in range(m):
        Rk = torch.rand((n, n), dtype=f64, device=device, generator=g)
        Ak = Rk @ Rk.T
        M[k] = Ak[ii, jj] * w
        b[k] = (Ak * Xbar).sum()
    c = (C[ii, jj] * w).cpu().numpy()
    if varbounds:
        r = np.arange(2 * n)
        cidx = np.repeat(np.arange(n), 2)
        G = sp.csc_matrix((np.tile([-1.0, 1.0], n), (r, cidx)), shape=(2 * n, N))
        h = np.full(2 * n, 10.0)
    else:
        G, h = _empty(N), np.zeros(0)
    return Problem(n=N, A=sp.csc_matrix((m, N)), b=b.cpu().numpy(), G=G, h=h, c=c,
                   psd=[np.arange(N, dtype=np.int64)], name=f"randsdp-gpu-n{n}-m{m}-s{seed}", M_dense=M)


# --------------------------------------------------------------------------- MIMO
def mimo_data(n, seed):
    """test/base_mimo.jl:3-17 (own RNG)."""
    rng = np.random.default_rng(seed)
    m = 10 * n
    H = rng.standard_normal((m, n))
    v = rng.standard_normal((m, 1))
    s = rng.choice([-1.0, 1.0], size=n)
    y = H @ s.reshape(-1, 1) + 1e-4 * v
    L = np.block([[H.T @ H, -H.T @ y], [-y.T @ H, y.T @ y]])
    return s, H, y, L


def mimo(n, seed=0):
    """test/moi_mimo.jl: min <L,X>, diag(X)=1, -1 <= X_v <= 1 for every triangle
    variable (rows: all `X_v <= 1`, then all `-X_v <= 1`)."""
    _, _, _, L = mimo_data(n, seed)
    side = n + 1
    N = sympackedlen(side)
    I = sp.identity(N, format="csc")
    G = sp.vstack([I, -I], format="csc")
    h = np.ones(2 * N)
    diag = tri_index(np.arange(side), np.arange(side))
    A = sp.csc_matrix((np.ones(side), (np.arange(side), diag)), shape=(side, N))
    return Problem(n=N, A=A, b=np.ones(side), G=G, h=h, c=_sym_coeff_vector(L),
                   psd=[np.arange(N, dtype=np.int64)], name=f"mimo-n{n}-s{seed}")


# --------------------------------------------------------------------------- sensor localisation
def sensorloc_data(n, seed):
    """test/base_sensorloc.jl:2-22 (own RNG): n sensors and m = n // 10 anchors in the unit square, all pairwise distances."""
    rng = np.random.default_rng(seed)
    m = int(np.floor(0.1 * n))
    x_true = rng.random((2, n))
    a = rng.random((m, 2))
    return m, x_true, a


def sensorloc(n, seed=0, keep=0.1):
    """test/moi_sensorloc.jl / test/jump_sensorloc.jl (the SENSORLOC set of the reference's benchmark, test/runbench.jl:
    n = 100 .. 400): a FEASIBILITY SDP on one (n + 2) x (n + 2) block Z = [I X; X' Y]:
        anchor k - sensor j:  a_k1^2 Z11 + a_k2^2 Z22 - 2 a_k1 Z[1, j+2] - 2 a_k2 Z[2, j+2] + Z[j+2, j+2] = dbar_kj^2   (all k, j)
        sensor i - sensor j:  Z[i+2, i+2] + Z[j+2, j+2] - 2 Z[i+2, j+2] = d_ij^2       (a random tenth of the pairs)
        Z11 = 1, Z12 = 0, Z21 = 0, Z22 = 1
    zero objective; rows in the reference's order (VectorAffineFunction-in-Zeros, one row each; Z12 appears twice as the
    reference writes it)."""
    m, x_true, a = sensorloc_data(n, seed)
    rng = np.random.default_rng(seed + 1)
    side = n + 2
    N = sympackedlen(side)
    T = lambda i, j: int(tri_index(min(i, j), max(i, j)))
    rows, cols, vals, b = [], [], [], []

    def eq(terms, rhs):
        r = len(b)
        acc = {}
        for coef, col in terms:
            acc[col] = acc.get(col, 0.0) + coef
        for col, coef in acc.items():
            rows.append(r); cols.append(col); vals.append(coef)
        b.append(rhs)
    for j in range(n):
        for k in range(m):
            dbar2 = float(np.sum((x_true[:, j] - a[k]) ** 2))
            eq([(a[k, 0] ** 2, T(0, 0)), (a[k, 1] ** 2, T(1, 1)), (-2 * a[k, 0], T(0, j + 2)), (-2 * a[k, 1], T(1, j + 2)),
                (1.0, T(j + 2, j + 2))], dbar2)
    for i in range(n):
        for j in range(i):
            if rng.random() > 1.0 - keep:
                d2 = float(np.sum((x_true[:, i] - x_true[:, j]) ** 2))
                eq([(1.0, T(i + 2, i + 2)), (1.0, T(j + 2, j + 2)), (-2.0, T(i + 2, j + 2))], d2)
    for (i, j, v) in ((0, 0, 1.0), (0, 1, 0.0), (1, 0, 0.0), (1, 1, 1.0)):
        eq([(1.0, T(i, j))], v)
    A = sp.csc_matrix((vals, (rows, cols)), shape=(len(b), N))
    pr = Problem(n=N, A=A, b=np.array(b), G=_empty(N), h=np.zeros(0), c=np.zeros(N),
                 psd=[np.arange(N, dtype=np.int64)], name=f"sensorloc-n{n}-s{seed}")
    pr.x_true = x_true
    return pr


def block_diag_problems(probs, name="blockdiag"):
    """Several independent models in one (what "8 blocks" means for the MIMO
    config: one block-diagonal model)."""
    off = 0
    As, Gs, bs, hs, cs, psd, soc = [], [], [], [], [], [], []
    for pr in probs:
        As.append(pr.A); Gs.append(pr.G); bs.append(pr.b); hs.append(pr.h); cs.append(pr.c)
        psd += [v + off for v in pr.psd]
        soc += [v + off for v in pr.soc]
        off += pr.n
    return Problem(n=off, A=sp.block_diag(As, format="csc"), b=np.concatenate(bs),
                   G=sp.block_diag(Gs, format="csc"), h=np.concatenate(hs),
                   c=np.concatenate(cs), psd=psd, soc=soc,
                   max_sense=probs[0].max_sense, name=name)


# --------------------------------------------------------------------------- SDPLIB
def read_sdpa(path):
    """SDPA sparse reader with the reference harness's semantics
    (test/base_sdplib.jl:1-45): all blocks are merged into ONE n x n PSD
    variable and `n = length(c)` overrides the block sizes (so gpp500-1 is
    solved as 501 x 501); objective matrix stored negated."""
    with open(path) as f:
        lines = [ln.strip() for ln in f if ln.strip() and ln.strip()[0] not in '"*']
    m = int(lines[0].split()[0])

    def numbers(s):
        for ch in "{}(),":
            s = s.replace(ch, " ")
        return [float(t) for t in s.split()]

    blks = numbers(lines[2])
    cvec = np.array(numbers(lines[3])[:m])
    cum = np.concatenate([[0], np.cumsum(blks)]).astype(np.int64)
    n = len(cvec)                                    # the reference's quirk
    ks, ii, jj, vv = [], [], [], []
    for ln in lines[4:]:
        t = ln.split()
        if len(t) < 5:
            continue
        k, blk, i, j, val = int(t[0]), int(t[1]), int(t[2]), int(t[3]), float(t[4])
        off = cum[blk - 1]
        ks.append(k); ii.append(i + off - 1); jj.append(j + off - 1); vv.append(val)
    ks = np.array(ks); ii = np.array(ii); jj = np.array(jj); vv = np.array(vv)
    F = []
    for k in range(m + 1):
        sel = ks == k
        i, j, v = ii[sel], jj[sel], vv[sel]
        if k == 0:
            v = -v
        # F[k][i,j] = v; F[k][j,i] = v  (assignment, later entries overwrite)
        M = sp.lil_matrix((n, n))
        M[i, j] = v
        M[j, i] = v
        F.append(M.tocsr())
    return n, m, F, cvec


def sdplib(path, name=None):
    """test/moi_sdplib.jl: min <F0,X>  s.t. <Fk,X> = c_k, X PSD (one block)."""
    n, m, F, cvec = read_sdpa(path)
    N = sympackedlen(n)
    rows, cols, vals = [], [], []
    for k in range(1, m + 1):
        coef = sp.triu(sp.coo_matrix(F[k]))
        i, j, v = coef.row.astype(np.int64), coef.col.astype(np.int64), coef.data
        rows.append(np.full(len(v), k - 1)); cols.append(tri_index(i, j))
        vals.append(np.where(i == j, v, 2.0 * v))
    A = sp.csc_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(m, N))
    return Problem(n=N, A=A, b=cvec.copy(), G=_empty(N), h=np.zeros(0),
                   c=_sym_coeff_vector(F[0]), psd=[np.arange(N, dtype=np.int64)],
                   name=name or str(path))


def sdplib_blocks(path, name=None):
    """The same SDPA model, min <-F0, X> s.t. <Fk, X> = c_k, X PSD, with the file's BLOCK STRUCTURE
    kept (the reference's harness merges all blocks into one n x n variable and overrides n by
    length(c), test/base_sdplib.jl:24-26 -- `sdplib()` above reproduces that): one
    PositiveSemidefiniteConeTriangle per SDPA block of side s > 0, and |s| 1x1 PSD cones (nonnegative
    scalars) per diagonal block (s < 0).  This is what a JuMP user writes for arch / control / truss /
    qap / theta instances; it exercises many small PSD blocks."""
    with open(path) as f:
        lines = [ln.strip() for ln in f if ln.strip() and ln.strip()[0] not in '"*']
    m = int(lines[0].split()[0])

    def numbers(s):
        for ch in "{}(),":
            s = s.replace(ch, " ")
        return [float(t) for t in s.split()]

    blks = [int(v) for v in numbers(lines[2])]
    cvec = np.array(numbers(lines[3])[:m])
    # variable layout: for every block, its triangle (s > 0) or its |s| scalars (s < 0)
    start, psd, pos = [], [], 0
    for s_ in blks:
        start.append(pos)
        if s_ > 0:
            L = sympackedlen(s_)
            psd.append(np.arange(pos, pos + L, dtype=np.int64))
            pos += L
        else:
            for q in range(-s_):
                psd.append(np.arange(pos + q, pos + q + 1, dtype=np.int64))
            pos += -s_
    nvar = pos
    rows, cols, vals = [], [], []
    cobj = {}
    seen = {}
    for ln in lines[4:]:
        t = ln.split()
        if len(t) < 5:
            continue
        k, b, i, j, val = int(t[0]), int(t[1]) - 1, int(t[2]) - 1, int(t[3]) - 1, float(t[4])
        s_ = blks[b]
        if s_ > 0:
            lo, hi = min(i, j), max(i, j)
            var = start[b] + hi * (hi + 1) // 2 + lo
            coef = val if lo == hi else 2.0 * val
        else:
            if i != j:
                continue                                # off-diagonal entry of a diagonal block: ignored by SDPA too
            var = start[b] + i
            coef = val
        if k == 0:
            cobj[var] = -coef                           # objective matrix stored negated (base_sdplib.jl:37-38)
        else:
            seen[(k - 1, var)] = coef                   # later entries overwrite (assignment semantics)
    for (rk, var), coef in seen.items():
        rows.append(rk); cols.append(var); vals.append(coef)
    A = sp.csc_matrix((vals, (rows, cols)), shape=(m, nvar))
    c = np.zeros(nvar)
    for var, coef in cobj.items():
        c[var] = coef
    return Problem(n=nvar, A=A, b=cvec.copy(), G=_empty(nvar), h=np.zeros(0), c=c, psd=psd,
                   name=name or (str(path) + " (blocks kept)"))


def unpack_psd(x, side):
    """ivec (src/util.jl:18-38): triangle vector -> full symmetric matrix."""
    X = np.zeros((side, side))
    jj = np.repeat(np.arange(side), np.arange(1, side + 1))
    ii = np.concatenate([np.arange(j + 1) for j in range(side)])
    X[ii, jj] = x
    X[jj, ii] = x
    return X
