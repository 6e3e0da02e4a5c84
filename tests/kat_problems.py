"""The reference's hand-written known-answer problems
(/root/reference/test/moi_proxsdp_unit.jl, test/test_terminationstatus.jl),
rebuilt directly in the standard form that MOI hands to `chambolle_pock`
(/root/reference/src/MOI_wrapper.jl:229-292).  Used by the oracle tests (CPU)
and by the HIP parity tests (GPU) so both solve identical arrays.

MOI conventions reproduced here: `EqualTo(v)` on `f(x)` becomes a Zeros row
`f(x) - v`, so b = v; `GreaterThan(0)` on x becomes the Nonpositives row `-x <= 0`;
a VectorAffineFunction-in-Nonpositives row `g(x) + k <= 0` gives G row g, h = -k.
Variable indices below are 1-based in the comments (as in the Julia tests) and
0-based in the code."""
import numpy as np
import scipy.sparse as sp

from proxsdp_jl_amd.problems import Problem


def _mat(rows, ncols):
    """rows: list of {col: coef}."""
    r, c, v = [], [], []
    for i, row in enumerate(rows):
        for j, val in row.items():
            r.append(i); c.append(j); v.append(float(val))
    return sp.csc_matrix((v, (r, c)), shape=(len(rows), ncols))


def _cvec(n, terms):
    c = np.zeros(n)
    for j, v in terms.items():
        c[j] += v
    return c


def simple_lp():
    """moi_proxsdp_unit.jl:1-49 / test_terminationstatus.jl:1-38.
    min -4x1-3x2, 2x1+x2=4, x1+2x2=4, x>=0  ->  -9.33333, x=(1.3333,1.3333)."""
    return Problem(n=2, A=_mat([{0: 2, 1: 1}, {0: 1, 1: 2}], 2), b=np.array([4.0, 4.0]),
                   G=_mat([{0: -1}, {1: -1}], 2), h=np.zeros(2),
                   c=_cvec(2, {0: -4, 1: -3}), name="simple_lp")


def simple_lp_2_1d_sdp():
    """moi_proxsdp_unit.jl:51-95: the same LP with two 1x1 PSD cones."""
    return Problem(n=2, A=_mat([{0: 2, 1: 1}, {0: 1, 1: 2}], 2), b=np.array([4.0, 4.0]),
                   G=_mat([], 2), h=np.zeros(0), c=_cvec(2, {0: -4, 1: -3}),
                   psd=[np.array([0]), np.array([1])], name="simple_lp_2_1d_sdp")


def lp_in_SDP_equality_form():
    """moi_proxsdp_unit.jl:97-138: 4x4 PSD (10 vars); 2X1+X3+X6=4, X1+2X3+X10=4,
    min -4X1-3X3 -> -9.33333, X=[1.3333,0,1.3333,0,...]."""
    return Problem(n=10, A=_mat([{0: 2, 2: 1, 5: 1}, {0: 1, 2: 2, 9: 1}], 10),
                   b=np.array([4.0, 4.0]), G=_mat([], 10), h=np.zeros(0),
                   c=_cvec(10, {0: -4, 2: -3}), psd=[np.arange(10)],
                   name="lp_in_SDP_equality_form")


def lp_in_SDP_inequality_form():
    """moi_proxsdp_unit.jl:140-182: 2x2 PSD; 2X1+X3<=4, X1+2X3<=4; MAX 4X1+3X3 -> 9.33333."""
    return Problem(n=3, A=_mat([], 3), b=np.zeros(0),
                   G=_mat([{0: 2, 2: 1}, {0: 1, 2: 2}], 3), h=np.array([4.0, 4.0]),
                   c=_cvec(3, {0: -4, 2: -3}), psd=[np.arange(3)], max_sense=True,
                   name="lp_in_SDP_inequality_form")


def sdp_from_moi():
    """moi_proxsdp_unit.jl:184-223: 2x2 PSD, X2=1, min X1+X3 -> 2, X=ones(3)."""
    return Problem(n=3, A=_mat([{1: 1}], 3), b=np.array([1.0]), G=_mat([], 3), h=np.zeros(0),
                   c=_cvec(3, {0: 1, 2: 1}), psd=[np.arange(3)], name="sdp_from_moi")


def double_sdp_from_moi():
    """moi_proxsdp_unit.jl:225-271: two such blocks -> 4."""
    return Problem(n=6, A=_mat([{1: 1}, {4: 1}], 6), b=np.array([1.0, 1.0]),
                   G=_mat([], 6), h=np.zeros(0), c=_cvec(6, {0: 1, 2: 1, 3: 1, 5: 1}),
                   psd=[np.arange(3), np.arange(3, 6)], name="double_sdp_from_moi")


def sdp_wiki(max_sense=False):
    """moi_proxsdp_unit.jl:302-338 (Wikipedia SDP): 3x3 PSD, X1=X3=X6=1,
    -0.2<=X2<=-0.1, 0.4<=X5<=0.5; min X4 -> -0.978, max X4 -> 0.872."""
    A = _mat([{0: 1}, {2: 1}, {5: 1}], 6)
    G = _mat([{1: 1}, {1: -1}, {4: 1}, {4: -1}], 6)
    h = np.array([-0.1, 0.2, 0.5, -0.4])
    sign = -1.0 if max_sense else 1.0
    return Problem(n=6, A=A, b=np.ones(3), G=G, h=h, c=_cvec(6, {3: sign}),
                   psd=[np.arange(6)], max_sense=max_sense,
                   name="sdp_wiki_max" if max_sense else "sdp_wiki_min")


# name -> (builder, expected user-sense objective, atol, expected primal or None)
KATS = {
    "simple_lp": (simple_lp, -9.33333, 1e-2, np.array([1.3333, 1.3333])),
    "simple_lp_2_1d_sdp": (simple_lp_2_1d_sdp, -9.33333, 1e-2, np.array([1.3333, 1.3333])),
    "lp_in_SDP_equality_form": (lp_in_SDP_equality_form, -9.33333, 1e-2,
                                np.array([1.3333, 0, 1.3333, 0, 0, 0, 0, 0, 0, 0])),
    "lp_in_SDP_inequality_form": (lp_in_SDP_inequality_form, 9.33333, 1e-2,
                                  np.array([1.3333, 0, 1.3333])),
    "sdp_from_moi": (sdp_from_moi, 2.0, 1e-2, np.ones(3)),
    "double_sdp_from_moi": (double_sdp_from_moi, 4.0, 1e-2, np.ones(6)),
    "sdp_wiki_min": (lambda: sdp_wiki(False), -0.978, 1e-2, None),
    "sdp_wiki_max": (lambda: sdp_wiki(True), 0.872, 1e-2, None),
}


# ----------------------------------------------------------------- extra instances (own, not from the reference)
def soc_norm():
    """min t  s.t. (t, x1, x2) in SOC, x1 = 3, x2 = 4  ->  t = 5 (soc_projection!, prox_operators.jl:138-158)."""
    return Problem(n=3, A=_mat([{1: 1}, {2: 1}], 3), b=np.array([3.0, 4.0]), G=_mat([], 3), h=np.zeros(0),
                   c=_cvec(3, {0: 1}), soc=[np.arange(3)], name="soc_norm")


def sdp_plus_soc():
    """2x2 PSD block (vars 0..2) with X12 = 1, SOC (t, u1, u2) (vars 3..5) with u = (X11, 2),
    min X11 + X22 + t, plus a free variable z (var 6) pinned to 1.5 by an equality."""
    A = _mat([{1: 1}, {4: 1, 0: -1}, {5: 1}, {6: 1}], 7)
    b = np.array([1.0, 0.0, 2.0, 1.5])
    return Problem(n=7, A=A, b=b, G=_mat([], 7), h=np.zeros(0), c=_cvec(7, {0: 1, 2: 1, 3: 1, 6: 0.5}),
                   psd=[np.arange(3)], soc=[np.arange(3, 6)], name="sdp_plus_soc")


def infeasible_lp():
    """x1 + x2 = -1 with x >= 0: primal infeasible (status INFEASIBLE = 6)."""
    return Problem(n=2, A=_mat([{0: 1, 1: 1}], 2), b=np.array([-1.0]), G=_mat([{0: -1}, {1: -1}], 2),
                   h=np.zeros(2), c=_cvec(2, {0: 1, 1: 1}), name="infeasible_lp")


def unbounded_lp():
    """min -x1 - x2  s.t. x1 - x2 = 0, x >= 0: dual infeasible (status DUAL_INFEASIBLE = 5)."""
    return Problem(n=2, A=_mat([{0: 1, 1: -1}], 2), b=np.array([0.0]), G=_mat([{0: -1}, {1: -1}], 2),
                   h=np.zeros(2), c=_cvec(2, {0: -1, 1: -1}), name="unbounded_lp")


def mixed_cones(seed=0, sides=(1, 3, 104, 1, 5), soc_len=4, nfree=3, p=30, m=12):
    """Random feasible model with every variable class at once, variables deliberately NOT in
    solver order: PSD blocks of the given sides (1x1, full-eig-sized and Lanczos-sized), one SOC,
    free variables; equalities and inequalities built from a known strictly feasible point so that
    the optimum exists; objective = trace-like positive weights (bounded below on the cones)."""
    rng = np.random.default_rng(seed)
    lens = [s_ * (s_ + 1) // 2 for s_ in sides]
    n = sum(lens) + soc_len + nfree
    perm = rng.permutation(n)                      # user variable ids of the cone slots
    psd, pos = [], 0
    for L in lens:
        psd.append(perm[pos:pos + L].astype(np.int64)); pos += L
    soc = [perm[pos:pos + soc_len].astype(np.int64)]; pos += soc_len
    free = perm[pos:]
    x0 = np.zeros(n)
    c = np.zeros(n)
    for s_, idx in zip(sides, psd):
        G = rng.standard_normal((s_, s_ + 2))
        X = G @ G.T / (s_ + 2) + 0.5 * np.eye(s_)
        W = rng.standard_normal((s_, s_)); W = W @ W.T / s_ + np.eye(s_)       # PD objective weight
        k = 0
        for j in range(s_):
            for i in range(j + 1):
                x0[idx[k]] = X[i, j]
                c[idx[k]] = W[i, j] if i == j else 2.0 * W[i, j]
                k += 1
    u = rng.standard_normal(soc_len - 1)
    x0[soc[0][0]] = np.linalg.norm(u) + 1.0
    x0[soc[0][1:]] = u
    c[soc[0][0]] = 1.5
    x0[free] = rng.standard_normal(nfree)
    A = sp.random(p, n, density=0.08, random_state=rng, format="csc")
    # every free variable must be pinned by an equality, or the problem is unbounded in it
    rows = [{int(v): 1.0} for v in free]
    A = sp.vstack([A, _mat(rows, n)]).tocsc()
    b = A @ x0
    Gm = sp.random(m, n, density=0.1, random_state=rng, format="csc")
    h = Gm @ x0 + rng.uniform(0.1, 1.0, m)
    return Problem(n=n, A=A, b=b, G=Gm, h=h, c=c, psd=psd, soc=soc, name=f"mixed-cones-s{seed}")


def _tri(i, j):
    """0-based index of entry (i <= j) in MOI triangle order (column-major upper triangle)."""
    return j * (j + 1) // 2 + i


def infeasible_sdp(n=110):
    """Max-Cut-shaped model on a ring graph with one contradictory row: diag(X) = 1 AND X_00 = -1
    (a PSD matrix has X_00 >= 0): primal infeasible, PSD side > 100 so the block takes the Lanczos
    path (and the operator-form mat-vec on the support path)."""
    N = n * (n + 1) // 2
    rows = [{_tri(i, i): 1.0} for i in range(n)] + [{_tri(0, 0): 1.0}]
    b = np.concatenate([np.ones(n), [-1.0]])
    c = {}
    for i in range(n):
        j = (i + 1) % n
        c[_tri(min(i, j), max(i, j))] = 0.5          # -1/4 L on the ring's edges (off-diagonals doubled)
        c[_tri(i, i)] = -0.5
    return Problem(n=N, A=_mat(rows, N), b=b, G=_mat([], N), h=np.zeros(0), c=_cvec(N, c),
                   psd=[np.arange(N)], name=f"infeasible_sdp_n{n}")


def unbounded_sdp(n=110):
    """min -tr(X) + ring terms with only X_00 = 1 fixed: the other diagonal entries can grow without
    bound inside the PSD cone: dual infeasible (unbounded), PSD side > 100 (Lanczos path)."""
    N = n * (n + 1) // 2
    c = {_tri(i, i): -1.0 for i in range(n)}
    for i in range(n - 1):
        c[_tri(i, i + 1)] = 0.2
    return Problem(n=N, A=_mat([{_tri(0, 0): 1.0}], N), b=np.array([1.0]), G=_mat([], N), h=np.zeros(0),
                   c=_cvec(N, c), psd=[np.arange(N)], name=f"unbounded_sdp_n{n}")
