"""TEST INFRASTRUCTURE ONLY -- CPU oracle, never imported by the product path.

Line-faithful NumPy/SciPy restatement of the reference's PDHG hot path:

    chambolle_pock      /root/reference/src/pdhg.jl:1-530
    linesearch!         pdhg.jl:532-582      dual_step!   pdhg.jl:584-609
    primal_step!        pdhg.jl:611-637      certificates pdhg.jl:639-676
    cone_feas/get_duals/dual_feas/fix_diag_scaling/cache_solution  pdhg.jl:678-787
    psd_projection! &c. /root/reference/src/prox_operators.jl
    compute_gap!/compute_residual!/convergedrank  /root/reference/src/residuals.jl
    preprocess!/norm_scaling  /root/reference/src/scaling.jl
    structs             /root/reference/src/structs.jl

It operates on the standard form (A, G, b, h, c, cones) that
MOI_wrapper.jl:229-292 assembles.  All indices are 0-based here; the iteration
counter `k` stays 1-based as in the reference (it indexes the circular
histories through mod1).

Parity status: the reference is Julia and cannot run in this image, so this
restatement is pinned by the reference's own known-answer tests
(test/moi_proxsdp_unit.jl, test/test_terminationstatus.jl, test/moi_mimo.jl,
test/moi_sdplib.jl) rebuilt in standard form in tests/test_oracle_kat.py.  The
eigen-solver third-party layer is "parity unpinned" (see oracle/eig.py).
"""
import math
import time
from dataclasses import dataclass, field

import numpy as np
import scipy.sparse as sp

from . import eig as _eig
from .options import Options


# --------------------------------------------------------------------------- structs.jl
class CircularVector:
    """structs.jl:2-30; index i is 1-based, mod1(i, l)."""

    def __init__(self, l):
        self.v = np.zeros(l)
        self.l = l

    def __getitem__(self, i):
        return self.v[(i - 1) % self.l]

    def __setitem__(self, i, val):
        self.v[(i - 1) % self.l] = val

    def max_abs_diff(self):
        # structs.jl:14-20: for i in 1:l  abs(v[i]-v[i-1]), v[0] wraps to v[l]
        return float(np.max(np.abs(self.v - np.roll(self.v, 1))))


@dataclass
class AffineSets:            # structs.jl:32-42
    n: int
    p: int
    m: int
    A: sp.csc_matrix
    G: sp.csc_matrix
    b: np.ndarray
    h: np.ndarray
    c: np.ndarray
    extra: int = 0


@dataclass
class SDPSet:                # structs.jl:44-48 (vec_i 0-based here)
    vec_i: np.ndarray
    tri_len: int
    sq_side: int


@dataclass
class SOCSet:                # structs.jl:50-53
    idx: np.ndarray
    len: int


@dataclass
class ConicSets:             # structs.jl:55-58
    sdpcone: list = field(default_factory=list)
    socone: list = field(default_factory=list)


@dataclass
class Result:                # structs.jl:60-81
    status: int = 0
    status_string: str = "Problem not solved"
    primal: np.ndarray = None
    dual_cone: np.ndarray = None
    dual_eq: np.ndarray = None
    dual_in: np.ndarray = None
    slack_eq: np.ndarray = None
    slack_in: np.ndarray = None
    primal_residual: float = math.nan
    dual_residual: float = math.nan
    objval: float = math.nan
    dual_objval: float = math.nan
    gap: float = math.nan
    time: float = math.nan
    iter: int = -1
    final_rank: int = -1
    primal_feasible_user_tol: bool = False
    dual_feasible_user_tol: bool = False
    certificate_found: bool = False
    result_count: int = 0
    # not in the reference: instrumentation used by tests / bench
    stats: dict = field(default_factory=dict)
    trace: list = field(default_factory=list)
    state: dict = None           # export_state() at capture_iteration (chambolle_pock's test seam)


class Residuals:             # structs.jl:100-125
    def __init__(self, window):
        self.dual_gap = CircularVector(2 * window)
        self.prim_obj = CircularVector(2 * window)
        self.dual_obj = CircularVector(2 * window)
        self.equa_feasibility = 0.0
        self.ineq_feasibility = 0.0
        self.feasibility = CircularVector(2 * window)
        self.primal_residual = CircularVector(2 * window)
        self.dual_residual = CircularVector(2 * window)
        self.comb_residual = CircularVector(2 * window)


class Params:                # structs.jl:159-192
    pass


class Aux:                   # AuxiliaryData structs.jl:127-151
    def __init__(self, aff, cones):
        self.m = [np.zeros((s.sq_side, s.sq_side), order="F") for s in cones.sdpcone]
        self.Mty = np.zeros(aff.n)
        self.Mty_old = np.zeros(aff.n)
        self.Mx = np.zeros(aff.p + aff.m)
        self.Mx_old = np.zeros(aff.p + aff.m)
        self.y_half = np.zeros(aff.p + aff.m)
        self.y_temp = np.zeros(aff.p + aff.m)
        self.soc = []        # (start, len) of each SOC inside x  (util.jl:2-16)
        self.tri_idx = []    # per block: (iu_rows, iu_cols, offdiag mask, offset)


# --------------------------------------------------------------------------- scaling.jl
def preprocess(aff, cones):
    """preprocess! (scaling.jl:2-26): cone variables first, in cone order."""
    if cones.sdpcone or cones.socone:
        allv = []
        for s in cones.sdpcone:
            allv.extend(int(i) for i in s.vec_i)
        for s in cones.socone:
            allv.extend(int(i) for i in s.idx)
        extra = sorted(set(range(aff.n)) - set(allv))
        ord_ = np.array(allv + extra, dtype=np.int64)
    else:
        ord_ = np.arange(aff.n, dtype=np.int64)
    c_orig = aff.c.copy()
    aff.A = aff.A[:, ord_].tocsc()
    aff.G = aff.G[:, ord_].tocsc()
    aff.c = aff.c[ord_].copy()
    return c_orig[ord_], np.argsort(ord_, kind="stable")


def _offdiag_mask(cones, n):
    mask = np.zeros(n, dtype=bool)
    cont = 0
    for s in cones.sdpcone:
        for j in range(s.sq_side):
            mask[cont:cont + j] = True       # i < j
            cont += j + 1
    return mask


def norm_scaling(aff, cones):
    """norm_scaling (scaling.jl:28-58): x sqrt(2)/2 on off-diagonal PSD columns."""
    cte = math.sqrt(2.0) / 2.0
    mask = _offdiag_mask(cones, aff.n)
    scale = np.where(mask, cte, 1.0)
    aff.A = (aff.A @ sp.diags(scale)).tocsc() if aff.A.shape[0] else aff.A
    aff.G = (aff.G @ sp.diags(scale)).tocsc() if aff.G.shape[0] else aff.G
    aff.c = aff.c * scale


def fix_diag_scaling(v, cones, num):
    """fix_diag_scaling (pdhg.jl:734-743), in place."""
    mask = _offdiag_mask(cones, v.shape[0])
    v[mask] /= num


# --------------------------------------------------------------------------- prox_operators.jl
def _setup_blocks(a, cones):
    cont = 0
    for s in cones.sdpcone:
        n = s.sq_side
        cols, rows = np.triu_indices(n)          # (j outer, i inner) of the upper triangle
        # np.triu_indices is row-major over (r,c) with r<=c; swapping names gives
        # column-major upper order: for j, for i<=j  -> (i=cols?, ...) -- build explicitly
        jj = np.repeat(np.arange(n), np.arange(1, n + 1))
        ii = np.concatenate([np.arange(j + 1) for j in range(n)]) if n else np.zeros(0, int)
        a.tri_idx.append((ii, jj, ii != jj, cont))
        cont += n * (n + 1) // 2
    a.sdplen = cont
    for s in cones.socone:
        a.soc.append((cont, s.len))
        cont += s.len
    a.conelen = cont


def psd_vec_to_square(v, a, cones, sqrt_2=math.sqrt(2.0)):
    """prox_operators.jl:1-16."""
    for idx, s in enumerate(cones.sdpcone):
        ii, jj, off, start = a.tri_idx[idx]
        seg = v[start:start + s.tri_len]
        a.m[idx][ii, jj] = np.where(off, seg / sqrt_2, seg)
    return a.sdplen + 1


def psd_square_to_vec(v, a, cones, sqrt_2=math.sqrt(2.0)):
    """prox_operators.jl:17-31."""
    for idx, s in enumerate(cones.sdpcone):
        ii, jj, off, start = a.tri_idx[idx]
        vals = a.m[idx][ii, jj]
        v[start:start + s.tri_len] = np.where(off, vals * sqrt_2, vals)


def _rank1_accumulate(X, vals, vecs):
    """fill!(X,0); X += val*v*v' for each pair (prox_operators.jl:92-106,115-124),
    done as one dgemm (same result up to summation order)."""
    if len(vals) == 0:
        X[:, :] = 0.0
        return
    X[:, :] = (vecs * vals) @ vecs.T


def full_eig(a, idx, opt, p):
    """full_eig! (prox_operators.jl:111-126)."""
    p.current_rank[idx] = 0
    w, Q = _eig.full_eigh(a.m[idx])
    p.min_eig[idx] = 0.0
    pos = w > 0.0
    _rank1_accumulate(a.m[idx], w[pos], Q[:, pos])
    p.current_rank[idx] = int(np.sum(w > opt.tol_psd))
    p.stats["full_eigs"] += 1


def krylovkit_eig_proj(arc, a, idx, opt, p):
    """krylovkit_eig! wrapper (prox_operators.jl:89-109)."""
    _eig.krylovkit_eig(arc, a.m[idx], int(p.target_rank[idx]), opt)
    if arc.converged:
        p.min_eig[idx] = float(np.min(arc.vals))
        k = min(int(p.target_rank[idx]), arc.converged_eigs)
        vals = arc.vals[:k]
        pos = vals > 0.0
        p.current_rank[idx] += int(np.sum(pos))
        _rank1_accumulate(a.m[idx], vals[pos], arc.vecs[:, :k][:, pos])


def arpack_eig_proj(arc, a, idx, opt, p):
    """arpack_eig! wrapper (prox_operators.jl:68-87); values ascending."""
    _eig.arpack_eig(arc, a.m[idx], int(p.target_rank[idx]), opt)
    if arc.converged:
        p.min_eig[idx] = float(np.min(arc.vals))
        k = int(p.target_rank[idx])
        vals = arc.vals[:k]
        pos = vals > 0.0
        p.current_rank[idx] += int(np.sum(pos))
        _rank1_accumulate(a.m[idx], vals[pos], arc.vecs[:, :k][:, pos])


def psd_projection(v, a, cones, opt, p, arc_list, it):
    """psd_projection! (prox_operators.jl:33-66)."""
    p.min_eig = np.zeros(len(cones.sdpcone))
    psd_vec_to_square(v, a, cones)
    for idx, s in enumerate(cones.sdpcone):
        p.current_rank[idx] = 0
        if s.sq_side == 1:
            a.m[idx][0, 0] = max(0.0, a.m[idx][0, 0])
            p.min_eig[idx] = a.m[idx][0, 0]
        elif (not opt.full_eig_decomp
              and p.target_rank[idx] <= opt.max_target_rank_krylov_eigs
              and s.sq_side > opt.min_size_krylov_eigs
              and (p.iter % opt.full_eig_freq) > opt.full_eig_len):
            if opt.eigsolver == 1:
                arpack_eig_proj(arc_list[idx], a, idx, opt, p)
            else:
                krylovkit_eig_proj(arc_list[idx], a, idx, opt, p)
            if not arc_list[idx].converged:
                p.stats["krylov_fallbacks"] += 1
                full_eig(a, idx, opt, p)
        else:
            full_eig(a, idx, opt, p)
    psd_square_to_vec(v, a, cones)


def soc_projection(v, a):
    """soc_projection! (prox_operators.jl:138-158)."""
    for (start, ln) in a.soc:
        s = v[start]
        vec = v[start + 1:start + ln]
        nv = float(np.linalg.norm(vec))
        if nv <= -s:
            v[start] = 0.0
            vec[:] = 0.0
        elif nv <= s:
            pass
        else:
            val = 0.5 * (1.0 + s / nv)
            vec *= val
            v[start] = val * nv


def box_projection(v, aff, step):
    """box_projection! (prox_operators.jl:160-170)."""
    v[:aff.p] = aff.b
    v[aff.p:] = np.minimum(v[aff.p:] / step, aff.h)


# --------------------------------------------------------------------------- residuals.jl
def compute_gap(res, pair_x, pair_y, a, aff, p):
    """compute_gap! (residuals.jl:2-35)."""
    if aff.p > 0:
        res.equa_feasibility = float(np.max(np.abs(a.Mx[:aff.p] - aff.b), initial=0.0)) / (1.0 + p.norm_b)
    if aff.m > 0:
        res.ineq_feasibility = float(np.max(a.Mx[aff.p:] - aff.h, initial=0.0)) / (1.0 + p.norm_h)
    res.feasibility[p.iter] = max(res.equa_feasibility, res.ineq_feasibility)
    po = float(aff.c @ pair_x)
    do = 0.0
    if aff.p > 0:
        do -= float(aff.b @ pair_y[:aff.p])
    if aff.m > 0:
        do -= float(aff.h @ pair_y[aff.p:])
    res.prim_obj[p.iter] = po
    res.dual_obj[p.iter] = do
    res.dual_gap[p.iter] = abs(po - do) / (1.0 + abs(po) + abs(do))


def _norm_inf(v):
    return float(np.max(np.abs(v), initial=0.0))


def compute_residual(res, st, a, p, aff):
    """compute_residual! (residuals.jl:37-71); st holds x, x_old, y, y_old."""
    a.Mty_old = st.x_old - p.primal_step * a.Mty_old
    st.x_old = st.x - p.primal_step * a.Mty
    st.x_old -= a.Mty_old
    res.primal_residual[p.iter] = (math.sqrt(aff.n) * _norm_inf(st.x_old)
                                   / max(_norm_inf(a.Mty_old), p.norm_b, p.norm_h, 1.0))
    a.Mx_old = st.y_old - p.dual_step * a.Mx_old
    st.y_old = st.y - p.dual_step * a.Mx
    st.y_old -= a.Mx_old
    res.dual_residual[p.iter] = (math.sqrt(aff.m + aff.p) * _norm_inf(st.y_old)
                                 / max(_norm_inf(a.Mx_old), p.norm_c, 1.0))
    res.comb_residual[p.iter] = max(res.primal_residual[p.iter], res.dual_residual[p.iter])
    st.x_old = st.x.copy()
    st.y_old = st.y.copy()
    a.Mty_old = a.Mty.copy()
    a.Mx_old = a.Mx.copy()


def soc_gap(v, start, ln):
    return float(np.linalg.norm(v[start + 1:start + ln])) - v[start]


def soc_convergence(a, st, opt):
    """residuals.jl:73-86."""
    for (start, ln) in a.soc:
        if soc_gap(st.x, start, ln) >= opt.tol_soc:
            return False
    return True


def convergedrank(p, cones, opt):
    """residuals.jl:88-101."""
    for idx, s in enumerate(cones.sdpcone):
        if not (s.sq_side < opt.min_size_krylov_eigs
                or p.target_rank[idx] > opt.max_target_rank_krylov_eigs
                or p.min_eig[idx] < opt.tol_psd):
            return False
    return True


# --------------------------------------------------------------------------- pdhg.jl
class _State:
    pass


def primal_step(st, a, cones, M, c, opt, p, arc_list, proj_callback=None):
    """primal_step! (pdhg.jl:611-637).  proj_callback(iter, x_in, x_out, p, arc_list): test hook that
    sees the vector handed to psd_projection! and what came back (captured-iterate fixtures)."""
    st.x -= p.primal_step * (a.Mty + c)
    if cones.sdpcone:
        x_in = st.x.copy() if proj_callback is not None else None
        psd_projection(st.x, a, cones, opt, p, arc_list, p.iter)
        if proj_callback is not None:
            proj_callback(p.iter, x_in, st.x, p, arc_list)
    if cones.socone:
        soc_projection(st.x, a)
    a.Mx = M @ st.x


def linesearch(st, a, aff, Mt, opt, p):
    """linesearch! (pdhg.jl:532-582)."""
    p.primal_step = p.primal_step * math.sqrt(1.0 + p.theta)
    trials = 0
    for _ in range(opt.max_linsearch_steps):
        trials += 1
        p.theta = p.primal_step / p.primal_step_old
        bt = p.beta * p.primal_step
        a.y_half = st.y + bt * ((1.0 + p.theta) * a.Mx - p.theta * a.Mx_old)
        a.y_temp = a.y_half.copy()
        box_projection(a.y_half, aff, bt)
        a.y_temp -= bt * a.y_half
        a.Mty = Mt @ a.y_temp
        # "In-place norm" (pdhg.jl:559-564): the differences overwrite Mty and y_temp ...
        a.Mty -= a.Mty_old
        a.y_temp -= st.y_old
        y_norm = float(np.linalg.norm(a.y_temp))
        Mty_norm = float(np.linalg.norm(a.Mty))
        if math.sqrt(p.beta) * p.primal_step * Mty_norm <= opt.delta * y_norm:
            break
        p.primal_step *= opt.linsearch_decay
    # ... and are "reverted" after the loop (pdhg.jl:573-575): the accepted Mty and y carry
    # fl(fl(v - v_old) + v_old), not v -- restated because it is what the reference iterates on
    a.Mty += a.Mty_old
    a.y_temp += st.y_old
    st.y = a.y_temp.copy()
    p.primal_step_old = p.primal_step
    p.dual_step = p.beta * p.primal_step
    p.stats["linesearch_trials"] += trials
    p.last_trials = trials


def dual_step(st, a, aff, Mt, opt, p):
    """dual_step! (pdhg.jl:584-609)."""
    a.y_half = st.y + p.dual_step * (2.0 * a.Mx - a.Mx_old)
    a.y_temp = a.y_half.copy()
    box_projection(a.y_half, aff, p.dual_step)
    a.y_temp -= p.dual_step * a.y_half
    a.Mty = Mt @ a.y_temp
    st.y = a.y_temp.copy()
    p.primal_step_old = p.primal_step
    p.last_trials = 1


def certificate_parameters(p, opt):
    """pdhg.jl:670-676."""
    p.certificate_search_min_iter = p.iter + 2 * opt.convergence_window + p.iter // 5 + 1000
    p.certificate_search = True
    opt.time_limit *= 1.1
    opt.max_iter_local = opt.max_iter_local + opt.max_iter_local // 10


def cone_feas(v, cones, a, num=math.sqrt(2.0)):
    """cone_feas (pdhg.jl:678-699)."""
    sdp_viol = 0.0
    sdplen = psd_vec_to_square(v, a, cones, num) - 1
    for idx, s in enumerate(cones.sdpcone):
        if s.sq_side == 1:
            sdp_viol = max(sdp_viol, -min(0.0, a.m[idx][0, 0]))
        else:
            w, _ = _eig.full_eigh(a.m[idx])
            sdp_viol = max(sdp_viol, -min(0.0, float(np.min(w))))
    cont = sdplen
    for s in cones.socone:
        ln = s.len
        sv = v[cont]
        sdp_viol = max(sdp_viol, -min(0.0, sv - float(np.linalg.norm(v[cont + 1:cont + ln]))))
        cont += ln
    return sdp_viol, cont


def get_duals(y, cones, aff, c, A, G):
    """get_duals (pdhg.jl:701-710)."""
    dual_eq = y[:aff.p].copy()
    dual_in = y[aff.p:].copy()
    dual_cone = c + A.T @ dual_eq + G.T @ dual_in
    dual_cone = np.asarray(dual_cone, dtype=float).copy()
    fix_diag_scaling(dual_cone, cones, 2.0)
    return dual_eq, dual_in, dual_cone


def dual_feas_parts(dual_in, dual_cone, cones, a):
    """dual_feas (pdhg.jl:716-732)."""
    ineq_viol = 0.0
    if len(dual_in) > 0:
        ineq_viol = -min(0.0, float(np.min(dual_in)))
    cone_viol, cont = cone_feas(dual_cone, cones, a)
    zero_viol = 0.0
    dual_zr = dual_cone[cont:]
    if len(dual_zr) > 0:
        zero_viol = float(np.max(np.abs(dual_zr)))
    return max(cone_viol, ineq_viol, zero_viol)


def dual_feas(y, cones, aff, c, A, G, a):
    """dual_feas (pdhg.jl:712-715)."""
    _, dual_in, dual_cone = get_duals(y, cones, aff, c, A, G)
    return dual_feas_parts(dual_in, dual_cone, cones, a)


def equilibrate(M, aff, opt):
    """equilibrate! (equilibration.jl:1-72): projected-gradient row/column scaling.  The reference
    replaces v by its mean in every iteration (:56-58), so D is a multiple of the identity: kept.
    In the reference `E = Diagonal(u)`, `D = Diagonal(v)` (:16-17) wrap u and v WITHOUT copying, so
    `E.diag .= exp.(u)` (:25-26) overwrites u with exp(u) (v with exp(v)) at the top of every iteration and
    gradient, step and running average continue from there.  opt.equilibration_reference_aliasing = True
    (default) restates exactly that; False = the iteration the code evidently intends (u, v kept, E = exp(u),
    D = exp(v)) -- what rounds 1-4 implemented.  With the aliasing the scaling comes out badly conditioned
    (maxcut n=30: E ~ 0.023, D ~ 110) and some known answers the unscaled solver passes are missed
    (tests/test_oracle_kat.py records which); the option is off by default and never switched on by the
    reference's tests, so neither variant can be checked against Julia here ("parity unpinned").
    Returns the diagonals (E over rows, D over columns)."""
    M = sp.csc_matrix(M)
    nQ, n = aff.m + aff.p, aff.n
    alpha = (n / nQ) ** 0.25
    beta = (nQ / n) ** 0.25
    alpha2, beta2 = alpha ** 2, beta ** 2
    gamma = 0.1
    u, v = np.zeros(nQ), np.zeros(n)
    u_, v_ = np.zeros(nQ), np.zeros(n)
    rows = M.indices
    cols = np.repeat(np.arange(n), np.diff(M.indptr))
    alias = bool(getattr(opt, "equilibration_reference_aliasing", True))
    for it in range(1, opt.equilibration_iters + 1):
        Ed, Dd = np.exp(u), np.exp(v)
        if alias:                                 # E.diag === u, D.diag === v (equilibration.jl:16-17,25-26)
            u, v = Ed.copy(), Dd.copy()
        d2 = (M.data * Dd[cols] * Ed[rows]) ** 2
        step_size = 2.0 / (gamma * (it + 1.0))
        row_norms = np.bincount(rows, weights=d2, minlength=nQ)
        col_norms = np.bincount(cols, weights=d2, minlength=n)
        u_grad = row_norms - alpha2 + gamma * u
        v_grad = col_norms - beta2 + gamma * v
        u = np.clip(u - step_size * u_grad, opt.equilibration_lb, opt.equilibration_ub)
        v = v - step_size * v_grad
        v = np.full(n, np.sum(v) / n)
        v = np.clip(v, 0.0, opt.equilibration_ub)
        u_ = 2 * u / (it + 2) + it * u_ / (it + 2)
        v_ = 2 * v / (it + 2) + it * v_ / (it + 2)
    return np.exp(u_), np.exp(v_)


def _sparse_extrema(M):
    """maximum(M), minimum(M) of a SparseMatrixCSC: implicit zeros count (pdhg.jl:68-69)."""
    M = sp.csc_matrix(M)
    has_zero = M.nnz < M.shape[0] * M.shape[1]
    if M.nnz == 0:
        return 0.0, 0.0
    hi, lo = float(M.data.max()), float(M.data.min())
    if has_zero:
        hi, lo = max(hi, 0.0), min(lo, 0.0)
    return hi, lo


def cache_solution(st, res, cones, aff, p, opt, c, A, b, G, h, var_ordering, a, ED=None):
    """cache_solution (pdhg.jl:745-787).  NB: mutates st.x in place exactly as
    the reference does (fix_diag_scaling on pair.x)."""
    fix_diag_scaling(st.x, cones, math.sqrt(2.0))
    if opt.equilibration and ED is not None:     # "Remove equilibrating" (pdhg.jl:751-755); the scaled
        st.x = ED[1] * st.x                      # iterates replace pair.x / pair.y, as in the reference
        st.y = ED[0] * st.y
    slack_eq = A @ st.x - b
    slack_in = G @ st.x - h
    dual_eq, dual_in, dual_cone = get_duals(st.y, cones, aff, c, A, G)
    dfeas = dual_feas_parts(dual_in, dual_cone, cones, a)
    return Result(
        status=p.stop_reason,
        status_string=p.stop_reason_string,
        primal=st.x[var_ordering].copy(),
        dual_cone=dual_cone[var_ordering].copy(),
        dual_eq=dual_eq, dual_in=dual_in,
        slack_eq=np.asarray(slack_eq).ravel(), slack_in=np.asarray(slack_in).ravel(),
        primal_residual=res.equa_feasibility,
        dual_residual=res.ineq_feasibility,
        objval=res.prim_obj[p.iter], dual_objval=res.dual_obj[p.iter],
        gap=res.dual_gap[p.iter], time=time.time() - p.time0, iter=p.iter,
        final_rank=int(np.sum(p.current_rank)),
        primal_feasible_user_tol=bool(res.feasibility[p.iter] <= opt.tol_feasibility),
        dual_feasible_user_tol=bool(dfeas <= opt.tol_feasibility_dual),
        certificate_found=p.certificate_found, result_count=1,
        stats=dict(p.stats, dual_feasibility=dfeas))


STATE_HIST = ("dual_gap", "prim_obj", "dual_obj", "feasibility", "primal_residual", "dual_residual", "comb_residual")
STATE_SCAL = ("primal_step", "primal_step_old", "dual_step", "beta", "theta", "adapt_level")


def export_state(st, a, p, res, ada_count):
    """Everything chambolle_pock carries across an iteration boundary (after the control logic of iteration p.iter,
    pdhg.jl:246-483): the same dictionary proxsdp_hip_solve_ex captures / resumes (include/proxsdp_hip.h
    proxsdp_state).  x_old = x, y_old = y, Mty_old = Mty, Mx_old = Mx hold there (residuals.jl:65-68); the
    KrylovKit start vector is fixed (krylovkit_reset_resid = false), so no eigensolver state is carried."""
    d = dict(iteration=int(p.iter), x=st.x.copy(), y=st.y.copy(), Mty=np.asarray(a.Mty, float).copy(),
             Mx=np.asarray(a.Mx, float).copy(),
             target_rank=np.asarray(p.target_rank, np.int64).copy(),
             current_rank=np.asarray(p.current_rank, np.int64).copy(),
             min_eig=np.asarray(p.min_eig, float).copy(),
             hist=np.stack([getattr(res, nme).v.copy() for nme in STATE_HIST]),
             rank_update=int(p.rank_update), update_cont=int(p.update_cont), ada_count=int(ada_count),
             equa_feasibility=float(res.equa_feasibility), ineq_feasibility=float(res.ineq_feasibility),
             dual_feasibility=float(p.dual_feasibility))
    for nme in STATE_SCAL:
        d[nme] = float(getattr(p, nme))
    return d


def _import_state(state, st, a, p, res, aff):
    nQ = aff.p + aff.m
    if len(state["x"]) != aff.n or len(state["y"]) != nQ or len(state["Mty"]) != aff.n or len(state["Mx"]) != nQ:
        raise ValueError("resume state does not match the problem")
    st.x = np.array(state["x"], dtype=float)
    st.x_old = st.x.copy()
    st.y = np.array(state["y"], dtype=float)
    st.y_old = st.y.copy()
    a.Mty = np.array(state["Mty"], dtype=float)
    a.Mty_old = a.Mty.copy()
    a.Mx = np.array(state["Mx"], dtype=float)
    a.Mx_old = a.Mx.copy()
    for nme in STATE_SCAL:
        setattr(p, nme, float(state[nme]))
    p.target_rank = np.array(state["target_rank"], dtype=np.int64)
    p.current_rank = np.array(state["current_rank"], dtype=np.int64)
    p.min_eig = np.array(state["min_eig"], dtype=float)
    p.rank_update, p.update_cont = int(state["rank_update"]), int(state["update_cont"])
    p.dual_feasibility = float(state.get("dual_feasibility", -1.0))
    res.equa_feasibility = float(state.get("equa_feasibility", 0.0))
    res.ineq_feasibility = float(state.get("ineq_feasibility", 0.0))
    h = np.asarray(state["hist"], dtype=float)
    if h.shape != (len(STATE_HIST), 2 * p.window):
        raise ValueError("resume state: hist must be 7 x 2*convergence_window")
    for q, nme in enumerate(STATE_HIST):
        getattr(res, nme).v[:] = h[q]
    p.iter = int(state["iteration"])
    return int(state["ada_count"])


def chambolle_pock(aff_in, cones, opt_in=None, *, eig_resid=None, trace=False,
                   iter_callback=None, proj_callback=None, resume=None, capture_iteration=None):
    """chambolle_pock (pdhg.jl:1-530).  `aff_in` and `opt_in` are not mutated
    (the reference mutates both; the C ABI must not -- SURVEY.md section 8b).
    eig_resid: optional list of start vectors, one per PSD block.
    resume: a state dictionary (export_state, or the library's proxsdp_hip_solve_ex capture): the loop continues with
    iteration state['iteration'] + 1 from that iterate instead of pdhg.jl:54-142's initial point (test seam: late
    windows of long solves).  capture_iteration: k -> Result.state = export_state after iteration k."""
    opt = (opt_in or Options()).copy()
    aff = AffineSets(aff_in.n, aff_in.p, aff_in.m,
                     sp.csc_matrix(aff_in.A, dtype=float).copy(), sp.csc_matrix(aff_in.G, dtype=float).copy(),
                     np.array(aff_in.b, dtype=float), np.array(aff_in.h, dtype=float),
                     np.array(aff_in.c, dtype=float))
    p = Params()
    p.theta = opt.initial_theta
    p.adapt_level = opt.initial_adapt_level
    p.window = opt.convergence_window
    p.beta = opt.initial_beta
    p.time0 = time.time()
    p.norm_b = float(np.linalg.norm(aff.b))
    p.norm_h = float(np.linalg.norm(aff.h))
    p.norm_c = float(np.linalg.norm(aff.c))
    p.rank_update, p.stop_reason, p.update_cont = 0, 0, 0
    p.stop_reason_string = "Not optimized"
    nb = len(cones.sdpcone)
    p.target_rank = np.array([min(max(int(opt.initial_target_rank), 1), s_.sq_side) for s_ in cones.sdpcone], dtype=np.int64) \
        if nb else np.zeros(0, dtype=np.int64)
    p.current_rank = 2 * np.ones(nb, dtype=np.int64)
    p.min_eig = np.zeros(nb)
    p.dual_feasibility = -1.0
    p.dual_feasibility_check = False
    p.certificate_search = False
    p.certificate_search_min_iter = 0
    p.certificate_found = False
    p.iter = 0
    p.stats = {"linesearch_trials": 0, "full_eigs": 0, "krylov_fallbacks": 0}
    p.last_trials = 0
    sol = []
    arc_list = [_eig.EigSolverAlloc(s.sq_side, opt,
                                    None if eig_resid is None else eig_resid[i])
                for i, s in enumerate(cones.sdpcone)]
    ada_count = 0

    if opt.max_iter <= 0:
        opt.max_iter_local = opt.max_iter_conic if (cones.socone or cones.sdpcone) else opt.max_iter_lp
    else:
        opt.max_iter_local = opt.max_iter

    # ---- Init (pdhg.jl:54-142)
    c_orig, var_ordering = preprocess(aff, cones)
    A_orig, b_orig = aff.A.copy(), aff.b.copy()
    G_orig, h_orig = aff.G.copy(), aff.h.copy()
    # Diagonal preconditioning (pdhg.jl:64-92)
    ED = None
    if opt.equilibration:
        UB, LB = _sparse_extrema(sp.vstack([aff.A, aff.G]))
        if UB == 0.0 or LB / UB <= opt.equilibration_limit:
            opt.equilibration = False
    if opt.equilibration_force:
        opt.equilibration = True
    if opt.equilibration:
        M0 = sp.vstack([aff.A, aff.G], format="csc")
        Ed, Dd = equilibrate(M0, aff, opt)
        Ms = sp.csc_matrix(sp.diags(Ed) @ M0 @ sp.diags(Dd))
        aff.A = sp.csc_matrix(Ms[:aff.p, :])
        aff.G = sp.csc_matrix(Ms[aff.p:, :])
        rhs = Ed * np.concatenate([b_orig, h_orig])
        aff.b, aff.h = rhs[:aff.p].copy(), rhs[aff.p:].copy()
        aff.c = Dd * aff.c
        ED = (Ed, Dd)
    norm_scaling(aff, cones)

    st = _State()
    st.x = np.zeros(aff.n)
    st.x_old = np.zeros(aff.n)
    st.y = np.zeros(aff.p + aff.m)
    st.y_old = np.zeros(aff.p + aff.m)
    a = Aux(aff, cones)
    _setup_blocks(a, cones)
    res = Residuals(p.window)

    M = sp.vstack([aff.A, aff.G], format="csr")
    Mt = M.T.tocsr()
    if not opt.approx_norm:                                       # pdhg.jl:108-119
        if min(M.shape) >= 2:
            try:
                import scipy.sparse.linalg as spla
                spectral_norm = float(spla.svds(M.astype(float), k=1, return_singular_vectors=False)[0])
            except Exception:
                spectral_norm = float(np.sqrt(np.sum(M.data ** 2)))
        else:
            spectral_norm = float(np.linalg.svd(M.toarray(), compute_uv=False).max()) if M.shape[0] * M.shape[1] else 0.0
    else:
        spectral_norm = float(np.sqrt(np.sum(M.data ** 2)))      # LinearAlgebra.norm(M)
    if spectral_norm < 1e-10:
        spectral_norm = 1.0
    p.primal_step = 1.0 / spectral_norm
    p.primal_step_old = p.primal_step
    p.dual_step = p.primal_step

    if opt.advanced_initialization:
        st.x = p.primal_step * aff.c
        a.Mx = M @ st.x
        a.Mx_old = M @ st.x_old

    tr = []
    captured = None
    k = 0
    if resume is not None:
        if opt.equilibration:
            raise ValueError("resume: not with equilibration")
        ada_count = _import_state(resume, st, a, p, res, aff)
        k = p.iter
    t_loop0 = time.time()
    kmax = 2 * opt.max_iter_local
    while k < kmax:
        if capture_iteration is not None and captured is None and k == capture_iteration and k > 0 \
                and not (resume is not None and k == int(resume["iteration"])):
            captured = export_state(st, a, p, res, ada_count)
        k += 1
        p.iter = k
        primal_step(st, a, cones, M, aff.c, opt, p, arc_list, proj_callback)
        if opt.line_search_flag:
            linesearch(st, a, aff, Mt, opt, p)
        else:
            dual_step(st, a, aff, Mt, opt, p)
        compute_residual(res, st, a, p, aff)
        compute_gap(res, st.x, st.y, a, aff, p)

        if trace:
            tr.append(dict(iter=k, prim_obj=res.prim_obj[k], dual_obj=res.dual_obj[k],
                           gap=res.dual_gap[k], feas=res.feasibility[k],
                           prim_res=res.primal_residual[k], dual_res=res.dual_residual[k],
                           primal_step=p.primal_step, beta=p.beta, theta=p.theta,
                           target_rank=[int(t) for t in p.target_rank],
                           current_rank=[int(t) for t in p.current_rank],
                           min_eig=[float(t) for t in p.min_eig],
                           trials=p.last_trials))
        if iter_callback is not None:
            iter_callback(k, st, a, p, res)

        if (opt.check_dual_feas and k % opt.check_dual_feas_freq == 0) or \
                (opt.log_verbose and k % opt.log_freq == 0 and opt.extended_log2):
            cc = (0.0 if p.stop_reason == 6 else 1.0) * c_orig
            p.dual_feasibility = dual_feas(st.y, cones, aff, cc, A_orig, G_orig, a)
            p.dual_feasibility_check = True
        else:
            p.dual_feasibility_check = False

        if p.iter < p.certificate_search_min_iter:
            continue

        if opt.certificate_search and p.certificate_search:
            if p.stop_reason == 6:
                if res.dual_obj[k] > opt.certificate_obj_tol:
                    p.dual_feasibility = dual_feas(st.y, cones, aff, 0 * c_orig, A_orig, G_orig, a)
                    p.dual_feasibility_check = True
                    if p.dual_feasibility < opt.tol_feasibility_dual:
                        p.certificate_found = True
                        p.stop_reason_string += " [Dual ray found]"
                        break
            else:
                if res.prim_obj[k] < -opt.certificate_obj_tol:
                    if res.feasibility[p.iter] < opt.tol_feasibility:
                        p.certificate_found = True
                        p.stop_reason_string += " [Primal ray found]"
                        break
            if ((res.prim_obj[k] < -opt.certificate_fail_tol
                 and res.dual_obj[k] < -opt.certificate_fail_tol
                 and res.feasibility[p.iter] < -opt.certificate_fail_tol)
                    or math.isnan(res.comb_residual[k])):
                p.stop_reason_string += " [Failed to find certificate]"
                break

        # ---- convergence / rank update / divergence / adaptive steps (pdhg.jl:246-332)
        p.rank_update += 1
        if (res.dual_gap[p.iter] <= opt.tol_gap and res.feasibility[p.iter] <= opt.tol_feasibility
                and (not opt.check_dual_feas or p.dual_feasibility < opt.tol_feasibility_dual)):
            if convergedrank(p, cones, opt) and soc_convergence(a, st, opt) and p.iter > opt.min_iter:
                if not p.certificate_search:
                    p.stop_reason = 1
                    p.stop_reason_string = "Optimal solution found"
                else:
                    p.stop_reason_string += " [Failed to find certificate - type 2]"
                    break
                break
            elif p.rank_update > p.window:
                p.update_cont += 1
                if p.update_cont > 0:
                    for idx, s in enumerate(cones.sdpcone):
                        if p.current_rank[idx] + opt.rank_slack >= p.target_rank[idx]:
                            if p.min_eig[idx] > opt.tol_psd:
                                if opt.rank_increment == 0:
                                    p.target_rank[idx] = min(opt.rank_increment_factor * p.target_rank[idx], s.sq_side)
                                else:
                                    p.target_rank[idx] = min(opt.rank_increment_factor + p.target_rank[idx], s.sq_side)
                    p.rank_update, p.update_cont = 0, 0
        elif (k > p.window and res.comb_residual[k - p.window] < res.comb_residual[k]
              and p.rank_update > p.window):
            p.update_cont += 1
            if p.update_cont > opt.divergence_min_update:
                for idx, s in enumerate(cones.sdpcone):
                    if p.target_rank[idx] < s.sq_side:
                        p.rank_update, p.update_cont = 0, 0
                    if p.current_rank[idx] + opt.rank_slack >= p.target_rank[idx]:
                        if p.min_eig[idx] > opt.tol_psd:
                            if opt.rank_increment == 0:
                                p.target_rank[idx] = min(opt.rank_increment_factor * p.target_rank[idx], s.sq_side)
                            else:
                                p.target_rank[idx] = min(opt.rank_increment_factor + p.target_rank[idx], s.sq_side)
        elif (res.primal_residual[k] > opt.tol_primal and res.dual_residual[k] < opt.tol_dual
              and k > p.window):
            ada_count += 1
            if ada_count > opt.adapt_window:
                ada_count = 0
                if opt.line_search_flag:
                    p.beta *= (1.0 - p.adapt_level)
                    p.primal_step /= math.sqrt(1.0 - p.adapt_level)
                else:
                    p.primal_step /= (1.0 - p.adapt_level)
                    p.dual_step *= (1.0 - p.adapt_level)
                p.adapt_level *= opt.adapt_decay
        elif (res.primal_residual[k] < opt.tol_primal and res.dual_residual[k] > opt.tol_dual
              and k > p.window):
            ada_count += 1
            if ada_count > opt.adapt_window:
                ada_count = 0
                if opt.line_search_flag:
                    p.beta /= (1.0 - p.adapt_level)
                    p.primal_step *= math.sqrt(1.0 - p.adapt_level)
                else:
                    p.primal_step *= (1.0 - p.adapt_level)
                    p.dual_step /= (1.0 - p.adapt_level)
                p.adapt_level *= opt.adapt_decay

        def _cache():
            return cache_solution(st, res, cones, aff, p, opt, c_orig, A_orig, b_orig,
                                  G_orig, h_orig, var_ordering, a, ED)

        def _cert_infeas():            # certificate_infeasibility (pdhg.jl:655-668)
            aff.c[:] = 0.0
            certificate_parameters(p, opt)

        def _cert_dual_infeas():       # certificate_dual_infeasibility (pdhg.jl:639-653)
            aff.b[:] = 0.0
            aff.h[:] = 0.0
            certificate_parameters(p, opt)

        # ---- iteration / time limits (pdhg.jl:334-382)
        if p.iter >= opt.max_iter_local or time.time() - p.time0 >= opt.time_limit:
            if (p.iter > opt.min_iter_time_infeas
                    and res.dual_gap.max_abs_diff() < opt.infeas_stable_gap_tol
                    and res.dual_gap[k] > opt.infeas_limit_gap_tol):
                if res.feasibility[p.iter] <= opt.tol_feasibility / 100:
                    p.stop_reason = 5
                    p.stop_reason_string = "Problem declared unbounded due to lack of improvement"
                    if opt.certificate_search and not p.certificate_search:
                        _cert_dual_infeas()
                        sol.append(_cache())
                    elif opt.certificate_search and p.certificate_search:
                        pass
                    else:
                        break
                elif res.feasibility[p.iter] > opt.infeas_feasibility_tol:
                    p.stop_reason = 6
                    p.stop_reason_string = "Problem declared infeasible due to lack of improvement"
                    if opt.certificate_search and not p.certificate_search:
                        _cert_infeas()
                        sol.append(_cache())
                    elif opt.certificate_search and p.certificate_search:
                        pass
                    else:
                        break
            elif p.iter >= opt.max_iter_local:
                p.stop_reason = 3
                p.stop_reason_string = f"Iteration limit of {opt.max_iter_local} was hit"
            else:
                p.stop_reason = 2
                p.stop_reason_string = f"Time limit hit, limit: {opt.time_limit} time: {time.time() - p.time0}"
            if p.iter >= opt.max_iter_local or time.time() - p.time0 >= opt.time_limit:
                break

        if opt.certificate_search and p.certificate_search:
            continue

        # ---- objective blow-up / stalls (pdhg.jl:389-483)
        if (p.iter > opt.min_iter_max_obj and res.dual_obj[k] > opt.max_obj) or math.isnan(res.dual_obj[k]):
            p.stop_reason = 6
            p.stop_reason_string = f"Infeasible: |Dual objective| = {res.dual_obj[k]} > maximum allowed = {opt.max_obj}"
            if opt.certificate_search and not p.certificate_search:
                _cert_infeas()
                sol.append(_cache())
            else:
                break
        if (p.iter > opt.min_iter_max_obj and res.prim_obj[k] < -opt.max_obj) or math.isnan(res.prim_obj[k]):
            p.stop_reason = 5
            p.stop_reason_string = f"Unbounded: |Primal objective| = {res.prim_obj[k]} > maximum allowed = {opt.max_obj}"
            if opt.certificate_search and not p.certificate_search:
                _cert_dual_infeas()
                sol.append(_cache())
            else:
                break
        if (p.iter > opt.min_iter_max_obj
                and res.dual_gap[k] > opt.infeas_limit_gap_tol
                and res.feasibility[p.iter] > opt.infeas_feasibility_tol
                and res.feasibility.max_abs_diff() < opt.infeas_stable_feasibility_tol):
            p.stop_reason = 6
            p.stop_reason_string = f"Infeasible: feasibility stalled at {res.feasibility[p.iter]}"
            if opt.certificate_search and not p.certificate_search:
                _cert_infeas()
                sol.append(_cache())
            else:
                break
        if (p.iter > opt.min_iter_max_obj
                and res.dual_gap[k] > 1 - opt.infeas_gap_tol
                and res.dual_gap.max_abs_diff() < opt.infeas_stable_gap_tol):
            if abs(res.dual_obj[k]) > abs(res.prim_obj[k]) and res.feasibility[p.iter] > opt.infeas_feasibility_tol:
                p.stop_reason = 6
                p.stop_reason_string = "Infeasible: duality gap stalled at 100 % with |Dual objective| >> |Primal objective|"
                if opt.certificate_search and not p.certificate_search:
                    _cert_infeas()
                    sol.append(_cache())
                else:
                    break
            elif abs(res.prim_obj[k]) > abs(res.dual_obj[k]) and res.feasibility[p.iter] <= opt.tol_feasibility:
                p.stop_reason = 5
                p.stop_reason_string = "Unbounded: duality gap stalled at 100 % with |Dual objective| << |Primal objective|"
                if opt.certificate_search and not p.certificate_search:
                    _cert_dual_infeas()
                    sol.append(_cache())
                else:
                    break

    loop_time = time.time() - t_loop0
    if capture_iteration is not None and captured is None and p.iter == capture_iteration:
        captured = export_state(st, a, p, res, ada_count)
    p.stats["loop_time"] = loop_time
    p.stats["lanczos_matvecs"] = int(sum(arc.matvecs for arc in arc_list))
    p.stats["lanczos_restarts"] = int(sum(arc.restarts for arc in arc_list))

    def _final_cache(cvec):
        return cache_solution(st, res, cones, aff, p, opt, cvec, A_orig, b_orig,
                              G_orig, h_orig, var_ordering, a, ED)

    if opt.certificate_search and p.certificate_search:
        assert len(sol) == 1
        if p.certificate_found:
            if p.stop_reason == 6:
                c_orig = c_orig * 0.0
            sol.pop()
            sol.append(_final_cache(c_orig))
    else:
        assert len(sol) == 0
        sol.append(_final_cache(c_orig))
    out = sol[0]
    out.trace = tr
    out.state = captured
    out.stats.update(p.stats)
    return out
