"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle, the
committed golden fixtures, the reference's known answers, and -- at full size --
size-independent properties.  Tolerances: fp64 kernels, deterministic
reductions; kernel-level results agree to ~1e-12 relative, eigenpairs to the
solvers' 1e-12/1e-10 residual tolerance, solves to the solver's own
tol_gap/tol_feasibility (BASELINE.json north_star)."""
import json
import math

import numpy as np
import pytest
import scipy.sparse as sp

import oracle
from oracle import Options, eig as oeig
from proxsdp_jl_amd import binding as B
from proxsdp_jl_amd import problems as P
from proxsdp_jl_amd.optimizer import Optimizer

from helpers import PROJ_CASES, check_truncated_projection, oracle_project, planted_packed, smat, svec
from kat_problems import KATS, sdp_wiki, simple_lp, soc_norm, sdp_plus_soc, unbounded_lp, infeasible_lp, mixed_cones

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    assert B.device_count() > 0, "no HIP device: the product path has no CPU fallback"


def _rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(1e-300, np.linalg.norm(b))


# ----------------------------------------------------------------- kernels
@pytest.mark.parametrize("n", [1, 2, 3, 63, 64, 65, 127, 128, 200, 513, 1000])
def test_symv_packed_matches_dense(n):
    rng = np.random.default_rng(n)
    x = rng.standard_normal(n * (n + 1) // 2)
    v = rng.standard_normal(n)
    y = B.symv_packed(x, n, v)
    ref = smat(x, n) @ v
    assert _rel(y, ref) < 1e-13


def test_symv_is_symmetric_and_linear_at_full_size():
    """n = 4000 (the metric's size): u'(Xv) == v'(Xu), X(av+bu) == aXv+bXu, and
    agreement with the packed arithmetic done on the CPU for a few rows."""
    n = 4000
    rng = np.random.default_rng(0)
    x = rng.standard_normal(n * (n + 1) // 2)
    u, v = rng.standard_normal(n), rng.standard_normal(n)
    Xu, Xv = B.symv_packed(x, n, u), B.symv_packed(x, n, v)
    assert abs(v @ Xu - u @ Xv) <= 1e-11 * np.linalg.norm(Xu) * np.linalg.norm(v)
    Xc = B.symv_packed(x, n, 2.0 * u - 3.0 * v)
    assert _rel(Xc, 2.0 * Xu - 3.0 * Xv) < 1e-12
    for i in (0, 1, 63, 64, 2047, 3999):
        col = np.arange(n)
        lo, hi = np.minimum(i, col), np.maximum(i, col)
        vals = x[hi * (hi + 1) // 2 + lo]
        vals = np.where(col == i, vals, vals / math.sqrt(2))
        assert abs(vals @ u - Xu[i]) <= 1e-11 * np.linalg.norm(vals) * np.linalg.norm(u)


@pytest.mark.parametrize("n,r", [(5, 0), (5, 1), (64, 3), (65, 16), (200, 17), (257, 40), (513, 2), (130, 130), (333, 7)])
def test_reconstruct_matches_dense(n, r):
    """both reconstruction kernels: the LDS-staged scalar-FMA one and the fp64 MFMA SYRK
    (v_mfma_f64_16x16x4_f64; asymmetric Z x lambda, so a transposed C/D map would show)"""
    rng = np.random.default_rng(n + r)
    Z = rng.standard_normal((n, r))
    lam = rng.uniform(0.1, 5.0, r)
    ref = svec((Z * lam) @ Z.T) if r else np.zeros(n * (n + 1) // 2)
    for mfma in (0, 1, -1):
        out = B.reconstruct(Z, lam, n, mfma=mfma)
        assert np.allclose(out, ref, rtol=1e-13, atol=1e-13 * max(1.0, np.abs(ref).max())), mfma


def test_reconstruct_mfma_at_full_size_and_high_rank():
    """n = 4000 at r = 63 (target rank ~ sqrt n) and r = 600 (full_eig!-sized positive part): MFMA
    kernel == scalar kernel to rounding, and a few entries against the definition"""
    n = 4000
    rng = np.random.default_rng(7)
    for r in (63, 600):
        Z = rng.standard_normal((n, r)) / np.sqrt(n)
        lam = rng.uniform(0.1, 5.0, r)
        a = B.reconstruct(Z, lam, n, mfma=0)
        b = B.reconstruct(Z, lam, n, mfma=1)
        assert np.allclose(a, b, rtol=1e-12, atol=1e-13 * np.abs(a).max())
        for (i, j) in ((0, 0), (5, 77), (63, 64), (1999, 3999), (3999, 3999)):
            v = (Z[i] * lam) @ Z[j] * (1.0 if i == j else math.sqrt(2.0))
            assert abs(b[j * (j + 1) // 2 + i] - v) <= 1e-12 * max(1.0, abs(v))


def test_spmv_both_orientations():
    rng = np.random.default_rng(1)
    for (rows, cols, dens) in [(50, 400, 0.05), (300, 40, 0.3), (7, 5000, 0.9), (1000, 1000, 0.002)]:
        M = sp.random(rows, cols, density=dens, random_state=rng, format="csc")
        x, y = rng.standard_normal(cols), rng.standard_normal(rows)
        assert _rel(B.spmv(M, x), M @ x) < 1e-13
        assert _rel(B.spmv(M, y, transpose=True), M.T @ y) < 1e-13
    # one very long row (gpp500-1's all-ones constraint) and empty rows/columns
    M = sp.lil_matrix((4, 20000))
    M[1, :] = 1.0
    M[3, 5] = 2.0
    M = M.tocsc()
    x = rng.standard_normal(20000)
    assert _rel(B.spmv(M, x), M @ x) < 1e-13
    # several rows beyond the segment threshold (8192 entries), random values, mixed with short
    # and medium rows: segmented workgroup-per-4096-entries path + fixed-order combination
    cols = 140000
    rows = [rng.choice(cols, size=k, replace=False) for k in (9000, 130000, 3, 700, 8192, 8193)]
    r = np.concatenate([np.full(len(c), i) for i, c in enumerate(rows)])
    M = sp.csc_matrix((rng.standard_normal(len(r)), (r, np.concatenate(rows))), shape=(len(rows) + 2, cols))
    x = rng.standard_normal(cols)
    assert _rel(B.spmv(M, x), M @ x) < 1e-13
    y = rng.standard_normal(M.shape[0])
    assert _rel(B.spmv(M, y, transpose=True), M.T @ y) < 1e-13


def test_vector_kernels_against_oracle_functions():
    """Kernel-level seams of SURVEY section 8b: primal update (pdhg.jl:622), one linesearch
    trial in y-space (pdhg.jl:547-553 + box_projection!) and the residual / gap reductions
    (residuals.jl:2-71) against the oracle's own functions on random data."""
    rng = np.random.default_rng(5)
    n, p, m = 70001, 3001, 2500
    Q = p + m
    x, x_old, Mty, Mty_old, c = (rng.standard_normal(n) for _ in range(5))
    y, y_old, Mx, Mx_old = (rng.standard_normal(Q) for _ in range(4))
    b, h = rng.standard_normal(p), rng.standard_normal(m)
    bh = np.concatenate([b, h])
    tau, beta, theta = 0.37, 1.7, 0.8
    # primal update: the oracle statement is x .-= tau .* (Mty .+ c)
    assert np.array_equal(B.primal_update(x, Mty, c, tau), x - tau * (Mty + c))
    # dual trial
    bt = beta * tau
    ybar = y + bt * ((1.0 + theta) * Mx - theta * Mx_old)
    proj = np.concatenate([b, np.minimum(ybar[p:] / bt, h)])     # box_projection!(ybar/bt) (prox_operators.jl:160-170)
    yref = ybar - bt * proj
    yout, nrm = B.dual_trial(y, Mx, Mx_old, bh, p, bt, theta)
    # linesearch! keeps fl(fl(y+ - y_old) + y_old) (in-place norm + revert, pdhg.jl:560-575): the same bits
    assert np.array_equal(yout, (yref - y) + y)
    assert nrm == pytest.approx(np.sum((yref - y) ** 2), rel=1e-12)
    # residuals + gap reductions
    sigma = bt
    out = B.residuals(x, x_old, Mty, Mty_old, c, tau, y, y_old, Mx, Mx_old, bh, p, sigma)
    Px, Pxo = x - tau * Mty, x_old - tau * Mty_old
    Py, Pyo = y - sigma * Mx, y_old - sigma * Mx_old
    ref = [np.abs(Px - Pxo).max(), np.abs(Pxo).max(), c @ x, np.abs(Py - Pyo).max(), np.abs(Pyo).max(),
           np.abs(Mx[:p] - b).max(), max(0.0, (Mx[p:] - h).max()), b @ y[:p], h @ y[p:]]
    assert np.allclose(out, ref, rtol=1e-12, atol=1e-12)


# ----------------------------------------------------------------- eigen layer
@pytest.mark.parametrize("n,nev,top", [
    (101, 2, [40.0, 25.0, 9.0, 4.0]),
    (257, 4, [90.0, 60.0, 33.0, 12.0, 5.0]),
    (300, 12, [50, 40, 30, 20, 10, 9, 8, 7, 6.5, 6, 5.5, 5.2]),
    (1000, 3, [500.0, 20.0, 19.5, 19.0]),
    # Krylov basis beyond 128 and 192 columns (NCH = 3 / 4 step kernels): nev = 80 -> krylovdim 161,
    # nev = 120 -> krylovdim 241 (the library's limit is krylovdim 255, target rank 127 = sqrt(16000))
    (600, 80, list(np.linspace(400.0, 40.0, 84))),
    (1000, 120, list(np.linspace(900.0, 60.0, 124))),
])
def test_lanczos_matches_oracle_and_dense(n, nev, top):
    x = planted_packed(n, 11, top, bulk=(-5.0, 1.0))
    X = smat(x, n)
    vals, vecs, info = B.eigsolve(x, n, nev)
    ncv = max(2 * nev + 1, 25)
    ovals, ovecs, oconv, onumiter, onumops = oeig.krylovkit_eigsolve(
        lambda v: X @ v, oeig.start_vector(n), nev, ncv, 100, 1e-12)
    ref = np.sort(np.linalg.eigvalsh(X))[::-1]
    scale = abs(ref[0])
    assert info["converged"] >= nev and len(vals) >= nev
    assert np.allclose(vals[:nev], ref[:nev], rtol=0, atol=1e-11 * scale)
    assert np.allclose(vals[:nev], ovals[:nev], rtol=0, atol=1e-11 * scale)
    resid = np.linalg.norm(X @ vecs - vecs * vals, axis=0)
    assert np.all(resid[:nev] < 1e-9 * scale)
    assert np.allclose(vecs.T @ vecs, np.eye(vecs.shape[1]), atol=1e-10)
    # same algorithm, same start vector: same number of restarts / mat-vecs
    assert info["numiter"] == onumiter and info["nmatvec"] == onumops


def test_lanczos_edge_cases():
    # zero matrix: invariant subspace at K = 1, howmany reduced (first PDHG iteration: x - tau*(0+c) with x = tau*c)
    vals, vecs, info = B.eigsolve(np.zeros(101 * 102 // 2), 101, 2)
    assert list(vals) == [0.0] and info["converged"] == 1 and info["nmatvec"] == 1
    # tiny matrix, krylovdim > n
    X3 = np.array([[1, -0.15, 0.0], [-0.15, 1, 0.45], [0.0, 0.45, 1.0]])
    vals, vecs, info = B.eigsolve(svec(X3), 3, 2)
    assert info["converged"] == 3 and info["nmatvec"] == 3
    assert np.allclose(vals, np.sort(np.linalg.eigvalsh(X3))[::-1], atol=1e-13)
    # explicit start vector
    rng = np.random.default_rng(5)
    x = planted_packed(150, 2, [30.0, 10.0])
    r = rng.standard_normal(150)
    v1, _, i1 = B.eigsolve(x, 150, 2, resid=r)
    v2, _, i2 = B.eigsolve(x, 150, 2, resid=3.0 * r)          # normalised inside, as KrylovKit does
    assert np.allclose(v1[:2], v2[:2], atol=1e-12) and i1["nmatvec"] == i2["nmatvec"]


def test_krylovkit_eager_mode_against_oracle():
    """krylovkit_eager = true (options.jl:112 -> KrylovKit's `eager`, eigsolver.jl:809; off by default): the
    convergence test runs after every expansion step once `howmany` vectors exist, so a projection stops as soon as
    its pairs have converged.  Library vs oracle: the same mat-vec and restart counts on planted spectra, the same
    eigenvalues, and a PDHG run (Max-Cut n = 200) with identical per-iteration mat-vec counts and trace; the eager
    run needs fewer mat-vecs than the default one."""
    o = B.default_options()
    B.set_option(o, "krylovkit_eager", 1)
    for n, nev, top in [(101, 2, [40.0, 25.0, 9.0, 4.0]), (300, 6, [50.0, 40.0, 30.0, 20.0, 10.0, 9.0, 8.0])]:
        x = planted_packed(n, 11, top, bulk=(-5.0, 1.0))
        vals, vecs, info = B.eigsolve(x, n, nev, options=o)
        X = smat(x, n)
        ovals, ovecs, oconv, onumiter, onumops = oeig.krylovkit_eigsolve(
            lambda v: X @ v, oeig.start_vector(n), nev, max(2 * nev + 1, 25), 100, 1e-12, eager=True)
        assert info["nmatvec"] == onumops and info["numiter"] == onumiter, (n, info, onumops, onumiter)
        k = min(len(vals), len(ovals), nev)
        assert np.allclose(vals[:k], ovals[:k], rtol=0, atol=1e-10 * max(top))
        assert np.allclose(vals[:k], np.sort(np.linalg.eigvalsh(X))[::-1][:k], rtol=0, atol=1e-10 * max(top))
        vals0, _, info0 = B.eigsolve(x, n, nev)
        assert info["nmatvec"] <= info0["nmatvec"]
    pr = P.maxcut(200, seed=0)
    iters = 60
    oo = Options(); oo.max_iter = iters; oo.krylovkit_eager = True
    omv = []
    ref = oracle.solve(pr, oo, trace=True, proj_callback=lambda it, xi, xo, p_, arc: omv.append(int(arc[0].matvecs)))
    per_it = np.diff(np.array([0] + omv))
    sol = Optimizer(max_iter=iters, krylovkit_eager=1).optimize(pr, trace_capacity=iters)
    assert sol.status == ref.status and sol.iter == ref.iter
    assert np.array_equal(sol.trace[:, 13], per_it.astype(float)), (sol.trace[:, 13], per_it)
    G, T = _trace_cols(ref.trace), sol.trace[:, [1, 2, 3, 4, 7, 11]]
    assert np.array_equal(T[:, 5], G[:, 5])
    assert np.allclose(T, G, rtol=1e-8, atol=1e-10 * np.abs(G).max())


@pytest.mark.parametrize("case", PROJ_CASES, ids=[c[0] for c in PROJ_CASES])
def test_psd_projection_matches_golden_and_oracle(case, golden_dir):
    name, n, seed, top, tr, full = case
    gold = np.load(golden_dir / "psd_projection.npz")
    x = planted_packed(n, seed, top)
    out, info = B.psd_project(x, n, tr, mode=1 if full else 0)
    g_out, meta = gold[name + "__out"], gold[name + "__meta"]
    scale = np.abs(x).max()
    assert np.allclose(out, g_out, rtol=0, atol=2e-9 * scale), np.abs(out - g_out).max()
    assert info["rank"] == int(meta[4])
    assert abs(info["min_eig"] - meta[5]) <= 1e-9 * scale
    if not full:
        assert info["fell_back"] == 0
        assert info["nmatvec"] == int(meta[6])
    # and live against the oracle
    o_out, o_rank, o_min, _ = oracle_project(x, n, tr, full)
    assert np.allclose(out, o_out, rtol=0, atol=2e-9 * scale)


def test_projection_properties_at_full_size():
    """n = 4000: a planted rank-3 PSD matrix is a fixed point of the rank-4
    projection; projecting X and -X splits X (Moreau) on the captured subspace."""
    n, r = 4000, 3
    rng = np.random.default_rng(3)
    Z, _ = np.linalg.qr(rng.standard_normal((n, r)))
    lam = np.array([300.0, 120.0, 45.0])
    ii = np.concatenate([np.arange(j + 1) for j in range(n)])
    jj = np.repeat(np.arange(n), np.arange(1, n + 1))
    W = Z * lam
    x = np.einsum("ik,ik->i", W[ii], Z[jj]) * np.where(ii == jj, 1.0, math.sqrt(2.0))
    out, info = B.psd_project(x, n, 4)
    # the 4th Ritz value is 0 up to rounding; `val > 0` (prox_operators.jl:101) may count it
    assert info["rank"] in (3, 4) and info["fell_back"] == 0
    assert np.abs(out - x).max() <= 1e-9 * np.abs(x).max()
    out2, _ = B.psd_project(out, n, 4)                         # idempotence
    assert np.abs(out2 - out).max() <= 1e-9 * np.abs(x).max()
    neg, info_neg = B.psd_project(-x, n, 4)                    # -X has no positive part
    assert np.abs(neg).max() <= 1e-9 * np.abs(x).max()


# ----------------------------------------------------------------- solves
def _gopt(**kw):
    o = Optimizer(tol_gap=1e-6, tol_feasibility=1e-6, time_limit=30.0)
    for k, v in kw.items():
        o.set_attribute(k, v)
    return o


@pytest.mark.parametrize("support_path", [0, 1], ids=["dense", "support"])
@pytest.mark.parametrize("name", list(KATS))
def test_known_answers_and_golden_results(name, support_path, golden_dir):
    """The reference's KATs (test/moi_proxsdp_unit.jl) on the HIP path, and the
    oracle's committed final Result for the same problem -- with the dense vector
    passes and with the support-aware ones forced on (falls back to dense where the
    path is not legal: LPs, 1x1 blocks)."""
    build, expected, atol, xexp = KATS[name]
    gold = json.loads((golden_dir / "kat_results.json").read_text())[name]
    opt = _gopt(support_path=support_path)
    sol = opt.optimize(build())
    assert opt.termination_status() == "OPTIMAL"
    assert opt.primal_status() == "FEASIBLE_POINT" and opt.dual_status() == "FEASIBLE_POINT"
    assert abs(opt.objective_value() - expected) <= atol
    if xexp is not None:
        assert np.allclose(opt.variable_primal(), xexp, atol=atol)
    # against the oracle: identical algorithm in fp64 -> same iteration count, same numbers
    assert sol.iter == gold["iter"] and sol.final_rank == gold["final_rank"]
    assert abs(sol.objval - gold["objval"]) <= 1e-9 * (1 + abs(gold["objval"]))
    assert abs(sol.dual_objval - gold["dual_objval"]) <= 1e-9 * (1 + abs(gold["dual_objval"]))
    for key in ("primal", "dual_cone", "dual_eq", "dual_in", "slack_eq", "slack_in"):
        assert np.allclose(getattr(sol, key), gold[key], rtol=0, atol=1e-8), key


@pytest.mark.parametrize("settings", [
    dict(eigsolver=1, min_size_krylov_eigs=1),
    dict(eigsolver=2, min_size_krylov_eigs=1),
    dict(full_eig_decomp=1),
])
def test_sdp_wiki_all_eig_paths(settings):
    """moi_proxsdp_unit.jl:358-370."""
    for mx, exp in ((False, -0.978), (True, 0.872)):
        opt = _gopt(**settings)
        sol = opt.optimize(sdp_wiki(mx))
        assert opt.termination_status() == "OPTIMAL" and abs(opt.objective_value() - exp) <= 1e-2
        if settings.get("eigsolver") == 2:
            assert sol.stats["lanczos_matvecs"] > 0 and sol.stats["full_eigs"] == 0
        if settings.get("eigsolver") == 1:
            assert sol.stats["krylov_fallbacks"] == sol.iter      # ncv > n -> ARPACK error -> full_eig!


def test_termination_statuses():
    """test_terminationstatus.jl:40-73."""
    opt = Optimizer()
    opt.optimize(simple_lp())
    assert opt.termination_status() == "OPTIMAL"
    opt = Optimizer(max_iter=1)
    opt.optimize(simple_lp())
    assert opt.termination_status() == "ITERATION_LIMIT" and opt.pdhg_iterations() >= 0
    opt = Optimizer(tol_gap=1e-16, tol_feasibility=1e-16, tol_primal=1e-16, tol_dual=1e-16,
                    tol_feasibility_dual=1e-16, tol_psd=1e-16, time_limit=0.0)
    opt.optimize(simple_lp())
    assert opt.termination_status() == "TIME_LIMIT"


@pytest.mark.parametrize("build", [soc_norm, sdp_plus_soc], ids=["soc_norm", "sdp_plus_soc"])
def test_soc_cones_against_oracle(build):
    """soc_projection! / soc_convergence (prox_operators.jl:138-158, residuals.jl:73-86) and the
    cones-first variable order with free variables (scaling.jl:2-26)."""
    pr = build()
    opt = _gopt()
    sol = opt.optimize(pr)
    o = Options()
    o.tol_gap = o.tol_feasibility = 1e-6
    ref = oracle.solve(pr, o)
    assert sol.status == ref.status == 1 and sol.iter == ref.iter
    assert abs(sol.objval - ref.objval) <= 1e-9 * (1 + abs(ref.objval))
    assert np.allclose(sol.primal, ref.primal, atol=1e-8) and np.allclose(sol.dual_cone, ref.dual_cone, atol=1e-8)
    if build is soc_norm:
        assert abs(opt.objective_value() - 5.0) < 1e-4


@pytest.mark.parametrize("seed", [0, 1])
def test_mixed_cone_model_against_oracle(seed):
    """Every variable class in one model, user variable ids shuffled: 1x1 PSD blocks, a 3x3 and a
    5x5 block (full_eig!), a 104x104 block (Lanczos), one SOC, free variables, sparse equalities
    and inequalities.  Exercises preprocess! reordering, the concurrent block projections and the
    un-permutation of the results."""
    pr = mixed_cones(seed)
    opt = _gopt()
    sol = opt.optimize(pr, trace_capacity=300)
    o = Options()
    o.tol_gap = o.tol_feasibility = 1e-6
    ref = oracle.solve(pr, o, trace=True)
    assert sol.status == ref.status == 1
    assert abs(sol.iter - ref.iter) <= max(3, 0.05 * ref.iter)
    m = min(len(ref.trace), len(sol.trace), 30)
    G, T = _trace_cols(ref.trace)[:m], sol.trace[:m, [1, 2, 3, 4, 7, 11]]
    assert np.allclose(T, G, rtol=1e-7, atol=1e-9 * max(1.0, np.abs(G).max()))
    assert abs(sol.objval - ref.objval) <= 2e-6 * (1 + abs(ref.objval))
    sc = max(1.0, np.abs(ref.primal).max())
    assert np.allclose(sol.primal, ref.primal, atol=2e-5 * sc)
    assert np.allclose(sol.slack_eq, ref.slack_eq, atol=2e-5 * sc) and np.allclose(sol.slack_in, ref.slack_in, atol=2e-5 * sc)
    assert sol.stats["lanczos_matvecs"] > 0 and sol.stats["full_eigs"] > 0
    # the answer itself: cone membership in USER variable order
    for idx, side in zip(pr.psd, pr.psd_sides()):
        assert np.linalg.eigvalsh(P.unpack_psd(sol.primal[idx], side)).min() >= -1e-6
    t = sol.primal[pr.soc[0]]
    assert t[0] >= np.linalg.norm(t[1:]) - 1e-6


def test_unbounded_lp_certificate_search():
    """Unbounded LP: objective blow-up -> certificate search with b,h zeroed -> primal ray
    (pdhg.jl:184-244, 407-422, 639-676) -- same branch sequence as the oracle."""
    pr = unbounded_lp()
    opt = _gopt()
    sol = opt.optimize(pr)
    o = Options()
    o.tol_gap = o.tol_feasibility = 1e-6
    ref = oracle.solve(pr, o)
    assert sol.status == ref.status == 5 and sol.certificate_found and ref.certificate_found
    assert opt.termination_status() == "DUAL_INFEASIBLE" and opt.primal_status() == "INFEASIBILITY_CERTIFICATE"
    assert sol.iter == ref.iter
    assert "Primal ray found" in sol.status_string


def test_certificate_search_on_a_psd_model_support_and_dense_paths():
    """Certificate search on an infeasible SDP whose PSD block (side 110) takes the Lanczos path: the
    snapshot taken when infeasibility is declared rescales the iterate IN PLACE (pdhg.jl:749-755), after
    which the operator-form factors of the support path no longer describe it -- they are dropped
    (ADVICE round 1).  Support path (operator-form mat-vec), dense path and the oracle: status
    INFEASIBLE with a dual ray found, iteration counts within 5 %."""
    from kat_problems import infeasible_sdp
    pr = infeasible_sdp()
    ref = oracle.solve(pr, Options())
    assert ref.status == 6 and ref.certificate_found
    for sp in (0, 1):
        opt = Optimizer(support_path=sp)
        sol = opt.optimize(pr)
        print("support_path", sp, sol.status, sol.iter, sol.status_string, "| oracle", ref.iter)
        assert sol.status == 6 and sol.certificate_found
        assert "[Dual ray found]" in sol.status_string
        assert abs(sol.iter - ref.iter) <= 0.05 * ref.iter
        if sp == 1:
            assert sol.stats["fop_projections"] > 0


def test_infeasible_lp_prefix_matches_oracle():
    """Infeasible LP (the reference needs ~1e6 iterations to declare it): the first 3000
    iterations follow the oracle exactly (no eigen-solver involved)."""
    pr = infeasible_lp()
    opt = Optimizer(max_iter=3000)
    sol = opt.optimize(pr, trace_capacity=3000)
    o = Options()
    o.max_iter = 3000
    ref = oracle.solve(pr, o, trace=True)
    assert sol.status == ref.status and sol.iter == ref.iter
    G = np.array([[t["prim_obj"], t["dual_obj"], t["feas"], t["primal_step"]] for t in ref.trace])
    T = sol.trace[:, [1, 2, 4, 7]]
    assert np.allclose(T, G, rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("n", [2, 3, 4, 5])
def test_mimo_property(n):
    """moi_mimo.jl:71-75."""
    opt = _gopt()
    sol = opt.optimize(P.mimo(n, seed=123))
    X = P.unpack_psd(sol.primal, n + 1)
    assert opt.termination_status() == "OPTIMAL"
    assert np.all(np.abs(X) > 0.99) and np.all(np.abs(X) < 1.01)


def _assert_trace_rows(T, G, upto, what):
    """rows [0, upto) of a library trace against oracle rows: discrete columns equal, continuous
    columns to 1e-9 (objectives, steps) / 1e-7 (residual-type quantities, quotients of differences)."""
    assert np.array_equal(T[:upto, 0], G[:upto, 0]), what                     # iter
    assert np.array_equal(T[:upto, 10], G[:upto, 10]), what                   # target_rank schedule
    assert np.array_equal(T[:upto, 11], G[:upto, 11]), what                   # linesearch trials
    for col, nm in ((1, "prim_obj"), (2, "dual_obj"), (7, "primal_step"), (8, "beta"), (9, "theta")):
        assert np.allclose(T[:upto, col], G[:upto, col], rtol=1e-9, atol=1e-12), (what, nm)
    for col, nm in ((3, "gap"), (4, "feas"), (5, "prim_res"), (6, "dual_res")):
        assert np.allclose(T[:upto, col], G[:upto, col], rtol=1e-7, atol=1e-12), (what, nm)


@pytest.mark.parametrize("merge", [-1, 1], ids=["eigensolve-auto", "eigensolve-by-merge"])
@pytest.mark.parametrize("name", ["maxcut_readme_n4", "sdplib_mcp124-1", "maxcut_er_n200_s0"])
def test_iteration_traces_match_golden(name, merge, golden_dir):
    """Per-iteration parity with the oracle's committed traces (fp64 both sides), THROUGH the Lanczos
    restarts.  The comparison is tight on every row up to the first iteration whose projection input
    has lambda_r == lambda_{r+1} (recorded by the fixture generator with LAPACK, `degenerate_iters`):
    there the reference's rank-r truncation (prox_operators.jl:99-106) is defined only up to a rotation
    inside the eigenspace (mcp124-1, iteration 53: lambda_2 = ... = lambda_6 = 36.83154802, gap
    4e-11), so two correct eigensolvers legitimately continue on different trajectories.  From that
    row on the per-iterate criterion of test_projection_parity_on_oracle_iterates applies instead
    (same input to both projections); no sanity bounds are used.
    `eigensolve-by-merge` (host_eig_merge = 1): the K x K Rayleigh-quotient eigensolves of these runs (krylovdim 25)
    go through the split + rank-one merge path that auto mode reserves for krylovdim >= 64 -- same rows."""
    gold = json.loads((golden_dir / "traces.json").read_text())[name]
    if name == "maxcut_readme_n4":
        pr, iters = P.maxcut_readme(), 200
    elif name == "sdplib_mcp124-1":
        pr, iters = P.sdplib(golden_dir / "sdplib" / "mcp124-1.dat-s"), 120
    else:
        pr, iters = P.maxcut(200, seed=0), 120
    opt = Optimizer(max_iter=iters, host_eig_merge=merge)
    sol = opt.optimize(pr, trace_capacity=iters)
    rows = np.array(gold["rows"])
    assert sol.status == gold["status"] and sol.iter == gold["iter"]
    if merge == 1 and name != "maxcut_readme_n4":
        assert sol.stats["host_eig_merges"] > 0
    k = min(len(rows), len(sol.trace))
    assert k == len(rows)
    deg = gold["degenerate_iters"]
    upto = (deg[0] - 1) if deg else k                 # rows before the first degenerate projection
    restarts_before = [it for it in gold["restart_iters"] if it <= upto]
    if name != "maxcut_readme_n4":
        assert restarts_before, "the tight window must include Lanczos restarts"
        assert upto >= 50
    _assert_trace_rows(sol.trace, rows, upto, name)
    if not deg:
        assert abs(opt.objective_value() - gold["objval"]) <= 1e-9 * (1 + abs(gold["objval"]))


def test_metric_instance_first_iterations_match_oracle_trace(golden_dir):
    """The metric's own instance (Max-Cut ER n=4000, seed 0, reference default options): the first
    PDHG iterations (as many as the committed fixture holds: TRACE4000_ITERS of
    tests/golden/make_golden_large.py, minutes of CPU there) against the oracle trace.  Includes
    iteration 2 (55 mat-vecs: three thick restarts), iteration 10 (one restart) and every later restart
    and rank update in the window; same mat-vec count per iteration, same linesearch trials and rank
    schedule, objectives / steps to 1e-9."""
    gold = json.loads((golden_dir / "trace_maxcut_n4000.json").read_text())
    pr = P.maxcut(gold["n"], seed=gold["seed"])
    rows = np.array(gold["rows"])
    for kw in (dict(), dict(lanczos_operator=0), dict(support_path=0)):
        opt = Optimizer(max_iter=len(rows), **kw)
        sol = opt.optimize(pr, trace_capacity=len(rows))
        assert sol.status == gold["status"] and sol.iter == gold["iter"]
        assert np.array_equal(sol.trace[:, 13], np.array(gold["matvecs"], float)), kw   # Lanczos mat-vecs per iteration
        assert max(gold["matvecs"]) > 25
        _assert_trace_rows(sol.trace, rows, len(rows), str(kw))
        assert abs(opt.objective_value() - gold["objval"]) <= 1e-9 * (1 + abs(gold["objval"]))


def test_headline_regime_rank63_matches_oracle_trace(golden_dir):
    """the HEADLINE regime of bench.py at its own size -- Max-Cut n=4000 started at target rank
    63 ~ sqrt(n) (initial_target_rank = 63, max_target_rank_krylov_eigs = 64: nev = 63, krylovdim = 127, up to
    518 mat-vecs = 7 thick restarts in one projection) -- against a committed oracle trace of the same options
    (tests/golden/make_golden_large.py trace4000r63, 40 iterations).  Identical Lanczos mat-vec counts per
    iteration and objectives / steps to 1e-9 on the operator-form path (the default), on the packed-triangle
    operator and on the dense-vector path."""
    gold = json.loads((golden_dir / "trace_maxcut_n4000_rank63.json").read_text())
    pr = P.maxcut(gold["n"], seed=gold["seed"])
    rows = np.array(gold["rows"])
    assert gold["initial_target_rank"] == 63 and max(gold["matvecs"]) > 400 and len(rows) >= 10
    base = dict(initial_target_rank=gold["initial_target_rank"], max_target_rank_krylov_eigs=gold["max_target_rank_krylov_eigs"])
    for kw in (dict(), dict(lanczos_operator=0), dict(support_path=0)):
        opt = Optimizer(max_iter=len(rows), **base, **kw)
        sol = opt.optimize(pr, trace_capacity=len(rows))
        assert sol.status == gold["status"] and sol.iter == gold["iter"]
        assert np.array_equal(sol.trace[:, 13], np.array(gold["matvecs"], float)), (kw, sol.trace[:, 13], gold["matvecs"])
        assert np.array_equal(sol.trace[:, 10], rows[:, 10])                          # target rank stays 63
        _assert_trace_rows(sol.trace, rows, len(rows), "rank63 " + str(kw))
        assert abs(opt.objective_value() - gold["objval"]) <= 1e-9 * (1 + abs(gold["objval"]))
        if not kw:
            assert sol.stats["fop_projections"] >= len(rows) - 1, "operator-form mat-vec not taken on the default path"


def test_captured_iterate_projection_fixtures(golden_dir):
    """projection pairs for CAPTURED PDHG ITERATES.  The committed fixtures hold, for
    the first three restart-needing iterations of mcp124-1 and Max-Cut n=200, the vector handed to
    psd_projection! and the oracle's result; the same input goes through proxsdp_hip_psd_project.
    Criterion: 1e-9 |X| agreement, same rank / mat-vec count / min_eig -- or the input is degenerate at
    the truncation rank and the outputs differ only by a rotation inside that eigenspace."""
    z = np.load(golden_dir / "captured_projections.npz")
    keys = sorted(k[:-4] for k in z.files if k.endswith("__in"))
    assert len(keys) == 6
    kinds = {}
    for key in keys:
        n, it, tr, rank, mineig, mv, rs, conv = z[key + "__meta"]
        n, tr = int(n), int(tr)
        x_in, vals, vecs = z[key + "__in"], z[key + "__vals"], z[key + "__vecs"]
        ref_out = svec((vecs * vals) @ vecs.T)
        o = B.default_options()
        out, info = B.psd_project(x_in, n, tr, mode=0, options=o)
        kind = check_truncated_projection(x_in, n, tr, out, ref_out)
        kinds[key] = kind
        assert info["fell_back"] == 0 and info["rank"] == int(rank)
        assert info["nmatvec"] > 25                                   # a restart was needed here too
        if kind == "tight":
            assert info["nmatvec"] == int(mv), (key, info, mv)
            assert info["min_eig"] == pytest.approx(mineig, rel=1e-9, abs=1e-10 * np.linalg.norm(x_in))
            assert info["converged"] == int(conv)
    print(kinds)
    assert sum(v == "tight" for v in kinds.values()) >= 5


@pytest.mark.parametrize("case", ["sdplib_mcp124-1", "maxcut_er_n200_s0", "maxcut_er_n1000_s0"])
def test_projection_parity_on_oracle_iterates(case, golden_dir):
    """The truncated-projection criterion applied ITERATION BY ITERATION on a live oracle run: every
    projection input the oracle meets from its first restart on (mcp124-1 / n=200: all 120 iterations;
    n=1000: iteration 2 -- 65 mat-vecs, four restarts -- and the 11 iterations after it, one dense
    iterate is 4 MB) is handed to the library's
    projection; each result matches the oracle's to 1e-9 |X| or sits on a LAPACK-verified degenerate
    cluster.  Together with the bit-exact vector-kernel seams this is per-iteration parity of the loop
    after trajectories may have separated."""
    from helpers import capture_restart_projections
    if case == "sdplib_mcp124-1":
        pr, iters, cnt, every = P.sdplib(golden_dir / "sdplib" / "mcp124-1.dat-s"), 120, 10 ** 6, True
    elif case == "maxcut_er_n200_s0":
        pr, iters, cnt, every = P.maxcut(200, seed=0), 400, 10 ** 6, True
    else:
        pr, iters, cnt, every = P.maxcut(1000, seed=0), 40, 12, True
    caps = capture_restart_projections(pr, iters, cnt, all_after_first=every)
    assert len(caps) >= (12 if case == "maxcut_er_n1000_s0" else 60)
    ntight = ndeg = 0
    for c in caps:
        o = B.default_options()
        out, info = B.psd_project(c["x_in"], c["n"], c["target_rank"], mode=0, options=o)
        kind = check_truncated_projection(c["x_in"], c["n"], c["target_rank"], out, c["x_out"])
        assert info["fell_back"] == 0 and info["rank"] == c["rank"], (c["iter"], info)
        if kind == "tight":
            ntight += 1
            assert info["nmatvec"] == c["matvecs"], (c["iter"], info["nmatvec"], c["matvecs"])
        else:
            ndeg += 1
    print(case, "iterates", len(caps), "tight", ntight, "degenerate", ndeg)
    assert ntight >= len(caps) - 3


def test_maxcut_n1000_objective_matches_oracle_solve(golden_dir):
    """BASELINE config 2 (Max-Cut ER n=1000) solved to tol 1e-4 with REFERENCE DEFAULT options on both
    sides; the oracle's result (status, iterations, objective; ~25 min of CPU in the build container)
    is committed by tests/golden/make_golden_large.py.  north_star: same objective within 1e-4."""
    gold = json.loads((golden_dir / "solve_maxcut_n1000.json").read_text())
    pr = P.maxcut(gold["n"], seed=gold["seed"])
    opt = Optimizer()
    sol = opt.optimize(pr)
    print("gpu", sol.status, sol.iter, opt.objective_value(), sol.gap, "oracle", gold["status"], gold["iter"],
          gold["objval"], gold["gap"])
    assert sol.status == gold["status"] == 1
    assert abs(opt.objective_value() - gold["objval"]) <= 1e-4 * (1 + abs(gold["objval"]))
    assert sol.gap <= 1e-4 and sol.primal_feasible_user_tol
    assert sol.iter == gold["iter"]                      # 5921 = 5921 (measured every round since the oracle solve was committed)


def _first_departure(T, G, gm, mv, rtol):
    """first row where the library's trace leaves the oracle's: objective / gap / residual columns beyond rtol, another
    number of linesearch trials, another target rank or another Lanczos mat-vec count"""
    m = min(len(T), len(G))
    sc = np.abs(G[:m, 1:8]).max(axis=0)
    ok = (np.all(np.abs(T[:m, 1:8] - G[:m, 1:8]) <= rtol * np.abs(G[:m, 1:8]) + 1e-3 * rtol * sc, axis=1)
          & (T[:m, 11] == G[:m, 11]) & (T[:m, 10] == G[:m, 10]) & (mv[:m] == gm[:m]))
    return int(np.argmin(ok)) if not ok.all() else m


def test_sdplib_500_instances_follow_the_oracle_trace_until_the_degenerate_iterates(golden_dir):
    """SDPLIB mcp500-1 and gpp500-1 with REFERENCE DEFAULT options (Krylov path) against the oracle's first 400 iterations
    (tests/golden/trace_sdplib500.json, make_golden_sdplib500_trace.py) -- ADVICE r3: say WHERE the two sides part instead
    of comparing the end states of two 5000-iteration solves.
    Measured (round 4, tools/gpurun_ab.py): on mcp500-1 library and oracle agree to 1e-13 (objectives; all seven trace
    columns to 1e-9 for 117 iterations) with identical linesearch trials and Lanczos mat-vec counts for the first 125
    iterations; at iteration 126 the truncated projection
    meets a (near-)repeated eigenvalue at the truncation edge (any orthonormal basis of that
    eigenspace is a valid KrylovKit answer, the two sides return different ones), the mat-vec counts differ and the
    trajectories are 1e-3 apart 70 iterations later (gpp500-1: 169 iterations to 1e-9).  From there on EVERY build is its own trajectory (round 3's and
    round 4's library, bit-identical to each other, and the oracle): end states are compared only through the solver's
    own criteria and the literature optimum (next test)."""
    gold = json.loads((golden_dir / "trace_sdplib500.json").read_text())
    measured = {"mcp500-1": 117, "gpp500-1": 169}
    for name in ("mcp500-1", "gpp500-1"):
        g = gold[name]
        G = np.array(g["rows"]); gm = np.array(g["matvecs"])
        pr = P.sdplib(golden_dir / "sdplib" / f"{name}.dat-s")
        sol = Optimizer(max_iter=len(G)).optimize(pr, trace_capacity=len(G))
        T = sol.trace[:, :12]
        first = _first_departure(T, G, gm, sol.trace[:, 13], 1e-9)
        loose = _first_departure(T, G, gm * 0, sol.trace[:, 13] * 0, 1e-6)
        print(name, "trace + trial counts + mat-vec counts equal to 1e-9 for the first", first, "iterations; to 1e-6 without the mat-vec counts:", loose)
        assert first >= 100, (name, first)
        if measured[name] is not None:
            assert first >= measured[name] - 5


def test_sdplib_500_instances_solved_to_tolerance_against_the_oracle_solves(golden_dir):
    """SDPLIB mcp500-1 and gpp500-1 with REFERENCE DEFAULT options (Krylov path), solved to tol 1e-4 by the oracle
    (tests/golden/make_golden_sdplib500.py: 42 and 9 min of CPU) and by the library.  Both trajectories leave the
    oracle's after ~125 iterations (previous test), so the end states are compared through what the solver itself
    promises: status OPTIMAL by the reference's rule on both sides, the first rank update at the same iteration, the
    objective inside the stop rule's slack around the literature optimum (mcp500-1 598.15: the rule stops both sides
    ~1e-3 relative short of it -- oracle 597.13 after 5182 iterations, library 597.15 after 5086; gpp500-1 25.3205: the
    library reaches it after 7385 iterations, the ORACLE stops at 26.89 after 4403 because the reference's rule does
    not test dual feasibility and fires where primal and dual objective cross on the way down -- DESIGN.md section 7;
    recorded, not asserted: it is the oracle's end state, not a property of the library)."""
    gold = json.loads((golden_dir / "solve_sdplib500.json").read_text())

    def schedule(sol):
        out = []
        for row in sol.trace:
            if not out or out[-1][1] != int(row[10]):
                out.append([int(row[0]), int(row[10])])
        return out
    g = gold["mcp500-1"]
    pr = P.sdplib(golden_dir / "sdplib" / "mcp500-1.dat-s")
    sol = Optimizer().optimize(pr, trace_capacity=20000)
    print("mcp500-1 gpu", sol.status, sol.iter, sol.objval, schedule(sol), "oracle", g["status"], g["iter"], g["objval"], g["rank_schedule"])
    assert sol.status == g["status"] == 1
    assert sol.gap <= 1e-4 and sol.primal_feasible_user_tol
    assert schedule(sol)[:2] == g["rank_schedule"][:2]
    assert abs(abs(sol.objval) - 598.15) <= 3.5e-3 * 598.15 and abs(abs(g["objval"]) - 598.15) <= 3.5e-3 * 598.15
    g = gold["gpp500-1"]
    pr = P.sdplib(golden_dir / "sdplib" / "gpp500-1.dat-s")
    sol = Optimizer().optimize(pr, trace_capacity=20000)
    print("gpp500-1 gpu", sol.status, sol.iter, sol.objval, schedule(sol), "oracle", g["status"], g["iter"], g["objval"], g["rank_schedule"])
    assert sol.status == g["status"] == 1
    assert sol.gap <= 1e-4 and sol.primal_feasible_user_tol
    assert schedule(sol)[:2] == g["rank_schedule"][:2]
    assert abs(sol.objval - 25.3205) <= 1e-3 * 25.3205                      # the literature optimum


def test_round4_step_kernels_reproduce_round3_bit_for_bit(golden_dir):
    """Round 4 rebuilt the Lanczos step kernels (producer-major partial records reduced lane <-> column, no loads beyond
    column k, unconditional row sums) WITHOUT moving a bit: the record sums keep the balanced-tree order of round 3's
    fold network (kernels.hip.hpp tree_in_wave / tree_across).  Pinned against traces written by the ROUND-3 library
    (tools/gpurun_ab.py on the build of commit af9cfd5): objectives, residuals and Lanczos mat-vec counts of the first 900
    iterations of Max-Cut n = 2000 (default options: packed operator at first, operator form, 45 restarts, two rank
    updates) and of the first 40 iterations of the headline regime (n = 4000, nev 63, krylovdim 127) must be EQUAL as
    float64 bit patterns.  (A plain sequential record sum is 3 % faster and moved one rank update of the n = 2000 golden
    solve by one iteration -- DESIGN.md section 5.)  An intentional change of summation order has to regenerate these files."""
    for fname, pr, kw in (("bits_maxcut_n2000_round3.npy", P.maxcut(2000, seed=0), dict(max_iter=900)),
                          ("bits_maxcut_n4000_rank63_round3.npy", P.maxcut(4000, seed=0),
                           dict(max_iter=40, initial_target_rank=63, max_target_rank_krylov_eigs=64))):
        gold = np.load(golden_dir / fname)
        sol = Optimizer(**kw).optimize(pr, trace_capacity=kw["max_iter"])
        got = sol.trace[:len(gold)][:, [1, 2, 5, 6, 13]]
        same = (got.view(np.uint64) == gold.view(np.uint64)).all(axis=1)
        print(fname, "rows equal as bit patterns:", int(same.sum()), "of", len(gold))
        assert same.all(), np.nonzero(~same)[0][:5]


def test_rank_one_merge_on_helper_threads_does_not_move_a_bit():
    """options.host_merge_threads: the secular roots, Gu-Eisenstat weights and eigenvector columns of the K x K eigensolve's
    rank-one merge are independent per root / column and go to spinning helper threads in chunks; who computes a root must
    not matter.  Headline regime (n = 4000, nev 63, krylovdim 127: every restart and every final eigensolve is a merge),
    60 iterations: trace columns and mat-vec counts equal as bit patterns with 0, 3 (the default) and 6 helpers."""
    pr = P.maxcut(4000, seed=0)
    ref = None
    for ht in (0, 3, 6):
        sol = Optimizer(max_iter=60, initial_target_rank=63, max_target_rank_krylov_eigs=64, host_merge_threads=ht).optimize(pr, trace_capacity=60)
        assert sol.stats["host_eig_merges"] > 0
        got = sol.trace[:, [1, 2, 3, 4, 5, 6, 7, 13]].copy()
        if ref is None:
            ref = got
        else:
            assert np.array_equal(got.view(np.uint64), ref.view(np.uint64)), ht


def test_maxG51_default_options_follows_the_oracle_into_the_100_restart_regime(golden_dir):
    """VERDICT r3 item 4 (maxG51 with default options ends at the time limit): what the reference's algorithm does on
    this instance, pinned by 12 min of oracle CPU (tests/golden/trace_maxG51_default.json).  4012 iterations of 25-42
    Lanczos mat-vecs; at iteration 4013 the target rank goes 8 -> 9 and from then on EVERY projection is KrylovKit's
    full 100 restarts (krylovdim = max(2 nev + 1, 25) = 25 cannot separate the clustered ninth eigenvalue: 719 mat-vecs,
    8 of 9 pairs converged, the unconverged one dropped by prox_operators.jl:99).  Library vs oracle: the SAME mat-vec
    count in every one of the 4150 iterations, the same seven rank updates at the same iterations, objectives to 1e-7
    (1e-9 from iteration 2000 on).
    (With full_eig_decomp = true the same instance is OPTIMAL after 1933 iterations on both sides: config 5's test.)"""
    gold = json.loads((golden_dir / "trace_maxG51_default.json").read_text())
    pr = P.sdplib(golden_dir / "sdplib" / "maxG51.dat-s")
    sol = Optimizer(max_iter=gold["iter"]).optimize(pr, trace_capacity=gold["iter"])
    T = sol.trace
    gm = np.array(gold["matvecs"])
    assert sol.iter == gold["iter"]
    same = T[:, 13] == gm
    print("mat-vec counts equal in", int(same.sum()), "of", len(gm), "; last 137 iterations:", int(gm[-1]), "mat-vecs each")
    assert same.all(), np.nonzero(~same)[0][:10]
    sched = []
    for row in T:
        if not sched or sched[-1][1] != int(row[10]):
            sched.append([int(row[0]), int(row[10])])
    assert sched == gold["rank_schedule"]
    G = np.array(gold["rows_every_25"])
    R = T[24::25, :12]
    assert np.array_equal(R[:, 0], G[:, 0]) and np.array_equal(R[:, 11], G[:, 11])
    # measured: 1.4e-8 at worst (the oscillating transient around iteration 700), 1e-11 from iteration 2000 on
    assert np.allclose(R[:, 1:3], G[:, 1:3], rtol=1e-7, atol=1e-7)
    assert np.allclose(R[80:, 1:3], G[80:, 1:3], rtol=1e-9, atol=1e-9)
    assert gm[4012] > 700 and gm[4011] < 50                                # the onset of the 100-restart regime
    assert sol.stats["krylov_fallbacks"] == 0


def test_maxG51_krylov_path_with_min_lanczos_40_takes_the_oracles_iterations(golden_dir):
    """The way out of the previous test's stall that the REFERENCE offers: its own option eigsolver_min_lanczos
    (options.jl; default 25) = 40 -- a Krylov space wide enough to converge the clustered pairs.  maxG51 solved to tol 1e-4
    on the Krylov path by both sides (oracle: 29 min of CPU, tests/golden/solve_maxG51_krylov40.json): OPTIMAL after the
    SAME 6488 iterations, the same 13 rank updates at the same iterations, objectives 1.7e-13 apart, Lanczos mat-vec
    totals within 0.1 % (428 116 vs 427 986); library: 5.3 s (VERDICT r3 item 4 asked for <= 60 s)."""
    gold = json.loads((golden_dir / "solve_maxG51_krylov40.json").read_text())
    pr = P.sdplib(golden_dir / "sdplib" / "maxG51.dat-s")
    opt = Optimizer(eigsolver_min_lanczos=40)
    sol = opt.optimize(pr, trace_capacity=gold["iter"] + 50)
    print("gpu", sol.status, sol.iter, sol.objval, sol.stats["lanczos_matvecs"], round(sol.time, 2), "s; oracle", gold["status"], gold["iter"],
          gold["objval"], gold["matvecs"])
    assert sol.status == gold["status"] == 1
    assert sol.iter == gold["iter"]
    assert abs(sol.objval - gold["objval"]) <= 1e-9 * abs(gold["objval"])
    assert abs(sol.stats["lanczos_matvecs"] - gold["matvecs"]) <= 0.005 * gold["matvecs"]
    sched = []
    for row in sol.trace:
        if not sched or sched[-1][1] != int(row[10]):
            sched.append([int(row[0]), int(row[10])])
    assert sched == gold["rank_schedule"]
    G = np.array(gold["rows_every_50"])
    R = sol.trace[49::50, :12]
    assert np.array_equal(R[:, 0], G[:, 0]) and np.array_equal(R[:, 11], G[:, 11])
    assert np.allclose(R[:, 1:3], G[:, 1:3], rtol=1e-6, atol=1e-6)
    assert abs(abs(sol.objval) - 4003.81) <= 1e-3 * 4003.81             # inside the stop rule's slack around the literature optimum
    assert sol.time <= 60.0


def test_config4_mimo_8x512_solved_to_tolerance_against_the_oracle_solve(golden_dir):
    """BASELINE config 4 at its own shape, solved by BOTH sides: eight MIMO detection SDPs (n = 512: PSD side 513,
    box rows on every entry) as one block-diagonal model, reference default options, tol 1e-4.  The oracle's solve
    (35 s of CPU; tests/golden/solve_mimo_n512_x8.json, generated by the script in its `generator` field) against the
    library's batched multi-block Lanczos on the general vector path: same status and iteration count, the same
    linesearch trial counts, traces to 1e-6 up to where the degenerate iterates (repeated eigenvalues, DESIGN.md
    section 6) let rounding through, objective within the solver's tolerance."""
    gold = json.loads((golden_dir / "solve_mimo_n512_x8.json").read_text())
    pr = P.block_diag_problems([P.mimo(512, seed=s_) for s_ in range(8)], name="mimo-x8")
    sol = Optimizer().optimize(pr, trace_capacity=gold["iter"] + 20)
    G = np.array(gold["rows"])
    T = sol.trace[:, :12]
    print("gpu", sol.status, sol.iter, sol.objval, sol.stats["lanczos_matvecs"], "oracle", gold["status"], gold["iter"],
          gold["objval"], gold["matvecs"])
    assert sol.status == gold["status"] == 1
    assert sol.iter == gold["iter"]                      # 76 = 76
    assert abs(sol.objval - gold["objval"]) <= 1e-4 * (1 + abs(gold["objval"]))
    assert sol.stats["batched_block_steps"] > 0
    m = min(len(T), len(G))
    close = np.all(np.isclose(T[:m, 1:8], G[:m, 1:8], rtol=1e-6, atol=1e-9), axis=1) & (T[:m, 11] == G[:m, 11])
    first_off = int(np.argmin(close)) if not close.all() else m
    print("traces agree (1e-6, trial counts) for the first", first_off, "of", m, "iterations")
    assert first_off >= 25                               # measured: 27 (rounds 4 and 5)
    assert abs(sol.stats["lanczos_matvecs"] - gold["matvecs"]) <= 0.05 * gold["matvecs"]


@pytest.mark.parametrize("row", [-1, 0])
def test_config5_maxG51_solved_to_tolerance_takes_the_oracles_iterations(row, golden_dir):
    """BASELINE config 5 at its own size, solved by BOTH sides: SDPLIB maxG51 (n = 1000) with full_eig_decomp = true,
    tol 1e-4.  The oracle (LAPACK full_eig! every iteration; ~15 min of CPU, tests/golden/make_golden_large.py
    solvemaxg51) and the library's sign-function projection -- on its shortened, tested schedule (default) and on the
    full table (sign_start_row = 0) -- must stop at the SAME iteration with the same objective, and every 50th trace
    row along the way must agree: a projection error of 1e-10 of the spectral scale per iteration does not move a
    1933-iteration trajectory off LAPACK's."""
    gold = json.loads((golden_dir / "solve_maxG51_full_eig.json").read_text())
    pr = P.sdplib(golden_dir / "sdplib" / "maxG51.dat-s")
    opt = Optimizer(full_eig_decomp=1, sign_start_row=row)
    sol = opt.optimize(pr, trace_capacity=gold["iter"] + 10)
    print("gpu", sol.status, sol.iter, sol.objval, "oracle", gold["status"], gold["iter"], gold["objval"],
          "short pass/fail", sol.stats["sign_short_pass"], sol.stats["sign_short_fail"])
    assert sol.status == gold["status"] == 1
    assert sol.iter == gold["iter"]
    assert abs(sol.objval - gold["objval"]) <= 1e-8 * abs(gold["objval"])
    assert abs(sol.dual_objval - gold["dual_objval"]) <= 1e-8 * abs(gold["dual_objval"])
    assert sol.final_rank == gold["final_rank"]
    G = np.array(gold["rows_every_50"])
    T = sol.trace[49::50, :12][:len(G)]
    assert np.array_equal(T[:, [0, 10, 11]], G[:, [0, 10, 11]])                 # iteration, target rank, linesearch trials
    assert np.allclose(T[:, 1:10], G[:, 1:10], rtol=1e-6, atol=1e-9 * np.abs(G[:, 1:3]).max())
    assert sol.stats["full_eigs_sign"] == sol.iter
    if row == 0:
        assert sol.stats["sign_products"] == 57 * sol.iter
    else:
        assert sol.stats["sign_short_pass"] >= 0.98 * sol.iter


@pytest.mark.parametrize("name", ["mcp250-1", "mcp500-1", "maxG11"])
def test_sdplib_full_eig_solves_take_the_oracles_iterations(name, golden_dir):
    """The regime of BASELINE config 5 (full_eig_decomp = true: every projection is full_eig!) on three more SDPLIB
    instances (Max-Cut family; the gpp instances need > 100 000 full_eig! iterations), solved to tol 1e-4 by the oracle (LAPACK; tests/golden/make_golden_full_eig.py) and by the library's
    sign-function projection on its shortened, tested schedule: same status, the same iteration count, objective to
    1e-7 relative, every 50th trace row to 1e-5."""
    gold = json.loads((golden_dir / "solve_sdplib_full_eig.json").read_text())
    if name not in gold:
        pytest.skip(f"{name} not in the committed fixture")
    g = gold[name]
    pr = P.sdplib(golden_dir / "sdplib" / f"{name}.dat-s")
    sol = Optimizer(full_eig_decomp=1, time_limit=600.0).optimize(pr, trace_capacity=g["iter"] + 10)
    print(name, "gpu", sol.status, sol.iter, sol.objval, "oracle", g["status"], g["iter"], g["objval"],
          "short pass/fail", sol.stats["sign_short_pass"], sol.stats["sign_short_fail"])
    assert sol.status == g["status"]
    assert sol.iter == g["iter"]
    assert abs(sol.objval - g["objval"]) <= 1e-7 * (1 + abs(g["objval"]))
    G = np.array(g["rows_every_50"])
    if len(G):
        T = sol.trace[49::50, :12][:len(G)]
        assert np.array_equal(T[:, [0, 10, 11]], G[:, [0, 10, 11]])
        assert np.allclose(T[:, 1:10], G[:, 1:10], rtol=1e-5, atol=1e-8 * np.abs(G[:, 1:3]).max())


def test_maxcut_n2000_solve_matches_oracle_through_the_implicit_full_eig_regime(golden_dir):
    """Max-Cut ER n = 2000 solved to tol 1e-4 with REFERENCE DEFAULT options by the oracle
    (tests/golden/make_golden_large.py solve2000: 7098 iterations, 30 min of CPU; from iteration 6369 on target_rank
    is 17 > max_target_rank_krylov_eigs, so its last 730 iterations are LAPACK full_eig! calls) against the
    library, whose implicit regime is served by the Lanczos engine (verified against the sign projection): same
    status, same rank schedule (16 rank updates at identical iterations), the same 7098 iterations, the same 730
    full_eig! calls, objective within 1e-6 relative (measured 3.1e-7; all THREE full_eig! engines of the library --
    Lanczos-served at posres 1e-7 / 1e-9 / 1e-11, sign function, rocSOLVER dsyevd -- agree with each other to 1e-12
    on this solve, so the 3e-7 is oracle-vs-library trajectory drift, 300x inside north_star's 1e-4)."""
    gold = json.loads((golden_dir / "solve_maxcut_n2000.json").read_text())
    pr = P.maxcut(gold["n"], seed=gold["seed"])
    opt = Optimizer()
    sol = opt.optimize(pr, trace_capacity=gold["iter"] + 200)
    sched = []
    for row in sol.trace:
        if not sched or sched[-1][1] != int(row[10]):
            sched.append([int(row[0]), int(row[10])])
    print("gpu", sol.status, sol.iter, opt.objective_value(), sol.stats["full_eigs"], sol.stats["full_eigs_lanczos"],
          sol.stats["full_eigs_lanczos_checks"], "oracle", gold["iter"], gold["objval"], gold["full_eigs"])
    assert sol.status == gold["status"] == 1
    assert sched == gold["rank_schedule"]
    assert sol.iter == gold["iter"]
    assert sol.stats["full_eigs"] == gold["full_eigs"] and sol.stats["full_eigs_lanczos"] >= gold["full_eigs"] - 5
    assert sol.stats["full_eigs_lanczos_mismatches"] == 0
    # every Lanczos-served call carried its certificate (deflated 10-step run): none failed
    assert sol.stats["full_eigs_lanczos_cert_failed"] == 0
    assert sol.stats["full_eigs_lanczos_certified"] == sol.stats["full_eigs_lanczos"]
    assert abs(opt.objective_value() - gold["objval"]) <= 1e-6 * (1 + abs(gold["objval"]))
    assert sol.final_rank == gold["final_rank"]


@pytest.mark.parametrize("fname,lit,tol", [("mcp124-1", -141.99, 1e-3), ("gpp124-2", 46.8623, 1e-3),
                                           ("mcp124-1", -141.99, 1e-4)])
def test_sdplib_against_oracle(fname, lit, tol, golden_dir):
    """moitest.jl:119-143 on the HIP path, side by side with the oracle run on this
    box's CPU: same status, both within the solver's own tolerances, objectives within
    tol_gap*(1+|po|+|do|) of each other (the solver's own gap measure), lambda_min >= -1e-4."""
    pr = P.sdplib(golden_dir / "sdplib" / f"{fname}.dat-s")
    opt = Optimizer(tol_gap=tol, tol_feasibility=tol)
    sol = opt.optimize(pr)
    o = Options()
    o.tol_gap = o.tol_feasibility = tol
    ref = oracle.solve(pr, o)
    X = P.unpack_psd(sol.primal, pr.psd_sides()[0])
    assert sol.status == ref.status == 1
    assert np.linalg.eigvalsh(X).min() >= -1e-4
    diff = abs(opt.objective_value() - ref.objval)
    print(f"{fname} tol={tol}: gpu obj {opt.objective_value():.8f} it {sol.iter}  oracle obj {ref.objval:.8f} "
          f"it {ref.iter}  |diff| {diff:.3e}")
    # Both runs stop as soon as gap <= tol and feas <= tol; on the Lanczos path they stop at
    # different iterations (truncated projection, see test_iteration_traces_match_golden).
    # Two eps-feasible, eps-gap points can differ in objective by the gap term plus the
    # dual-weighted infeasibility:  tol*(1+|po|+|do|) + |y|_1 * tol*(1+|b|)   (residuals.jl:2-35).
    bound = tol * (1 + abs(ref.objval) + abs(ref.dual_objval)) + \
        np.abs(ref.dual_eq).sum() * tol * (1 + np.linalg.norm(pr.b))
    assert diff <= bound, (diff, bound)
    assert abs(opt.objective_value() - lit) <= bound + 5 * tol * (1 + abs(lit))
    assert sol.gap <= tol and sol.primal_feasible_user_tol
    assert sol.stats["lanczos_matvecs"] > 0 and sol.stats["full_eigs"] == 0
    assert abs(sol.iter - ref.iter) <= 0.25 * ref.iter


@pytest.mark.parametrize("case", ["maxcut_n260", "mcp124-1"])
def test_arpack_path_against_oracle(case, golden_dir):
    """eigsolver = 1 (arpack_eig!, eigsolver.jl:748-770) on a Lanczos-sized block against the
    oracle, whose ARPACK path is SciPy's wrapper of the SAME dsaupd/dseupd Fortran the reference
    calls through Arpack.jl.  The library is not ARPACK: it runs its thick-restart engine with
    dsaupd's acceptance rule (|resid_i| <= tol max(eps^(2/3), |theta_i|) on all nev wanted pairs) and
    dsaupd's restart size.  Identical status, both within the solver's tolerances, objectives within
    the solver's own gap measure, iteration counts within 25 % (the bound of the KrylovKit test), and
    -- eigen layer -- the projections of the first iterates agree to 1e-8 |X|."""
    tol = 1e-4 if case == "maxcut_n260" else 1e-3
    pr = P.maxcut(260, seed=3) if case == "maxcut_n260" else P.sdplib(golden_dir / "sdplib" / "mcp124-1.dat-s")
    n = pr.psd_sides()[0]
    opt = Optimizer(eigsolver=1, tol_gap=tol, tol_feasibility=tol)
    sol = opt.optimize(pr)
    o = Options()
    o.eigsolver = 1
    o.tol_gap = o.tol_feasibility = tol
    ref = oracle.solve(pr, o)
    print(f"{case}: gpu it {sol.iter} obj {opt.objective_value():.8f} fallbacks {sol.stats['krylov_fallbacks']} | "
          f"oracle it {ref.iter} obj {ref.objval:.8f} fallbacks {ref.stats['krylov_fallbacks']}")
    assert sol.status == ref.status == 1
    assert sol.stats["lanczos_matvecs"] > 0
    bound = tol * (1 + abs(ref.objval) + abs(ref.dual_objval)) + \
        np.abs(ref.dual_eq).sum() * tol * (1 + np.linalg.norm(pr.b))
    assert abs(opt.objective_value() - ref.objval) <= bound
    assert sol.gap <= tol and sol.primal_feasible_user_tol
    # iteration counts: mcp124-1's iterates have 5-fold degenerate eigenvalues at the truncation rank (DESIGN.md
    # section 6), so its count to tolerance is chaotic -- measured on the ORACLE ITSELF: restating the
    # reference's in-place norm (an ulp-level change of y and Mty, pdhg.jl:560-575) moved its count from 2653
    # to 1925 (the library: 2692).  Band 50 % there, 25 % on the non-degenerate instance.
    assert abs(sol.iter - ref.iter) <= (0.5 if case == "mcp124-1" else 0.25) * ref.iter
    X = P.unpack_psd(sol.primal, n)
    assert np.linalg.eigvalsh(X).min() >= -1e-4
    # eigen layer on captured iterates: same input, ARPACK (SciPy) vs the library's dsaupd-rule engine
    from helpers import capture_restart_projections
    caps = capture_restart_projections(pr, 60, 10 ** 6, all_after_first=True, any_iter=True, eigsolver=1)
    assert len(caps) >= 40
    for c in caps[:40]:
        ob = B.default_options()
        B.set_option(ob, "eigsolver", 1)
        out, info = B.psd_project(c["x_in"], c["n"], c["target_rank"], mode=0, options=ob)
        kind = check_truncated_projection(c["x_in"], c["n"], c["target_rank"], out, c["x_out"], tol_rel=1e-8)
        assert info["fell_back"] == 0 and info["rank"] == c["rank"], (c["iter"], info, kind)


@pytest.mark.parametrize("support_path", [0, 1], ids=["dense", "support"])
def test_block_diagonal_model_two_blocks(support_path):
    """Two independent MIMO instances in one block-diagonal model (the shape of the
    'MIMO x 8 blocks' config) against the oracle."""
    pr = P.block_diag_problems([P.mimo(6, seed=1), P.mimo(7, seed=2)])
    opt = _gopt(support_path=support_path)
    sol = opt.optimize(pr)
    o = Options()
    o.tol_gap = o.tol_feasibility = 1e-6
    ref = oracle.solve(pr, o)
    assert sol.status == ref.status == 1 and sol.iter == ref.iter
    assert abs(sol.objval - ref.objval) <= 1e-8 * (1 + abs(ref.objval))
    assert np.allclose(sol.primal, ref.primal, atol=1e-7)


@pytest.mark.parametrize("kind", ["mimo_dense_passes", "maxcut_support_packed"])
def test_batched_multi_block_lanczos_equals_the_per_block_path_and_the_oracle(kind):
    """Row "batched symmetric mat-vec across blocks" (north_star; reference loop prox_operators.jl:40-61): PSD
    blocks of EQUAL side advance their Lanczos recurrences in ONE launch per step (grid.z = block,
    options.block_batch) instead of one stream + host thread per block.  Per block the kernels' bodies are the
    same, so the whole trace must be BIT-identical to the per-block path (block_batch = 0), the Lanczos mat-vec
    totals per iteration must be the oracle's (blocks restart at different basis sizes and drop out of the
    launches at different cycles), and the trace must follow the oracle's as the single-block tests do."""
    if kind == "mimo_dense_passes":
        # three MIMO instances of side 121 (> 100: Lanczos path; box rows on every entry: dense vector passes)
        pr = P.block_diag_problems([P.mimo(120, seed=s_) for s_ in (1, 2, 3)])
        kw, iters = dict(), 90
    else:
        # four Max-Cut blocks of side 150 on the support path with the packed-triangle operator
        pr = P.block_diag_problems([P.maxcut(150, seed=s_) for s_ in (0, 1, 2, 3)])
        kw, iters = dict(support_path=1, lanczos_operator=0), 120
    o = Options()
    o.max_iter = iters
    omv = []
    ref = oracle.solve(pr, o, trace=True, proj_callback=lambda it, xi, xo, p_, arc: omv.append(sum(int(a.matvecs) for a in arc)))
    per_it = np.diff(np.array([0] + omv))
    sb = Optimizer(max_iter=iters, block_batch=1, **kw).optimize(pr, trace_capacity=iters)
    ss = Optimizer(max_iter=iters, block_batch=0, **kw).optimize(pr, trace_capacity=iters)
    assert sb.stats["batched_block_steps"] > 0 and ss.stats["batched_block_steps"] == 0
    cols = [c for c in range(14) if c != 12]                               # (column 12 is the wall clock)
    assert np.array_equal(sb.trace[:, cols], ss.trace[:, cols]), "batched launch changed the arithmetic of a block"
    assert sb.stats["lanczos_matvecs"] == ss.stats["lanczos_matvecs"]
    assert sb.stats["lanczos_restarts"] == ss.stats["lanczos_restarts"]
    assert sb.status == ref.status and sb.iter == ref.iter
    G, T = _trace_cols(ref.trace), sb.trace[:, [1, 2, 3, 4, 7, 11]]
    assert np.array_equal(T[:, 5], G[:, 5])                                # linesearch trials
    mv = sb.trace[:, 13]
    same = mv == per_it[:len(mv)]
    tight = int(np.argmin(same)) if not same.all() else len(mv)            # degenerate truncation may part the runs later
    assert tight >= 10, (mv[:12], per_it[:12])
    assert np.allclose(T[:tight], G[:tight], rtol=1e-8, atol=1e-10 * np.abs(G).max())
    print(kind, "iterations with the oracle's mat-vec totals:", tight, "of", len(mv),
          "| batched block-steps", sb.stats["batched_block_steps"], "| restarts", sb.stats["lanczos_restarts"])


def test_mimo_config4_at_its_full_shape_batched_against_per_block():
    """BASELINE config 4 at its ACTUAL shape (VERDICT r2 config note): 8 MIMO detection SDPs of n = 512 (PSD side 513,
    263 682 box rows each: Nx = 1.05 M, Q = 2.11 M) as one block-diagonal model on one GPU.  The oracle needs ~1 s per
    iteration here, so the checks are: the batched launches (grid.z = block) and the per-block stream path give the
    same bits over the whole solve; OPTIMAL; and the reference's own MIMO criterion on every block
    (test/moi_mimo.jl:71-75: every |X_ij| in (0.99, 1.01))."""
    nb, n = 8, 512
    pr = P.block_diag_problems([P.mimo(n, seed=s_) for s_ in range(nb)])
    a = Optimizer(block_batch=1, time_limit=120.0, tol_gap=1e-5, tol_feasibility=1e-5).optimize(pr, trace_capacity=2000)
    b = Optimizer(block_batch=0, time_limit=120.0, tol_gap=1e-5, tol_feasibility=1e-5).optimize(pr, trace_capacity=2000)
    assert a.status == b.status == 1 and a.iter == b.iter
    cols = [c for c in range(14) if c != 12]
    assert np.array_equal(a.trace[:, cols], b.trace[:, cols])
    assert a.stats["batched_block_steps"] > 0 and b.stats["batched_block_steps"] == 0
    assert a.stats["lanczos_matvecs"] == b.stats["lanczos_matvecs"]
    side = n + 1
    L = side * (side + 1) // 2
    for k in range(nb):
        X = P.unpack_psd(a.primal[k * L:(k + 1) * L], side)
        assert 0.99 < np.abs(X).min() and np.abs(X).max() < 1.01, k


def test_support_and_dense_paths_agree_on_maxcut():
    """Max-Cut n=300, 150 iterations: the support-aware passes and the dense passes are the
    same arithmetic on the support and exact zeros elsewhere."""
    pr = P.maxcut(300, seed=2)
    sols = []
    for sp in (0, 1):
        opt = Optimizer(max_iter=150, support_path=sp)
        sols.append(opt.optimize(pr, trace_capacity=150))
    a, b = sols
    assert a.iter == b.iter == 150
    assert np.array_equal(a.trace[:, 11], b.trace[:, 11])                 # linesearch trials
    mv = a.trace[:, 13]
    tight = max(3, int(np.argmax(mv > 25)) if np.any(mv > 25) else 150)
    for col in (1, 2, 3, 4, 5, 6, 7, 9):
        assert np.allclose(a.trace[:tight, col], b.trace[:tight, col], rtol=1e-10, atol=1e-13), col
    assert abs(a.objval - b.objval) <= 1e-3 * (1 + abs(a.objval))


@pytest.mark.parametrize("case", ["maxcut", "two_blocks", "periodic_full_eig", "arpack_rule", "hub_rows"])
def test_operator_form_matvec_matches_packed_matvec(case):
    """lanczos_operator=1 (A v = Vp Lam Vp' v + E v from the previous projection's factors and the
    sparse support update) against lanczos_operator=0 (the packed triangle, what dsymv('U')
    reads) and against the oracle: same linesearch decisions, same mat-vec counts, iterates equal
    to rounding until the first Lanczos restart, same optimum.  'periodic_full_eig' interleaves
    full_eig! iterations (full_eig_freq/len), after which no factors exist and the packed
    mat-vec must take over for one iteration."""
    kw = dict(max_iter=160, support_path=1)
    if case == "maxcut":
        pr = P.maxcut(300, seed=2)
    elif case == "two_blocks":
        pr = P.block_diag_problems([P.maxcut(180, seed=1), P.maxcut(130, seed=4)])
    elif case == "hub_rows":                     # vertices of degree 150 and 90: rows wider than the ELL part
        L = P.erdos_renyi_laplacian(300, 6).tolil()
        W = -L
        W.setdiag(0)
        for hub, deg in ((7, 150), (200, 90)):
            for v in np.random.default_rng(hub).choice(300, size=deg, replace=False):
                if v != hub:
                    W[hub, v] = W[v, hub] = 1.0
        W = W.tocsr()
        import scipy.sparse as sp
        pr = P.maxcut_from_laplacian((sp.diags(np.asarray(W.sum(axis=1)).ravel()) - W).tocsr(), name="maxcut-hubs")
    elif case == "arpack_rule":                  # eigsolver=1: dsaupd's acceptance rule on the same engine
        pr = P.maxcut(260, seed=5)
        kw.update(eigsolver=1)
    else:
        pr = P.maxcut(220, seed=3)
        kw.update(full_eig_freq=12, full_eig_len=1)
    sols = [Optimizer(lanczos_operator=op, **kw).optimize(pr, trace_capacity=160) for op in (0, 1)]
    a, b = sols
    assert a.iter == b.iter == 160
    assert a.stats["fop_projections"] == 0 and b.stats["fop_projections"] > 0
    nblk = len(pr.psd)
    if case == "periodic_full_eig":
        assert 0 < b.stats["full_eigs"] and b.stats["fop_projections"] < b.stats["lanczos_calls"]
    elif case == "arpack_rule":                  # a non-converged dsaupd falls back to full_eig!: no factors next time
        assert b.stats["fop_projections"] >= b.stats["lanczos_calls"] - 2 * b.stats["krylov_fallbacks"] - 1 >= 140
    else:
        assert b.stats["fop_projections"] == b.stats["lanczos_calls"] == 160 * nblk
    assert np.array_equal(a.trace[:, 11], b.trace[:, 11])                 # linesearch trials
    mv = a.trace[:, 13]
    tight = max(3, int(np.argmax(mv > 25 * nblk)) if np.any(mv > 25 * nblk) else 160)
    assert np.array_equal(a.trace[:tight, 13], b.trace[:tight, 13])       # mat-vecs per iteration
    for col in (1, 2, 3, 4, 5, 6, 7, 9):
        assert np.allclose(a.trace[:tight, col], b.trace[:tight, col], rtol=1e-9, atol=1e-12), col
    assert abs(a.objval - b.objval) <= 1e-3 * (1 + abs(a.objval))
    if case == "maxcut":
        o = Options()
        o.max_iter = 160
        ref = oracle.solve(pr, o, trace=True)
        G = _trace_cols(ref.trace)
        assert np.allclose(b.trace[:tight, [1, 2, 3, 4, 7, 11]], G[:tight], rtol=1e-7, atol=1e-10)


@pytest.mark.parametrize("kw", [dict(equilibration_force=1, equilibration_reference_aliasing=0), dict(approx_norm=0),
                                dict(equilibration_force=1, approx_norm=0, equilibration_reference_aliasing=0),
                                dict(equilibration_force=1), dict(equilibration_force=1, approx_norm=0)])
@pytest.mark.parametrize("build", ["sdp_wiki_min", "lp_in_SDP_inequality_form", "maxcut30", "mimo6"])
def test_equilibration_and_spectral_norm_against_oracle(build, kw):
    """equilibrate! (host, once) + un-scaling at the exit (pdhg.jl:64-92,751-755) and the
    sigma_max step size (pdhg.jl:108-119: Arpack.svds in the reference, SciPy's ARPACK svds in
    the oracle, a host Lanczos on M'M in the library)."""
    pr = {"maxcut30": lambda: P.maxcut(30, seed=2), "mimo6": lambda: P.mimo(6, seed=1)}.get(build, None)
    pr = pr() if pr else KATS[build][0]()
    opt = Optimizer(tol_gap=1e-6, tol_feasibility=1e-6, **kw)
    sol = opt.optimize(pr, trace_capacity=400)
    o = Options()
    o.tol_gap = o.tol_feasibility = 1e-6
    for k, v in kw.items():
        o.set(k, bool(v))
    ref = oracle.solve(pr, o, trace=True)
    # (equilibration_reference_aliasing: 1 = default = equilibrate!'s Diagonal(u) aliasing restated exactly on both
    # sides, 0 = the intended iteration; oracle/pdhg.py:equilibrate, csrc/prep.hpp:equilibrate_host)
    assert sol.status == ref.status
    aliased = bool(kw.get("equilibration_force")) and kw.get("equilibration_reference_aliasing", 1) != 0
    if aliased and build == "mimo6":
        # WITH the aliasing the scaling iteration is not contractive: on this instance ulp-level arithmetic variants
        # (libm's exp vs NumPy's, summation order) move E by 4 % after 1000 iterations (measured on the oracle alone,
        # 1e-16 without the aliasing; maxcut30 is stable either way) -- the two restatements, like Julia's own run,
        # are each "a" result of the reference's code.  What must agree is what diagonal scaling cannot change:
        assert ref.status == 1 and abs(sol.objval - ref.objval) <= 1e-5 * (1 + abs(ref.objval))
        assert abs(sol.iter - ref.iter) <= 0.25 * ref.iter
        return
    assert abs(sol.iter - ref.iter) <= max(2, 0.02 * ref.iter)
    m = min(len(ref.trace), len(sol.trace), 40)
    G, T = _trace_cols(ref.trace)[:m], sol.trace[:m, [1, 2, 3, 4, 7, 11]]
    assert np.allclose(T, G, rtol=1e-6, atol=1e-9 * max(1.0, np.abs(G).max()))
    if ref.status != 1:
        return
    assert abs(sol.objval - ref.objval) <= 1e-6 * (1 + abs(ref.objval))
    sc = max(1.0, np.abs(ref.primal).max())
    assert np.allclose(sol.primal, ref.primal, atol=2e-5 * sc)
    assert np.allclose(sol.slack_eq, ref.slack_eq, atol=2e-5 * sc)
    assert np.allclose(sol.dual_eq, ref.dual_eq, atol=2e-4 * max(1.0, np.abs(ref.dual_eq).max(initial=0.0)))
    assert np.allclose(sol.dual_in, ref.dual_in, atol=2e-4 * max(1.0, np.abs(ref.dual_in).max(initial=0.0)))


def test_blocks_wider_than_64_workgroups():
    """n = 4200 -> 66 row groups: the per-workgroup partial arrays no longer fit one wave-wide
    load (pld = 128) and the `n > 4096` branches of the Lanczos kernels run.  Eigenpairs against
    LAPACK, and the three solver configurations (dense vector passes / support path with the
    packed mat-vec / support path with the operator form) against each other."""
    n = 4200
    x = planted_packed(n, 5, [90.0, 70.0, 55.0, 30.0], bulk=(-4.0, 1.0))
    vals, vecs, info = B.eigsolve(x, n, 3)
    X = smat(x, n)
    ref = np.sort(np.linalg.eigvalsh(X))[::-1]
    assert info["converged"] >= 3
    assert np.allclose(vals[:3], ref[:3], rtol=0, atol=1e-11 * ref[0])
    assert np.allclose(vecs.T @ vecs, np.eye(vecs.shape[1]), atol=1e-10)
    pr = P.maxcut(n, seed=1)
    runs = [Optimizer(max_iter=40, **kw).optimize(pr, trace_capacity=40)
            for kw in (dict(support_path=0), dict(support_path=1, lanczos_operator=0), dict(support_path=1, lanczos_operator=1))]
    a = runs[0]
    for b in runs[1:]:
        assert np.array_equal(a.trace[:, 11], b.trace[:, 11]) and np.array_equal(a.trace[:, 13], b.trace[:, 13])
        for col in (1, 2, 3, 4, 5, 6, 7, 9):
            assert np.allclose(a.trace[:, col], b.trace[:, col], rtol=1e-8, atol=1e-11), col
    assert runs[2].stats["fop_projections"] == 40 and runs[1].stats["fop_projections"] == 0


@pytest.mark.parametrize("n,seed,rank0,maxrank,iters", [
    (101, 1, 2, 16, 260),        # smallest Lanczos-sized block, 2 row groups, rank updates after iteration 200
    (333, 2, 30, 40, 60),        # krylovdim 61..81: 2 chunks of basis columns per wave
    (300, 3, 70, 78, 40),        # rp > 64: two chunks of previous factors (NCHP = 2), krylovdim 141 (NCH = 3)
    (1000, 4, 12, 16, 80),       # BASELINE config 2 size
])
def test_operator_form_across_ranks_and_sizes(n, seed, rank0, maxrank, iters):
    """The operator-form kernels are templated on the number of 16-column chunks of the Krylov
    basis (NCH) and of the previous factors (NCHP); this sweeps the combinations against the
    packed-triangle mat-vec on the same instances."""
    pr = P.maxcut(n, seed=seed)
    kw = dict(max_iter=iters, support_path=1, initial_target_rank=rank0, max_target_rank_krylov_eigs=maxrank)
    a = Optimizer(lanczos_operator=0, **kw).optimize(pr, trace_capacity=iters)
    b = Optimizer(lanczos_operator=1, **kw).optimize(pr, trace_capacity=iters)
    assert a.iter == b.iter == iters and b.stats["fop_projections"] > 0 and b.stats["full_eigs"] == a.stats["full_eigs"]
    assert np.array_equal(a.trace[:, 10], b.trace[:, 10])                 # target-rank schedule
    assert np.array_equal(a.trace[:, 11], b.trace[:, 11])                 # linesearch trials
    krylovdim = max(2 * rank0 + 1, 25)
    mv = a.trace[:, 13]
    tight = max(3, int(np.argmax(mv > krylovdim)) if np.any(mv > krylovdim) else iters)
    assert np.array_equal(a.trace[:tight, 13], b.trace[:tight, 13])
    for col in (1, 2, 3, 4, 5, 6, 7, 9):
        assert np.allclose(a.trace[:tight, col], b.trace[:tight, col], rtol=1e-8, atol=1e-11), col
    sc = np.abs(a.trace[:, 1]).max()
    assert abs(a.trace[-1, 1] - b.trace[-1, 1]) <= 2e-2 * sc              # same neighbourhood after the separation
    assert np.all(np.isfinite(b.trace))


def test_operator_form_converges_to_the_same_optimum():
    """Full solves (tol 1e-4) of a Max-Cut instance with both operators: same status, objectives
    within the solver's own gap measure, iterate feasible by the solver's own criterion
    (residuals.jl:2-35) and PSD."""
    pr = P.maxcut(400, seed=7)
    kw = dict(tol_gap=1e-4, tol_feasibility=1e-4, support_path=1)
    a = Optimizer(lanczos_operator=0, **kw).optimize(pr)
    b = Optimizer(lanczos_operator=1, **kw).optimize(pr)
    assert a.status == b.status == 1
    assert abs(a.objval - b.objval) <= 2e-4 * (1 + abs(a.objval))
    assert abs(a.iter - b.iter) <= 0.25 * a.iter
    assert b.stats["fop_projections"] == b.stats["lanczos_calls"] and b.stats["full_eigs"] == 0
    X = P.unpack_psd(b.primal, 400)
    assert np.abs(np.diag(X) - 1).max() <= 1e-4 * (1 + np.sqrt(400.0)) and np.linalg.eigvalsh(X).min() >= -1e-6


@pytest.mark.parametrize("fname,lit", [("mcp250-1", 317.2643), ("mcp500-1", 598.1485)])
def test_sdplib_literature_optima(fname, lit, golden_dir):
    """SDPLIB Max-Cut instances the oracle is too slow for, solved to tol 1e-4 on the default
    (operator-form) path: OPTIMAL, objective within the feasibility-induced slack of the
    literature optimum, iterate PSD with unit diagonal to tolerance.  (tools/sdplib_sweep.py runs
    the whole family incl. maxG11 / maxG32 / maxG55.)"""
    pr = P.sdplib(golden_dir / "sdplib" / f"{fname}.dat-s")
    opt = Optimizer(tol_gap=1e-4, tol_feasibility=1e-4, max_target_rank_krylov_eigs=64, time_limit=120.0)
    sol = opt.optimize(pr)
    n = pr.psd_sides()[0]
    assert sol.status == 1 and sol.stats["fop_projections"] == sol.stats["lanczos_calls"] > 0
    # the stop rule bounds gap and diag(X) = 1 to tol (1 + |b|), not dual feasibility: the oracle itself (reference-
    # faithful linesearch) stops mcp250-1 after 3074 iterations at 316.537, 2.3e-3 below the literature optimum
    assert abs(abs(sol.objval) - lit) <= 3.5e-3 * lit
    X = P.unpack_psd(sol.primal, n)
    assert np.abs(np.diag(X) - 1).max() <= 1e-4 * (1 + np.sqrt(n)) and np.linalg.eigvalsh(X).min() >= -1e-6


@pytest.mark.parametrize("fname,lit,budget", [("truss1", 8.999996, 0), ("control1", -17.784627, 3000), ("theta1", -23.0, 3000)])
def test_sdplib_multiblock_instances_with_batched_small_blocks(fname, lit, budget, golden_dir):
    """SDPLIB instances with their block structure KEPT (truss1: six 2x2 blocks + a 1x1;
    control1: 10x10 + 5x5; theta1: one 50x50) -- every block is below min_size_krylov_eigs, i.e.
    full_eig! per block per iteration.  The library projects all blocks of side 2..64 in ONE batched
    Jacobi launch; oracle = LAPACK dsyevr per block.  truss1 is solved to tol 1e-4 and checked against
    the literature optimum (SDPLIB README; the model minimises -F0.X).  control1 / theta1 need ~1e6 PDHG
    iterations (oracle: theta1 937 006 iterations to -23.0047, control1 hits the 1e6 limit), so they run
    on a fixed budget of 3000 iterations: same trace as the oracle.  The batched path and the per-block
    rocSOLVER path agree."""
    pr = P.sdplib_blocks(golden_dir / "sdplib" / f"{fname}.dat-s")
    o = Options()
    if budget:
        o.max_iter = budget
    ref = oracle.solve(pr, o, trace=True)
    G = _trace_cols(ref.trace)
    sols = {}
    for sbb in (-1, 0):
        kw = dict(max_iter=budget) if budget else {}
        opt = Optimizer(small_block_batch=sbb, **kw)
        sol = opt.optimize(pr, trace_capacity=len(G))
        sols[sbb] = (opt, sol)
        print(fname, "small_block_batch", sbb, "status", sol.status, "iter", sol.iter, "obj", opt.objective_value(),
              "batched", sol.stats["batched_small_eigs"], "| oracle", ref.status, ref.iter, ref.objval, "| lit", lit)
        assert sol.status == ref.status == (3 if budget else 1)
        assert abs(sol.iter - ref.iter) <= max(2, 0.02 * ref.iter)
        m = min(len(G), len(sol.trace), 400)
        T = sol.trace[:m, [1, 2, 3, 4, 7, 11]]
        assert np.allclose(T, G[:m], rtol=1e-6, atol=1e-9 * max(1.0, np.abs(G[:m]).max()))
        assert abs(opt.objective_value() - ref.objval) <= 1e-5 * (1 + abs(ref.objval))
        if not budget:
            assert abs(opt.objective_value() - lit) <= 2e-3 * (1 + abs(lit))
        assert sol.final_rank == ref.final_rank
    nsmall = sum(1 for s in pr.psd_sides() if 2 <= s <= 64)
    if nsmall >= 2:
        assert sols[-1][1].stats["batched_small_eigs"] == nsmall * sols[-1][1].iter
    assert sols[0][1].stats["batched_small_eigs"] == 0


@pytest.mark.parametrize("n", [2, 3, 9, 16, 17, 22, 31, 33, 48, 50, 64])
def test_small_block_sign_projection_against_lapack(n):
    """k_small_sign_project (csrc/small_sign.hip.hpp: the whole sign-function projection of a block of side <= 64 in ONE
    launch, one workgroup, iterates in LDS, fp64 MFMA products) through psd_project mode 5: == LAPACK's projection to
    1e-10 of the spectral scale (the floor of the iteration, as for the tiled sign projection), the count of positive
    eigenvalues exact; indefinite, definite (both signs), low-rank and zero inputs; spectra spread over eight decades."""
    rng = np.random.default_rng(100 + n)
    cases = []
    M = rng.standard_normal((n, n)); cases.append((M + M.T) / 2)
    cases.append(cases[0] @ cases[0].T)                                   # positive definite: X+ = X
    cases.append(-(cases[0] @ cases[0].T))                                # negative definite: X+ = 0
    U, _ = np.linalg.qr(rng.standard_normal((n, n)))
    lam = np.concatenate([10.0 ** rng.uniform(-6, 2, n // 2), -10.0 ** rng.uniform(-6, 2, n - n // 2)])
    cases.append((U * lam) @ U.T)                                         # eight decades, both signs
    k = max(1, n // 3)
    cases.append((U[:, :k] * np.linspace(1.0, 2.0, k)) @ U[:, :k].T - (U[:, k:2 * k] * 0.5) @ U[:, k:2 * k].T)   # low rank
    cases.append(np.zeros((n, n)))
    for X in cases:
        X = (X + X.T) / 2
        w, Q = np.linalg.eigh(X)
        ref = (Q * np.maximum(w, 0.0)) @ Q.T
        scale = max(np.abs(w).max(), 1e-300)
        resolved = np.abs(w) > 1e-9 * scale
        for start_row in (0, 8):                  # the full table | the shortened, tested schedule the solver uses (with its in-kernel fall-back)
            out, info = B.psd_project(svec(X), n, start_row + 1, mode=5)
            assert np.abs(out - svec(ref)).max() <= 1e-10 * scale + 1e-300, (n, start_row, np.abs(out - svec(ref)).max() / scale)
            assert info["rank"] >= int((w[resolved] > 0).sum()) and info["rank"] <= int((w > 0).sum()) + int((~resolved).sum())


def test_small_models_take_the_one_launch_projections_and_follow_the_oracle():
    """Models whose PSD blocks never take the Krylov path (side <= min_size_krylov_eigs): one launch per iteration for
    all of them -- Jacobi at side 2, the LDS-resident sign projection for sides 3 .. 64 (auto) -- against the oracle's
    LAPACK full_eig!: same iteration counts and traces on a 22 x 22 sensor-localisation-shaped block (MIMO 21),
    Max-Cut 60 and a mixed model; and the same solves with the batch switched off (rocSOLVER / tiled sign)."""
    import oracle
    for pr, iters in ((P.mimo(21, seed=3), 150), (P.maxcut(60, seed=2), 150), (P.maxcut(33, seed=1), 150)):
        o = oracle.Options(); o.max_iter = iters
        ref = oracle.solve(pr, o, trace=True)
        exp = np.array([t["prim_obj"] for t in ref.trace])
        a = Optimizer(max_iter=iters).optimize(pr, trace_capacity=iters)
        b = Optimizer(max_iter=iters, small_block_batch=0).optimize(pr, trace_capacity=iters)
        assert a.stats["batched_small_eigs"] == a.iter and a.stats["full_eigs_sign"] == a.iter and b.stats["batched_small_eigs"] == 0
        for sol in (a, b):
            assert sol.iter == ref.iter and sol.status == ref.status
            got = np.array([r[1] for r in sol.trace])
            # (the sign projections resolve X+ to 1e-10 of its spectral scale: the objective <c, x> sees that times |c|)
            atol = 1e-9 * (1.0 + float(np.linalg.norm(pr.c)) * np.sqrt(pr.n))
            assert np.allclose(got, exp, rtol=1e-7, atol=atol), (float(np.abs(got - exp).max()), atol)


@pytest.mark.parametrize("name,build", [("randsdp-5x5", lambda: P.randsdp(5, 5, seed=1)), ("mimo-60", lambda: P.mimo(60, seed=0)),
                                        ("mimo-100", lambda: P.mimo(100, seed=0))])
def test_reference_benchmark_small_instances_take_the_oracles_iterations(name, build):
    """The small ends of the reference's benchmark sets (test/runbench.jl: RANDSDP 5 x 5, MIMO 100; MIMO 60 for the one-launch
    small-block projection) with reference default options: the library stops OPTIMAL after the oracle's iteration count
    (1216 / 45 / 52) at the oracle's objective."""
    import oracle
    pr = build()
    ref = oracle.solve(pr, oracle.Options())
    sol = Optimizer().optimize(pr)
    print(name, "gpu", sol.status, sol.iter, sol.objval, "oracle", ref.status, ref.iter, ref.objval)
    assert sol.status == ref.status == 1 and sol.iter == ref.iter
    # (MIMO's optimal value is a sum of terms of size |c| |x| ~ 1e6 that cancel to ~1e-5; the sign projections resolve X+ to 1e-10 of
    # its scale -- measured here: 1.5e-5 absolute, with the tiled and with the one-launch kernel alike; Jacobi: 8e-12)
    scale = float(np.linalg.norm(pr.c) * np.linalg.norm(sol.primal))
    assert abs(sol.objval - ref.objval) <= 1e-6 * (1 + abs(ref.objval)) + 1e-10 * scale


@pytest.mark.parametrize("n", [50, 100])
def test_sensorloc_benchmark_family_takes_the_oracles_iterations(n):
    """The SENSORLOC set of the reference's benchmark (test/runbench.jl:103-108, test/jump_sensorloc.jl; problems.sensorloc):
    a feasibility SDP on one (n + 2) block.  n = 50 (side 52: the one-launch sign projection of small blocks) and n = 100
    (side 102: the Krylov path, ~70 mat-vecs per iteration) with reference default options against the oracle's solve:
    OPTIMAL after the SAME number of iterations, zero objective, the sensors' positions recovered."""
    import oracle
    pr = P.sensorloc(n, seed=0)
    ref = oracle.solve(pr, oracle.Options())
    sol = Optimizer().optimize(pr)
    print("sensorloc", n, "gpu", sol.status, sol.iter, sol.time, "oracle", ref.status, ref.iter)
    assert sol.status == ref.status == 1 and sol.iter == ref.iter
    assert abs(sol.objval) <= 1e-10 and sol.primal_feasible_user_tol
    X = P.unpack_psd(sol.primal, n + 2)
    assert np.abs(X[:2, 2:] - pr.x_true).max() <= 1e-3 and np.linalg.eigvalsh(X).min() >= -1e-6
    assert np.allclose(sol.primal, ref.primal, atol=1e-6)


def test_block_cycle_kernel_in_a_multi_block_model_with_one_stream_per_block():
    """Blocks of different sides are projected concurrently, one worker thread and stream per block (run_blocks): each block's
    one-workgroup cycle kernel, its deferred rotation and its pinned staging buffers are its own.  Two Max-Cut blocks (sides 120 and
    150, operator form) and a third whose objective is dense on its block (side 90: the packed operator, resident in LDS): the
    step kernels and the one-workgroup kernel give the same bits."""
    a_, b_ = P.maxcut(120, seed=1), P.maxcut(150, seed=2)
    c_ = P.maxcut(90, seed=3)
    rng = np.random.default_rng(7)
    c_.c[:] = c_.c + 0.05 * rng.standard_normal(c_.n)               # dense objective: this block's update is not sparse
    pr = P.block_diag_problems([a_, b_, c_], name="three-blocks")
    kw = dict(max_iter=300, min_size_krylov_eigs=50)
    r0 = Optimizer(lanczos_cycle_kernel=0, **kw).optimize(pr, trace_capacity=300)
    r2 = Optimizer(lanczos_cycle_kernel=2, **kw).optimize(pr, trace_capacity=300)
    assert r0.stats["cycle_launches"] == 0 and r2.stats["cycle_launches"] > 600
    cols = [c for c in range(r0.trace.shape[1]) if c != 12]
    assert r0.iter == r2.iter and np.array_equal(r0.trace[:, cols], r2.trace[:, cols])
    assert np.array_equal(r0.primal, r2.primal) and np.array_equal(r0.dual_eq, r2.dual_eq)
    assert r0.stats["lanczos_matvecs"] == r2.stats["lanczos_matvecs"] and r0.stats["lanczos_restarts"] == r2.stats["lanczos_restarts"]


def _long_column_model(n=120, copies=6, seed=3):
    """Max-Cut n plus `copies` x (n - 1) redundant equality rows X_00 + X_jj = 2 (consistent with diag = 1) and as many inequality
    rows X_00 - X_jj <= 0.5: the column of X_00 carries 2 * copies * (n - 1) + 1 entries (1429 at the defaults), every other diagonal
    column 2 * copies + 1 -- the shape that made one thread of the transposed products walk thousands of entries (sensor
    localisation's identity block, profiles/r06_medium_blocks.md)."""
    pr = P.maxcut(n, seed=seed)
    diag = np.array([j * (j + 1) // 2 + j for j in range(n)])
    rows, cols, vals = [], [], []
    r = 0
    for _ in range(copies):
        for j in range(1, n):
            rows += [r, r]; cols += [diag[0], diag[j]]; vals += [1.0, 1.0]; r += 1
    A2 = sp.csr_matrix((vals, (rows, cols)), shape=(r, pr.n))
    G2 = sp.csr_matrix((np.array(vals) * np.tile([1.0, -1.0], r), (rows, cols)), shape=(r, pr.n))
    return P.Problem(n=pr.n, A=sp.vstack([pr.A, A2]).tocsc(), b=np.concatenate([pr.b, np.full(r, 2.0)]),
                     G=sp.vstack([pr.G, G2]).tocsc() if pr.G.shape[0] else G2.tocsc(), h=np.concatenate([pr.h, np.full(r, 0.5)]),
                     c=pr.c, psd=pr.psd, name="maxcut-long-column")


@pytest.mark.parametrize("support", [0, 1])
def test_long_columns_of_the_constraint_matrix_follow_the_oracle(support):
    """Round 6: columns of M beyond 192 entries are summed by a wave -- products in parallel, additions in the scalar loop's order
    (kernels.hip.hpp `wave_col_dot_inorder`) -- in the transposed products of both vector paths (`k_spmv_csc_norm_batch`: dense
    passes, `k_spmvT_S_batch`: support path).  A model with one column of 1429 entries against the oracle: identical linesearch
    trials and mat-vec counts, traces to 1e-9; and the two paths against each other."""
    import oracle
    pr = _long_column_model()
    assert np.diff(pr.A.indptr).max() + np.diff(pr.G.indptr).max() > 1400
    o = oracle.Options(); o.max_iter = 80
    mv, prev = [], [0]
    ref = oracle.solve(pr, o, trace=True, proj_callback=lambda it, xin, xout, p, arc: (mv.append(arc[0].matvecs - prev[0]), prev.__setitem__(0, arc[0].matvecs)))
    sol = Optimizer(max_iter=80, support_path=support).optimize(pr, trace_capacity=80)
    exp = np.array([[t["prim_obj"], t["dual_obj"], t["gap"], t["feas"], t["primal_step"], t["trials"]] for t in ref.trace])
    m = min(len(exp), len(sol.trace))
    assert m >= 60
    assert np.array_equal(sol.trace[:m, 11], exp[:m, 5])
    assert np.array_equal(sol.trace[:m, 13], np.array(mv[:m], float))
    scale = 1.0 + np.abs(exp[:m, :5]).max(axis=0)
    assert (np.abs(sol.trace[:m][:, [1, 2, 3, 4, 7]] - exp[:m, :5]) / scale).max() <= 1e-9


@pytest.mark.parametrize("n", [150, 200, 300, 400])
def test_sensorloc_larger_sizes_take_the_committed_oracle_counts(n, golden_dir):
    """SENSORLOC at sizes where the oracle takes minutes to an hour (tests/golden/sensorloc_oracle.json, made by
    tests/golden/make_golden_sensorloc.py): OPTIMAL after the oracle's iteration count (with its Lanczos mat-vec total where recorded) -- except n = 150 and 400, where the two
    trajectories part (recorded in the golden file): there status and solution are compared."""
    gold = json.loads((golden_dir / "sensorloc_oracle.json").read_text())
    if str(n) not in gold:
        pytest.skip("no committed oracle solve at this size")
    g = gold[str(n)]
    pr = P.sensorloc(n, seed=0)
    sol = Optimizer().optimize(pr)
    print("sensorloc", n, sol.status, sol.iter, sol.stats["lanczos_matvecs"], "oracle", g)
    assert sol.status == g["status"] == 1
    X = P.unpack_psd(sol.primal, n + 2)
    # positions at the default tolerances (tol_gap = tol_feasibility = 1e-4): 1.4e-3 at n = 150, below 1e-3 elsewhere
    assert np.abs(X[:2, 2:] - pr.x_true).max() <= 5e-3 and abs(sol.objval) <= 1e-10
    if g.get("exact", True):
        assert sol.iter == g["iterations"]
        if "lanczos_matvecs" in g:
            assert sol.stats["lanczos_matvecs"] == g["lanczos_matvecs"]
    else:
        # n = 150, 400: the two sides agree to 1e-10 for ~155 iterations (n = 150) and then part like any two roundings of this iteration do
        # (degenerate truncations): same status, same solution, iteration counts of the same order
        assert 0.5 * g["iterations"] <= sol.iter <= 2.0 * g["iterations"]


@pytest.mark.parametrize("n", [2, 3, 7, 16, 31, 50, 64])
def test_batched_small_block_projection_against_lapack(n):
    """k_small_psd_project (parallel-order cyclic Jacobi in LDS) on many random blocks of one size in a
    single model: result == LAPACK projection to 1e-13, through a 1-iteration solve seam is not
    available, so this goes through psd_project mode 3 (batched kernel on one block)."""
    rng = np.random.default_rng(n)
    for trial in range(3):
        M = rng.standard_normal((n, n)); X = (M + M.T) / 2
        if trial == 2:
            X = X @ X.T * (1 if n % 2 else -1)            # definite cases
        x = svec(X)
        out, info = B.psd_project(x, n, 1, mode=3)
        w, Q = np.linalg.eigh(X)
        ref = (Q * np.maximum(w, 0.0)) @ Q.T
        assert np.allclose(out, svec(ref), rtol=0, atol=1e-13 * max(1.0, np.abs(w).max()))
        assert info["rank"] == int((w > 1e-7).sum())


@pytest.mark.parametrize("n,maxrank,rank0", [(700, 16, 2), (1500, 40, 30)])
def test_persistent_cycle_kernel_matches_the_step_kernels(n, maxrank, rank0):
    """lanczos_cycle_kernel = 1: one persistent launch per Lanczos cycle (workgroups of one XCD keep their
    rows of the basis in LDS, partial dots exchanged through that XCD's L2).  Same arithmetic per step as
    the two step kernels, only the grouping of partial sums differs: same mat-vec / restart counts, traces
    to 1e-9.  (If the placement assumption failed, the bounded spins would time out and the library would
    fall back to the step kernels: cycle_launches would still be counted but the test's equality holds
    either way; the second assertion checks that cycles really ran.)"""
    pr = P.maxcut(n, seed=5)
    kw = dict(max_iter=160, support_path=1, max_target_rank_krylov_eigs=maxrank, initial_target_rank=rank0)
    a = Optimizer(lanczos_cycle_kernel=0, **kw).optimize(pr, trace_capacity=160)
    b = Optimizer(lanczos_cycle_kernel=1, **kw).optimize(pr, trace_capacity=160)
    assert a.stats["cycle_launches"] == 0 and b.stats["cycle_launches"] > 100
    assert b.stats["cycle_steps"] >= 0.9 * b.stats["lanczos_matvecs"]
    assert a.iter == b.iter and a.status == b.status
    assert np.array_equal(a.trace[:, 13], b.trace[:, 13])                         # mat-vecs per iteration
    assert np.array_equal(a.trace[:, 11], b.trace[:, 11])                         # linesearch trials
    assert a.stats["lanczos_restarts"] == b.stats["lanczos_restarts"]
    for col in (1, 2, 7):
        assert np.allclose(a.trace[:, col], b.trace[:, col], rtol=1e-9, atol=1e-12), col


@pytest.mark.parametrize("name", ["maxcut150", "maxcut300", "maxcut500", "sensorloc100", "sensorloc200", "mcp124-1", "mcp250-1", "gpp124-2"])
def test_block_cycle_kernel_reproduces_the_step_kernels_bit_for_bit(name, golden_dir):
    """Round 6: blocks of side <= 512 run a whole Lanczos cycle in ONE launch of ONE workgroup (lanczos_block1.hip.hpp;
    lanczos_cycle_kernel = 2, and auto where it applies) -- the step kernels' arithmetic term by term, with LDS and registers
    in place of the global records and a workgroup barrier in place of a kernel boundary.  Both operators (packed triangle:
    sensor localisation / gpp, operator form: Max-Cut / mcp), one and two row groups per virtual workgroup, with thick
    restarts: the traces are EQUAL, bit for bit, and so are the mat-vec and restart counts."""
    pr = {"maxcut150": lambda: P.maxcut(150, seed=2), "maxcut300": lambda: P.maxcut(300, seed=3), "maxcut500": lambda: P.maxcut(500, seed=4),
          "sensorloc100": lambda: P.sensorloc(100, seed=0), "sensorloc200": lambda: P.sensorloc(200, seed=0),
          "mcp124-1": lambda: P.sdplib(golden_dir / "sdplib" / "mcp124-1.dat-s"), "mcp250-1": lambda: P.sdplib(golden_dir / "sdplib" / "mcp250-1.dat-s"),
          "gpp124-2": lambda: P.sdplib(golden_dir / "sdplib" / "gpp124-2.dat-s")}[name]()
    kw = dict(max_iter=400)
    a = Optimizer(lanczos_cycle_kernel=0, **kw).optimize(pr, trace_capacity=400)
    b = Optimizer(lanczos_cycle_kernel=2, **kw).optimize(pr, trace_capacity=400)
    c = Optimizer(**kw).optimize(pr, trace_capacity=400)                           # auto
    side = pr.psd_sides()[0]
    packed_fits = side <= 140                                                      # (the packed triangle must be resident in LDS)
    fop = name.startswith("maxcut") or name.startswith("mcp")
    assert a.stats["cycle_launches"] == 0
    if fop or packed_fits:
        assert b.stats["cycle_launches"] > 300 and b.stats["cycle_steps"] >= b.stats["symv_launches"] >= b.stats["lanczos_matvecs"] > 0
        assert c.stats["cycle_launches"] == (b.stats["cycle_launches"] if side <= 256 else 0)     # auto: one row group per virtual workgroup
    else:
        assert b.stats["cycle_launches"] == 0 == c.stats["cycle_launches"]
    assert a.iter == b.iter == c.iter and a.status == b.status == c.status
    assert a.stats["lanczos_matvecs"] == b.stats["lanczos_matvecs"] and a.stats["lanczos_restarts"] == b.stats["lanczos_restarts"] > 0
    cols = [c_ for c_ in range(a.trace.shape[1]) if c_ != 12]                         # (column 12 is the wall clock)
    assert np.array_equal(a.trace[:, cols], b.trace[:, cols]) and np.array_equal(a.trace[:, cols], c.trace[:, cols]), np.abs(a.trace - b.trace).max(axis=0)
    assert np.array_equal(a.primal, b.primal) and np.array_equal(a.dual_eq, b.dual_eq)


def test_warm_start_knob_reaches_the_same_optimum_with_fewer_restarts():
    """lanczos_warm_start = 1 (library-only): start vector = normalised sum of the previous projection's Ritz
    vectors + 1e-3 x the fixed vector.  Eigenpairs are still converged to krylovkit_tol, so the iterates
    agree until rounding-level differences grow; the solve converges to the same objective (1e-4, the
    north-star criterion) with fewer Lanczos restarts."""
    pr = P.maxcut(600, seed=4)
    a = Optimizer(support_path=1)
    sa = a.optimize(pr, trace_capacity=400)
    b = Optimizer(support_path=1, lanczos_warm_start=1)
    sb = b.optimize(pr, trace_capacity=400)
    print("default", sa.iter, a.objective_value(), sa.stats["lanczos_restarts"], sa.stats["lanczos_matvecs"],
          "| warm", sb.iter, b.objective_value(), sb.stats["lanczos_restarts"], sb.stats["lanczos_matvecs"], sb.stats["warm_starts"])
    assert sa.status == sb.status == 1
    assert abs(a.objective_value() - b.objective_value()) <= 1e-4 * (1 + abs(a.objective_value()))
    assert sb.stats["warm_starts"] > 0 and sa.stats["warm_starts"] == 0
    assert sb.stats["lanczos_matvecs"] < sa.stats["lanczos_matvecs"]
    assert np.allclose(sa.trace[:40, 1], sb.trace[:40, 1], rtol=1e-6, atol=1e-9)


def _trace_cols(ref_trace):
    return np.array([[t["prim_obj"], t["dual_obj"], t["gap"], t["feas"], t["primal_step"], t["trials"]]
                     for t in ref_trace])


def test_randsdp_config_against_oracle():
    """BASELINE config 'randSDP' (test/base_randsdp.jl + moi_randsdp.jl) at a CPU-comparable
    size: dense equality rows, variable bounds, n=60 (< 100 -> full_eig! every iteration)."""
    pr = P.randsdp(60, 40, seed=3)
    opt = Optimizer(max_iter=400)
    sol = opt.optimize(pr, trace_capacity=400)
    o = Options()
    o.max_iter = 400
    ref = oracle.solve(pr, o, trace=True)
    assert sol.status == ref.status and sol.iter == ref.iter
    G, T = _trace_cols(ref.trace), sol.trace[:, [1, 2, 3, 4, 7, 11]]
    assert np.array_equal(T[:, 5], G[:, 5])
    assert np.allclose(T[:60], G[:60], rtol=1e-7, atol=1e-10)
    assert np.allclose(T[:, :2], G[:, :2], rtol=1e-4, atol=1e-6 * np.abs(G[:, :2]).max())
    assert sol.stats["full_eigs"] == sol.iter


def test_randsdp_dense_matrix_entry_matches_sparse_and_oracle():
    """proxsdp_problem.M_dense (host pointer): the dense A streamed as a row-major array must
    reproduce the CSC path (same data, summation order differs) and the oracle's trace."""
    pr_s = P.randsdp(60, 40, seed=3)
    pr_d = P.randsdp(60, 40, seed=3, dense=True)
    assert pr_d.A.nnz == 0 and pr_d.M_dense.shape == (40, 1830)
    s_s = Optimizer(max_iter=300).optimize(pr_s, trace_capacity=300)
    s_d = Optimizer(max_iter=300).optimize(pr_d, trace_capacity=300)
    assert s_d.status == s_s.status and s_d.iter == s_s.iter
    cols = [1, 2, 3, 4, 5, 6, 7, 11]
    assert np.array_equal(s_d.trace[:, 11], s_s.trace[:, 11])
    assert np.allclose(s_d.trace[:60, cols], s_s.trace[:60, cols], rtol=1e-7, atol=1e-10)
    assert np.allclose(s_d.trace[:, 1:3], s_s.trace[:, 1:3], rtol=1e-4, atol=1e-6 * np.abs(s_s.trace[:, 1:3]).max())
    assert np.allclose(s_d.primal, s_s.primal, rtol=1e-4, atol=1e-6 * np.abs(s_s.primal).max())
    assert np.allclose(s_d.slack_eq, s_s.slack_eq, atol=1e-6 * (1 + np.abs(pr_s.b).max()))
    assert np.allclose(s_d.dual_cone, s_s.dual_cone, rtol=1e-4, atol=1e-6 * np.abs(s_s.dual_cone).max())
    assert s_d.stats["dense_passes"] >= 2 * s_d.iter            # one A x and >= one batched A'y per iteration
    o = Options()
    o.max_iter = 300
    ref = oracle.solve(pr_s, o, trace=True)
    G = _trace_cols(ref.trace)
    assert np.allclose(s_d.trace[:60, [1, 2, 3, 4, 7, 11]], G[:60], rtol=1e-7, atol=1e-10)


def test_randsdp_config3_cpu_comparable_size_matches_oracle_trace(golden_dir):
    """BASELINE config 3 at the size SURVEY.md section 8 calls CPU-comparable: randSDP n = 500, m = 1000 (dense A:
    1000 x 125 250 doubles = 1 GB), seed 0, reference default options.  The oracle's first 200 iterations
    (tests/golden/make_golden_randsdp500.py) against the library on its dense-A path (M_dense, streamed twice per
    iteration) and on the CSC path fed the same numbers: identical linesearch trials and Lanczos mat-vec counts,
    traces to 1e-7 over the first 60 iterations and to 1e-4 over all 200."""
    gold = json.loads((golden_dir / "trace_randsdp_n500_m1000.json").read_text())
    G = np.array(gold["rows"]); gm = np.array(gold["matvecs"])
    iters = len(G)
    for dense in (True, False):
        pr = P.randsdp(gold["n"], gold["m"], seed=gold["seed"], dense=dense)
        sol = Optimizer(max_iter=iters).optimize(pr, trace_capacity=iters)
        T = sol.trace[:, :12]
        assert sol.iter == iters
        assert np.array_equal(T[:, 11], G[:, 11]), "linesearch trials"
        assert np.array_equal(T[:, 10], G[:, 10]), "target rank"
        same = sol.trace[:, 13] == gm
        print("dense" if dense else "csc", "mat-vec counts equal in", int(same.sum()), "of", iters)
        assert same.all(), np.nonzero(~same)[0]            # 200 of 200 (measured on both paths)
        sc = np.abs(G[:, 1:8]).max(axis=0)
        assert np.allclose(T[:60, 1:8], G[:60, 1:8], rtol=1e-7, atol=1e-10 * sc)
        assert np.allclose(T[:, 1:8], G[:, 1:8], rtol=1e-4, atol=1e-6 * sc)
        if dense:
            assert sol.stats["dense_passes"] >= 2 * iters


def test_randsdp_dense_matrix_on_device_lanczos_size():
    """M_dense as a DEVICE pointer (generated on the GPU, as the 64 GB BASELINE size must be),
    n = 150 so the Lanczos path runs; checked against the CSC path fed the same numbers and
    against the solver's own optimality measures."""
    import torch
    pr_d = P.randsdp_device(150, 120, seed=5, device="cuda:0")
    M = pr_d.M_dense.cpu().numpy()
    import scipy.sparse as sp
    pr_s = P.Problem(n=pr_d.n, A=sp.csc_matrix(M), b=pr_d.b, G=pr_d.G, h=pr_d.h, c=pr_d.c, psd=pr_d.psd)
    kw = dict(max_iter=600)                  # (PDHG needs > 20 000 iterations on this family)
    s_d = Optimizer(**kw).optimize(pr_d, trace_capacity=600)
    s_s = Optimizer(**kw).optimize(pr_s, trace_capacity=600)
    assert s_d.status == s_s.status == 3 and s_d.iter == s_s.iter == 600
    assert np.array_equal(s_d.trace[:, 11], s_s.trace[:, 11])          # same linesearch decisions
    sc = np.abs(s_s.trace[:, 1:5]).max(axis=0)
    assert np.allclose(s_d.trace[:40, 1:5], s_s.trace[:40, 1:5], rtol=1e-6, atol=1e-9 * sc)
    assert np.allclose(s_d.trace[:, 1:3], s_s.trace[:, 1:3], rtol=1e-3, atol=1e-5 * sc[:2])
    assert np.allclose(s_d.primal, s_s.primal, rtol=1e-3, atol=1e-5 * np.abs(s_s.primal).max())
    assert np.allclose(s_d.slack_eq, s_s.slack_eq, rtol=1e-3, atol=1e-6 * (1 + np.abs(pr_d.b).max()))
    assert s_d.stats["lanczos_matvecs"] > 0
    assert torch.equal(pr_d.M_dense.cpu(), torch.from_numpy(M))        # borrowed matrix untouched


def test_randsdp_config3_at_its_full_size_dense_passes_against_torch():
    """BASELINE config 3 at its ACTUAL size (VERDICT r2 config note): randSDP n = 2000, m = 4000 -- the coefficient
    matrix is 4000 x 2 001 000 doubles = 64 GB, generated in HBM and borrowed by the library as a device pointer.
    No oracle can hold it, so the two dense passes are checked against PLAIN TORCH on the same tensor: the exit
    path's A x (slack_eq + b) against `M @ x`, and its c + A'y + G'y (dual_cone) against `M.T @ y`; plus: the
    borrowed matrix is bit-identical afterwards, the projection ran at target rank 50 on the Lanczos path, and
    every iteration streamed the matrix (dense_passes)."""
    import torch
    free, _ = torch.cuda.mem_get_info()
    if free < 90 * 2 ** 30:
        pytest.skip("needs 90 GB of free HBM")
    n, m = 2000, 4000
    pr = P.randsdp_device(n, m, seed=0, device="cuda:0")
    M = pr.M_dense
    chk0 = (float(M[::97].sum()), float(M[:, ::1013].abs().sum()))
    iters = 12
    sol = Optimizer(max_iter=iters, initial_target_rank=50, max_target_rank_krylov_eigs=50).optimize(pr, trace_capacity=iters)
    assert sol.iter == iters and sol.stats["dense_passes"] >= 2 * iters and sol.stats["lanczos_matvecs"] > 0
    assert np.all(np.isfinite(sol.trace[:, 1:7]))
    x = torch.from_numpy(sol.primal).to("cuda:0")
    ax = (M @ x).cpu().numpy()
    sc = max(1.0, np.abs(ax).max())
    assert np.abs((sol.slack_eq + pr.b) - ax).max() <= 1e-11 * sc * n              # A x, 2.0e6-term dot products
    jj = np.repeat(np.arange(n), np.arange(1, n + 1))
    ii = np.arange(pr.n) - jj * (jj + 1) // 2
    w = np.where(ii == jj, 1.0, 2.0)
    ye = torch.from_numpy(sol.dual_eq).to("cuda:0")
    aty = (M.T @ ye).cpu().numpy() + pr.G.T @ sol.dual_in
    lhs = sol.dual_cone * w - pr.c                                                   # = +-(A'y + G'y): the sign is the dual's convention
    err = min(np.abs(lhs - aty).max(), np.abs(lhs + aty).max())
    assert err <= 1e-11 * max(1.0, np.abs(aty).max()) * m
    chk1 = (float(M[::97].sum()), float(M[:, ::1013].abs().sum()))
    assert chk0 == chk1, "the borrowed 64 GB matrix was modified"
    del M, x, ye
    torch.cuda.empty_cache()


@pytest.mark.parametrize("fname,iters", [("maxG51", 40), ("gpp500-1", 40)])
def test_sdplib_full_eig_fallback_config(fname, iters, golden_dir):
    """BASELINE config 'SDPLIB maxG51 / gpp500-1, full-rank fallback eig path'
    (full_eig_decomp=true: rocSOLVER dsyevd + rank-r+ reconstruction every iteration);
    gpp500-1 also has the all-ones constraint row of 125 250 entries (long-row SpMV) and,
    with the reference reader's n = length(c) quirk, side 501."""
    pr = P.sdplib(golden_dir / "sdplib" / f"{fname}.dat-s")
    o = Options()
    o.max_iter = iters
    o.full_eig_decomp = True
    ref = oracle.solve(pr, o, trace=True)
    # an explicit full_eig_decomp = true never goes to the Lanczos engine, whatever full_eig_lanczos
    # says in auto mode (the Lanczos-served full_eig! is for the IMPLICIT regime only, see
    # test_implicit_full_eig_regime_served_by_lanczos).  full_eig_sign: 0 = rocSOLVER dsyevd + rank-r+
    # reconstruction, 1 / auto = the sign-function projection (fp64 MFMA products, no eigenpairs; sign_start_row = 0: 57);
    # both reproduce the oracle's LAPACK trace.
    for fel, sign, row in ((0, 0, -1), (-1, 0, -1), (0, 1, 0), (0, 1, -1), (-1, -1, -1)):
        opt = Optimizer(max_iter=iters, full_eig_decomp=1, full_eig_lanczos=fel, full_eig_sign=sign, sign_start_row=row)
        sol = opt.optimize(pr, trace_capacity=iters)
        assert sol.status == ref.status == 3 and sol.iter == ref.iter == iters
        G, T = _trace_cols(ref.trace), sol.trace[:, [1, 2, 3, 4, 7, 11]]
        assert np.array_equal(T[:, 5], G[:, 5])
        assert np.allclose(T, G, rtol=1e-6, atol=1e-8 * np.abs(G).max())
        assert sol.stats["full_eigs"] == iters
        assert sol.stats["lanczos_matvecs"] == 0 and sol.stats["full_eigs_lanczos"] == 0
        assert sol.stats["full_eigs_sign"] == (iters if sign != 0 else 0)
        # the full table: 57 products per call; the shortened schedule (default): 34 when its test passes
        # (start row 8), 64 when it fails and the skipped rows are run after all
        short = sol.stats["sign_short_pass"] + sol.stats["sign_short_fail"]
        if sign == 0: assert sol.stats["sign_products"] == 0 and short == 0
        elif row == 0: assert sol.stats["sign_products"] == 57 * iters and short == 0
        else:
            assert short == iters and sol.stats["sign_short_pass"] >= iters // 2
            assert 29 * iters <= sol.stats["sign_products"] <= 57 * iters
        assert sol.final_rank == ref.final_rank


def test_sign_function_projection_of_several_blocks_side_by_side():
    """Four PSD blocks of different sides (150, 97, 64, 40) in one model with full_eig_decomp = true: the blocks'
    sign-function projections run concurrently (one host worker thread and one HIP stream per block, each with its
    own five work matrices) and must reproduce the oracle's LAPACK trace; with the support-aware vector passes the
    final product also fills the blocks' residual partials."""
    model = P.block_diag_problems([P.maxcut(150, seed=1), P.maxcut(97, seed=2), P.maxcut(64, seed=3), P.maxcut(40, seed=4)],
                                  name="four-blocks")
    o = Options()
    o.max_iter = 60
    o.full_eig_decomp = True
    ref = oracle.solve(model, o, trace=True)
    G = _trace_cols(ref.trace)
    for kw in (dict(), dict(support_path=1)):
        sol = Optimizer(max_iter=60, full_eig_decomp=1, **kw).optimize(model, trace_capacity=60)
        assert sol.status == ref.status and sol.iter == ref.iter == 60
        T = sol.trace[:, [1, 2, 3, 4, 7, 11]]
        assert np.array_equal(T[:, 5], G[:, 5])
        assert np.allclose(T, G, rtol=1e-6, atol=1e-8 * np.abs(G).max())
        assert sol.stats["full_eigs_sign"] == 4 * 60
        assert sol.stats["sign_short_pass"] + sol.stats["sign_short_fail"] == 4 * 60      # shortened schedule, tested
        assert sol.stats["sign_short_pass"] >= 2 * 60 and sol.stats["sign_products"] < 57 * 4 * 60
        assert sol.final_rank == ref.final_rank


def test_sign_function_projection_on_64_tiles_large_block():
    """Blocks wider than 3072 run the products on the 64 x 64-tile kernel (k_sym_gemm<SG_PLAIN / SG_POLY>; smaller
    ones only use it for the final product): n = 3100 against LAPACK, forced through psd_project mode 4."""
    n = 3100
    rng = np.random.default_rng(7)
    for name in ("gauss", "lowrank_pos"):
        if name == "gauss":
            M = rng.standard_normal((n, n)); X = (M + M.T) / 2
        else:
            Z = rng.standard_normal((n, 40)); M = rng.standard_normal((n, n)) * 0.05
            X = Z @ Z.T - (M @ M.T)                                  # 40 large positive directions over a negative bulk
        w, V = np.linalg.eigh(X)
        ref = (V * np.maximum(w, 0.0)) @ V.T
        out, info = B.psd_project(svec(X), n, 1, mode=4)
        assert np.abs(out - svec(ref)).max() <= 1e-9 * np.abs(w).max(), name
        assert info["rank"] == int((w > 0).sum())


def test_sign_function_projection_beyond_side_4096_in_auto_mode():
    """the auto window of full_eig_sign ended at side 4096, so full_eig! of maxG55 / maxG60-sized
    blocks (5000 / 7000) fell to rocSOLVER's dsyevd (~0.2 % of the fp64 peak).  The window now ends at 16384 (with a
    free-memory check): n = 4600 through the full_eig! test entry with the solver's AUTOMATIC engine choice, against
    LAPACK; the product count says which engine ran."""
    n = 4600
    rng = np.random.default_rng(3)
    Z = rng.standard_normal((n, 50)); M = rng.standard_normal((n, n)) * 0.03
    X = Z @ Z.T - (M @ M.T)
    w, V = np.linalg.eigh(X)
    ref = (V * np.maximum(w, 0.0)) @ V.T
    out, ms, rank, products = B.full_eig_kernel(svec(X), n, sign=-1)
    print("n", n, "auto engine:", products, "products,", round(ms, 1), "ms per projection")
    assert products >= 30                                          # the MFMA sign iteration ran, not dsyevd
    assert np.abs(out - svec(ref)).max() <= 1e-9 * np.abs(w).max()
    assert rank == int((w > 0).sum())


def test_sign_engine_on_the_krylov_branch_reproduces_the_lanczos_solve(golden_dir):
    """psd_sign_engine = 1: when fewer than target_rank eigenvalues are positive, the truncated projection of
    prox_operators.jl:89-109 IS the exact one and min_eig <= 0, so the sign-function projection may replace the
    Lanczos engine where it is measured to be cheaper (mcp500-1 needs > 100 mat-vecs per projection late in the
    solve: 531 k mat-vecs in 11.9 s -> 154 k in 2.7 s, 523 projections served, 9 verification rounds, the same 5086
    iterations).  Same linesearch decisions, same optimum, iteration count within rounding; every stand-in is verified
    (#positive < target_rank) or redone by Lanczos, and the engine itself is checked against the Lanczos engine on
    first use and every 64th projection (test_sign_engine_steps_aside_where_lanczos_is_not_the_exact_projection)."""
    pr = P.sdplib(golden_dir / "sdplib" / "mcp500-1.dat-s")
    a = Optimizer(tol_gap=1e-4, tol_feasibility=1e-4).optimize(pr, trace_capacity=20000)
    b = Optimizer(tol_gap=1e-4, tol_feasibility=1e-4, psd_sign_engine=1).optimize(pr, trace_capacity=20000)
    # the two engines agree to rounding per projection; over thousands of iterations that can move the stopping
    # test by a few iterations (round 3: 5086 = 5086, objective to 1e-12; on mcp250-1 the engine never engages
    # since the linesearch-fidelity change: its Lanczos projections cost 27 mat-vecs)
    assert a.status == b.status == 1 and abs(a.iter - b.iter) <= 0.01 * a.iter
    assert a.stats["sign_engine_projections"] == 0 and b.stats["sign_engine_projections"] > 100
    assert b.stats["lanczos_matvecs"] < a.stats["lanczos_matvecs"]
    m = min(a.iter, b.iter, 2000)
    assert np.array_equal(a.trace[:m, 11], b.trace[:m, 11])              # linesearch trials per iteration
    sc = np.abs(a.trace[:m, 1:5]).max(axis=0)
    assert np.allclose(a.trace[:m, 1:5], b.trace[:m, 1:5], rtol=0, atol=1e-6 * sc)
    assert abs(a.objval - b.objval) <= 1e-5 * (1 + abs(a.objval))
    assert a.final_rank == b.final_rank


def test_sign_engine_steps_aside_where_lanczos_is_not_the_exact_projection():
    """Single-vector Lanczos returns ONE eigenvector per distinct eigenvalue.  MIMO's first iterates have repeated
    positive eigenvalues, so the reference's (KrylovKit) projection is not the exact projection there -- and parity
    means reproducing the reference.  With psd_sign_engine = 1 the first verification round (both engines on the
    same input) sees the difference and leaves the block to the Lanczos engine: identical solve."""
    pr = P.mimo(512, seed=0)
    a = Optimizer(tol_gap=1e-4, tol_feasibility=1e-4).optimize(pr, trace_capacity=2000)
    b = Optimizer(tol_gap=1e-4, tol_feasibility=1e-4, psd_sign_engine=1).optimize(pr, trace_capacity=2000)
    assert b.stats["sign_engine_checks"] == 1 and b.stats["sign_engine_mismatches"] == 1
    assert b.stats["sign_engine_projections"] == 0
    assert a.status == b.status and a.iter == b.iter
    assert np.array_equal(a.trace[:, 1:5], b.trace[:, 1:5])
    assert b.stats["lanczos_matvecs"] == a.stats["lanczos_matvecs"]


def _spectrum_cases(n, rng):
    lam = {}
    lam["gauss"] = rng.standard_normal(n) * 3
    v = -np.abs(rng.standard_normal(n)); k = max(2, n // 6); v[:k] = np.abs(rng.standard_normal(k)) * 20 + 1
    lam["lowrank_pos"] = v
    v = rng.standard_normal(n); v[: n // 4] = 0.0
    lam["zeros"] = v
    v = rng.standard_normal(n); v[:5] = 36.83154802; v[5:10] = -2.5; v[10:14] = [1e-12, -1e-12, 3e-9, -3e-9]
    lam["degenerate_tiny"] = v
    lam["negdef"] = -np.abs(rng.standard_normal(n)) - 0.1
    lam["posdef"] = np.abs(rng.standard_normal(n)) + 0.1
    lam["null"] = np.zeros(n)
    return lam


@pytest.mark.parametrize("n", [33, 64, 100, 257, 501, 1000, 1001, 1430])
def test_sign_function_projection_against_lapack(n):
    """full_eig! by the matrix sign function (sign_project.hip.hpp; psd_project mode 4): X+ = (X + X sign X)/2
    from fp64 MFMA products (34 .. 64, see the shortened-schedule test below), no eigenpairs.  Against LAPACK's projection: every |eigenvalue| >= 1e-10 ||X||
    is resolved, smaller ones cost at most their own size; the count of positive eigenvalues comes from
    tr S and tr S^2.  Cases: generic, low-rank positive part, a 25 % null space, repeated eigenvalues with
    tiny ones next to zero, definite matrices, the zero matrix; sides that are not multiples of 32 / 64; 1000 / 1001 / 1430:
    the products run on 48 x 48 tiles (k_sym_gemm48, chosen by makespan)."""
    rng = np.random.default_rng(n)
    Qm, _ = np.linalg.qr(rng.standard_normal((n, n)))
    for name, lam in _spectrum_cases(n, rng).items():
        X = (Qm * lam) @ Qm.T
        X = (X + X.T) / 2
        w, V = np.linalg.eigh(X)
        ref = (V * np.maximum(w, 0.0)) @ V.T
        out, info = B.psd_project(svec(X), n, 1, mode=4)
        sc = max(np.abs(w).max(), 1e-300)
        err = np.abs(out - svec(ref)).max() / sc
        assert err <= 1e-9, (name, err)
        assert info["nmatvec"] == 0 and info["min_eig"] == 0.0
        if name in ("gauss", "lowrank_pos", "negdef", "posdef", "null"):
            assert info["rank"] == int((lam > 0).sum()), name
        elif name == "zeros":
            assert info["rank"] == int((lam > 1e-7).sum()), name       # the exact null space counts as zero
        dense, _ = B.psd_project(svec(X), n, 1, mode=1)
        assert np.abs(out - dense).max() / sc <= 1e-9


@pytest.mark.parametrize("n", [1000, 1430, 2000])
def test_sign_projection_tile_shapes_agree(n, monkeypatch):
    """The product kernels on 32 x 32 and on 48 x 48 tiles (PROXSDP_HIP_SIGN_TILE48 = 0 / 1; auto = by makespan: 48 at these
    sides) are the same iteration with another grouping of the K sums: projections agree to rounding, same positive count."""
    rng = np.random.default_rng(7 * n)
    M = rng.standard_normal((n, n)); X = (M + M.T) / 2
    outs = []
    for knob in ("0", "1", None):
        if knob is None:
            monkeypatch.delenv("PROXSDP_HIP_SIGN_TILE48", raising=False)
        else:
            monkeypatch.setenv("PROXSDP_HIP_SIGN_TILE48", knob)
        out, info = B.psd_project(svec(X), n, 1, mode=4)
        outs.append((out, info["rank"]))
    sc = np.abs(np.linalg.eigvalsh(X)).max()
    assert np.abs(outs[0][0] - outs[1][0]).max() <= 1e-12 * sc and outs[0][1] == outs[1][1]
    assert np.array_equal(outs[1][0], outs[2][0])                  # auto picks the 48-tiles here


@pytest.mark.parametrize("n", [100, 501, 1000])
def test_shortened_sign_schedule_is_tested_and_completed_when_the_test_fails(n):
    """options.sign_start_row: the sign iteration starts at row 8 of its coefficient table (34 products instead of
    57), which resolves |eigenvalues| >= 1.1e-5 s (s = the spectral scale (sum lambda^4)^(1/4)); the result is then
    TESTED (sum t^2 (1 - t^2) over the eigenvalues t of the computed sign matrix) and the skipped rows are run only
    when the test fails.  Spectra: (a) nothing below 1e-3 s -> 34 products; (b) a symmetric PAIR +-1e-7 s (every
    odd trace functional cancels on it) and (c) a single +3e-10 s: below what row 8 resolves, above the 1e-10 s the
    full table resolves -> the test must fail, the run is completed (64 products: the final product, enqueued ahead of the test, runs twice) and the result is as accurate as
    the full table's; (d) +-3e-12 s and an exact null direction: below what ANY schedule resolves -> the test passes,
    the error stays below 1e-10 s.  Every case against LAPACK and against the full table (sign_start_row = 0)."""
    rng = np.random.default_rng(100 + n)
    Qm, _ = np.linalg.qr(rng.standard_normal((n, n)))
    base = rng.standard_normal(n) * 3.0
    base[np.abs(base) < 0.1] = 0.5                                   # nothing small by accident
    s0 = float((base ** 4).sum() ** 0.25)
    cases = {"clear": (base.copy(), True)}
    lam = base.copy(); lam[0], lam[1] = 1e-7 * s0, -1e-7 * s0
    cases["pair_1e-7"] = (lam, False)
    lam = base.copy(); lam[0] = 3e-10 * s0
    cases["single_3e-10"] = (lam, False)
    lam = base.copy(); lam[0], lam[1], lam[2] = 3e-12 * s0, -3e-12 * s0, 0.0
    cases["harmless"] = (lam, True)
    for name, (lam, passes) in cases.items():
        X = (Qm * lam) @ Qm.T
        X = (X + X.T) / 2
        w, V = np.linalg.eigh(X)
        ref = svec((V * np.maximum(w, 0.0)) @ V.T)
        sc = float((w ** 4).sum() ** 0.25)
        full, _, rk_full, p_full = B.full_eig_kernel(svec(X), n, sign=100, repeat=1)
        short, _, rk_short, p_short = B.full_eig_kernel(svec(X), n, sign=108, repeat=1)
        print(f"n={n} {name}: products full {p_full} short {p_short} | err full {np.abs(full - ref).max() / sc:.2e} "
              f"short {np.abs(short - ref).max() / sc:.2e}")
        assert p_full == 57
        assert p_short == (34 if passes else 64), name
        assert np.abs(full - ref).max() <= 1e-10 * sc, name
        assert np.abs(short - ref).max() <= 1e-10 * sc, name
        if name != "harmless":
            assert rk_short == rk_full == int((w > 0).sum()), name


@pytest.mark.parametrize("n,seed", [(420, 1), (1000, 0)])
def test_implicit_full_eig_regime_served_by_lanczos(n, seed):
    """target_rank beyond max_target_rank_krylov_eigs: the reference's psd_projection! falls back to
    full_eig! (LAPACK dsyevr, prox_operators.jl:46-59).  full_eig! needs every POSITIVE eigenpair; the
    library computes them with the Lanczos engine when the previous projection of the block had few
    (full_eig_lanczos auto), with the dense solver as fallback.  Same projection => same trace as the
    all-dense run of the library and as the oracle (dsyevr)."""
    pr = P.maxcut(n, seed=seed)
    iters = 80
    kw = dict(max_iter=iters, max_target_rank_krylov_eigs=2, initial_target_rank=3)
    o = Options()
    o.max_iter = iters
    o.max_target_rank_krylov_eigs = 2
    o.initial_target_rank = 3
    ref = oracle.solve(pr, o, trace=True)
    G = _trace_cols(ref.trace)
    for fel in (0, -1):
        opt = Optimizer(full_eig_lanczos=fel, **kw)
        sol = opt.optimize(pr, trace_capacity=iters)
        T = sol.trace[:, [1, 2, 3, 4, 7, 11]]
        assert sol.status == ref.status and sol.iter == ref.iter == iters
        assert sol.stats["full_eigs"] == iters == ref.stats["full_eigs"]
        assert np.array_equal(T[:, 5], G[:, 5])
        assert np.allclose(T, G, rtol=1e-6, atol=1e-8 * np.abs(G).max()), fel
        print(n, "full_eig_lanczos", fel, "served by Lanczos", sol.stats["full_eigs_lanczos"], "of", iters,
              "mat-vecs", sol.stats["lanczos_matvecs"])
        if fel == 0:
            assert sol.stats["full_eigs_lanczos"] == 0
        elif n == 420:                         # (n = 1000: > n/8 positive eigenvalues in these early iterations -> dense)
            assert sol.stats["full_eigs_lanczos"] >= 10
        assert sol.final_rank == ref.final_rank


@pytest.mark.parametrize("n", [150, 420])
def test_lanczos_served_full_eig_is_verified_against_the_dense_engine(n):
    """full_eig! served by the Lanczos engine is the library's own algorithm; a single-vector
    Krylov space that is DEFICIENT in an eigendirection (a repeated eigenvalue shows one copy; a start vector with no
    component along an eigenvector never finds it) silently drops positive eigenpairs from X+.  Made exact here: a
    diagonal matrix and a start vector with zeros at two of the positive coordinates -- those coordinates stay
    exactly zero through every mat-vec and re-orthogonalisation, so the engine cannot see lambda = 7 and lambda = 2.
    options.full_eig_lanczos_verify (auto: the first call of a block, then periodically) runs the dense engine on
    the same input, detects the difference and hands the dense result back (psd_project mode 2: fell_back = 1).
    With a generic start vector the check passes and the Lanczos result is kept."""
    rng = np.random.default_rng(n)
    lam = -rng.uniform(0.5, 3.0, n)
    top = [9.0, 7.0, 5.0, 3.0, 2.0, 1.0]
    lam[:len(top)] = top
    X = np.diag(lam)
    ref = svec(np.diag(np.maximum(lam, 0.0)))
    generic = rng.standard_normal(n)
    deficient = generic.copy()
    deficient[[1, 4]] = 0.0
    out, info = B.psd_project(svec(X), n, len(top), mode=2, resid=generic)
    assert np.abs(out - ref).max() <= 1e-9 * 9.0 and info["rank"] == len(top) and info["fell_back"] == 0
    # the hazard, unverified: two positive eigenpairs are lost
    o = B.default_options()
    B.set_option(o, "full_eig_lanczos_verify", 0)
    B.set_option(o, "full_eig_lanczos_certify", 0)   # (the per-call certificate has its own test below)
    raw, info0 = B.psd_project(svec(X), n, len(top), mode=2, options=o, resid=deficient)
    assert info0["fell_back"] == 0 and info0["rank"] == len(top) - 2
    assert abs(np.abs(raw - ref).max() - 7.0) <= 1e-9
    # verified by the dense engine (first call of a block): its projection is returned
    o = B.default_options()
    B.set_option(o, "full_eig_lanczos_certify", 0)
    out, info = B.psd_project(svec(X), n, len(top), mode=2, options=o, resid=deficient)
    assert info["fell_back"] == 1 and info["rank"] == len(top) and info["min_eig"] == 0.0
    assert np.abs(out - ref).max() <= 1e-9 * 9.0


def test_lanczos_served_full_eig_is_certified_on_every_call():
    """the dense check of the previous test runs on the first call and every 256th after it; the
    CERTIFICATE (options.full_eig_lanczos_certify, default on) runs on EVERY call: a second, independent vector is
    orthogonalised against the returned Ritz vectors and pushed through 10 steps of the same recurrence with them locked
    -- Lanczos on the deflated operator -- and its largest Ritz value must not be positive.  Exact hazards: an eigenvalue
    5 of MULTIPLICITY 3 on coordinates the start vector sees only one of (a single-vector Krylov space contains one
    eigenvector per distinct eigenvalue: two copies are lost) plus an eigenvalue 7 it does not see at all.  With the
    periodic dense check switched off: the uncertified engine returns a projection that lacks 7 and two copies of 5;
    the certified one detects it and hands the input to the dense engine.  A generic start vector passes."""
    n = 257
    rng = np.random.default_rng(11)
    lam = -rng.uniform(0.5, 3.0, n)
    top = [9.0, 7.0, 5.0, 5.0, 5.0, 3.0, 1.0]
    lam[:len(top)] = top
    X = np.diag(lam)
    ref = svec(np.diag(np.maximum(lam, 0.0)))
    generic = rng.standard_normal(n)
    deficient = generic.copy()
    deficient[[1, 3, 4]] = 0.0                       # lambda = 7 and two of the three copies of lambda = 5
    o = B.default_options()
    B.set_option(o, "full_eig_lanczos_verify", 0)
    B.set_option(o, "full_eig_lanczos_certify", 0)
    # (previous positive count = 4 = what the deficient Krylov space can see: the count-collapse guard stays quiet)
    raw, info0 = B.psd_project(svec(X), n, len(top) - 3, mode=2, options=o, resid=deficient)
    assert info0["fell_back"] == 0 and info0["rank"] == len(top) - 3          # the hazard: three positive pairs dropped
    assert abs(np.abs(raw - ref).max() - 7.0) <= 1e-9
    o = B.default_options()
    B.set_option(o, "full_eig_lanczos_verify", 0)    # certificate alone (default: 10 steps)
    out, info = B.psd_project(svec(X), n, len(top) - 3, mode=2, options=o, resid=deficient)
    assert info["fell_back"] == 1 and info["rank"] == len(top)
    assert np.abs(out - ref).max() <= 1e-9 * 9.0
    out, info = B.psd_project(svec(X), n, len(top), mode=2, options=o, resid=generic)
    assert info["fell_back"] == 0 and info["rank"] == len(top)               # nothing left outside the returned pairs
    assert np.abs(out - ref).max() <= 1e-9 * 9.0
    # a repeated eigenvalue in a GENERIC basis and a dense matrix: certificate passes or fails, the result is right either way
    Qm, _ = np.linalg.qr(rng.standard_normal((n, n)))
    Xd = (Qm * lam) @ Qm.T
    Xd = (Xd + Xd.T) / 2
    w, V = np.linalg.eigh(Xd)
    refd = svec((V * np.maximum(w, 0.0)) @ V.T)
    out, info = B.psd_project(svec(Xd), n, len(top), mode=2, options=o)
    assert np.abs(out - refd).max() <= 1e-8 * 9.0 and info["rank"] == len(top)


def test_final_rank_of_the_sign_path_is_bounded_against_the_reference_count():
    """full_eig! counts current_rank = #{lambda > tol_psd} (prox_operators.jl:123); the sign-function path counts
    #{lambda > 0} from tr S and tr S^2 (ADVICE r2 / VERDICT r2 item 9).  Eigenvalues in (0, tol_psd] are the only
    difference: the sign path's count is >= the reference's and exceeds it by at most the number of such
    eigenvalues that the iteration resolves (those >= 1e-10 ||X||; smaller ones contribute fractions)."""
    n = 257
    rng = np.random.default_rng(5)
    Qm, _ = np.linalg.qr(rng.standard_normal((n, n)))
    lam = -rng.uniform(0.5, 3.0, n)
    lam[:6] = [9.0, 7.0, 5.0, 3.0, 2.0, 1.0]
    lam[6:10] = [5e-8, 2e-8, 8e-9, 1e-9]            # inside (0, tol_psd = 1e-7]
    X = (Qm * lam) @ Qm.T
    X = (X + X.T) / 2
    out_s, info_s = B.psd_project(svec(X), n, 1, mode=4)
    out_d, info_d = B.psd_project(svec(X), n, 1, mode=1)
    assert info_d["rank"] == 6                       # the reference's count
    assert 6 <= info_s["rank"] <= 10                 # the documented deviation, bounded
    assert np.abs(out_s - out_d).max() <= 1e-9 * 9.0


def test_mimo_dense_vector_path_against_oracle():
    """BASELINE config 'MIMO' shape at n=120 (side 121 > 100: Lanczos path; every triangle
    entry is box-constrained, so Mty is dense and the dense vector passes are used)."""
    pr = P.mimo(120, seed=4)
    opt = Optimizer(max_iter=150)
    sol = opt.optimize(pr, trace_capacity=150)
    o = Options()
    o.max_iter = 150
    ref = oracle.solve(pr, o, trace=True)
    assert sol.status == ref.status and sol.iter == ref.iter
    G, T = _trace_cols(ref.trace), sol.trace[:, [1, 2, 3, 4, 7, 11]]
    assert np.array_equal(T[:, 5], G[:, 5])
    mv = sol.trace[:, 13]
    tight = max(3, int(np.argmax(mv > 25)) if np.any(mv > 25) else len(T))
    assert np.allclose(T[:tight], G[:tight], rtol=1e-8, atol=1e-11)
    assert sol.stats["lanczos_matvecs"] > 0


@pytest.mark.parametrize("case", ["mimo", "mixed_cones", "gpp124-2", "maxcut_dense_passes", "trial_limit"])
def test_general_path_batched_linesearch_is_bit_identical_to_one_trial_per_synchronisation(case, golden_dir):
    """options.general_batch: on models without a support set (full-vector passes) up to three linesearch candidates,
    their residual and gap reductions are evaluated in one batch of launches with ONE read-back per iteration; per
    candidate the arithmetic is that of the one-trial-at-a-time path (general_batch = 0), so every trace column --
    trial counts included -- must be IDENTICAL, not close.  Cases: MIMO (box rows on every entry), a mixed-cone
    model (SOC + PSD + LP rows), SDPLIB gpp124-2 with full_eig! every iteration (all-ones row: full support),
    Max-Cut forced onto the dense passes, and max_linsearch_steps = 2 (the reference's trial-limit quirk:
    the step is decayed once more and the residual re-evaluated with it)."""
    kw, iters = {}, 120
    if case == "mimo":
        pr = P.mimo(60, seed=2)
    elif case == "mixed_cones":
        pr = mixed_cones(3)
    elif case == "gpp124-2":
        pr = P.sdplib(golden_dir / "sdplib" / "gpp124-2.dat-s"); kw = dict(full_eig_decomp=1)
    elif case == "maxcut_dense_passes":
        pr = P.maxcut(150, seed=5); kw = dict(support_path=0)
    else:
        pr = P.mimo(40, seed=7); kw = dict(max_linsearch_steps=2)
    a = Optimizer(max_iter=iters, general_batch=0, **kw).optimize(pr, trace_capacity=iters)
    b = Optimizer(max_iter=iters, **kw).optimize(pr, trace_capacity=iters)
    assert a.status == b.status and a.iter == b.iter
    cols = [c for c in range(a.trace.shape[1]) if c != 12]            # (12 = elapsed seconds)
    assert np.array_equal(a.trace[:, cols], b.trace[:, cols])
    assert a.objval == b.objval and np.array_equal(a.primal, b.primal) and np.array_equal(a.dual_eq, b.dual_eq)
    if case == "trial_limit":
        assert a.trace[:, 11].max() == 2


def test_maxcut_n1000_reaches_tolerance():
    """BASELINE config 1 (Max-Cut ER n=1000, single PSD cone): converges to the solver's
    own tolerances; the solution is feasible and PSD; weak duality holds."""
    pr = P.maxcut(1000, seed=0)
    opt = Optimizer(tol_gap=1e-4, tol_feasibility=1e-4, time_limit=600.0)
    sol = opt.optimize(pr)
    assert opt.termination_status() == "OPTIMAL"
    assert sol.gap <= 1e-4 and sol.primal_feasible_user_tol
    X = P.unpack_psd(sol.primal, 1000)
    assert np.abs(np.diag(X) - 1).max() <= 1e-4 * (1 + math.sqrt(1000.0))     # equa_feasibility, residuals.jl:6-11
    assert np.linalg.eigvalsh(X).min() >= -1e-6
    assert sol.stats["lanczos_matvecs"] > 0


def test_krylov_dimension_beyond_255_is_served_by_the_dense_eigensolver_not_refused():
    """The reference accepts any max_target_rank_krylov_eigs / eigsolver_min_lanczos (options.jl:76,88); the step kernels
    hold 255 basis columns.  Beyond that (target rank > 127, or eigsolver_min_lanczos > 255) the library used to return
    PROXSDP_E_INVALID (VERDICT r4 missing #4); now the same truncated projection -- top target_rank pairs, positive ones
    kept, min_eig the smallest of them (prox_operators.jl:89-109) -- comes from the dense eigensolver.  Max-Cut n = 400
    started at target rank 150 (krylovdim 301) against the oracle's KrylovKit run: same iterations, traces to 1e-8."""
    pr = P.maxcut(400, seed=1)
    kw = dict(max_target_rank_krylov_eigs=200, initial_target_rank=150, max_iter=25)
    sol = Optimizer(**kw).optimize(pr, trace_capacity=25)
    o = Options()
    o.max_target_rank_krylov_eigs, o.initial_target_rank, o.max_iter = 200, 150, 25
    ref = oracle.solve(pr, o, trace=True)
    assert sol.status == ref.status and sol.iter == ref.iter == 25
    assert sol.stats["dense_truncated_projections"] == 25 and sol.stats["lanczos_matvecs"] == 0
    assert "dense eigensolver served" in sol.status_string
    G = _trace_cols(ref.trace)
    assert np.allclose(sol.trace[:, [1, 2, 3, 4, 7, 11]], G, rtol=1e-8, atol=1e-10)
    assert sol.final_rank == ref.final_rank
    # eigsolver_min_lanczos = 300 (> 255) with the reference's default ranks: every Krylov-branch projection goes the same way
    sol2 = Optimizer(eigsolver_min_lanczos=300, max_iter=30).optimize(pr, trace_capacity=30)
    o2 = Options()
    o2.eigsolver_min_lanczos, o2.max_iter = 300, 30
    ref2 = oracle.solve(pr, o2, trace=True)
    assert sol2.stats["dense_truncated_projections"] == 30
    assert np.allclose(sol2.trace[:, [1, 2, 3, 4, 7, 11]], _trace_cols(ref2.trace), rtol=1e-8, atol=1e-10)
    # the kernel-level entry too: target rank 130 on one block, against LAPACK's truncated projection.  min_eig is the
    # smallest of the top target_rank eigenvalues (prox_operators.jl:95 takes the minimum over everything KrylovKit
    # returned, which can be a few pairs MORE than asked for when they converge in the same cycle: not reproducible
    # without running KrylovKit's dimension; the truncated projection itself is)
    x = planted_packed(300, 7, list(np.linspace(90.0, 1.0, 140)), bulk=(-3.0, -0.5))
    out, info = B.psd_project(x, 300, 130, mode=0)
    w, Q = np.linalg.eigh(smat(x, 300))
    w, Q = w[::-1], Q[:, ::-1]
    assert info["rank"] == 130 and abs(info["min_eig"] - w[129]) <= 1e-9 * 90
    assert np.abs(out - svec((Q[:, :130] * w[:130]) @ Q[:, :130].T)).max() <= 1e-9 * 90
    ref_out, rk, mn, _ = oracle_project(x, 300, 130, False)
    assert rk == 130 and np.abs(out - ref_out).max() <= 1e-9 * 90


@pytest.mark.parametrize("name,iters", [("arch0", 120), ("qap5", 120), ("truss2", 120), ("control2", 120), ("thetaG11", 100)])
def test_sdplib_multi_block_families_follow_the_oracle_trace(name, iters, golden_dir):
    """The SDPLIB families the reference ships data for but rounds 1-4 never ran (VERDICT r4 missing #6), with the
    file's block structure kept (problems.sdplib_blocks: what a JuMP user writes): arch0 (one 161 x 161 block + 174
    nonnegative scalars: the Lanczos path beside 1 x 1 cones), qap5 (one 26 x 26 block: full_eig! below
    min_size_krylov_eigs), truss2 (33 blocks of side 4 + one scalar: the batched small-block Jacobi kernel), control2
    (sides 10 and 20), thetaG11 (one 801 x 801 block, 2401 rows).  Against a live oracle trace: same linesearch trials
    in every iteration, trace to 1e-7, same Lanczos mat-vec total where KrylovKit runs."""
    pr = P.sdplib_blocks(golden_dir / "sdplib" / f"{name}.dat-s")
    sol = Optimizer(max_iter=iters).optimize(pr, trace_capacity=iters)
    o = Options()
    o.max_iter = iters
    ref = oracle.solve(pr, o, trace=True)
    assert sol.iter == ref.iter == iters and sol.status == ref.status
    G = _trace_cols(ref.trace)
    T = sol.trace[:, [1, 2, 3, 4, 7, 11]]
    assert np.array_equal(T[:, 5], G[:, 5]), "linesearch trials differ"
    sc = max(1.0, np.abs(G[:, :2]).max())
    assert np.allclose(T[:, :5], G[:, :5], rtol=1e-7, atol=1e-9 * sc), np.abs(T[:, :5] - G[:, :5]).max()
    assert sol.stats["lanczos_matvecs"] == ref.stats["lanczos_matvecs"]
    if name == "truss2":
        assert sol.stats["batched_small_eigs"] > 0
    assert sol.final_rank == ref.final_rank


def test_weighted_warm_start_of_the_lanczos_served_full_eig_changes_work_not_results():
    """options.full_eig_lanczos_warm_pow (round 5): the start vector of a positive-part run weights the previous Ritz vectors
    by (lam_0 / lam_c)^p.  It is the library's own engine, converged to the same tolerances and certified either way: p = 0
    (rounds 3-4) and p = 1 (default) must give the same solve -- iterations, rank schedule, objective -- with different
    mat-vec totals.  Max-Cut n = 600 with max_target_rank_krylov_eigs = 4 (the implicit full_eig! regime starts at rank 5)."""
    pr = P.maxcut(600, seed=4)
    out = {}
    for p in (0.0, 1.0):
        opt = Optimizer(max_target_rank_krylov_eigs=4, full_eig_lanczos_warm_pow=p)
        sol = opt.optimize(pr, trace_capacity=20000)
        out[p] = (sol, opt.objective_value())
        assert sol.status == 1 and sol.stats["full_eigs_lanczos"] > 100
        assert sol.stats["full_eigs_lanczos_cert_failed"] == 0
    a, b = out[0.0][0], out[1.0][0]
    print("p = 0:", a.iter, a.stats["lanczos_matvecs"], "| p = 1:", b.iter, b.stats["lanczos_matvecs"])
    assert a.iter == b.iter and np.array_equal(a.trace[:, 10], b.trace[:, 10]) and np.array_equal(a.trace[:, 11], b.trace[:, 11])
    assert abs(out[0.0][1] - out[1.0][1]) <= 1e-9 * abs(out[0.0][1])
    assert np.allclose(a.trace[:, 1:8], b.trace[:, 1:8], rtol=1e-8, atol=1e-10)
    assert a.stats["lanczos_matvecs"] != b.stats["lanczos_matvecs"]


@pytest.mark.parametrize("seed", range(12))
def test_random_mixed_cone_models_with_rotating_options_follow_the_oracle(seed):
    """a fixed dozen of tools/fuzz_differential.py's random models (1-6 PSD blocks with sides from 1 to 150, one SOC, free variables,
    equalities and inequalities, shuffled variable ids), each with one path-changing REFERENCE option: same iterations and status,
    identical linesearch trials, trace to 1e-6, identical mat-vec totals while KrylovKit runs (round 5's sweep: 282 solves, none off)"""
    rng = np.random.default_rng(1000 + seed)
    pool = [1, 1, 2, 3, 4, 5, 8, 17, 31, 32, 33, 48, 64, 65, 101, 104, 130, 150]
    sides = tuple(int(v) for v in rng.choice(pool, int(rng.integers(1, 7))))
    pr = mixed_cones(seed=seed, sides=sides, soc_len=int(rng.integers(2, 9)), nfree=int(rng.integers(0, 5)),
                     p=int(rng.integers(3, 60)), m=int(rng.integers(0, 30)))
    ref_opt = [dict(), dict(line_search_flag=0), dict(full_eig_decomp=1), dict(approx_norm=0), dict(min_size_krylov_eigs=20),
               dict(max_target_rank_krylov_eigs=3, convergence_window=40), dict(tol_gap=1e-7, tol_feasibility=1e-7),
               dict(krylovkit_eager=1)][seed % 8]
    iters = 250
    o = Options()
    o.max_iter = iters
    for k_, v_ in ref_opt.items():
        o.set(k_, bool(v_) if isinstance(getattr(o, k_), bool) else v_)
    ref = oracle.solve(pr, o, trace=True)
    sol = Optimizer(max_iter=iters, **ref_opt).optimize(pr, trace_capacity=iters)
    assert sol.iter == ref.iter and sol.status == ref.status, (sides, ref_opt)
    G = _trace_cols(ref.trace)
    T = sol.trace[:, [1, 2, 3, 4, 7, 11]]
    assert np.array_equal(T[:, 5], G[:, 5]), (sides, ref_opt)
    sc = max(1.0, np.abs(G[:, :2]).max())
    assert np.abs(T[:, :5] - G[:, :5]).max() <= (1e-5 if "approx_norm" in ref_opt else 1e-6) * sc, (sides, ref_opt)
    if sol.stats["full_eigs_lanczos"] == 0:
        assert sol.stats["lanczos_matvecs"] == ref.stats["lanczos_matvecs"], (sides, ref_opt)
