"""CPU-only tests of the C-ABI library: it loads without a GPU, exports every
symbol include/proxsdp_hip.h declares, its options struct matches the
reference's Options names/defaults, and its host-side logic (start vector,
K x K eigen-solver, preprocess!/norm_scaling) agrees with the oracle.  No
compute entry point is called successfully here (there is no GPU); they must
fail loudly instead of falling back to a CPU path."""
import ctypes
import dataclasses

import numpy as np
import pytest
import scipy.sparse as sp

import oracle
from oracle import eig as oeig
from oracle import pdhg as opdhg
from proxsdp_jl_amd import binding as B
from proxsdp_jl_amd import problems as P
from proxsdp_jl_amd.optimizer import Optimizer

from kat_problems import KATS, sdp_wiki


def test_library_loads_and_exports_every_declared_symbol():
    L = B.lib()
    names = B.header_symbols()
    assert len(names) >= 14
    for name in names:
        assert hasattr(L, name), f"{name} declared in include/proxsdp_hip.h but not exported"
    assert L.proxsdp_hip_abi_version() == 9


def test_options_struct_layout_and_defaults_match_reference_options():
    o = B.default_options()
    assert o.struct_size == ctypes.sizeof(B.Options)
    ref = oracle.Options()
    for f in dataclasses.fields(ref):
        got = B.get_option(o, f.name)                      # by-name getter on the C side
        assert got == pytest.approx(float(getattr(ref, f.name))), f.name
        assert float(getattr(o, f.name)) == got, f"ctypes offset of {f.name} differs from the C struct"
    # round trip through the by-name setter hits the same field ctypes sees
    for k, (name, ctype) in enumerate(B.Options._fields_):
        if name.startswith("pad") or name.startswith("reserved") or name == "struct_size":
            continue
        B.set_option(o, name, 3 + k)
        assert float(getattr(o, name)) == 3 + k, name


def test_unknown_option_is_an_error():
    """MOI_wrapper.jl:84-93 / moitest.jl:153-156."""
    with pytest.raises(KeyError):
        Optimizer(unsupportedarg=10)
    opt = Optimizer(tol_gap=1e-6, max_iter=7)
    assert opt.get_attribute("tol_gap") == 1e-6 and opt.get_attribute("max_iter") == 7


def test_optimizer_attribute_plumbing():
    """moitest.jl:25-32,158-171."""
    m = Optimizer()
    assert m.SOLVER_NAME == "ProxSDP" and m.termination_status() == "OPTIMIZE_NOT_CALLED"
    assert m.time_limit_sec() is None
    m.set_time_limit_sec(0.0)
    assert m.time_limit_sec() == 0.0
    m.set_time_limit_sec(None)
    assert m.time_limit_sec() is None
    m.set_time_limit_sec(1.0)
    assert m.time_limit_sec() == 1.0
    assert m.silent()
    m.set_silent(False)
    assert not m.silent()


def test_no_cpu_fallback_without_a_device():
    if B.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(B.ProxSDPHipError) as e:
        B.solve(sdp_wiki(False))
    assert e.value.code == -2                               # PROXSDP_E_HIP
    with pytest.raises(B.ProxSDPHipError):
        B.symv_packed(np.ones(6), 3, np.ones(3))


def test_invalid_problem_is_rejected_before_touching_the_device():
    pr = sdp_wiki(False)
    pr.psd = [np.array([0, 1, 2, 3, 4])]                    # not a triangular number
    with pytest.raises(B.ProxSDPHipError) as e:
        B.solve(pr)
    assert e.value.code == -1
    pr = sdp_wiki(False)
    pr.psd = [np.array([0, 1, 2, 3, 4, 4])]                 # repeated variable
    with pytest.raises(B.ProxSDPHipError) as e:
        B.solve(pr)
    assert e.value.code == -1


def test_start_vector_bit_identical_to_oracle():
    for n, seed, init in [(1, 1234, 3), (7, 1234, 3), (1000, 1234, 3), (513, 99, 3), (64, 5, 2), (10, 1, 1)]:
        a = B.host_start_vector(n, seed, init)
        b = oeig.start_vector(n, seed, init)
        assert np.array_equal(a, b), (n, seed, init)


@pytest.mark.parametrize("k", [1, 2, 3, 5, 12, 25, 33, 64, 129])
def test_small_symmetric_eigensolver(k):
    rng = np.random.default_rng(k)
    A = rng.standard_normal((k, k))
    A = A + A.T
    d, V = B.host_symeig(A)
    ref = np.linalg.eigvalsh(A)
    scale = max(1.0, np.abs(ref).max())
    assert np.allclose(d, ref, rtol=0, atol=5e-14 * scale * k)
    assert np.allclose(V.T @ V, np.eye(k), atol=1e-13 * k)
    assert np.allclose(A @ V, V * d, atol=1e-12 * scale * k)


def test_small_eigensolver_arrowhead_and_degenerate():
    # the Rayleigh quotient right after a thick restart: diag + arrow row + tridiagonal tail
    k = 25
    rng = np.random.default_rng(0)
    T = np.diag(np.sort(rng.uniform(-1, 40, k))[::-1])
    T[15, :15] = T[:15, 15] = rng.standard_normal(15) * 1e-3
    for j in range(15, k - 1):
        T[j, j + 1] = T[j + 1, j] = rng.uniform(0.1, 2)
    d, V = B.host_symeig(T)
    assert np.allclose(d, np.linalg.eigvalsh(T), atol=1e-12)
    assert np.allclose(T @ V, V * d, atol=1e-11)
    # repeated eigenvalues and an exactly diagonal matrix
    d, V = B.host_symeig(np.diag([2.0, 2.0, 2.0, -1.0]))
    assert np.allclose(d, [-1, 2, 2, 2]) and np.allclose(V.T @ V, np.eye(4))
    d, V = B.host_symeig(np.zeros((6, 6)))
    assert np.allclose(d, 0) and np.allclose(V.T @ V, np.eye(6))


@pytest.mark.parametrize("build", [lambda: P.maxcut(9, seed=3), lambda: P.mimo(4, seed=1),
                                   lambda: KATS["double_sdp_from_moi"][0](),
                                   lambda: P.randsdp(5, 4, seed=2)])
def test_preprocess_matches_oracle(build):
    """preprocess! + norm_scaling (scaling.jl) and ||M||_F (pdhg.jl:121)."""
    pr = build()
    order, inv, c_scaled, fro = B.host_preprocess(pr)
    aff, cones = oracle.to_standard_form(pr)
    c_orig, var_ordering = opdhg.preprocess(aff, cones)
    opdhg.norm_scaling(aff, cones)
    M = sp.vstack([aff.A, aff.G])
    assert np.array_equal(inv, var_ordering)
    assert np.array_equal(order, np.argsort(var_ordering))
    assert np.allclose(c_scaled, aff.c, rtol=1e-15, atol=0)
    assert fro == pytest.approx(np.sqrt((M.data ** 2).sum()), rel=1e-14)


def test_preprocess_reorders_cone_variables_first():
    # variables 0,1 free; PSD cone on (4,2,3) given out of order; SOC on (5,6)
    A = sp.csc_matrix(np.arange(14, dtype=float).reshape(2, 7) + 1)
    pr = P.Problem(n=7, A=A, b=np.ones(2), G=sp.csc_matrix((0, 7)), h=np.zeros(0),
                   c=np.arange(7, dtype=float), psd=[np.array([4, 2, 3])], soc=[np.array([5, 6])])
    order, inv, c_scaled, fro = B.host_preprocess(pr)
    assert list(order) == [4, 2, 3, 5, 6, 0, 1]
    assert list(inv) == [5, 6, 1, 2, 0, 3, 4]
    s = np.sqrt(2) / 2
    assert np.allclose(c_scaled, [4, 2 * s, 3, 5, 6, 0, 1])


def test_dense_matrix_entry_validation():
    """proxsdp_problem.M_dense (include/proxsdp_hip.h): accepted only with the variables in
    solver order; the sparse structures then hold G alone."""
    pr = P.randsdp(5, 4, seed=2, dense=True)
    assert pr.A.nnz == 0 and pr.M_dense.shape == (4, 15)
    order, inv, c_scaled, fro = B.host_preprocess(pr)
    assert np.array_equal(order, np.arange(15)) and np.array_equal(inv, np.arange(15))
    assert fro == pytest.approx(np.sqrt(2 * (1 + .5 + 1 + .5 + .5)))    # the +-1 bound rows of G only (off-diagonals x sqrt(2)/2)
    # a free variable placed BEFORE the cone variables is not in solver order
    bad = P.Problem(n=4, A=sp.csc_matrix((1, 4)), b=np.ones(1), G=sp.csc_matrix((0, 4)), h=np.zeros(0),
                    c=np.ones(4), psd=[np.array([1, 2, 3])], M_dense=np.ones((1, 4)))
    with pytest.raises(B.ProxSDPHipError, match="solver order"):
        B.host_preprocess(bad)
    with pytest.raises(ValueError, match="wrong shape"):
        B._Marshalled(P.Problem(n=4, A=sp.csc_matrix((1, 4)), b=np.ones(1), G=sp.csc_matrix((0, 4)),
                                h=np.zeros(0), c=np.ones(4), M_dense=np.ones((2, 4))))


def _c_struct_fields(header, name):
    import re
    body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), header, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    out = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        fp = re.search(r"\(\*(\w+)\)", decl)                  # function pointer member
        if fp:
            out.append(fp.group(1))
            continue
        for part in decl.split(","):
            out.append(re.findall(r"(\w+)\s*(?:\[\w+\])?$", part.strip())[0])
    return out


@pytest.mark.parametrize("cname,jname,cls", [("proxsdp_problem", "Problem", "Problem"),
                                             ("proxsdp_stats", "Stats", "Stats"),
                                             ("proxsdp_result", "CResult", "Result")])
def test_bindings_mirror_the_header_field_by_field(cname, jname, cls):
    """The ctypes structures and the Julia shim's structs (julia/ProxSDPHip.jl, which cannot be
    executed here) must list exactly the header's members, in order."""
    import re
    header = B.HEADER_PATH.read_text()
    cf = _c_struct_fields(header, cname)
    pyf = [f for f, _ in getattr(B, cls)._fields_]
    assert pyf == cf
    jl = (B.HEADER_PATH.parent.parent / "julia" / "ProxSDPHip.jl").read_text()
    body = re.search(r"struct %s\b.*?\n(.*?)\n(?:    \w+\(\) = new\(\)\n)?end" % jname, jl, re.S).group(1)
    jf = [re.match(r"\s*(\w+)::", ln).group(1) for ln in body.splitlines() if re.match(r"\s*\w+::", ln)]
    assert jf == cf


@pytest.mark.parametrize("k", [96, 127, 190, 255])
def test_threaded_eigenvector_accumulation_is_bit_identical_to_serial(k):
    """host_util.hpp ql_implicit: from k >= 96 the rotation recurrence runs alone on the calling thread
    while helper threads replay the logged rotations on disjoint row slices of the eigenvector matrix.
    Same arithmetic per entry => the same bits as the serial loop; and a correct decomposition."""
    import time
    rng = np.random.default_rng(k)
    T = np.diag(rng.standard_normal(k)) + np.diag(rng.uniform(0.5, 1.5, k - 1), 1)
    T = T + np.triu(T, 1).T
    T[: k // 2, k // 2] += 0.3 * rng.standard_normal(k // 2)        # an arrow column, as after a thick restart
    T = np.triu(T) + np.triu(T, 1).T
    t0 = time.time(); d0, U0 = B.host_symeig(T, threads=0); t_serial = time.time() - t0
    t0 = time.time(); d1, U1 = B.host_symeig(T, threads=3); t_thr = time.time() - t0         # three helper threads, on request
    assert np.array_equal(d0, d1) and np.array_equal(U0, U1)
    assert np.allclose(d1, np.linalg.eigvalsh(T), rtol=0, atol=1e-12 * np.abs(T).max())
    assert np.allclose(U1.T @ U1, np.eye(k), atol=1e-13)
    assert np.abs(T @ U1 - U1 * d1).max() <= 1e-12 * np.abs(T).max() * k
    print(k, "serial %.3f ms, with helpers %.3f ms" % (1e3 * t_serial, 1e3 * t_thr))


@pytest.mark.parametrize("K,m", [(25, 15), (53, 31), (127, 76), (5, 1), (9, 8)])
def test_two_phase_eigensolver_of_the_restarted_rayleigh_quotient(K, m):
    """host_util.hpp symeig_tridiag_from: the arrow part [diag(D) f; f' .] is Householder-reduced
    first (in the solver: while the GPU runs the cycle), the alpha/beta tail appended, then one QL
    sweep -- must equal the dense eigen-decomposition of the assembled matrix."""
    rng = np.random.default_rng(K + m)
    D = np.sort(rng.uniform(1, 60, m))[::-1].copy()
    f = rng.standard_normal(m) * np.where(np.arange(m) < m // 3, 1e-13, 1e-2)     # converged + open pairs
    al = np.zeros(K); be = np.zeros(K)
    al[m:] = rng.standard_normal(K - m)
    be[m:K - 1] = 1 + 0.1 * rng.standard_normal(K - 1 - m)
    T = np.zeros((K, K))
    T[np.arange(m), np.arange(m)] = D
    T[m, :m] = T[:m, m] = f
    for j in range(m, K):
        T[j, j] = al[j]
        if j + 1 < K:
            T[j, j + 1] = T[j + 1, j] = be[j]
    d, U = B.host_symeig_arrow(D, f, al, be)
    assert np.allclose(d, np.linalg.eigvalsh(T), rtol=0, atol=1e-12 * np.abs(T).max())
    assert np.allclose(U.T @ U, np.eye(K), atol=1e-13)
    assert np.abs(T @ U - U * d).max() <= 1e-12 * np.abs(T).max()


def _rq(D, f, al, be):
    K, m = len(al), len(D)
    T = np.zeros((K, K))
    T[np.arange(m), np.arange(m)] = D
    if m:
        T[m, :m] = T[:m, m] = f
    for j in range(m, K):
        T[j, j] = al[j]
        if j + 1 < K:
            T[j, j + 1] = T[j + 1, j] = be[j]
    return T


@pytest.mark.parametrize("K", [5, 9, 25, 53, 97, 127, 190, 255])
def test_split_and_rank_one_merge_eigensolver_against_lapack(K):
    """csrc/host_eig_merge.hpp (options.host_eig_merge): the K x K Rayleigh quotient of the thick-restart Lanczos
    decomposed by a split (arrow + hub | tail, or the two halves of the first cycle's tridiagonal) and ONE rank-one
    merge -- deflation, secular equation per root from the nearer pole, Gu-Eisenstat eigenvectors.  Against LAPACK
    on: plain tridiagonals, restarted quotients with converged (|f| ~ 1e-13) and open pairs, repeated Ritz values,
    a vanishing coupling (everything deflates) and a graded spectrum."""
    rng = np.random.default_rng(K)
    m = (3 * K) // 5
    cases = []
    al = rng.standard_normal(K) * 3
    be = np.abs(1 + 0.3 * rng.standard_normal(K))
    cases.append(("tridiagonal", np.zeros(0), np.zeros(0), al, be, K // 2))
    D = np.sort(rng.uniform(1, 60, m))[::-1].copy()
    f = rng.standard_normal(m) * np.where(np.arange(m) < m // 3, 1e-13, 1e-2)
    al2 = np.zeros(K); be2 = np.zeros(K)
    al2[m:] = rng.standard_normal(K - m)
    be2[m:] = np.abs(1 + 0.1 * rng.standard_normal(K - m))
    cases.append(("restart", D, f, al2, be2, m + 1))
    D3 = D.copy(); D3[1] = D3[0]
    cases.append(("repeated", D3, np.abs(f) + 1e-3, al2, be2, m + 1))
    be4 = be2.copy(); be4[m] = 1e-14
    cases.append(("decoupled", D, f, al2, be4, m + 1))
    cases.append(("graded", np.zeros(0), np.zeros(0), np.geomspace(1e3, 1e-6, K), np.geomspace(1, 1e-8, K), K // 2))
    for name, D_, f_, al_, be_, k1 in cases:
        T = _rq(D_, f_, al_, be_)
        d, U, info = B.host_symeig_split(D_, f_, al_, be_, k1)
        sc = max(1.0, np.abs(T).max())
        assert np.abs(d - np.linalg.eigvalsh(T)).max() <= 1e-13 * K * sc, name
        assert np.abs(U.T @ U - np.eye(K)).max() <= 1e-13, name
        assert np.abs(T @ U - U * d).max() <= 1e-13 * K * sc, name
        assert info["nondeflated"] + info["deflated"] == K and info["max_secular_iterations"] <= 40, (name, info)
        if name == "decoupled":
            assert info["nondeflated"] == 0
        # the merge's independent pieces on helper threads (options.host_merge_threads): the same bits
        d3, U3, info3 = B.host_symeig_split(D_, f_, al_, be_, k1, threads=3)
        assert np.array_equal(d3, d) and np.array_equal(U3, U) and info3 == info, name


def test_julia_shim_constructs_the_problem_struct_with_every_field():
    """julia/ProxSDPHip.jl cannot be executed here; at least its positional `Problem(...)` constructor call must
    pass exactly one argument per field of the struct it mirrors (a field added to proxsdp_problem and to the Julia
    struct but not to the call would be a MethodError in Julia), with index_base = 1 in the 15th position."""
    import re
    jl = (B.HEADER_PATH.parent.parent / "julia" / "ProxSDPHip.jl").read_text()
    body = re.search(r"struct Problem\b.*?\n(.*?)\nend", jl, re.S).group(1)
    nfields = len([ln for ln in body.splitlines() if re.match(r"\s*\w+::", ln)])
    call = jl[jl.index("prob = Problem("):]
    depth, args, cur = 0, [], ""
    for ch in call[len("prob = Problem("):]:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            if depth == 0:
                args.append(cur)
                break
            depth -= 1
        if ch == "," and depth == 0:
            args.append(cur); cur = ""
        else:
            cur += ch
    args = [re.sub(r"#.*", "", a, flags=re.M).strip() for a in args]
    assert len(args) == nfields == len(B.Problem._fields_), (len(args), nfields)
    assert args[14] == "Int32(1)"                        # index_base: Julia indices are 1-based
