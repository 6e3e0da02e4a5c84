"""ctypes binding of include/proxsdp_hip.h (libproxsdp_hip.so).

This is the Python twin of the Julia `ccall` shim in julia/ProxSDPHip.jl: it
marshals the standard form that `_optimize!` builds
(/root/reference/src/MOI_wrapper.jl:229-292) into `proxsdp_problem`, calls
`proxsdp_hip_solve` -- the replacement for `chambolle_pock(aff, con, options)`
at MOI_wrapper.jl:310 -- and copies `proxsdp_result` out.

There is no CPU fallback: a missing library raises ImportError-like
`LibraryNotBuilt`, and every compute entry point raises `ProxSDPHipError` when
the HIP runtime reports no device.
"""
import ctypes as C
import os
import pathlib
import re

import numpy as np
import scipy.sparse as sp

_HERE = pathlib.Path(__file__).resolve().parent
LIB_PATH = _HERE / "libproxsdp_hip.so"
HEADER_PATH = _HERE.parent / "include" / "proxsdp_hip.h"

TRACE_COLS = 14
TRACE_NAMES = ("iter", "prim_obj", "dual_obj", "gap", "feas", "prim_res", "dual_res",
               "primal_step", "beta", "theta", "target_rank", "trials", "elapsed", "matvecs")


class LibraryNotBuilt(RuntimeError):
    pass


class ProxSDPHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libproxsdp_hip error {code}: {msg}")
        self.code = code


i32, i64, f64 = C.c_int32, C.c_int64, C.c_double
pi64 = C.POINTER(C.c_int64)
pf64 = C.POINTER(C.c_double)


class CSC(C.Structure):
    _fields_ = [("nrows", i64), ("ncols", i64), ("colptr", pi64), ("rowval", pi64), ("nzval", pf64)]


class Problem(C.Structure):
    _fields_ = [("n", i64), ("p", i64), ("m", i64), ("A", CSC), ("G", CSC),
                ("b", pf64), ("h", pf64), ("c", pf64),
                ("n_psd", i64), ("psd_ptr", pi64), ("psd_idx", pi64),
                ("n_soc", i64), ("soc_ptr", pi64), ("soc_idx", pi64),
                ("index_base", i32), ("reserved0", i32), ("eig_resid", pf64),
                ("reduce_ctx", C.c_void_p), ("reduce_fn", C.c_void_p),
                ("M_dense", C.c_void_p), ("M_dense_on_device", i32), ("reserved1", i32),
                ("n_coupling", i64), ("coupling_rows", pi64), ("coupling_owned", C.POINTER(i32)),
                ("reduce_vec_fn", C.c_void_p), ("reduce_vec_on_device", i32), ("reserved2", i32),
                ("nccl_comm", C.c_void_p), ("reserved3", i64)]


REDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, pf64, i32, pf64, i32)
REDUCE_VEC_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, i64, i32)


def _opt_fields():
    F = []
    a = lambda name, t: F.append((name, t))
    a("struct_size", i64)
    for nme in ("log_verbose", "log_freq", "timer_verbose", "timer_file", "disable_julia_logger",
                "warn_on_limit", "extended_log", "extended_log2", "log_repeat_header", "pad0"):
        a(nme, i32)
    a("time_limit", f64)
    for nme in ("tol_gap", "tol_feasibility", "tol_feasibility_dual", "tol_primal", "tol_dual",
                "tol_psd", "tol_soc"):
        a(nme, f64)
    a("check_dual_feas", i32); a("check_dual_feas_freq", i32)
    a("max_obj", f64); a("min_iter_max_obj", i32); a("pad1", i32)
    a("min_iter_time_infeas", i32); a("pad2", i32)
    for nme in ("infeas_gap_tol", "infeas_limit_gap_tol", "infeas_stable_gap_tol",
                "infeas_feasibility_tol", "infeas_stable_feasibility_tol"):
        a(nme, f64)
    a("certificate_search", i32); a("pad3", i32)
    a("certificate_obj_tol", f64); a("certificate_fail_tol", f64)
    a("min_beta", f64); a("max_beta", f64); a("initial_beta", f64)
    a("initial_adapt_level", f64); a("adapt_decay", f64); a("adapt_window", i32); a("pad4", i32)
    a("convergence_window", i32); a("convergence_check", i32)
    for nme in ("max_iter", "min_iter", "divergence_min_update", "max_iter_lp", "max_iter_conic",
                "max_iter_local"):
        a(nme, i64)
    a("advanced_initialization", i32); a("line_search_flag", i32)
    a("max_linsearch_steps", i32); a("pad5", i32)
    a("delta", f64); a("initial_theta", f64); a("linsearch_decay", f64)
    a("full_eig_decomp", i32); a("max_target_rank_krylov_eigs", i32)
    a("min_size_krylov_eigs", i32); a("warm_start_eig", i32)
    a("rank_increment", i32); a("rank_increment_factor", i32)
    a("eigsolver", i32); a("eigsolver_min_lanczos", i32); a("eigsolver_resid_seed", i64)
    a("arpack_tol", f64); a("arpack_resid_init", i32); a("arpack_reset_resid", i32); a("arpack_max_iter", i64)
    a("krylovkit_reset_resid", i32); a("krylovkit_resid_init", i32)
    a("krylovkit_tol", f64); a("krylovkit_max_iter", i32); a("krylovkit_eager", i32); a("krylovkit_verbose", i32)
    a("reduce_rank", i32); a("rank_slack", i32); a("pad6", i32)
    a("full_eig_freq", i64); a("full_eig_len", i64)
    a("equilibration", i32); a("equilibration_iters", i32)
    a("equilibration_lb", f64); a("equilibration_ub", f64); a("equilibration_limit", f64)
    a("equilibration_force", i32); a("approx_norm", i32)
    a("device_id", i32); a("trace_capacity", i32); a("profile_symv_every", i32); a("support_path", i32)
    a("lanczos_operator", i32); a("initial_target_rank", i32)
    a("full_eig_lanczos", i32); a("lanczos_cycle_kernel", i32); a("lanczos_warm_start", i32)
    a("reconstruct_mfma", i32); a("small_block_batch", i32); a("full_eig_sign", i32); a("psd_sign_engine", i32)
    a("full_eig_lanczos_verify", i32); a("full_eig_lanczos_posres", f64); a("full_eig_lanczos_kdim10", i32)
    a("sign_small_tile_max", i32); a("host_eig_threads", i32); a("block_threads", i32)
    a("host_eig_merge", i32); a("block_batch", i32); a("rocsolver_warmup", i32); a("debug_fail_iteration", i32); a("host_wait_spin", i32)
    a("sign_start_row", i32); a("general_batch", i32); a("full_eig_lanczos_certify", i32); a("host_merge_threads", i32); a("reserved_i", i32 * 1)
    a("full_eig_lanczos_tol", f64); a("reserved_d", f64 * 1)
    a("equilibration_reference_aliasing", i32); a("reserved_i3", i32 * 2)
    a("block_batch_groups", i32); a("reserved_i2", i32 * 8); a("full_eig_lanczos_warm_pow", f64); a("reserved_d2", f64 * 3)
    return F


class Options(C.Structure):
    _fields_ = _opt_fields()


class Stats(C.Structure):
    _fields_ = [("lanczos_matvecs", i64), ("lanczos_restarts", i64), ("lanczos_calls", i64),
                ("full_eigs", i64), ("krylov_fallbacks", i64), ("linesearch_trials", i64),
                ("symv_launches", i64), ("symv_profiled", i64), ("symv_profiled_ms", f64),
                ("symv_bytes", f64), ("algorithmic_bytes", f64), ("init_time", f64),
                ("loop_time", f64), ("exit_time", f64), ("t_primal", f64), ("t_psd", f64),
                ("t_linesearch", f64), ("t_residual", f64), ("dense_passes", i64), ("dense_ms", f64), ("fop_projections", i64), ("exit_matvecs", i64),
                ("host_eig_time", f64), ("host_eigs", i64), ("device_eigs", i64), ("batched_small_eigs", i64),
                ("mfma_reconstructions", i64), ("orth_profiled", i64), ("orth_profiled_ms", f64),
                ("full_eig_solver_ms", f64), ("full_eig_recon_ms", f64), ("cycle_launches", i64),
                ("full_eigs_lanczos", i64), ("cycle_steps", i64), ("cycle_ms", f64), ("warm_starts", i64),
                ("full_eigs_sign", i64), ("sign_products", i64),
                ("sign_engine_projections", i64), ("sign_engine_rejected", i64),
                ("sign_engine_checks", i64), ("sign_engine_mismatches", i64),
                ("full_eigs_lanczos_checks", i64), ("full_eigs_lanczos_mismatches", i64),
                ("batched_block_steps", i64), ("rccl_reductions", i64),
                ("batched_profiled_blocks", i64), ("host_eig_merges", i64),
                ("host_eig_overlap_time", f64), ("sign_short_pass", i64), ("sign_short_fail", i64),
                ("full_eigs_lanczos_certified", i64), ("full_eigs_lanczos_cert_failed", i64), ("cert_matvecs", i64),
                ("dense_truncated_projections", i64), ("reserved_s", i64 * 7)]


class Result(C.Structure):
    _fields_ = [("status", i32), ("certificate_found", i32), ("primal_feasible_user_tol", i32),
                ("dual_feasible_user_tol", i32), ("result_count", i32), ("final_rank", i32),
                ("iter", i64), ("primal_residual", f64), ("dual_residual", f64),
                ("objval", f64), ("dual_objval", f64), ("gap", f64), ("time", f64),
                ("dual_feasibility", f64),
                ("primal", pf64), ("dual_cone", pf64), ("dual_eq", pf64), ("dual_in", pf64),
                ("slack_eq", pf64), ("slack_in", pf64), ("trace", pf64), ("trace_rows", i64),
                ("status_string", C.c_char * 256), ("stats", Stats)]


STATE_NHIST = 7
STATE_HIST_NAMES = ("dual_gap", "prim_obj", "dual_obj", "feasibility", "primal_residual", "dual_residual", "comb_residual")
STATE_SCAL_NAMES = ("primal_step", "primal_step_old", "dual_step", "beta", "theta", "adapt_level",
                    "equa_feasibility", "ineq_feasibility", "dual_feasibility")


class State(C.Structure):
    _fields_ = [("struct_size", i64), ("iteration", i64), ("n", i64), ("Q", i64), ("n_psd", i64), ("hist_len", i64),
                ("x", pf64), ("y", pf64), ("Mty", pf64), ("Mx", pf64),
                ("target_rank", pi64), ("current_rank", pi64), ("min_eig", pf64), ("hist", pf64),
                ("scal", f64 * 16), ("ints", i64 * 8)]


_lib = None


def header_symbols():
    """Every function the public header declares (used by the CPU symbol test)."""
    txt = HEADER_PATH.read_text()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(proxsdp_(?:hip|host)_\w+)\s*\(", txt)))


def lib():
    """Load libproxsdp_hip.so (built in-tree by __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise LibraryNotBuilt(
            f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C proxsdp.jl_amd/csrc`).  There is no CPU fallback.")
    # PyTorch-ROCm wheels bundle their own HIP/HSA runtime.  Measured on the MI355X box: if the
    # system runtime behind this library initialises first, torch can no longer see the GPU
    # ("No HIP GPUs are available"); the other order works, including torch device pointers
    # handed to the library (M_dense).  So when torch is already imported, let it go first.
    import sys
    if "torch" in sys.modules:
        try:
            tc = sys.modules["torch"].cuda
            if tc.is_available():
                tc.init()
        except Exception:
            pass
    L = C.CDLL(str(LIB_PATH))
    L.proxsdp_hip_abi_version.restype = C.c_int
    L.proxsdp_hip_last_error.restype = C.c_char_p
    L.proxsdp_hip_default_options.argtypes = [C.POINTER(Options)]
    L.proxsdp_hip_default_options.restype = None
    L.proxsdp_hip_set_option.argtypes = [C.POINTER(Options), C.c_char_p, f64]
    L.proxsdp_hip_get_option.argtypes = [C.POINTER(Options), C.c_char_p, pf64]
    L.proxsdp_hip_solve.argtypes = [C.POINTER(Problem), C.POINTER(Options), C.POINTER(Result)]
    L.proxsdp_hip_solve_ex.argtypes = [C.POINTER(Problem), C.POINTER(Options), C.POINTER(Result),
                                       C.POINTER(State), C.POINTER(State)]
    L.proxsdp_hip_psd_project.argtypes = [pf64, i64, i32, i32, C.POINTER(Options), pf64, pf64,
                                          C.POINTER(i32), pf64, pi64, C.POINTER(i32), C.POINTER(i32)]
    L.proxsdp_hip_eigsolve.argtypes = [pf64, i64, i32, C.POINTER(Options), pf64, i32, pf64, pf64,
                                       C.POINTER(i32), C.POINTER(i32), pi64, C.POINTER(i32)]
    L.proxsdp_hip_symv_packed.argtypes = [pf64, i64, pf64, pf64, i32, pf64]
    L.proxsdp_hip_reconstruct.argtypes = [pf64, pf64, i64, i32, pf64, i32, pf64]
    L.proxsdp_hip_reconstruct_kernel.argtypes = [pf64, pf64, i64, i32, i32, pf64, i32, pf64]
    L.proxsdp_hip_full_eig_kernel.argtypes = [pf64, i64, i32, pf64, i32, pf64, C.POINTER(i32), C.POINTER(i64)]
    L.proxsdp_hip_spmv.argtypes = [C.POINTER(CSC), i32, i32, pf64, pf64]
    L.proxsdp_host_symeig.argtypes = [i32, pf64, pf64]
    L.proxsdp_host_start_vector.argtypes = [i64, i64, i32, pf64]
    L.proxsdp_host_preprocess.argtypes = [C.POINTER(Problem), pi64, pi64, pf64, pf64]
    L.proxsdp_hip_rccl_unique_id.argtypes = [C.c_void_p]
    L.proxsdp_hip_rccl_comm_init.argtypes = [i32, C.c_void_p, i32, i32, C.POINTER(C.c_void_p)]
    L.proxsdp_hip_rccl_comm_destroy.argtypes = [C.c_void_p]
    if L.proxsdp_hip_abi_version() != 9:
        raise ProxSDPHipError(-1, "ABI version mismatch")
    _lib = L
    return L


def _check(rc):
    if rc != 0:
        raise ProxSDPHipError(rc, lib().proxsdp_hip_last_error().decode(errors="replace"))


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def _p(a, t=pf64):
    return a.ctypes.data_as(t)


def default_options():
    o = Options()
    lib().proxsdp_hip_default_options(C.byref(o))
    return o


def set_option(o, name, value):
    """RawOptimizerAttribute semantics (MOI_wrapper.jl:84-93): unknown name is an error."""
    rc = lib().proxsdp_hip_set_option(C.byref(o), name.encode(), float(value))
    if rc != 0:
        raise KeyError(f"No parameter matching {name}")


def get_option(o, name):
    v = f64()
    rc = lib().proxsdp_hip_get_option(C.byref(o), name.encode(), C.byref(v))
    if rc != 0:
        raise KeyError(f"No parameter matching {name}")
    return v.value


def device_count():
    return lib().proxsdp_hip_device_count()


class _Marshalled:
    """Keeps the numpy arrays behind a proxsdp_problem alive."""

    def __init__(self, prob, eig_resid=None, index_base=0):
        """index_base = 1 hands the library what Julia's ccall hands it: 1-based Int64 colptr / rowval
        (SparseMatrixCSC, structs.jl:36-37) and 1-based cone variable lists (SDPSet.vec_i, SOCSet.idx,
        structs.jl:44-53); the offsets psd_ptr / soc_ptr stay 0-based (julia/ProxSDPHip.jl)."""
        if index_base not in (0, 1):
            raise ValueError("index_base must be 0 or 1")
        keep = []

        def csc(M, ncols):
            M = sp.csc_matrix(M, dtype=np.float64)
            M.sort_indices()
            cp, rv, nz = _i(M.indptr) + index_base, _i(M.indices) + index_base, _f(M.data)
            keep.extend([cp, rv, nz])
            return CSC(M.shape[0], ncols, _p(cp, pi64), _p(rv, pi64), _p(nz))

        P = Problem()
        P.n, P.p, P.m = prob.n, prob.A.shape[0], prob.G.shape[0]
        P.A, P.G = csc(prob.A, prob.n), csc(prob.G, prob.n)
        b, h, c = _f(prob.b), _f(prob.h), _f(prob.c)
        keep.extend([b, h, c])
        P.b, P.h, P.c = _p(b), _p(h), _p(c)

        def cones(lst):
            ptr = np.zeros(len(lst) + 1, dtype=np.int64)
            for k, v in enumerate(lst):
                ptr[k + 1] = ptr[k] + len(v)
            idx = (_i(np.concatenate(lst)) + index_base) if lst else np.zeros(1, dtype=np.int64)
            keep.extend([ptr, idx])
            return len(lst), _p(ptr, pi64), _p(idx, pi64)

        P.n_psd, P.psd_ptr, P.psd_idx = cones(list(prob.psd))
        P.n_soc, P.soc_ptr, P.soc_idx = cones(list(prob.soc))
        P.index_base = index_base
        if eig_resid is not None:
            r = _f(np.concatenate([np.asarray(v, float).ravel() for v in eig_resid]))
            keep.append(r)
            P.eig_resid = _p(r)
        Md = getattr(prob, "M_dense", None)
        if Md is not None:
            if prob.A.nnz:
                raise ValueError("M_dense given: the sparse A must have no stored entries")
            if hasattr(Md, "data_ptr"):                      # torch tensor on the GPU
                if not (Md.is_cuda and Md.is_contiguous() and Md.dtype.is_floating_point and Md.element_size() == 8):
                    raise ValueError("M_dense tensor must be a contiguous float64 CUDA tensor")
                if tuple(Md.shape) != (P.p, P.n):
                    raise ValueError("M_dense has the wrong shape")
                P.M_dense, P.M_dense_on_device = Md.data_ptr(), 1
            else:
                Md = np.ascontiguousarray(Md, dtype=np.float64)
                if Md.shape != (P.p, P.n):
                    raise ValueError("M_dense has the wrong shape")
                P.M_dense, P.M_dense_on_device = Md.ctypes.data, 0
            keep.append(Md)
        self.P, self.keep = P, keep


class SolveResult:
    """Result (structs.jl:60-81) copied out of proxsdp_result."""

    def __init__(self, R, n, p, m, arrays, trace):
        for name, _ in Result._fields_:
            if name in ("primal", "dual_cone", "dual_eq", "dual_in", "slack_eq", "slack_in", "trace", "stats"):
                continue
            v = getattr(R, name)
            setattr(self, name, v.decode(errors="replace") if isinstance(v, bytes) else v)
        self.primal, self.dual_cone, self.dual_eq, self.dual_in, self.slack_eq, self.slack_in = arrays
        self.stats = {k: (list(getattr(R.stats, k)) if k.startswith("reserved") else getattr(R.stats, k))
                      for k, _ in Stats._fields_}
        self.trace = trace[:R.trace_rows].copy()
        for k in ("certificate_found", "primal_feasible_user_tol", "dual_feasible_user_tol"):
            setattr(self, k, bool(getattr(self, k)))

    def trace_dicts(self):
        return [dict(zip(TRACE_NAMES, row)) for row in self.trace]


def rccl_available():
    return lib().proxsdp_hip_rccl_available() == 1


def rccl_unique_id():
    """128 bytes from ncclGetUniqueId (one rank calls this and ships the bytes to the others)."""
    buf = C.create_string_buffer(128)
    _check(lib().proxsdp_hip_rccl_unique_id(buf))
    return bytes(buf.raw)


def rccl_comm_init(nranks, uid, rank, device_id=0):
    """ncclCommInitRank through the library's librccl; returns the opaque communicator handle (int)."""
    comm = C.c_void_p()
    buf = C.create_string_buffer(bytes(uid), 128)
    _check(lib().proxsdp_hip_rccl_comm_init(int(nranks), buf, int(rank), int(device_id), C.byref(comm)))
    return comm.value


def rccl_comm_destroy(comm):
    if comm:
        _check(lib().proxsdp_hip_rccl_comm_destroy(C.c_void_p(comm)))


def _state_struct(n, Q, n_psd, window, state=None, iteration=0):
    """proxsdp_state over fresh numpy arrays (filled from the dict `state` when given).  Returns (struct, arrays)."""
    hl = 2 * int(window)
    arr = dict(x=np.zeros(max(n, 1)), y=np.zeros(max(Q, 1)), Mty=np.zeros(max(n, 1)), Mx=np.zeros(max(Q, 1)),
               target_rank=np.zeros(max(n_psd, 1), dtype=np.int64), current_rank=np.zeros(max(n_psd, 1), dtype=np.int64),
               min_eig=np.zeros(max(n_psd, 1)), hist=np.zeros((STATE_NHIST, hl)))
    S = State()
    S.struct_size = C.sizeof(State)
    S.iteration, S.n, S.Q, S.n_psd, S.hist_len = int(iteration), n, Q, n_psd, hl
    if state is not None:
        S.iteration = int(state["iteration"])
        for k in ("x", "y", "Mty", "Mx", "min_eig"):
            v = _f(state[k]).ravel()
            if len(v) != {"x": n, "Mty": n, "y": Q, "Mx": Q, "min_eig": n_psd}[k]:
                raise ValueError(f"state[{k!r}] has the wrong length")
            arr[k][:len(v)] = v
        for k in ("target_rank", "current_rank"):
            v = _i(state[k]).ravel()
            if len(v) != n_psd:
                raise ValueError(f"state[{k!r}] has the wrong length")
            arr[k][:len(v)] = v
        h = np.asarray(state["hist"], dtype=np.float64)
        if h.shape != (STATE_NHIST, hl):
            raise ValueError("state['hist'] must be 7 x 2*convergence_window")
        arr["hist"][:] = h
        for q, nme in enumerate(STATE_SCAL_NAMES):
            S.scal[q] = float(state[nme])
        S.ints[0], S.ints[1], S.ints[2] = int(state["rank_update"]), int(state["update_cont"]), int(state["ada_count"])
    S.x, S.y, S.Mty, S.Mx = _p(arr["x"]), _p(arr["y"]), _p(arr["Mty"]), _p(arr["Mx"])
    S.target_rank, S.current_rank = _p(arr["target_rank"], pi64), _p(arr["current_rank"], pi64)
    S.min_eig, S.hist = _p(arr["min_eig"]), _p(arr["hist"])
    return S, arr


def _state_dict(S, arr, n, Q, n_psd):
    d = dict(iteration=int(S.iteration), x=arr["x"][:n].copy(), y=arr["y"][:Q].copy(), Mty=arr["Mty"][:n].copy(),
             Mx=arr["Mx"][:Q].copy(), target_rank=arr["target_rank"][:n_psd].copy(),
             current_rank=arr["current_rank"][:n_psd].copy(), min_eig=arr["min_eig"][:n_psd].copy(),
             hist=arr["hist"].copy(), rank_update=int(S.ints[0]), update_cont=int(S.ints[1]), ada_count=int(S.ints[2]))
    for q, nme in enumerate(STATE_SCAL_NAMES):
        d[nme] = float(S.scal[q])
    return d


def solve(prob, options=None, eig_resid=None, trace_capacity=0, reduce=None, coupling=None, index_base=0, nccl_comm=None,
          resume=None, capture_iteration=None):
    """proxsdp_hip_solve: replaces chambolle_pock(aff, con, options) (MOI_wrapper.jl:310).
    Returns the minimisation objective; sign/constant fix-up is the caller's
    (MOI_wrapper.jl:336-337), see optimizer.Optimizer.
    reduce: optional callable(sums: np.ndarray, maxs: np.ndarray) -> None that all-reduces
    the two arrays in place over the shards of a block-sharded solve (see sharded.py).
    coupling: optional dict(rows=int64 array of this shard's row numbers, owned=int32 0/1 array,
    reduce_vec=callable(ptr: int, length: int, on_device: bool) -> None summing the buffer in place over
    the shards, on_device=bool) -- the rows shared with other shards (proxsdp_problem.coupling_rows).
    nccl_comm: optional RCCL communicator handle (rccl_comm_init): the library then reduces the scalar record and
    the coupling rows itself on its own stream (proxsdp_problem.nccl_comm); `reduce` / reduce_vec are not used.
    resume: optional state dict (as returned in `.state`, or by oracle.export_state) -- the solve continues with iteration
    state['iteration'] + 1 (proxsdp_hip_solve_ex); capture_iteration: k >= 1 -- the state after iteration k comes back as
    `.state` (None when the solve ended before k).  Vectors are in the solver's internal order and scaling."""
    L = lib()
    o = options if options is not None else default_options()
    if trace_capacity:
        o.trace_capacity = int(trace_capacity)
    M = _Marshalled(prob, eig_resid, index_base)
    if reduce is not None:
        def _cb(ctx, ps, ns, pm, nm):
            try:
                sums = np.ctypeslib.as_array(ps, shape=(ns,)) if ns > 0 else np.zeros(0)
                maxs = np.ctypeslib.as_array(pm, shape=(nm,)) if nm > 0 else np.zeros(0)
                reduce(sums, maxs)
                return 0
            except Exception:                      # never unwind into C
                import traceback
                traceback.print_exc()
                return 1
        cb = REDUCE_FN(_cb)
        M.keep.append(cb)
        M.P.reduce_fn = C.cast(cb, C.c_void_p)
        M.P.reduce_ctx = None
    if nccl_comm:
        # native path: the library issues the collectives itself (RCCL, its own stream); no callbacks
        M.P.nccl_comm = C.c_void_p(int(nccl_comm))
    if coupling is not None and len(coupling["rows"]) > 0:
        rows = _i(coupling["rows"])                 # 0-based row numbers of [A;G], whatever index_base
        owned = np.ascontiguousarray(coupling["owned"], dtype=np.int32)
        M.keep += [rows, owned]
        M.P.n_coupling = len(rows)
        M.P.coupling_rows = _p(rows, pi64)
        M.P.coupling_owned = owned.ctypes.data_as(C.POINTER(i32))
        rv = coupling.get("reduce_vec")
        if rv is not None and not nccl_comm:
            def _cbv(ctx, ptr, length, on_device):
                try:
                    rv(int(ptr), int(length), bool(on_device))
                    return 0
                except Exception:
                    import traceback
                    traceback.print_exc()
                    return 1
            cbv = REDUCE_VEC_FN(_cbv)
            M.keep.append(cbv)
            M.P.reduce_vec_fn = C.cast(cbv, C.c_void_p)
            M.P.reduce_vec_on_device = 1 if coupling.get("on_device") else 0
    n, p, m = M.P.n, M.P.p, M.P.m
    arrays = [np.zeros(max(k, 1)) for k in (n, n, p, m, p, m)]
    trace = np.zeros((max(o.trace_capacity, 1), TRACE_COLS))
    R = Result()
    R.primal, R.dual_cone, R.dual_eq, R.dual_in, R.slack_eq, R.slack_in = [_p(a) for a in arrays]
    R.trace = _p(trace)
    if resume is None and capture_iteration is None:
        _check(L.proxsdp_hip_solve(C.byref(M.P), C.byref(o), C.byref(R)))
        arrays = [a[:k] for a, k in zip(arrays, (n, n, p, m, p, m))]
        return SolveResult(R, n, p, m, arrays, trace)
    Q, nb = p + m, int(M.P.n_psd)
    rs = cs = None
    if resume is not None:
        rs, rarr = _state_struct(n, Q, nb, o.convergence_window, state=resume)
    if capture_iteration is not None:
        cs, carr = _state_struct(n, Q, nb, o.convergence_window, iteration=capture_iteration)
    _check(L.proxsdp_hip_solve_ex(C.byref(M.P), C.byref(o), C.byref(R),
                                  C.byref(rs) if rs is not None else None, C.byref(cs) if cs is not None else None))
    arrays = [a[:k] for a, k in zip(arrays, (n, n, p, m, p, m))]
    out = SolveResult(R, n, p, m, arrays, trace)
    out.state = _state_dict(cs, carr, n, Q, nb) if (cs is not None and cs.ints[3] == 1) else None
    return out


# ----------------------------------------------------------------- kernel-level entry points
def psd_project(packed, n, target_rank, mode=0, options=None, resid=None):
    L = lib()
    x = _f(packed)
    out = np.zeros_like(x)
    rank, conv, fell = i32(), i32(), i32()
    mineig, nmv = f64(), i64()
    r = _f(resid) if resid is not None else None
    _check(L.proxsdp_hip_psd_project(_p(x), n, target_rank, mode,
                                     C.byref(options) if options is not None else None,
                                     _p(r) if r is not None else None, _p(out),
                                     C.byref(rank), C.byref(mineig), C.byref(nmv), C.byref(conv), C.byref(fell)))
    return out, dict(rank=rank.value, min_eig=mineig.value, nmatvec=nmv.value,
                     converged=conv.value, fell_back=fell.value)


def eigsolve(packed, n, nev, options=None, resid=None, cap=None):
    L = lib()
    x = _f(packed)
    cap = cap or max(2 * nev + 2, 26)
    vals = np.zeros(cap)
    vecs = np.zeros((cap, n))           # row k = k-th vector (column-major n x cap on the C side)
    cnt, conv, nit = i32(), i32(), i32()
    nmv = i64()
    r = _f(resid) if resid is not None else None
    _check(L.proxsdp_hip_eigsolve(_p(x), n, nev, C.byref(options) if options is not None else None,
                                  _p(r) if r is not None else None, cap, _p(vals), _p(vecs),
                                  C.byref(cnt), C.byref(conv), C.byref(nmv), C.byref(nit)))
    k = min(cnt.value, cap)
    return vals[:k].copy(), vecs[:k].T.copy(), dict(count=cnt.value, converged=conv.value,
                                                    nmatvec=nmv.value, numiter=nit.value)


def symv_packed(packed, n, v, repeat=0):
    L = lib()
    x, vv = _f(packed), _f(v)
    y = np.zeros(n)
    ms = f64(0.0)
    _check(L.proxsdp_hip_symv_packed(_p(x), n, _p(vv), _p(y), repeat, C.byref(ms)))
    return (y, ms.value) if repeat else y


def primal_update(x, Mty, c, tau):
    L = lib()
    x, Mty, c = _f(x), _f(Mty), _f(c)
    out = np.zeros_like(x)
    L.proxsdp_hip_primal_update.argtypes = [pf64, pf64, pf64, f64, i64, pf64]
    _check(L.proxsdp_hip_primal_update(_p(x), _p(Mty), _p(c), float(tau), len(x), _p(out)))
    return out


def dual_trial(y, Mx, Mx_old, bh, p, bt, theta):
    L = lib()
    y, Mx, Mx_old, bh = _f(y), _f(Mx), _f(Mx_old), _f(bh)
    out = np.zeros_like(y)
    nrm = f64(0.0)
    L.proxsdp_hip_dual_trial.argtypes = [pf64, pf64, pf64, pf64, i64, i64, f64, f64, pf64, pf64]
    _check(L.proxsdp_hip_dual_trial(_p(y), _p(Mx), _p(Mx_old), _p(bh), int(p), len(y), float(bt), float(theta),
                                    _p(out), C.byref(nrm)))
    return out, nrm.value


def residuals(x, x_old, Mty, Mty_old, c, tau, y, y_old, Mx, Mx_old, bh, p, sigma):
    L = lib()
    a = [_f(v) for v in (x, x_old, Mty, Mty_old, c)]
    b = [_f(v) for v in (y, y_old, Mx, Mx_old, bh)]
    out = np.zeros(9)
    L.proxsdp_hip_residuals.argtypes = [pf64] * 5 + [f64, i64] + [pf64] * 5 + [i64, i64, f64, pf64]
    _check(L.proxsdp_hip_residuals(*[_p(v) for v in a], float(tau), len(a[0]), *[_p(v) for v in b],
                                   int(p), len(b[0]), float(sigma), _p(out)))
    return out


def reconstruct(Z, lam, n, repeat=0, mfma=-1):
    """mfma: -1 the library's choice, 0 scalar-FMA kernel, 1 fp64 MFMA SYRK"""
    L = lib()
    Zc = np.asfortranarray(Z, dtype=np.float64)
    lam = _f(lam)
    r = len(lam)
    out = np.zeros(n * (n + 1) // 2)
    ms = f64(0.0)
    _check(L.proxsdp_hip_reconstruct_kernel(Zc.ctypes.data_as(pf64), _p(lam), n, r, mfma, _p(out), repeat, C.byref(ms)))
    return (out, ms.value) if repeat else out


def full_eig_kernel(packed, n, sign=1, repeat=1):
    """full_eig! of one packed block on device-resident data: (X+ packed, ms per call, rank, products per call).
    sign: 0 rocSOLVER dsyevd, 1 sign-function projection (default sign_start_row), 100 + k: sign_start_row = k,
    -1: the solver's automatic engine choice (full_eig_sign = -1)"""
    L = lib()
    xin = _f(packed)
    out = np.zeros(n * (n + 1) // 2)
    ms = f64(0.0)
    rk = i32(0)
    npr = i64(0)
    _check(L.proxsdp_hip_full_eig_kernel(_p(xin), n, int(sign), _p(out), repeat, C.byref(ms), C.byref(rk), C.byref(npr)))
    return out, ms.value, rk.value, npr.value


def spmv(M, x, transpose=False):
    L = lib()
    M = sp.csc_matrix(M, dtype=np.float64)
    M.sort_indices()
    cp, rv, nz = _i(M.indptr), _i(M.indices), _f(M.data)
    S = CSC(M.shape[0], M.shape[1], _p(cp, pi64), _p(rv, pi64), _p(nz))
    xin = _f(x)
    out = np.zeros(M.shape[1] if transpose else M.shape[0])
    _check(L.proxsdp_hip_spmv(C.byref(S), 0, 1 if transpose else 0, _p(xin), _p(out)))
    return out


# ----------------------------------------------------------------- host-only helpers (no GPU)
def host_symeig(A, threads=-1):
    """threads: -1 the library's choice (helper threads from k >= 96), 0 serial -- bit-identical results"""
    L = lib()
    a = np.asfortranarray(A, dtype=np.float64).copy(order="F")
    k = a.shape[0]
    d = np.zeros(k)
    _check(L.proxsdp_host_symeig_threads(k, a.ctypes.data_as(pf64), _p(d), threads))
    return d, a


def host_symeig_arrow(D, f, al, be):
    """Two-phase eigen-decomposition of a thick-restarted Rayleigh quotient (see the header)."""
    L = lib()
    D, f, al, be = _f(D), _f(f), _f(al), _f(be)
    K, m = len(al), len(D)
    U = np.zeros((K, K), order="F")
    d = np.zeros(K)
    L.proxsdp_host_symeig_arrow.argtypes = [i32, i32, pf64, pf64, pf64, pf64, pf64, pf64]
    _check(L.proxsdp_host_symeig_arrow(K, m, _p(D), _p(f), _p(al), _p(be), U.ctypes.data_as(pf64), _p(d)))
    return d, U


def host_start_vector(n, seed=1234, init=3):
    out = np.zeros(n)
    _check(lib().proxsdp_host_start_vector(n, seed, init, _p(out)))
    return out


def host_symeig_split(D, f, al, be, k1, threads=0):
    """proxsdp_host_symeig_split: eigen-decomposition of the restarted Rayleigh quotient (as host_symeig_arrow) by a
    split at k1 + rank-one merge.  Returns (d ascending, U, info dict)."""
    D = _f(D); f = _f(f); al = _f(al); be = _f(be)
    K, m = len(al), len(D)
    U = np.zeros((K, K))
    d = np.zeros(K)
    info = (i32 * 3)()
    L = lib()
    L.proxsdp_host_symeig_split_threads.argtypes = [i32, i32, i32, pf64, pf64, pf64, pf64, i32, pf64, pf64, C.POINTER(i32)]
    _check(L.proxsdp_host_symeig_split_threads(K, m, int(k1), _p(D) if m else None, _p(f) if m else None, _p(al), _p(be),
                                               int(threads), _p(U), _p(d), info))
    return d, U.T.copy(), dict(nondeflated=info[0], deflated=info[1], max_secular_iterations=info[2])


def host_preprocess(prob, index_base=0):
    M = _Marshalled(prob, index_base=index_base)
    n = prob.n
    order, inv = np.zeros(n, dtype=np.int64), np.zeros(n, dtype=np.int64)
    cs = np.zeros(n)
    fro = f64()
    _check(lib().proxsdp_host_preprocess(C.byref(M.P), _p(order, pi64), _p(inv, pi64), _p(cs), C.byref(fro)))
    return order, inv, cs, fro.value
