"""TEST INFRASTRUCTURE ONLY -- CPU oracle, never imported by the product path.

Restatement of the reference's `Options` keyword struct
(/root/reference/src/options.jl:1-132).  Field names and defaults are the
reference's; the C ABI struct `proxsdp_options` (include/proxsdp_hip.h) carries
the same names.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this package.
"""
from dataclasses import dataclass, fields


@dataclass
class Options:
    # printing (options.jl:4-15)
    log_verbose: bool = False
    log_freq: int = 1000
    timer_verbose: bool = False
    timer_file: bool = False
    disable_julia_logger: bool = True
    time_limit: float = 360000.0
    warn_on_limit: bool = False
    extended_log: bool = False
    extended_log2: bool = False
    log_repeat_header: bool = False
    # tolerances (options.jl:18-27)
    tol_gap: float = 1e-4
    tol_feasibility: float = 1e-4
    tol_feasibility_dual: float = 1e-4
    tol_primal: float = 1e-4
    tol_dual: float = 1e-4
    tol_psd: float = 1e-7
    tol_soc: float = 1e-7
    check_dual_feas: bool = False
    check_dual_feas_freq: int = 1000
    max_obj: float = 1e20
    min_iter_max_obj: int = 10
    # infeasibility (options.jl:33-42)
    min_iter_time_infeas: int = 1000
    infeas_gap_tol: float = 1e-4
    infeas_limit_gap_tol: float = 1e-1
    infeas_stable_gap_tol: float = 1e-4
    infeas_feasibility_tol: float = 1e-4
    infeas_stable_feasibility_tol: float = 1e-8
    certificate_search: bool = True
    certificate_obj_tol: float = 1e-1
    certificate_fail_tol: float = 1e-8
    # beta bounds (unused in src/, kept for name parity; options.jl:45-47)
    min_beta: float = 1e-5
    max_beta: float = 1e5
    initial_beta: float = 1.0
    # adaptive steps (options.jl:50-52)
    initial_adapt_level: float = 0.9
    adapt_decay: float = 0.8
    adapt_window: int = 50
    # PDHG (options.jl:55-64)
    convergence_window: int = 200
    convergence_check: int = 50
    max_iter: int = 0
    min_iter: int = 40
    divergence_min_update: int = 50
    max_iter_lp: int = 10_000_000
    max_iter_conic: int = 1_000_000
    max_iter_local: int = 0
    advanced_initialization: bool = True
    # linesearch (options.jl:67-71)
    line_search_flag: bool = True
    max_linsearch_steps: int = 5000
    delta: float = 0.9999
    initial_theta: float = 1.0
    linsearch_decay: float = 0.75
    # spectral decomposition (options.jl:74-80)
    full_eig_decomp: bool = False
    max_target_rank_krylov_eigs: int = 16
    min_size_krylov_eigs: int = 100
    warm_start_eig: bool = True
    rank_increment: int = 1
    rank_increment_factor: int = 1
    # eigsolver selection (options.jl:87-89)
    eigsolver: int = 2
    eigsolver_min_lanczos: int = 25
    eigsolver_resid_seed: int = 1234
    # Arpack (options.jl:94-105)
    arpack_tol: float = 1e-10
    arpack_resid_init: int = 3
    arpack_reset_resid: bool = True
    arpack_max_iter: int = 10_000
    # KrylovKit (options.jl:108-113)
    krylovkit_reset_resid: bool = False
    krylovkit_resid_init: int = 3
    krylovkit_tol: float = 1e-12
    krylovkit_max_iter: int = 100
    krylovkit_eager: bool = False
    krylovkit_verbose: int = 0
    # rank heuristics (options.jl:116-117)
    reduce_rank: bool = False
    rank_slack: int = 3
    full_eig_freq: int = 10_000_000
    full_eig_len: int = 0
    # equilibration (options.jl:123-128)
    equilibration: bool = False
    equilibration_iters: int = 1000
    equilibration_lb: float = -10.0
    equilibration_ub: float = 10.0
    equilibration_limit: float = 0.9
    equilibration_force: bool = False
    # no reference counterpart: True restates equilibrate!'s Diagonal(u) aliasing (equilibration.jl:16-17,25-26)
    # exactly, False runs the iteration the code evidently intends; mirrored by the library
    equilibration_reference_aliasing: bool = True
    approx_norm: bool = True
    # no reference counterpart (pdhg.jl:19-20 hard-codes 2): lets a benchmark start at the rank a
    # BASELINE config names ("rank ~ sqrt(n)", "target rank 50"); mirrored by the library
    initial_target_rank: int = 2

    def set(self, name, value):
        """MOI.set(::Optimizer, ::RawOptimizerAttribute) semantics
        (/root/reference/src/MOI_wrapper.jl:84-93): unknown name is an error."""
        if name not in {f.name for f in fields(self)}:
            raise KeyError(f"No parameter matching {name}")
        setattr(self, name, value)

    def copy(self):
        return Options(**{f.name: getattr(self, f.name) for f in fields(self)})
