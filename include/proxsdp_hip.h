/*
 * proxsdp_hip.h -- C ABI of libproxsdp_hip.so, the MI355X (gfx950) implementation
 * of ProxSDP's PDHG hot path.
 *
 * Drop-in boundary.  The reference has no FFI: its solver is entered through ONE
 * call, `sol = chambolle_pock(aff, con, options)` at
 * /root/reference/src/MOI_wrapper.jl:310 (inside `_optimize!`, :220-342).
 * `proxsdp_hip_solve` replaces exactly that call: its three arguments carry
 *     AffineSets + ConicSets   /root/reference/src/structs.jl:32-58
 *     Options                  /root/reference/src/options.jl:1-132
 *     Result                   /root/reference/src/structs.jl:60-81
 * A Julia binding calls it with `ccall` (see INTEGRATION.md and julia/); the
 * Python ctypes binding in proxsdp.jl_amd/binding.py is what the tests use.
 *
 * Conventions
 *   - plain C, no C++/torch types; pointers are HOST pointers unless a field says
 *     otherwise (the two exceptions: `M_dense` with M_dense_on_device = 1 and the
 *     buffer handed to `reduce_vec_fn` with reduce_vec_on_device = 1 are DEVICE
 *     pointers on options.device_id), borrowed for the duration of the call, never
 *     freed or written by the library (the reference mutates `aff` in place --
 *     scaling.jl:24, pdhg.jl:647-663 -- the library works on private copies).
 *   - index_base = 1 is the Julia calling convention (1-based Int64 colptr / rowval /
 *     cone variable lists, structs.jl:32-53); exercised without Julia by
 *     tests/test_julia_convention.py (ctypes with 1-based arrays + a plain-C caller).
 *   - return value: 0 = the solve ran to a solver status (whatever it is);
 *     < 0 = the call failed (PROXSDP_E_*), text in proxsdp_hip_last_error().
 *     Nothing unwinds across the ABI.  There is NO CPU fallback: without a
 *     usable HIP device every compute entry point returns PROXSDP_E_HIP.
 *   - Float64 arithmetic throughout (the reference is Float64 only).
 */
#ifndef PROXSDP_HIP_H
#define PROXSDP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 7 (round 3): proxsdp_options gained sign_start_row and general_batch (taken from reserved_i), proxsdp_stats gained
 * sign_short_pass / sign_short_fail (taken from reserved): same struct sizes and offsets as version 6
 * 8 (round 4): proxsdp_options gained full_eig_lanczos_certify (from reserved_i) and full_eig_lanczos_tol (from
 * reserved_d) and host_merge_threads (from reserved_i), proxsdp_stats full_eigs_lanczos_certified / _cert_failed /
 * cert_matvecs (the last reserved slots):
 * same struct sizes and offsets; debug_fail_iteration now needs PROXSDP_HIP_FAULT_INJECTION=1
 * 9 (round 5): proxsdp_options grew at its END (equilibration_reference_aliasing, block_batch_groups,
 * new reserved slots -- offsets of every earlier member unchanged, struct_size larger), proxsdp_stats grew at its end
 * (proxsdp_result with it: it is the LAST member); a Krylov dimension beyond 255 is served by the dense eigensolver
 * instead of PROXSDP_E_INVALID; new proxsdp_state and
 * proxsdp_hip_solve_ex (capture / resume of the solver state at an iteration boundary); PROXSDP_E_COMM_ABORTED */
#define PROXSDP_HIP_ABI_VERSION 9

/* error codes (negative return values) */
#define PROXSDP_E_INVALID  (-1)   /* invalid argument / inconsistent problem data */
#define PROXSDP_E_HIP      (-2)   /* HIP runtime / rocSOLVER failure, no device   */
#define PROXSDP_E_NOMEM    (-3)   /* host or device allocation failed             */
#define PROXSDP_E_UNSUPP   (-4)   /* option combination not implemented           */
#define PROXSDP_E_INTERNAL (-5)
#define PROXSDP_E_COMM_ABORTED (-6) /* block-sharded solve on the native RCCL path: the solve failed AND the library called
                                     * ncclCommAbort on proxsdp_problem.nccl_comm -- the communicator is already released:
                                     * it must NOT be destroyed or used again by the caller */

/* solver status: Result.status, /root/reference/src/MOI_wrapper.jl:381-399 */
#define PROXSDP_STATUS_NOT_CALLED       0
#define PROXSDP_STATUS_OPTIMAL          1
#define PROXSDP_STATUS_TIME_LIMIT       2
#define PROXSDP_STATUS_ITERATION_LIMIT  3
#define PROXSDP_STATUS_INFEAS_OR_UNBND  4
#define PROXSDP_STATUS_DUAL_INFEASIBLE  5
#define PROXSDP_STATUS_INFEASIBLE       6

/* SparseMatrixCSC{Float64,Int64} (structs.jl:36-37) */
typedef struct proxsdp_csc {
    int64_t nrows;
    int64_t ncols;
    const int64_t* colptr;   /* ncols+1 entries, base = problem.index_base */
    const int64_t* rowval;   /* nnz entries,     base = problem.index_base */
    const double*  nzval;    /* nnz entries */
} proxsdp_csc;

/* AffineSets + ConicSets as assembled at MOI_wrapper.jl:229-292:
 *   minimise c'x   s.t.  A x = b,  G x <= h,  x restricted to the cones.
 * PSD cones are MOI PositiveSemidefiniteConeTriangle variable lists (upper
 * triangle, column by column); SOC cones are (t, x...) variable lists. */
typedef struct proxsdp_problem {
    int64_t n;                 /* number of scalar variables            (aff.n) */
    int64_t p;                 /* equality rows                         (aff.p) */
    int64_t m;                 /* inequality rows                       (aff.m) */
    proxsdp_csc A;             /* p x n */
    proxsdp_csc G;             /* m x n */
    const double* b;           /* p */
    const double* h;           /* m */
    const double* c;           /* n, already sign-flipped for MAX sense (:252-255) */
    int64_t n_psd;
    const int64_t* psd_ptr;    /* n_psd+1 offsets into psd_idx (0-based offsets)   */
    const int64_t* psd_idx;    /* SDPSet.vec_i of every cone, concatenated         */
    int64_t n_soc;
    const int64_t* soc_ptr;    /* n_soc+1 */
    const int64_t* soc_idx;    /* SOCSet.idx concatenated */
    int32_t index_base;        /* 0 or 1: base of colptr/rowval/psd_idx/soc_idx    */
    int32_t reserved0;
    /* optional (may be NULL): Lanczos start vectors, one per PSD cone, sides
     * concatenated.  The reference uses normalize!(randn(MersenneTwister(seed), n))
     * (eigsolver.jl:392-411), which only Julia can generate; when NULL the library
     * uses its own counter-based generator (csrc/host_util.hpp, mirrored in
     * oracle/eig.py:start_vector). */
    const double* eig_resid;
    /* optional block-sharded solve (multi-GPU, one process per GPU).
     * When reduce_fn != NULL this problem is ONE SHARD of a block-diagonal model (its PSD
     * blocks/variables and the constraint rows that touch only them).  The library calls
     *     reduce_fn(reduce_ctx, sums, nsum, maxs, nmax)
     * with host arrays; the callee must replace sums[] by the element-wise SUM and maxs[] by
     * the element-wise MAX over all shards (e.g. two RCCL all-reduces) and return 0.  It is
     * called once at start-up and once per PDHG iteration (plus rare extra calls), by every
     * shard in the same order.  Only scalars cross shards: the reference's serial loop over
     * blocks (prox_operators.jl:40) becomes one block set per GPU. */
    void* reduce_ctx;
    int (*reduce_fn)(void* ctx, double* sums, int32_t nsum, double* maxs, int32_t nmax);
    /* optional dense equality block, for models whose A_k are all dense (test/base_randsdp.jl:
     * 4-23: n = 2000, m = 4000 gives 8.0e9 entries = 64 GB, beyond a sparse index type and
     * pointless to index).  When M_dense != NULL it is A as a ROW-MAJOR p x n array of doubles,
     * columns in the caller's variable order; the CSC `A` is then ignored (its pointers may be
     * NULL) while G stays sparse (e.g. the variable bounds of test/moi_randsdp.jl:33-45).
     * M_dense_on_device = 1: a device pointer on options.device_id (e.g. generated there);
     * 0: host memory, uploaded once.  Borrowed and never modified either way (the reference
     * scales aff.A in place, scaling.jl:24; here the column scaling is applied to the
     * vectors).  Restrictions: the variables must already be in solver order (cone variables
     * first, in cone order: psd_idx/soc_idx concatenated = 0,1,2,...), and reduce_fn must be
     * NULL. */
    const double* M_dense;
    int32_t M_dense_on_device;
    int32_t reserved1;
    /* optional, block-sharded solves only: COUPLING ROWS -- rows of [A;G] with entries in the variables
     * of more than one shard.  Every shard lists the same n_coupling rows (its own
     * 0-BASED row numbers, whatever index_base: 0..p-1 equalities, p..p+m-1 inequalities; a shard without entries in such a row still
     * carries it, empty, with the same right-hand side) in the SAME order.  Per iteration the library
     * hands the shard's partial (M x) of these rows to
     *     reduce_vec_fn(reduce_ctx, buf, n_coupling, on_device)
     * which must replace buf[] by the element-wise SUM over all shards (one RCCL all-reduce on a
     * device buffer when reduce_vec_on_device = 1, host memory otherwise) and return 0 after the
     * result is in place.  coupling_owned[k] = 1 on exactly ONE shard per row: that shard counts
     * the row in the scalar sums (b'y, |y+ - y|^2, ...), the others skip it. */
    int64_t n_coupling;
    const int64_t* coupling_rows;
    const int32_t* coupling_owned;
    int (*reduce_vec_fn)(void* ctx, double* buf, int64_t len, int32_t on_device);
    int32_t reduce_vec_on_device;
    int32_t reserved2;
    /* optional, block-sharded solves only: an RCCL communicator (ncclComm_t, one rank per shard, created by the
     * caller -- e.g. ncclCommInitRank in the process that owns the GPU).  When non-NULL the library issues BOTH
     * collectives itself on its own stream -- ncclAllReduce(sum) / ncclAllReduce(max) of the packed scalar record
     * and ncclAllReduce(sum) of the coupling buffer -- with no host callback and no host copy of the vectors;
     * reduce_fn / reduce_vec_fn may then be NULL (they are ignored).  librccl.so is loaded at run time
     * (dlopen): the library does not link it, and a process that never passes a communicator never loads it.
     * Every host wait behind one of these collectives is bounded (PROXSDP_HIP_COLLECTIVE_TIMEOUT_S seconds, default
     * 300): a rank whose peer left the solve aborts its communicator (ncclCommAbort) and returns
     * PROXSDP_E_COMM_ABORTED instead of waiting for ever; a rank that fails for its own reasons AFTER its first collective
     * was enqueued aborts its communicator on the way out and returns PROXSDP_E_COMM_ABORTED as well (a failure before any
     * collective -- argument errors -- leaves the communicator untouched and returns the ordinary error code).
     * ncclCommAbort RELEASES the communicator: after PROXSDP_E_COMM_ABORTED the handle is dead -- do not pass it to
     * proxsdp_hip_rccl_comm_destroy / ncclCommDestroy, do not reuse it. */
    void* nccl_comm;
    int64_t reserved3;
} proxsdp_problem;

/* Options (options.jl:1-132): same names, same defaults (proxsdp_hip_default_options).
 * Booleans are int32 0/1.  Fields that have no use in the reference's src/ are kept
 * for name parity and ignored (convergence_check, min_beta, max_beta, reduce_rank,
 * warm_start_eig, disable_julia_logger, timer_*). */
typedef struct proxsdp_options {
    int64_t struct_size;       /* = sizeof(proxsdp_options), set by default_options */
    /* printing */
    int32_t log_verbose;  int32_t log_freq;
    int32_t timer_verbose; int32_t timer_file; int32_t disable_julia_logger;
    int32_t warn_on_limit; int32_t extended_log; int32_t extended_log2; int32_t log_repeat_header;
    int32_t pad0;
    double  time_limit;
    /* tolerances */
    double tol_gap, tol_feasibility, tol_feasibility_dual, tol_primal, tol_dual, tol_psd, tol_soc;
    int32_t check_dual_feas; int32_t check_dual_feas_freq;
    double  max_obj; int32_t min_iter_max_obj; int32_t pad1;
    /* infeasibility */
    int32_t min_iter_time_infeas; int32_t pad2;
    double infeas_gap_tol, infeas_limit_gap_tol, infeas_stable_gap_tol,
           infeas_feasibility_tol, infeas_stable_feasibility_tol;
    int32_t certificate_search; int32_t pad3;
    double certificate_obj_tol, certificate_fail_tol;
    double min_beta, max_beta, initial_beta;
    /* adaptive steps */
    double initial_adapt_level, adapt_decay; int32_t adapt_window; int32_t pad4;
    /* PDHG */
    int32_t convergence_window; int32_t convergence_check;
    int64_t max_iter; int64_t min_iter; int64_t divergence_min_update;
    int64_t max_iter_lp; int64_t max_iter_conic; int64_t max_iter_local;
    int32_t advanced_initialization; int32_t line_search_flag;
    int32_t max_linsearch_steps; int32_t pad5;
    double delta, initial_theta, linsearch_decay;
    /* spectral decomposition */
    int32_t full_eig_decomp; int32_t max_target_rank_krylov_eigs;
    int32_t min_size_krylov_eigs; int32_t warm_start_eig;
    int32_t rank_increment; int32_t rank_increment_factor;
    int32_t eigsolver; int32_t eigsolver_min_lanczos; int64_t eigsolver_resid_seed;
    double  arpack_tol; int32_t arpack_resid_init; int32_t arpack_reset_resid; int64_t arpack_max_iter;
    int32_t krylovkit_reset_resid; int32_t krylovkit_resid_init;
    double  krylovkit_tol; int32_t krylovkit_max_iter; int32_t krylovkit_eager; int32_t krylovkit_verbose;
    int32_t reduce_rank; int32_t rank_slack; int32_t pad6;
    int64_t full_eig_freq; int64_t full_eig_len;
    /* equilibration (equilibration.jl, pdhg.jl:64-92): off by default; `equilibration` alone
     * survives only if min(M)/max(M) > equilibration_limit, `equilibration_force` always.
     * approx_norm = 0: step size from sigma_max(M) instead of ||M||_F (pdhg.jl:108-119).
     * Neither is available with a dense A or a block-sharded solve. */
    int32_t equilibration; int32_t equilibration_iters;
    double  equilibration_lb, equilibration_ub, equilibration_limit;
    int32_t equilibration_force; int32_t approx_norm;
    /* ---- library-only knobs (no reference counterpart) ---- */
    int32_t device_id;         /* HIP device ordinal, default 0                      */
    int32_t trace_capacity;    /* rows available in result.trace (0 = no trace)      */
    int32_t profile_symv_every;/* >0: bracket every k-th symv launch with HIP events  */
    int32_t support_path;      /* -1 auto (default), 0 dense vector passes, 1 force the
                                * support-aware passes when legal */
    int32_t lanczos_operator;  /* mat-vec inside the Lanczos projection: 0 = packed triangle of the
                                * iterate (what dsymv('U') reads, 8 N bytes), 1 = operator form
                                * x_prev(low rank) + sparse update when available (support path;
                                * DESIGN.md section 4), -1 auto = 1 (default) */
    int32_t initial_target_rank; /* reference: 2, hard-coded (pdhg.jl:19-20); a benchmark may start at the
                                  * rank a config names ("rank ~ sqrt(n)"); capped at the block side */
    int32_t full_eig_lanczos;    /* full_eig! (prox_operators.jl:111-126) needs every POSITIVE eigenpair, not the
                                  * whole spectrum: -1 auto / 1 = when the previous projection of the block had few
                                  * positive eigenvalues, compute them with the Lanczos engine (all pairs down to the
                                  * first eigenvalue <= 0, converged to krylovkit_tol) and fall back to the dense
                                  * eigensolver otherwise; 0 = always the dense eigensolver.  Same projection. */
    int32_t lanczos_cycle_kernel;/* 1 = run a whole Lanczos cycle (operator form) in ONE persistent launch whose <= 32
                                  * workgroups sit on one XCD, keep their rows of the basis in LDS and exchange partial
                                  * dots through that XCD's L2, instead of two launches per step (same arithmetic per
                                  * step; falls back to the step kernels if its bounded spins time out): measured gain
                                  * 1.3x per step, +5 % iterations/s, off in auto.
                                  * 2 (round 6) = ONE WORKGROUP per Lanczos cycle for blocks of side <= 512 (csrc/
                                  * lanczos_block1.hip.hpp: basis in registers, operator and records in LDS, the restart
                                  * rotation in the prologue; the step kernels' arithmetic term by term: BIT-IDENTICAL
                                  * results); applies to the operator form with <= 16 factor columns and to a packed
                                  * triangle that fits in LDS (side <= ~140), Krylov dimension <= 31 -- anything else runs
                                  * the step kernels.  -1 auto = 2 for sides <= 256 (where it is 2.2-2.6x faster per step),
                                  * step kernels beyond; 0 = step kernels always. */
    int32_t lanczos_warm_start;  /* 0 (default): every KrylovKit projection starts from the fixed start vector, as the
                                  * reference does (krylovkit_reset_resid = false).  1: start from the normalised sum of
                                  * the previous projection's Ritz vectors (+ 1e-3 x the fixed vector).  Changes the
                                  * Krylov space, not what is converged (krylovkit_tol).  The Lanczos-served full_eig!
                                  * (full_eig_lanczos: the library's own engine, not KrylovKit's call) warm-starts at 0 and
                                  * 1; -1 = fixed start vector there too. */
    int32_t reconstruct_mfma;    /* rank-r reconstruction V Lam+ V': -1 auto, 0 scalar-FMA kernel, 1 fp64 MFMA
                                  * (v_mfma_f64_16x16x4_f64) kernel */
    int32_t small_block_batch;   /* project the small PSD blocks (never on the Krylov path: side <= min_size_krylov_eigs) in
                                  * ONE launch instead of one dense eigensolver call each: -1 auto = blocks of side 2 by
                                  * the batched Jacobi kernel and blocks of side 3..64 by the one-workgroup, LDS-resident
                                  * sign-function projection (csrc/small_sign.hip.hpp; needs full_eig_sign != 0 and every
                                  * requested tolerance >= 1e-8 -- otherwise auto is ">= 2 blocks of side 2..32, Jacobi");
                                  * 1 = Jacobi for every block of side 2..64; 2 = the sign kernel for every block of
                                  * side 2..64; 0 off */
    int32_t full_eig_sign;       /* full_eig! of a dense block without an eigendecomposition: X+ = (X + X sign(X)) / 2
                                  * with sign(X) from an odd-polynomial iteration of fp64 MFMA products (34 .. 64 products of
                                  * n x n symmetric matrices, see sign_start_row; every |eigenvalue| >= 1e-10 ||X|| is resolved to 1e-15,
                                  * smaller ones contribute an error <= their own size): -1 auto (33 <= n <= 16384 with HBM for the work matrices, and
                                  * every requested tolerance >= 1e-8: below that the 1e-10 floor of this path could
                                  * stall a solve, and the dense eigensolver is used), 1 always, 0 = rocSOLVER dsyevd +
                                  * reconstruction */
    int32_t psd_sign_engine;     /* 1: on the Krylov branch, let the sign-function projection stand in for the Lanczos
                                  * engine when it is measured to be the cheaper way to the SAME matrix (fewer than
                                  * target_rank positive eigenvalues => the truncated projection is the exact one and
                                  * min_eig <= 0; verified after every such projection, redone by Lanczos otherwise;
                                  * on first use and every 64th projection BOTH engines run on the same input and a
                                  * disagreement -- repeated eigenvalues, of which single-vector Lanczos returns one
                                  * copy -- leaves the block to the Lanczos engine for the rest of the solve).
                                  * -1 auto = 0 = off: the reference's engine choice, mat-vec counts as KrylovKit's */
    int32_t full_eig_lanczos_verify; /* full_eig! served by the Lanczos engine (full_eig_lanczos) is the library's own
                                  * algorithm; single-vector Lanczos returns ONE eigenvector per distinct eigenvalue, so a
                                  * repeated positive eigenvalue would silently lose copies.  k > 0: the first such call of a
                                  * block and every k-th after it are ALSO computed by the reference's engine (sign-function /
                                  * dense eigensolver) on the same input and compared (max |difference| <= 1e-8 max |X+|); a
                                  * mismatch hands the block back to the dense engine for the rest of the solve
                                  * (stats.full_eigs_lanczos_checks / _mismatches).  -1 auto = 256, 0 = never */
    double  full_eig_lanczos_posres; /* acceptance of that engine: the first strictly negative Ritz pair must be resolved to
                                  * posres x spectral scale, i.e. a positive eigenvalue the run could have missed is smaller
                                  * than that.  Default 1e-6 -- two orders inside the solver's tolerances (rounds 2-3: 1e-7;
                                  * measured in round 4 on the n = 4000 default solve, with the per-call certificate behind it:
                                  * 1e-7 / 1e-6 / 1e-5 give the same 8651 iterations and objectives equal to 1.4e-14 in
                                  * 12.6 / 11.4 / 9.1 s) */
    int32_t full_eig_lanczos_kdim10; /* its Krylov dimension = max(2 g + 1, g x kdim10 / 10 + 8), default 30 */
    int32_t sign_small_tile_max; /* sign-function projection: 32 x 32 product tiles up to this padded side, 64 x 64
                                  * above (default 3072, measured cross-over ~ 3500) */
    int32_t host_eig_threads;    /* helper threads of the host K x K eigensolver's rotation replay (default 0:
                                  * measured slower on the MI355X host; results are bit-identical) */
    int32_t block_threads;       /* host threads for several eigensolver-sized PSD blocks: worker threads driving per-block
                                  * streams (blocks that are not batched) and the helper threads of a batched run's
                                  * per-block restart logic (they spin only while a batched projection is in progress):
                                  * -1 auto = min(8, blocks), 0 = none (blocks / restart logic in sequence) */
    int32_t host_eig_merge;      /* K x K Rayleigh-quotient eigensolve of the thick-restart Lanczos by SPLIT + RANK-ONE MERGE
                                  * (csrc/host_eig_merge.hpp: the arrow part / first half is decomposed while the GPU still
                                  * runs the cycle -- its coefficients are read back early on a side stream -- and only the
                                  * small tail + one secular-equation merge stay on the critical path; only the Ritz vectors
                                  * actually needed are formed): -1 auto = from krylovdim 64 on, 0 = implicit QL always,
                                  * 1 = from krylovdim 24 on.  Same decomposition to rounding (orthogonality ~1e-14). */
    int32_t block_batch;         /* PSD blocks of EQUAL side whose projections take the Krylov branch (KrylovKit mode,
                                  * packed-triangle operator, krylovdim <= 63) advance their Lanczos recurrences in ONE
                                  * launch per step (grid.z = block; groups of up to 8) instead of one stream + host thread
                                  * per block: -1 auto = 1 = on, 0 = off.  Per block the arithmetic is unchanged. */
    int32_t rocsolver_warmup;    /* 1 = load rocSOLVER's code objects from a background thread at start-up (default 0) */
    int32_t debug_fail_iteration;/* k > 0: FAULT INJECTION for tests -- this process throws inside the PSD projection of
                                  * iteration k (a block-sharded solve must then abort on EVERY shard after that
                                  * iteration's scalar reduce instead of leaving the peers in a collective); 0 = never.
                                  * Honoured only when PROXSDP_HIP_FAULT_INJECTION=1 is set in the environment as well
                                  * (otherwise PROXSDP_E_INVALID): a test switch, not a user option */
    int32_t host_wait_spin;      /* how the solver thread waits for the GPU at its per-cycle / per-iteration read-backs:
                                  * 1 = poll hipStreamQuery (no sleep: the wake-up of a blocked wait costs tens of
                                  * microseconds per synchronisation and leaves the core cold for the K x K eigensolve
                                  * that follows), 0 = hipStreamSynchronize, -1 auto = 1 */
    int32_t sign_start_row;      /* sign-function projection: row of the coefficient table the iteration starts at.  The
                                  * table resolves |eigenvalues| down to 1e-10 x the spectral scale in 19 steps; the
                                  * iterates of a solve rarely come closer to singular than 1e-5, so the iteration starts
                                  * further down the table, TESTS the result (sum t^2 (1 - t^2) over the eigenvalues t of
                                  * the computed sign matrix, from two Frobenius norms) and continues with the rows it
                                  * skipped only when the test fails -- same error bound either way.
                                  * -1 auto: adaptive per block (starts at row 8: 34 products instead of 57; a failure
                                  * sends the next 8 projections two rows down, a second one soon after lowers the row for good); 0 = the full table, no test
                                  * (round-2 behaviour); k > 0 = always start at row k (capped where the test stops
                                  * resolving 1e-10) */
    int32_t general_batch;       /* models without a support set (full-vector passes, sparse M): linesearch candidates,
                                  * residual and gap in one batch of launches and ONE read-back per iteration (up to 3
                                  * candidates evaluated side by side, the first the reference's loop would accept wins;
                                  * per candidate the same arithmetic): -1 auto = 1 = on, 0 = one trial per
                                  * synchronisation (round-2 path) */
    int32_t full_eig_lanczos_certify; /* PER-CALL certificate of a Lanczos-served full_eig! (single-vector Lanczos shows one
                                  * eigenvector per distinct eigenvalue the start vector sees: a repeated positive eigenvalue
                                  * or a deficient start vector would drop positive pairs from X+): after convergence a
                                  * second, independent vector is orthogonalised against the returned Ritz vectors and run
                                  * through m steps of the same recurrence with them locked (Lanczos on the deflated
                                  * operator); its largest Ritz value must be <= full_eig_lanczos_posres x the spectral
                                  * scale, otherwise the dense engine projects that input.  -1 auto = 10 steps, 0 = off,
                                  * m >= 2 = m steps.  The periodic dense check (full_eig_lanczos_verify) stays behind it. */
    int32_t host_merge_threads;  /* helper threads for the rank-one merge of the K x K eigensolve (host_eig_merge): the secular
                                  * roots, the Gu-Eisenstat weights and the eigenvector columns are independent per root /
                                  * column and are handed out in chunks; the helpers spin only while a projection with
                                  * krylovdim >= 64 is in progress.  Results are bit-identical to the serial merge.
                                  * -1 auto = 3, 0 = none */
    int32_t reserved_i[1];       /* zero */
    double  full_eig_lanczos_tol;/* convergence of the POSITIVE Ritz pairs of that engine: residual <= tol x the spectral
                                  * scale; 0 (default) = krylovkit_tol as an absolute residual (KrylovKit's rule) */
    double  reserved_d[1];       /* zero */
    /* ---- ABI 9 (appended) ---- */
    int32_t equilibration_reference_aliasing; /* equilibrate! (equilibration.jl:1-72) builds `E = Diagonal(u)`, `D = Diagonal(v)`
                                  * WITHOUT copying (:16-17), so `E.diag .= exp.(u)` (:25-26) overwrites u with exp(u) (v with
                                  * exp(v)) at the top of every iteration and the gradient steps start from there.  1 (default):
                                  * that arithmetic, line by line -- what the reference computes; 0: the iteration the code
                                  * evidently intends (u, v kept; E = exp(u), D = exp(v)): rounds 1-4 behaviour */
    int32_t reserved_i3[2];      /* zero (two designs of round 5 -- a device-side thick restart and a warm-started block filter for the
                                  * Lanczos-served full_eig! -- were measured out before they got an option: DESIGN.md section 9) */
    int32_t block_batch_groups;  /* batched multi-block Lanczos (block_batch): equal-side blocks are split into this many groups that
                                  * run CONCURRENTLY (own stream + host thread each): one group's restart logic on the host overlaps
                                  * the other groups' cycles on the GPU.  -1 auto = 1 = groups run ONE AFTER THE OTHER (rounds 3-4 behaviour; also when more than 8 equal-side
                                  * blocks force several groups), k >= 2 = k
                                  * groups (from 4 blocks on, when the block worker pool is up: block_threads != 0).  Per block
                                  * nothing changes.  MEASURED NEGATIVE on MIMO 8 x 513 (same session: 313 / 303 / 217 it/s with
                                  * 1 / 2 / 4 groups): a cycle is bound by its chain of dependent launches on the GPU (135 us of
                                  * the 200), kernels of different streams do not overlap enough to pay for twice the launches --
                                  * kept as an opt-in for models with many more blocks */
    int32_t reserved_i2[8];      /* zero */
    double  full_eig_lanczos_warm_pow; /* start vector of a Lanczos-served full_eig! (positive-part run, the library's own engine): the
                                  * previous projection's Ritz vectors are summed with weights (lam_0 / lam_c)^p -- the pairs with the SMALL
                                  * eigenvalues are the ones such a run converges last, so they get the larger share.  Default 1.0
                                  * (measured, default-options solves: n = 4000: 687 201 -> 666 808 mat-vecs, 11.3-11.4 -> 10.9-11.1 s; n = 2000:
                                  * 323 250 -> 316 127; n = 3000: 492 988 -> 486 157; the same iteration counts and objectives; every p in
                                  * 0.125 .. 3 tried was better than 0); 0 = the plain sum of rounds 3-4 */
    double  reserved_d2[3];      /* zero */
} proxsdp_options;

#define PROXSDP_TRACE_COLS 14
/* trace row: iter, prim_obj, dual_obj, gap, feas, prim_res, dual_res, primal_step,
 *            beta, theta, target_rank(block 0), linesearch trials,
 *            elapsed (s since the loop started, host clock after the iteration's last
 *            stream synchronisation), Lanczos mat-vecs of this iteration */

/* counters and timers (the reference's TimerOutputs sections) */
typedef struct proxsdp_stats {
    int64_t lanczos_matvecs;     /* symmetric mat-vecs inside Lanczos (all blocks)   */
    int64_t lanczos_restarts;
    int64_t lanczos_calls;
    int64_t full_eigs;           /* full_eig! calls                                  */
    int64_t krylov_fallbacks;    /* Krylov not converged -> full_eig! fallback       */
    int64_t linesearch_trials;
    int64_t symv_launches;       /* == lanczos_matvecs                               */
    int64_t symv_profiled;       /* launches bracketed by events                     */
    double  symv_profiled_ms;    /* sum of their durations                           */
    double  symv_bytes;          /* algorithmic bytes of all symv launches (8*N + 16*n each) */
    double  algorithmic_bytes;   /* B_iter summed over iterations        */
    double  init_time;           /* s: preprocess + upload ("Init")                  */
    double  loop_time;           /* s: the PDHG loop ("CP loop")                     */
    double  exit_time;           /* s: cache_solution                                */
    double  t_primal, t_psd, t_linesearch, t_residual;   /* s, host wall incl. syncs: the reference's TimerOutputs sections (pdhg.jl:150-164).
                                  * t_primal = primal_step! incl. the projection (t_psd); on the fused paths the residual / gap REDUCTIONS
                                  * ride in the linesearch candidates' batch and its one read-back (counted in t_linesearch), t_residual is
                                  * what is left of compute_residual! / compute_gap!: the host scalars */
    int64_t dense_passes;        /* passes over a dense A (A x or batched A' y), 8*p*n bytes each */
    double  dense_ms;            /* their summed durations (HIP events on the solve stream)    */
    int64_t fop_projections;     /* projections whose Lanczos mat-vecs ran in operator form     */
    int64_t exit_matvecs;        /* mat-vecs of the exit path's lambda_min(dual cone) Lanczos   */
    double  host_eig_time;       /* s: K x K Rayleigh-quotient eigensolves done on the HOST     */
    int64_t host_eigs;           /* how many of them                                            */
    int64_t device_eigs;         /* unused (0): a device-side K x K eigensolver was measured and dropped, DESIGN.md section 9 */
    int64_t batched_small_eigs;  /* small-block (n <= 32) projections done by the batched Jacobi kernel */
    int64_t mfma_reconstructions;/* reconstructions that took the MFMA (v_mfma_f64_16x16x4) SYRK */
    int64_t orth_profiled;       /* k_lz_orth launches bracketed by events (profile_symv_every)  */
    double  orth_profiled_ms;    /* sum of their durations                                       */
    double  full_eig_solver_ms;  /* full_eig!: dense eigensolver time (events; profile_symv_every > 0) */
    double  full_eig_recon_ms;   /* full_eig!: reconstruction kernel time (events)               */
    int64_t cycle_launches;      /* Lanczos cycles run by the persistent LDS-resident kernel     */
    int64_t full_eigs_lanczos;   /* full_eig! calls served by the Lanczos engine (all positive pairs) */
    int64_t cycle_steps;         /* Lanczos steps run inside those launches                       */
    double  cycle_ms;            /* their summed kernel time (events; profile_symv_every > 0)     */
    int64_t warm_starts;         /* projections started from the previous Ritz vectors (lanczos_warm_start) */
    int64_t full_eigs_sign;      /* full_eig! calls served by the sign-function projection (full_eig_sign) */
    int64_t sign_products;       /* symmetric n x n matrix products (fp64 MFMA) those calls took */
    int64_t sign_engine_projections; /* Krylov-branch projections served by the sign function (psd_sign_engine) */
    int64_t sign_engine_rejected;    /* ... computed but discarded: truncation was active, Lanczos redid them */
    int64_t sign_engine_checks;      /* verification rounds (both engines on the same input; first use, then every 64th) */
    int64_t sign_engine_mismatches;  /* ... that disagreed (repeated eigenvalues): the block stays with Lanczos */
    int64_t full_eigs_lanczos_checks;     /* Lanczos-served full_eig! calls verified against the dense engine */
    int64_t full_eigs_lanczos_mismatches; /* ... that disagreed: the block went back to the dense engine */
    int64_t batched_block_steps;     /* Lanczos steps launched for several blocks at once (block_batch) */
    int64_t rccl_reductions;         /* collectives issued by the library itself on its own stream (nccl_comm) */
    int64_t batched_profiled_blocks; /* block mat-vecs inside the event-bracketed batched launches (symv_profiled
                                      * counts launches; bytes of those launches = this x (8 N + 16 n)) */
    int64_t host_eig_merges;         /* K x K eigensolves done by split + rank-one merge (host_eig_merge); host_eig_time
                                      * then counts only their critical-path part */
    double  host_eig_overlap_time;   /* s: the part of those eigensolves done while the GPU was running the cycle */
    int64_t sign_short_pass;         /* sign-function projections whose shortened schedule passed its test (sign_start_row) */
    int64_t sign_short_fail;         /* ... that failed it and continued with the skipped rows */
    int64_t full_eigs_lanczos_certified;  /* Lanczos-served full_eig! calls whose certificate run passed (full_eig_lanczos_certify) */
    int64_t full_eigs_lanczos_cert_failed;/* ... whose certificate found a positive direction outside the returned pairs: the
                                           * dense engine projected that input instead */
    int64_t cert_matvecs;                 /* mat-vecs of those certificate runs (included in lanczos_matvecs) */
    /* ---- ABI 9 (appended) ---- */
    int64_t dense_truncated_projections;  /* Krylov-branch projections whose krylovdim = max(2 target_rank + 1, eigsolver_min_lanczos)
                                           * exceeds the step kernels' 255 columns: served by the dense eigensolver (top target_rank
                                           * pairs of dsyevd, the same truncated projection and min_eig); status_string says so */
    int64_t reserved_s[7];                /* zero */
} proxsdp_stats;

/* Result (structs.jl:60-81).  Arrays are caller-allocated with the stated
 * lengths (NULL = not wanted); values are unscaled and in USER variable order
 * (pdhg.jl:768-769).  objval/dual_objval are the minimisation objective; the
 * sign/constant fix-up of MOI_wrapper.jl:336-337 stays with the caller. */
typedef struct proxsdp_result {
    int32_t status;                 /* PROXSDP_STATUS_*                              */
    int32_t certificate_found;
    int32_t primal_feasible_user_tol;
    int32_t dual_feasible_user_tol;
    int32_t result_count;
    int32_t final_rank;
    int64_t iter;
    double  primal_residual;        /* carries equa_feasibility (sic, pdhg.jl:774)   */
    double  dual_residual;          /* carries ineq_feasibility (sic, pdhg.jl:775)   */
    double  objval, dual_objval, gap, time;
    double  dual_feasibility;       /* value behind dual_feasible_user_tol           */
    double* primal;                 /* n */
    double* dual_cone;              /* n */
    double* dual_eq;                /* p */
    double* dual_in;                /* m */
    double* slack_eq;               /* p */
    double* slack_in;               /* m */
    double* trace;                  /* trace_capacity x PROXSDP_TRACE_COLS, row-major */
    int64_t trace_rows;             /* rows written                                  */
    char    status_string[256];
    proxsdp_stats stats;
} proxsdp_result;

/* State of a solve at an ITERATION BOUNDARY (after iteration `iteration`'s control logic, pdhg.jl:246-483, before
 * primal_step! of the next one): everything chambolle_pock carries from one iteration to the next.  At that point
 * x_old = x, y_old = y, Mty_old = Mty, Mx_old = Mx (residuals.jl:65-68), so four vectors suffice.  Vectors are in the
 * solver's INTERNAL order and scaling (preprocess! + norm_scaling, scaling.jl:2-49: cone variables first, PSD
 * off-diagonals carrying sqrt(2)).  Test / measurement seam (oracle resume-from-state, steady-window baselines): the
 * eigensolver workspaces are not part of it -- the reference's KrylovKit start vector is fixed
 * (krylovkit_reset_resid = false) and the library's own caches (previous Ritz factors, engine statistics) are rebuilt.
 * Not available for block-sharded solves or with equilibration (PROXSDP_E_UNSUPP); an iteration inside a certificate
 * search (pdhg.jl:639-676) is not captured (ints[3] stays 0, the solve goes on). */
#define PROXSDP_STATE_NHIST 7
typedef struct proxsdp_state {
    int64_t struct_size;       /* = sizeof(proxsdp_state) */
    int64_t iteration;         /* capture: IN  the iteration to capture after (>= 1); resume: the iteration the state belongs to */
    int64_t n, Q, n_psd;       /* array lengths; must equal the problem's n, p + m, n_psd */
    int64_t hist_len;          /* = 2 * options.convergence_window */
    double* x;                 /* n */
    double* y;                 /* Q */
    double* Mty;               /* n */
    double* Mx;                /* Q */
    int64_t* target_rank;      /* n_psd (Params.target_rank, structs.jl:159-192) */
    int64_t* current_rank;     /* n_psd */
    double* min_eig;           /* n_psd */
    double* hist;              /* PROXSDP_STATE_NHIST x hist_len, raw circular storage (structs.jl:2-30: entry i at slot
                                * (i-1) mod hist_len): dual_gap | prim_obj | dual_obj | feasibility | primal_residual |
                                * dual_residual | comb_residual */
    double scal[16];           /* 0 primal_step 1 primal_step_old 2 dual_step 3 beta 4 theta 5 adapt_level
                                * 6 equa_feasibility 7 ineq_feasibility 8 dual_feasibility; rest zero */
    int64_t ints[8];           /* 0 rank_update 1 update_cont 2 ada_count (pdhg.jl:306-332) 3 OUT captured (1 = written);
                                * rest zero */
} proxsdp_state;

/* ------------------------------------------------------------------ drop-in */
int  proxsdp_hip_abi_version(void);
void proxsdp_hip_default_options(proxsdp_options* opt);          /* options.jl defaults */
/* set an option by its reference name (RawOptimizerAttribute semantics,
 * MOI_wrapper.jl:84-93): unknown name -> PROXSDP_E_INVALID */
int  proxsdp_hip_set_option(proxsdp_options* opt, const char* name, double value);
int  proxsdp_hip_get_option(const proxsdp_options* opt, const char* name, double* value);
/* replaces chambolle_pock(aff, con, opt) -- MOI_wrapper.jl:310, pdhg.jl:1-530 */
int  proxsdp_hip_solve(const proxsdp_problem* prob, const proxsdp_options* opt,
                       proxsdp_result* res);
/* the same solve with the state seam: `resume` (may be NULL) = continue from this state with iteration
 * resume->iteration + 1 instead of starting at pdhg.jl:54-142's initial point; `capture` (may be NULL) = write the state
 * after iteration capture->iteration into the caller-allocated arrays (capture->ints[3] = 1 when that iteration was
 * reached).  proxsdp_hip_solve(p, o, r) == proxsdp_hip_solve_ex(p, o, r, NULL, NULL). */
int  proxsdp_hip_solve_ex(const proxsdp_problem* prob, const proxsdp_options* opt, proxsdp_result* res,
                          const proxsdp_state* resume, proxsdp_state* capture);
const char* proxsdp_hip_last_error(void);
int  proxsdp_hip_device_count(void);          /* <0: PROXSDP_E_HIP */

/* ------------------------------------------- RCCL communicator helpers (block-sharded solves)
 * The library loads librccl at run time (dlopen; it does not link it).  One rank calls _unique_id and
 * ships the 128 bytes to the others by any means (MPI, torch.distributed, a file); every rank then calls
 * _comm_init on the GPU it owns and passes the handle as proxsdp_problem.nccl_comm.  A caller that already
 * has an ncclComm_t (created through the same librccl) passes it directly instead. */
int  proxsdp_hip_rccl_available(void);                      /* 1 = librccl loaded and usable */
int  proxsdp_hip_rccl_unique_id(void* id128);               /* NCCL_UNIQUE_ID_BYTES = 128 */
int  proxsdp_hip_rccl_comm_init(int32_t nranks, const void* id128, int32_t rank, int32_t device_id, void** comm);
int  proxsdp_hip_rccl_comm_destroy(void* comm);

/* ------------------------------------------- kernel-level test entry points
 * (not part of the drop-in; each is pinned against the oracle in tests/).
 * All pointers are host pointers; data is copied to the device, the kernel(s)
 * run, results are copied back. */

/* psd_projection! of ONE block given in packed svec form
 * (prox_operators.jl:33-66 with psd_vec_to_square/psd_square_to_vec :1-31).
 * mode 0: Lanczos path (krylovkit_eig!, :89-109) with nev = target_rank,
 *         falling back to full_eig! when not converged;
 * mode 1: full_eig! (:111-126) through the dense eigensolver;
 * mode 2: full_eig! served by the Lanczos engine (every positive eigenpair), target_rank = estimate
 *         of the number of positive eigenvalues; *out_fell_back = 1 if the dense solver had to run;
 * mode 3: full_eig! by the batched small-block Jacobi kernel (2 <= n <= 64); *out_converged = #{lambda > 0}.
 * mode 5: full_eig! by the one-workgroup, LDS-resident sign projection of small blocks (csrc/small_sign.hip.hpp; 2 <= n <= 64);
 *         *out_converged = #{lambda > 0} as in mode 3.
 * mode 4: full_eig! by the sign-function projection (options.full_eig_sign = 1): fp64 MFMA products, no eigenpairs;
 *         *out_rank = #{lambda > 0};
 * resid: start vector (n) or NULL.  out_*: rank (current_rank), min_eig,
 * nmatvec, converged eigenpairs, fell_back flag. */
int proxsdp_hip_psd_project(const double* packed_in, int64_t n, int32_t target_rank,
                            int32_t mode, const proxsdp_options* opt, const double* resid,
                            double* packed_out, int32_t* out_rank, double* out_min_eig,
                            int64_t* out_nmatvec, int32_t* out_converged, int32_t* out_fell_back);

/* Lanczos partial eigendecomposition of smat(packed) (eigsolver.jl:798-823):
 * vals (capacity cap), vecs (n x cap column-major); returns count in *out_count,
 * info.converged in *out_converged. */
int proxsdp_hip_eigsolve(const double* packed, int64_t n, int32_t nev,
                         const proxsdp_options* opt, const double* resid, int32_t cap,
                         double* vals, double* vecs, int32_t* out_count,
                         int32_t* out_converged, int64_t* out_nmatvec, int32_t* out_numiter);

/* y = smat(packed) * v, the Lanczos operator (dsymv('U') in the reference,
 * eigsolver.jl:678); repeat>1 re-launches for timing, *ms = mean kernel time. */
int proxsdp_hip_symv_packed(const double* packed, int64_t n, const double* v, double* y,
                            int32_t repeat, double* ms);

/* packed_out = svec(sum_k lambda_k z_k z_k') over lambda_k > 0
 * (fill! + rank-1 dgemm loop, prox_operators.jl:92-106, then :17-31) */
int proxsdp_hip_reconstruct(const double* Z, const double* lambda, int64_t n, int32_t r,
                            double* packed_out, int32_t repeat, double* ms);
/* the same with the kernel chosen explicitly: mfma = 0 scalar-FMA kernel (LDS-staged), 1 = fp64 MFMA
 * SYRK (v_mfma_f64_16x16x4_f64), -1 = the library's choice (options.reconstruct_mfma auto) */
int proxsdp_hip_reconstruct_kernel(const double* Z, const double* lambda, int64_t n, int32_t r, int32_t mfma,
                                   double* packed_out, int32_t repeat, double* ms);

/* full_eig! (prox_operators.jl:111-126) of one packed block, timed: sign = 0 rocSOLVER dsyevd + reconstruction,
 * 1 = the sign-function projection (options.full_eig_sign) with the default options.sign_start_row, 100 + k = the
 * same with sign_start_row = k (100: the full table); ms = wall time per call over `repeat` calls on
 * device-resident data, out_rank = #{lambda > tol_psd} (dsyevd) / #{lambda > 0} (sign), out_products = MFMA
 * products per call (averaged over the warm-up call and the timed ones) */
int proxsdp_hip_full_eig_kernel(const double* packed_in, int64_t n, int32_t sign, double* packed_out,
                                int32_t repeat, double* ms, int32_t* out_rank, int64_t* out_products);

/* Mx = M x (pdhg.jl:634) and Mty = M' y (pdhg.jl:556) for M given as CSC */
int proxsdp_hip_spmv(const proxsdp_csc* M, int32_t index_base, int32_t transpose,
                     const double* in, double* out);

/* x_out = x - tau (Mty + c)   (primal_step!, pdhg.jl:622; fused AXPY kernel, FP contraction off) */
int proxsdp_hip_primal_update(const double* x, const double* Mty, const double* c, double tau, int64_t n,
                              double* x_out);

/* one linesearch trial in y-space (pdhg.jl:547-553 + box_projection!, prox_operators.jl:160-170):
 *   ybar = y + bt ((1+theta) Mx - theta Mx_old);  y_out = ybar - bt * box(ybar / bt)
 * with box = b on the first p rows and min(., h) on the rest (bh = [b;h]);
 * ynorm2 = |y_out - y|^2 (the right-hand side of the acceptance test, :566) */
int proxsdp_hip_dual_trial(const double* y, const double* Mx, const double* Mx_old, const double* bh,
                           int64_t p, int64_t Q, double bt, double theta, double* y_out, double* ynorm2);

/* compute_residual! + compute_gap! (residuals.jl:2-71), the reductions only.  out[9]:
 *   0 max|(x - tau Mty) - (x_old - tau Mty_old)|   1 max|x_old - tau Mty_old|   2 c'x
 *   3 max|(y - sigma Mx) - (y_old - sigma Mx_old)| 4 max|y_old - sigma Mx_old|
 *   5 max|Mx - b| (equalities)  6 max(0, max(Mx - h)) (inequalities)  7 b'y_eq  8 h'y_in */
int proxsdp_hip_residuals(const double* x, const double* x_old, const double* Mty, const double* Mty_old,
                          const double* c, double tau, int64_t n,
                          const double* y, const double* y_old, const double* Mx, const double* Mx_old,
                          const double* bh, int64_t p, int64_t Q, double sigma, double* out);

/* ------------------------------------------- host-only helpers (no GPU needed;
 * exercised by the CPU test-suite) */
/* eigen-decomposition of a small dense symmetric matrix (column-major k x k,
 * overwritten by eigenvectors; d ascending) -- the K x K Rayleigh quotient of
 * the thick-restart Lanczos */
int proxsdp_host_symeig(int32_t k, double* a, double* d);
/* the same with the helper threads of the eigenvector accumulation chosen explicitly: 0 = serial,
 * t > 0 = at least t helper threads,
 * -1 = the library's choice (helpers from k >= 96 when PROXSDP_HIP_EIG_THREADS > 0; default 0: they were
 * measured slower on the MI355X host).  Both give bit-identical results. */
int proxsdp_host_symeig_threads(int32_t k, double* a, double* d, int32_t threads);
/* eigen-decomposition of a thick-restarted Rayleigh quotient
 *   T = [diag(D) f 0; f' al[m] be[m] e1'; 0 be[m] e1 tridiag(al[m+1..], be[m+1..])]   (K x K)
 * through the two-phase path the Lanczos driver uses (arrow part reduced first, QL afterwards);
 * U (K x K column-major) eigenvectors, d ascending eigenvalues.  al, be are indexed by step (entries
 * below m unused). */
int proxsdp_host_symeig_arrow(int32_t K, int32_t m, const double* D, const double* f,
                              const double* al, const double* be, double* U, double* d);
/* the same matrix decomposed by a split at k1 and ONE rank-one merge (one level of Cuppen's divide and conquer with
 * Gu-Eisenstat eigenvectors, csrc/host_eig_merge.hpp): what the Lanczos driver uses from krylovdim 64 on, with the first
 * part solved while the GPU still runs the cycle.  k1 = m + 1 when m > 0 (the arrow with its hub), 1 <= k1 < K when
 * m = 0.  info[3] (optional): non-deflated poles, deflated poles, most secular iterations of a root. */
int proxsdp_host_symeig_split(int32_t K, int32_t m, int32_t k1, const double* D, const double* f,
                              const double* al, const double* be, double* U, double* d, int32_t* info);
/* the same with `threads` helper threads for the merge's independent pieces (options.host_merge_threads): bit-identical */
int proxsdp_host_symeig_split_threads(int32_t K, int32_t m, int32_t k1, const double* D, const double* f,
                                      const double* al, const double* be, int32_t threads, double* U, double* d, int32_t* info);
/* the library's Lanczos start vector (init 3/2/1 as options.jl:98-103) */
int proxsdp_host_start_vector(int64_t n, int64_t seed, int32_t init, double* out);
/* preprocess!/norm_scaling (scaling.jl): returns the variable order, the
 * inverse permutation and the scaled c; for layout tests */
int proxsdp_host_preprocess(const proxsdp_problem* prob, int64_t* order, int64_t* var_ordering,
                            double* c_scaled, double* frobenius_norm_M);

#ifdef __cplusplus
}
#endif
#endif /* PROXSDP_HIP_H */
