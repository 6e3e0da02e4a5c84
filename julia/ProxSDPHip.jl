# ProxSDPHip.jl -- the reference-side binding of libproxsdp_hip.so.
#
# UNTESTED IN THIS REPOSITORY'S ENVIRONMENT: neither the build container nor the GPU box has a
# Julia toolchain (no `julia` binary, no package depot), so this file is shipped as the source a
# ProxSDP maintainer would add.  The C signatures below are exactly include/proxsdp_hip.h; the
# same ABI is exercised by the Python ctypes binding (proxsdp.jl_amd/binding.py), which is what
# the test-suite runs.
#
# How it plugs in.  ProxSDP reaches its solver through ONE call,
#     sol = @timeit "Main" chambolle_pock(aff, con, options)        # src/MOI_wrapper.jl:310
# inside `_optimize!` (src/MOI_wrapper.jl:220-342).  Replace that line by
#     sol = ProxSDPHip.chambolle_pock_hip(aff, con, options)
# and everything above it -- `Optimizer <: MOI.AbstractOptimizer`, copy_to, the attribute
# getters at :362-530, JuMP models -- stays unchanged.  `aff::AffineSets`, `con::ConicSets`,
# `options::Options` and the returned `Result` are the reference's own types
# (src/structs.jl:32-81, src/options.jl).
module ProxSDPHip

import SparseArrays

const libproxsdp_hip = get(ENV, "PROXSDP_HIP_LIB", "libproxsdp_hip.so")

struct CSC                       # proxsdp_csc
    nrows::Int64
    ncols::Int64
    colptr::Ptr{Int64}
    rowval::Ptr{Int64}
    nzval::Ptr{Float64}
end

struct Problem                   # proxsdp_problem
    n::Int64
    p::Int64
    m::Int64
    A::CSC
    G::CSC
    b::Ptr{Float64}
    h::Ptr{Float64}
    c::Ptr{Float64}
    n_psd::Int64
    psd_ptr::Ptr{Int64}
    psd_idx::Ptr{Int64}
    n_soc::Int64
    soc_ptr::Ptr{Int64}
    soc_idx::Ptr{Int64}
    index_base::Int32
    reserved0::Int32
    eig_resid::Ptr{Float64}
    reduce_ctx::Ptr{Cvoid}       # block-sharded solves only (INTEGRATION.md); C_NULL otherwise
    reduce_fn::Ptr{Cvoid}
    M_dense::Ptr{Float64}        # optional dense A (row-major p x n); C_NULL = use the CSC A
    M_dense_on_device::Int32
    reserved1::Int32
    n_coupling::Int64            # coupling rows of a block-sharded solve (0 otherwise)
    coupling_rows::Ptr{Int64}
    coupling_owned::Ptr{Int32}
    reduce_vec_fn::Ptr{Cvoid}
    reduce_vec_on_device::Int32
    reserved2::Int32
    nccl_comm::Ptr{Cvoid}        # RCCL communicator of a block-sharded solve (C_NULL otherwise)
    reserved3::Int64
end

struct Stats                     # proxsdp_stats
    lanczos_matvecs::Int64
    lanczos_restarts::Int64
    lanczos_calls::Int64
    full_eigs::Int64
    krylov_fallbacks::Int64
    linesearch_trials::Int64
    symv_launches::Int64
    symv_profiled::Int64
    symv_profiled_ms::Float64
    symv_bytes::Float64
    algorithmic_bytes::Float64
    init_time::Float64
    loop_time::Float64
    exit_time::Float64
    t_primal::Float64
    t_psd::Float64
    t_linesearch::Float64
    t_residual::Float64
    dense_passes::Int64
    dense_ms::Float64
    fop_projections::Int64
    exit_matvecs::Int64
    host_eig_time::Float64
    host_eigs::Int64
    device_eigs::Int64
    batched_small_eigs::Int64
    mfma_reconstructions::Int64
    orth_profiled::Int64
    orth_profiled_ms::Float64
    full_eig_solver_ms::Float64
    full_eig_recon_ms::Float64
    cycle_launches::Int64
    full_eigs_lanczos::Int64
    cycle_steps::Int64
    cycle_ms::Float64
    warm_starts::Int64
    full_eigs_sign::Int64
    sign_products::Int64
    sign_engine_projections::Int64
    sign_engine_rejected::Int64
    sign_engine_checks::Int64
    sign_engine_mismatches::Int64
    full_eigs_lanczos_checks::Int64
    full_eigs_lanczos_mismatches::Int64
    batched_block_steps::Int64
    rccl_reductions::Int64
    batched_profiled_blocks::Int64
    host_eig_merges::Int64
    host_eig_overlap_time::Float64
    sign_short_pass::Int64
    sign_short_fail::Int64
    full_eigs_lanczos_certified::Int64
    full_eigs_lanczos_cert_failed::Int64
    cert_matvecs::Int64
    dense_truncated_projections::Int64
    reserved_s::NTuple{7,Int64}
end

mutable struct CResult           # proxsdp_result
    status::Int32
    certificate_found::Int32
    primal_feasible_user_tol::Int32
    dual_feasible_user_tol::Int32
    result_count::Int32
    final_rank::Int32
    iter::Int64
    primal_residual::Float64
    dual_residual::Float64
    objval::Float64
    dual_objval::Float64
    gap::Float64
    time::Float64
    dual_feasibility::Float64
    primal::Ptr{Float64}
    dual_cone::Ptr{Float64}
    dual_eq::Ptr{Float64}
    dual_in::Ptr{Float64}
    slack_eq::Ptr{Float64}
    slack_in::Ptr{Float64}
    trace::Ptr{Float64}
    trace_rows::Int64
    status_string::NTuple{256,UInt8}
    stats::Stats
    CResult() = new()
end

# proxsdp_options is filled by name, exactly like MOI.RawOptimizerAttribute does for the
# reference (src/MOI_wrapper.jl:84-93): an opaque, suitably large and aligned buffer plus
# proxsdp_hip_default_options / proxsdp_hip_set_option keeps this file independent of the C
# struct layout.
const OPTIONS_BYTES = 1024

function _options_buffer(options)
    buf = zeros(UInt64, OPTIONS_BYTES ÷ 8)
    ccall((:proxsdp_hip_default_options, libproxsdp_hip), Cvoid, (Ptr{UInt64},), buf)
    for name in fieldnames(typeof(options))
        v = getfield(options, name)
        v isa Union{Bool,Integer,AbstractFloat} || continue
        rc = ccall((:proxsdp_hip_set_option, libproxsdp_hip), Cint,
                   (Ptr{UInt64}, Cstring, Cdouble), buf, String(name), Float64(v))
        rc == 0 || error("No parameter matching $(name)")   # same text as MOI_wrapper.jl:90
    end
    return buf
end

_csc(M::SparseArrays.SparseMatrixCSC{Float64,Int64}) =
    CSC(size(M, 1), size(M, 2), pointer(M.colptr), pointer(M.rowval), pointer(M.nzval))

"""
    chambolle_pock_hip(aff, con, options) -> Result

Drop-in for `chambolle_pock(aff, con, options)` (src/pdhg.jl:1-530).  `aff`/`con` are only read;
the library works on private copies (the reference mutates `aff`: src/scaling.jl:24,
src/pdhg.jl:647-663).
"""
function chambolle_pock_hip(aff, con, options; ResultType = Main.ProxSDP.Result)
    psd_ptr = Int64[0]; psd_idx = Int64[]
    for s in con.sdpcone
        append!(psd_idx, s.vec_i); push!(psd_ptr, length(psd_idx))
    end
    soc_ptr = Int64[0]; soc_idx = Int64[]
    for s in con.socone
        append!(soc_idx, s.idx); push!(soc_ptr, length(soc_idx))
    end
    n, p, m = aff.n, aff.p, aff.m
    primal = zeros(n); dual_cone = zeros(n)
    dual_eq = zeros(p); dual_in = zeros(m); slack_eq = zeros(p); slack_in = zeros(m)
    opt = _options_buffer(options)
    res = CResult()
    A, G = aff.A, aff.G
    GC.@preserve A G aff psd_ptr psd_idx soc_ptr soc_idx primal dual_cone dual_eq dual_in slack_eq slack_in opt begin
        prob = Problem(n, p, m, _csc(A), _csc(G), pointer(aff.b), pointer(aff.h), pointer(aff.c),
                       length(con.sdpcone), pointer(psd_ptr), pointer(psd_idx),
                       length(con.socone), pointer(soc_ptr), pointer(soc_idx),
                       Int32(1), Int32(0), Ptr{Float64}(C_NULL),         # Julia indices are 1-based
                       C_NULL, C_NULL, Ptr{Float64}(C_NULL), Int32(0), Int32(0),
                       0, Ptr{Int64}(C_NULL), Ptr{Int32}(C_NULL), C_NULL, Int32(0), Int32(0),   # no coupling rows
                       C_NULL, 0)                                                                # no RCCL communicator
        res.primal = pointer(primal); res.dual_cone = pointer(dual_cone)
        res.dual_eq = pointer(dual_eq); res.dual_in = pointer(dual_in)
        res.slack_eq = pointer(slack_eq); res.slack_in = pointer(slack_in)
        res.trace = Ptr{Float64}(C_NULL); res.trace_rows = 0
        rc = ccall((:proxsdp_hip_solve, libproxsdp_hip), Cint,
                   (Ref{Problem}, Ptr{UInt64}, Ref{CResult}), prob, opt, res)
        if rc != 0
            msg = unsafe_string(ccall((:proxsdp_hip_last_error, libproxsdp_hip), Cstring, ()))
            error("libproxsdp_hip: error $(rc): $(msg)")
        end
    end
    status_string = String(UInt8[c for c in res.status_string if c != 0x00])
    return ResultType(
        res.status, status_string, primal, dual_cone, dual_eq, dual_in, slack_eq, slack_in,
        res.primal_residual, res.dual_residual, res.objval, res.dual_objval, res.gap, res.time,
        res.iter, res.final_rank, res.primal_feasible_user_tol != 0,
        res.dual_feasible_user_tol != 0, res.certificate_found != 0, res.result_count)
end

end # module
