// C ABI of libproxsdp_hip.so (include/proxsdp_hip.h).  Nothing unwinds across
// this boundary: every entry point catches, records the message in a
// thread-local buffer and returns a negative PROXSDP_E_* code.  There is no CPU
// fallback: without a HIP device the compute entry points fail with PROXSDP_E_HIP.
#include <memory>
#include <new>
#include <string>

#include "pdhg_loop.hip.hpp"

namespace {
thread_local std::string g_last_error;

template <typename F>
int guarded(F&& f) {
    try {
        g_last_error.clear();
        return f();
    } catch (const std::bad_alloc&) {
        g_last_error = "out of memory";
        return PROXSDP_E_NOMEM;
    } catch (const proxsdp::HipError& e) {
        g_last_error = e.what();
        return PROXSDP_E_HIP;
    } catch (const std::invalid_argument& e) {
        g_last_error = e.what();
        return PROXSDP_E_INVALID;
    } catch (const std::domain_error& e) {
        g_last_error = e.what();
        return PROXSDP_E_UNSUPP;
    } catch (const std::exception& e) {
        g_last_error = e.what();
        return PROXSDP_E_INTERNAL;
    } catch (...) {
        g_last_error = "unknown error";
        return PROXSDP_E_INTERNAL;
    }
}

// engine-only solver for the kernel-level entry points: one block of side n
struct Engine {
    proxsdp_result dummy{};
    proxsdp::Solver S;
    Engine(const proxsdp_options* opt, int64_t n, int max_nev)
        : S(fix(opt), dummy) {
        if (n < 1 || n > 46340) throw std::invalid_argument("n out of range");
        S.setup_device();
        S.eig.resize(1);
        S.alloc_eigwork(S.eig[0], (int)n, std::min<int>(std::max(max_nev, 2), (int)n));
    }
    static proxsdp_options fix(const proxsdp_options* o) {
        proxsdp_options d;
        proxsdp::default_options(&d);
        if (o) {
            if (o->struct_size != (int64_t)sizeof(proxsdp_options))
                throw std::invalid_argument("proxsdp_options.struct_size mismatch (ABI version?)");
            d = *o;
        }
        return d;
    }
    void set_resid(const double* resid) {
        proxsdp::EigWork& W = S.eig[0];
        W.resid_host.assign(W.npad, 0.0);
        if (resid) std::copy(resid, resid + W.n, W.resid_host.begin());
        else proxsdp::start_vector(W.n, (uint64_t)S.opt.eigsolver_resid_seed,
                                   S.opt.eigsolver == 1 ? S.opt.arpack_resid_init : S.opt.krylovkit_resid_init,
                                   W.resid_host.data());
        double nr = proxsdp::norm2(W.resid_host.data(), W.n);
        if (!(nr > 0.0)) throw std::invalid_argument("Lanczos start vector has zero norm");
        for (int i = 0; i < W.n; ++i) W.resid_host[i] /= nr;
        W.resid.upload(W.resid_host.data(), W.npad, S.stream);
        PX_HIP(hipStreamSynchronize(S.stream));
    }
};
}  // namespace

extern "C" {

int proxsdp_hip_abi_version(void) { return PROXSDP_HIP_ABI_VERSION; }

void proxsdp_hip_default_options(proxsdp_options* opt) {
    if (opt) proxsdp::default_options(opt);
}

int proxsdp_hip_set_option(proxsdp_options* opt, const char* name, double value) {
    if (!opt || !name) { g_last_error = "NULL argument"; return PROXSDP_E_INVALID; }
    int rc = proxsdp::set_option(opt, name, value);
    if (rc != 0) g_last_error = std::string("No parameter matching ") + name;
    return rc;
}

int proxsdp_hip_get_option(const proxsdp_options* opt, const char* name, double* value) {
    if (!opt || !name || !value) { g_last_error = "NULL argument"; return PROXSDP_E_INVALID; }
    int rc = proxsdp::get_option(opt, name, value);
    if (rc != 0) g_last_error = std::string("No parameter matching ") + name;
    return rc;
}

// ---- RCCL communicator helpers (block-sharded solve, native collectives: proxsdp_problem.nccl_comm)
int proxsdp_hip_rccl_available(void) {
    return guarded([&]() -> int { return proxsdp::Rccl::get().ok() ? 1 : 0; });
}
int proxsdp_hip_rccl_unique_id(void* id128) {
    return guarded([&]() -> int {
        if (!id128) throw std::invalid_argument("NULL id buffer");
        proxsdp::Rccl& rc = proxsdp::Rccl::get();
        rc.require();
        ncclUniqueId id;
        rc.check(rc.GetUniqueId(&id), "ncclGetUniqueId");
        std::memcpy(id128, id.internal, NCCL_UNIQUE_ID_BYTES);
        return 0;
    });
}
int proxsdp_hip_rccl_comm_init(int32_t nranks, const void* id128, int32_t rank, int32_t device_id, void** comm) {
    return guarded([&]() -> int {
        if (!id128 || !comm || nranks < 1 || rank < 0 || rank >= nranks) throw std::invalid_argument("invalid argument");
        proxsdp::Rccl& rc = proxsdp::Rccl::get();
        rc.require();
        if (hipSetDevice(device_id) != hipSuccess) throw proxsdp::HipError("hipSetDevice failed");
        ncclUniqueId id;
        std::memcpy(id.internal, id128, NCCL_UNIQUE_ID_BYTES);
        ncclComm_t c = nullptr;
        rc.check(rc.CommInitRank(&c, nranks, id, rank), "ncclCommInitRank");
        *comm = c;
        return 0;
    });
}
int proxsdp_hip_rccl_comm_destroy(void* comm) {
    return guarded([&]() -> int {
        if (!comm) return 0;
        proxsdp::Rccl& rc = proxsdp::Rccl::get();
        rc.require();
        rc.check(rc.CommDestroy(static_cast<ncclComm_t>(comm)), "ncclCommDestroy");
        return 0;
    });
}

const char* proxsdp_hip_last_error(void) { return g_last_error.c_str(); }

int proxsdp_hip_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        g_last_error = std::string("hipGetDeviceCount: ") + hipGetErrorString(e);
        return PROXSDP_E_HIP;
    }
    return n;
}

int proxsdp_hip_solve(const proxsdp_problem* prob, const proxsdp_options* opt, proxsdp_result* res) {
    return proxsdp_hip_solve_ex(prob, opt, res, nullptr, nullptr);
}

int proxsdp_hip_solve_ex(const proxsdp_problem* prob, const proxsdp_options* opt, proxsdp_result* res,
                         const proxsdp_state* resume, proxsdp_state* capture) {
    bool comm_aborted = false;
    const int rc = guarded([&]() -> int {
        if (!prob || !res) throw std::invalid_argument("NULL problem or result");
        proxsdp_options o = Engine::fix(opt);
        res->status = PROXSDP_STATUS_NOT_CALLED;
        res->trace_rows = 0;
        res->result_count = 0;
        res->certificate_found = 0;
        res->status_string[0] = 0;
        if (o.trace_capacity > 0 && !res->trace) o.trace_capacity = 0;
        proxsdp::Solver S(*prob, o, *res);
        S.resume_state = resume;
        S.capture_state = capture;
        try {
            S.run();
        } catch (...) {
            // native RCCL path: this rank leaves the solve -- stop its own pending collectives so that its stream drains;
            // the peers' waits are bounded (Solver::wait_collective) and fail the same way.  Only when a collective of this
            // solve may be pending: an argument error raised before the first one leaves the caller's communicator alone
            // (ADVICE r4).  ncclCommAbort releases the communicator: the caller learns it through PROXSDP_E_COMM_ABORTED.
            if (S.nccl && !S.nccl_aborted && S.collective_enqueued) S.abort_comm();
            comm_aborted = S.nccl_aborted;
            throw;
        }
        return 0;
    });
    if (rc != 0 && comm_aborted) {
        g_last_error += " [the RCCL communicator was aborted (ncclCommAbort): it is released, do not destroy or reuse it]";
        return PROXSDP_E_COMM_ABORTED;
    }
    return rc;
}

int proxsdp_hip_psd_project(const double* packed_in, int64_t n, int32_t target_rank, int32_t mode,
                            const proxsdp_options* opt, const double* resid, double* packed_out,
                            int32_t* out_rank, double* out_min_eig, int64_t* out_nmatvec,
                            int32_t* out_converged, int32_t* out_fell_back) {
    return guarded([&]() -> int {
        if (!packed_in || !packed_out) throw std::invalid_argument("NULL buffer");
        if (target_rank < 1) throw std::invalid_argument("target_rank < 1");
        proxsdp_options o = Engine::fix(opt);
        if (mode == 1) { o.full_eig_decomp = 1; o.full_eig_lanczos = 0; o.full_eig_sign = 0; }
        // mode 4: full_eig! by the sign-function projection (fp64 MFMA products)
        if (mode == 4) { o.full_eig_decomp = 1; o.full_eig_lanczos = 0; o.full_eig_sign = 1; }
        // the test entry point takes the Krylov branch whenever mode == 0
        if (mode == 0) { o.min_size_krylov_eigs = 0; o.max_target_rank_krylov_eigs = std::max(o.max_target_rank_krylov_eigs, target_rank); }
        // mode 2: full_eig! served by the Lanczos engine, `target_rank` = the estimate of the number of
        // positive eigenvalues (in a solve: the count of the block's previous projection)
        if (mode == 2) { o.full_eig_decomp = 1; o.full_eig_lanczos = 1; o.min_size_krylov_eigs = 0; }
        const int ws = mode == 2 ? std::min<int>({(int)n, 94, 2 * (target_rank + std::max(3, target_rank / 8)) + 2}) : target_rank;
        Engine E(&o, n, ws);
        E.set_resid(resid);
        proxsdp::Solver& S = E.S;
        const int64_t N = n * (n + 1) / 2;
        proxsdp::DevBuf<double> x(N);
        x.upload(packed_in, N, S.stream);
        S.P.blocks.push_back({(int)n, N, 0});
        if (mode == 2) S.eig[0].last_npos = target_rank;
        if (mode == 3 || mode == 5) {                       // the small-block kernels on this one block: 3 batched Jacobi, 5 LDS-resident sign projection
            if (n < 2 || n > 64) throw std::invalid_argument("mode 3 / 5: 2 <= n <= 64");
            proxsdp::DevBuf<long long> off(1); proxsdp::DevBuf<int> side(1), rk(2);
            const long long o0 = 0; const int s0 = (int)n;
            off.upload(&o0, 1, S.stream); side.upload(&s0, 1, S.stream);
            if (mode == 3) {
                const size_t lds = ((size_t)2 * n * (n | 1) + 64) * sizeof(double) + 64 * sizeof(int);
                if (lds > 48 * 1024)
                    PX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(proxsdp::dev::k_small_psd_project),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                hipLaunchKernelGGL(proxsdp::dev::k_small_psd_project, dim3(1), dim3(proxsdp::dev::TPB), lds, S.stream,
                                   x.p, (const long long*)off.p, (const int*)side.p, o.tol_psd, rk.p, rk.p + 1, 2, 64);
            } else {
                const size_t lds = proxsdp::dev::small_sign_lds_bytes((int)n);
                if (lds > 48 * 1024)
                    PX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(proxsdp::dev::k_small_sign_project),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                // target_rank doubles as the start row of the schedule here (1 = full table ... 9 = row 8, the solver's default)
                static const proxsdp::dev::SignSchedule sched;
                const int j0 = std::max(0, std::min(target_rank - 1, proxsdp::dev::SIGN_STEPS - 3));
                int rfail = 0;
                for (int k = 0; k < proxsdp::dev::SIGN_STEPS; ++k) if (sched.l[k] <= 1e-10 * sched.gain[j0]) rfail = k;
                hipLaunchKernelGGL(proxsdp::dev::k_small_sign_project, dim3(1), dim3(proxsdp::dev::SS_TPB), lds, S.stream,
                                   x.p, (const long long*)off.p, (const int*)side.p, 2, 64, rk.p, rk.p + 1, j0, rfail, (int*)nullptr);
            }
            int hr[2] = {0, 0};
            rk.download(hr, 2, S.stream);
            x.download(packed_out, N, S.stream);
            PX_HIP(hipStreamSynchronize(S.stream));
            if (out_rank) *out_rank = hr[0];
            if (out_min_eig) *out_min_eig = 0.0;
            if (out_nmatvec) *out_nmatvec = 0;
            if (out_converged) *out_converged = hr[1];
            if (out_fell_back) *out_fell_back = 0;
            return 0;
        }
        S.test_project(0, x.p, target_rank);
        x.download(packed_out, N, S.stream);
        PX_HIP(hipStreamSynchronize(S.stream));
        if (out_rank) *out_rank = (int32_t)S.test_rank();
        if (out_min_eig) *out_min_eig = S.test_min_eig();
        S.merge_block_stats();
        if (out_nmatvec) *out_nmatvec = S.st.lanczos_matvecs;
        if (out_converged) *out_converged = S.eig[0].converged_eigs;
        if (out_fell_back) *out_fell_back = mode == 2 ? (int32_t)(S.st.full_eigs_lanczos == 0) : (int32_t)S.st.krylov_fallbacks;
        return 0;
    });
}

int proxsdp_hip_full_eig_kernel(const double* packed_in, int64_t n, int32_t sign, double* packed_out,
                                int32_t repeat, double* ms, int32_t* out_rank, int64_t* out_products) {
    return guarded([&]() -> int {
        if (!packed_in || !packed_out) throw std::invalid_argument("NULL buffer");
        proxsdp_options o = Engine::fix(nullptr);
        o.full_eig_decomp = 1; o.full_eig_lanczos = 0; o.full_eig_sign = sign < 0 ? -1 : (sign ? 1 : 0);   // (-1: the solver's own choice of engine)
        if (sign >= 100) o.sign_start_row = sign - 100;
        Engine E(&o, n, 2);
        proxsdp::Solver& S = E.S;
        const int64_t N = n * (n + 1) / 2;
        proxsdp::DevBuf<double> x(N), y(N);
        x.upload(packed_in, N, S.stream);
        S.P.blocks.push_back({(int)n, N, 0});
        S.test_full_eig(x.p, y.p);             // warm-up (allocations, rocSOLVER workspace)
        PX_HIP(hipStreamSynchronize(S.stream));
        const int reps = std::max(1, repeat);
        const double t0 = proxsdp::now_s();
        for (int i = 0; i < reps; ++i) S.test_full_eig(x.p, y.p);
        PX_HIP(hipStreamSynchronize(S.stream));
        if (ms) *ms = (proxsdp::now_s() - t0) * 1e3 / reps;
        y.download(packed_out, N, S.stream);
        PX_HIP(hipStreamSynchronize(S.stream));
        S.merge_block_stats();
        if (out_rank) *out_rank = (int32_t)S.test_rank();
        if (out_products) *out_products = S.st.sign_products / (reps + 1);
        return 0;
    });
}

int proxsdp_hip_eigsolve(const double* packed, int64_t n, int32_t nev, const proxsdp_options* opt,
                         const double* resid, int32_t cap, double* vals, double* vecs,
                         int32_t* out_count, int32_t* out_converged, int64_t* out_nmatvec,
                         int32_t* out_numiter) {
    return guarded([&]() -> int {
        if (!packed || !vals || !vecs) throw std::invalid_argument("NULL buffer");
        if (nev < 1) throw std::invalid_argument("nev < 1");
        Engine E(opt, n, nev);
        E.set_resid(resid);
        proxsdp::Solver& S = E.S;
        proxsdp::EigWork& W = S.eig[0];
        const int64_t N = n * (n + 1) / 2;
        proxsdp::DevBuf<double> x(N);
        x.upload(packed, N, S.stream);
        S.lanczos(W, x.p, nev);
        PX_HIP(hipStreamSynchronize(S.stream));
        const int cnt = std::min<int>(W.count, cap);
        for (int i = 0; i < cnt; ++i) vals[i] = W.vals[i];
        if (cnt > 0)
            PX_HIP(hipMemcpy2D(vecs, (size_t)n * 8, W.Z.p, (size_t)W.npad * 8, (size_t)n * 8, cnt,
                               hipMemcpyDeviceToHost));
        if (out_count) *out_count = W.count;
        if (out_converged) *out_converged = W.converged_eigs;
        S.merge_block_stats();
        if (out_nmatvec) *out_nmatvec = S.st.lanczos_matvecs;
        if (out_numiter) *out_numiter = W.numiter;
        return 0;
    });
}

int proxsdp_hip_symv_packed(const double* packed, int64_t n, const double* v, double* y,
                            int32_t repeat, double* ms) {
    return guarded([&]() -> int {
        if (!packed || !v || !y) throw std::invalid_argument("NULL buffer");
        Engine E(nullptr, n, 2);
        proxsdp::Solver& S = E.S;
        proxsdp::EigWork& W = S.eig[0];
        const int64_t N = n * (n + 1) / 2;
        proxsdp::DevBuf<double> x(N), vd(W.npad);
        x.upload(packed, N, S.stream);
        vd.zero(S.stream);
        vd.upload(v, n, S.stream);
        PX_HIP(hipMemsetAsync(W.ctl_p, 0, sizeof(proxsdp::dev::LanczosCtl), S.stream));
        // y = (sum of partials)/sqrt2
        S.launch_symv(W, x.p, vd.p, false);
        hipLaunchKernelGGL(proxsdp::dev::k_symv_collect, dim3(W.nt), dim3(proxsdp::dev::TPB), 0, S.stream,
                           W.Ppart.p, W.nt, W.npad, W.w.p);
        W.w.download(y, n, S.stream);
        PX_HIP(hipStreamSynchronize(S.stream));
        if (repeat > 0 && ms) {
            hipEvent_t a, b;
            PX_HIP(hipEventCreate(&a)); PX_HIP(hipEventCreate(&b));
            for (int i = 0; i < 3; ++i) S.launch_symv(W, x.p, vd.p, false);
            PX_HIP(hipEventRecord(a, S.stream));
            for (int i = 0; i < repeat; ++i) S.launch_symv(W, x.p, vd.p, false);
            PX_HIP(hipEventRecord(b, S.stream));
            PX_HIP(hipEventSynchronize(b));
            float t = 0.f;
            PX_HIP(hipEventElapsedTime(&t, a, b));
            *ms = (double)t / repeat;
            (void)hipEventDestroy(a); (void)hipEventDestroy(b);
        }
        return 0;
    });
}

int proxsdp_hip_reconstruct(const double* Z, const double* lambda, int64_t n, int32_t r,
                            double* packed_out, int32_t repeat, double* ms) {
    return proxsdp_hip_reconstruct_kernel(Z, lambda, n, r, -1, packed_out, repeat, ms);
}

int proxsdp_hip_reconstruct_kernel(const double* Z, const double* lambda, int64_t n, int32_t r, int32_t mfma,
                                   double* packed_out, int32_t repeat, double* ms) {
    return guarded([&]() -> int {
        if ((r > 0 && (!Z || !lambda)) || !packed_out) throw std::invalid_argument("NULL buffer");
        if (r < 0) throw std::invalid_argument("r < 0");
        proxsdp_options o = Engine::fix(nullptr);
        o.reconstruct_mfma = mfma;
        Engine E(&o, n, 2);
        proxsdp::Solver& S = E.S;
        proxsdp::EigWork& W = S.eig[0];
        const int64_t N = n * (n + 1) / 2;
        proxsdp::DevBuf<double> x(N), Zd((size_t)std::max<int64_t>(1, n * r)), ld(std::max(1, r));
        Zd.upload(Z, (size_t)n * r, S.stream);
        ld.upload(lambda, r, S.stream);
        S.launch_reconstruct(W, Zd.p, (int)n, ld.p, r, x.p);
        x.download(packed_out, N, S.stream);
        PX_HIP(hipStreamSynchronize(S.stream));
        if (repeat > 0 && ms) {
            hipEvent_t a, b;
            PX_HIP(hipEventCreate(&a)); PX_HIP(hipEventCreate(&b));
            PX_HIP(hipEventRecord(a, S.stream));
            for (int i = 0; i < repeat; ++i) S.launch_reconstruct(W, Zd.p, (int)n, ld.p, r, x.p);
            PX_HIP(hipEventRecord(b, S.stream));
            PX_HIP(hipEventSynchronize(b));
            float t = 0.f;
            PX_HIP(hipEventElapsedTime(&t, a, b));
            *ms = (double)t / repeat;
            (void)hipEventDestroy(a); (void)hipEventDestroy(b);
        }
        return 0;
    });
}

int proxsdp_hip_spmv(const proxsdp_csc* M, int32_t index_base, int32_t transpose,
                     const double* in, double* out) {
    return guarded([&]() -> int {
        if (!M || !in || !out) throw std::invalid_argument("NULL argument");
        // run through the same preparation + kernels as the solver
        proxsdp_problem pr{};
        pr.n = M->ncols; pr.p = M->nrows; pr.m = 0;
        pr.A = *M;
        std::vector<int64_t> zc((size_t)M->ncols + 1, index_base);
        pr.G.nrows = 0; pr.G.ncols = M->ncols; pr.G.colptr = zc.data();
        std::vector<double> zb((size_t)std::max<int64_t>(M->nrows, 1), 0.0), zcv((size_t)std::max<int64_t>(M->ncols, 1), 0.0);
        pr.b = zb.data(); pr.h = zb.data(); pr.c = zcv.data();
        pr.index_base = index_base;
        proxsdp_options o;
        proxsdp::default_options(&o);
        proxsdp_result dummy{};
        proxsdp::Solver S(pr, o, dummy);
        S.test_spmv(transpose != 0, in, out);
        return 0;
    });
}

int proxsdp_hip_primal_update(const double* x, const double* Mty, const double* c, double tau, int64_t n,
                              double* x_out) {
    return guarded([&]() -> int {
        if (n < 0 || (n > 0 && (!x || !Mty || !c || !x_out))) throw std::invalid_argument("invalid argument");
        if (n == 0) return 0;
        Engine E(nullptr, 2, 2);
        hipStream_t st = E.S.stream;
        proxsdp::DevBuf<double> dx(n), dm(n), dc(n), dout(n);
        dx.upload(x, n, st); dm.upload(Mty, n, st); dc.upload(c, n, st);
        hipLaunchKernelGGL(proxsdp::dev::k_primal_update, dim3(proxsdp::grid_for(n)), dim3(proxsdp::dev::TPB), 0, st,
                           dout.p, (const double*)dx.p, (const double*)dm.p, (const double*)dc.p, tau, (long long)n);
        dout.download(x_out, n, st);
        PX_HIP(hipStreamSynchronize(st));
        return 0;
    });
}

int proxsdp_hip_dual_trial(const double* y, const double* Mx, const double* Mx_old, const double* bh,
                           int64_t p, int64_t Q, double bt, double theta, double* y_out, double* ynorm2) {
    return guarded([&]() -> int {
        if (Q <= 0 || p < 0 || p > Q || !y || !Mx || !Mx_old || !bh || !y_out) throw std::invalid_argument("invalid argument");
        Engine E(nullptr, 2, 2);
        hipStream_t st = E.S.stream;
        const int g = std::min(proxsdp::PSTRIDE, proxsdp::grid_for(Q));
        proxsdp::DevBuf<double> dy(Q), d1(Q), d0(Q), dbh(Q), dout(Q), part(proxsdp::PSTRIDE), sc(2);
        dy.upload(y, Q, st); d1.upload(Mx, Q, st); d0.upload(Mx_old, Q, st); dbh.upload(bh, Q, st);
        part.zero(st);
        hipLaunchKernelGGL(proxsdp::dev::k_dual_trial, dim3(g), dim3(proxsdp::dev::TPB), 0, st,
                           (const double*)dy.p, (const double*)d1.p, (const double*)d0.p, (const double*)dbh.p,
                           (int)p, (int)Q, bt, theta, dout.p, part.p, 1);
        hipLaunchKernelGGL(proxsdp::dev::k_combine, dim3(1), dim3(proxsdp::dev::TPB), 0, st,
                           (const double*)part.p, proxsdp::PSTRIDE, g, 1, 0u, sc.p);
        dout.download(y_out, Q, st);
        double nrm = 0.0;
        sc.download(&nrm, 1, st);
        PX_HIP(hipStreamSynchronize(st));
        if (ynorm2) *ynorm2 = nrm;
        return 0;
    });
}

int proxsdp_hip_residuals(const double* x, const double* x_old, const double* Mty, const double* Mty_old,
                          const double* c, double tau, int64_t n,
                          const double* y, const double* y_old, const double* Mx, const double* Mx_old,
                          const double* bh, int64_t p, int64_t Q, double sigma, double* out) {
    return guarded([&]() -> int {
        if (n <= 0 || Q <= 0 || p < 0 || p > Q || !x || !x_old || !Mty || !Mty_old || !c || !y || !y_old || !Mx ||
            !Mx_old || !bh || !out) throw std::invalid_argument("invalid argument");
        Engine E(nullptr, 2, 2);
        hipStream_t st = E.S.stream;
        using proxsdp::DevBuf;
        const int gx = std::min(proxsdp::PSTRIDE, proxsdp::grid_for(n)), gq = std::min(proxsdp::PSTRIDE, proxsdp::grid_for(Q));
        DevBuf<double> dx(n), dxo(n), dm(n), dmo(n), dc(n), dy(Q), dyo(Q), d1(Q), d0(Q), dbh(Q);
        DevBuf<double> part((size_t)9 * proxsdp::PSTRIDE), sc(9);
        dx.upload(x, n, st); dxo.upload(x_old, n, st); dm.upload(Mty, n, st); dmo.upload(Mty_old, n, st); dc.upload(c, n, st);
        dy.upload(y, Q, st); dyo.upload(y_old, Q, st); d1.upload(Mx, Q, st); d0.upload(Mx_old, Q, st); dbh.upload(bh, Q, st);
        part.zero(st);
        // the kernels lay their quantities out with stride gridDim.x: run both with PSTRIDE-strided combines
        hipLaunchKernelGGL(proxsdp::dev::k_residual_x, dim3(gx), dim3(proxsdp::dev::TPB), 0, st,
                           (const double*)dx.p, (const double*)dxo.p, 1.0, (const double*)dm.p, (const double*)dmo.p,
                           (const double*)dc.p, tau, (long long)n, part.p);
        hipLaunchKernelGGL(proxsdp::dev::k_combine, dim3(1), dim3(proxsdp::dev::TPB), 0, st,
                           (const double*)part.p, gx, gx, 3, 0x3u, sc.p);
        hipLaunchKernelGGL(proxsdp::dev::k_residual_y, dim3(gq), dim3(proxsdp::dev::TPB), 0, st,
                           (const double*)dy.p, (const double*)dyo.p, (const double*)d1.p, (const double*)d0.p,
                           (const double*)dbh.p, (int)p, (int)Q, sigma, part.p + (size_t)3 * proxsdp::PSTRIDE);
        hipLaunchKernelGGL(proxsdp::dev::k_combine, dim3(1), dim3(proxsdp::dev::TPB), 0, st,
                           (const double*)(part.p + (size_t)3 * proxsdp::PSTRIDE), gq, gq, 6, 0xFu, sc.p + 3);
        sc.download(out, 9, st);
        PX_HIP(hipStreamSynchronize(st));
        return 0;
    });
}

int proxsdp_host_symeig(int32_t k, double* a, double* d) {
    return proxsdp_host_symeig_threads(k, a, d, -1);
}

int proxsdp_host_symeig_threads(int32_t k, double* a, double* d, int32_t threads) {
    if (k < 0 || !a || !d) { g_last_error = "invalid argument"; return PROXSDP_E_INVALID; }
    int rc = proxsdp::symeig_dense(k, a, d, false, threads);
    if (rc != 0) { g_last_error = "QL iteration did not converge"; return PROXSDP_E_INTERNAL; }
    return 0;
}

int proxsdp_host_symeig_arrow(int32_t K, int32_t m, const double* D, const double* f,
                              const double* al, const double* be, double* U, double* d) {
    if (K < 2 || m < 1 || m >= K || !D || !f || !al || !be || !U || !d) { g_last_error = "invalid argument"; return PROXSDP_E_INVALID; }
    const int n1 = m + 1;
    std::vector<double> Qa((size_t)n1 * n1, 0.0), da(n1), ea(n1);
    for (int j = 0; j < m; ++j) { Qa[(size_t)j * n1 + j] = D[j]; Qa[(size_t)j * n1 + m] = Qa[(size_t)m * n1 + j] = f[j]; }
    proxsdp::householder_tridiag(n1, Qa.data(), da.data(), ea.data());
    int rc = proxsdp::symeig_tridiag_from(K, m, Qa.data(), da.data(), ea.data(), al, be, U, d);
    if (rc != 0) { g_last_error = "QL iteration did not converge"; return PROXSDP_E_INTERNAL; }
    return 0;
}

int proxsdp_host_symeig_split(int32_t K, int32_t m, int32_t k1, const double* D, const double* f,
                              const double* al, const double* be, double* U, double* d, int32_t* info) {
    return proxsdp_host_symeig_split_threads(K, m, k1, D, f, al, be, 0, U, d, info);
}

int proxsdp_host_symeig_split_threads(int32_t K, int32_t m, int32_t k1, const double* D, const double* f,
                                      const double* al, const double* be, int32_t threads, double* U, double* d, int32_t* info) {
    return guarded([&]() -> int {
        if (K < 2 || m < 0 || m >= K || k1 < 1 || k1 >= K || !al || !be || !U || !d) throw std::invalid_argument("invalid argument");
        proxsdp::SplitEig S;
        std::unique_ptr<proxsdp::SpinPool> pool;
        proxsdp::ParFor par;
        struct Guard { proxsdp::SpinPool* p; ~Guard() { if (p) p->disarm(); } } guard{nullptr};
        if (threads > 0) {
            pool.reset(new proxsdp::SpinPool(std::min<int>(threads, 15)));
            pool->arm();
            guard.p = pool.get();
            par = [&pool](int n, const std::function<void(int)>& body) { pool->run(n, body); };
            S.M.par = &par;
            S.M.nchunk = std::min<int>(threads, 15) + 1;
        }
        if (S.first(k1, m, D, f, al, be) != 0 || S.second(K, al, be) != 0) throw std::invalid_argument("split eigensolver failed");
        std::vector<int> cols(K);
        for (int c = 0; c < K; ++c) { cols[c] = c; d[c] = S.M.evals[c]; }
        S.M.vectors(cols.data(), K, U);
        if (info) { info[0] = S.M.k; info[1] = (int)S.M.df.size(); info[2] = S.M.max_iter_seen; }
        return 0;
    });
}
int proxsdp_host_start_vector(int64_t n, int64_t seed, int32_t init, double* out) {
    if (n < 0 || !out) { g_last_error = "invalid argument"; return PROXSDP_E_INVALID; }
    proxsdp::start_vector(n, (uint64_t)seed, init, out);
    return 0;
}

int proxsdp_host_preprocess(const proxsdp_problem* prob, int64_t* order, int64_t* var_ordering,
                            double* c_scaled, double* frobenius_norm_M) {
    return guarded([&]() -> int {
        if (!prob) throw std::invalid_argument("NULL problem");
        proxsdp::Prep R = proxsdp::prepare(*prob);
        for (int64_t i = 0; i < R.n; ++i) {
            if (order) order[i] = R.ord[i];
            if (var_ordering) var_ordering[i] = R.inv[i];
            if (c_scaled) c_scaled[i] = R.c[i];
        }
        if (frobenius_norm_M) *frobenius_norm_M = R.frob;
        return 0;
    });
}

#ifdef PX_TIMELINE
// measurement builds only (tools/timeline/): copy of the step kernels' stamp table
int proxsdp_hip_debug_timeline(unsigned long long* out, int64_t count, int32_t clear) {
    return guarded([&]() -> int {
        const size_t total = sizeof(proxsdp::dev::g_tl) / sizeof(unsigned long long);
        if (out && count > 0)
            PX_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(proxsdp::dev::g_tl), std::min<size_t>(total, (size_t)count) * sizeof(unsigned long long)));
        if (clear) {
            void* p = nullptr;
            PX_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(proxsdp::dev::g_tl)));
            PX_HIP(hipMemset(p, 0, sizeof(proxsdp::dev::g_tl)));
        }
        return (int)std::min<size_t>(total, (size_t)1 << 30);
    });
}
#endif

}  // extern "C"
