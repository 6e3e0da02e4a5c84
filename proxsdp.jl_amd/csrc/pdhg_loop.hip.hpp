// The PDHG loop itself: primal_step!, linesearch!/dual_step!, compute_residual!,
// compute_gap!, the stop / rank-update / adaptive-step logic and the exit path.
// Each function cites the reference lines it replaces.
#pragma once
#include "solver.hip.hpp"

namespace proxsdp {

constexpr int PSTRIDE = 2048;       // stride of one quantity in the partials buffer
constexpr int NQ = 16;

// pdhg.jl:634 (Mx = M x)
inline void Solver::spmv(const double* x, double* y) {
    spmv_sparse(x, y);                                   // all Q rows (zeros for the rows of a dense A)
    if (P.dense()) dense_mv(x, y, true);                 // rows [0, p)
}
inline void Solver::spmv_sparse(const double* x, double* y) {
    if (P.Q == 0) return;
    if (csr_wave)
        hipLaunchKernelGGL(dev::k_spmv_csr_wave, dim3(ceil_div(P.Q, dev::NWAVE)), dim3(dev::TPB), 0, stream,
                           csr_ptr.p, csr_col.p, csr_val.p, x, y, (int)P.Q, LONG_ROW);
    else
        hipLaunchKernelGGL(dev::k_spmv_csr_thread, dim3(ceil_div(P.Q, dev::TPB)), dim3(dev::TPB), 0, stream,
                           csr_ptr.p, csr_col.p, csr_val.p, x, y, (int)P.Q, LONG_ROW);
    if (n_seg > 0) {
        hipLaunchKernelGGL(dev::k_spmv_csr_seg, dim3(n_seg), dim3(dev::TPB), 0, stream,
                           seg_lo_d.p, seg_hi_d.p, csr_col.p, csr_val.p, x, segpart_d.p);
        hipLaunchKernelGGL(dev::k_spmv_seg_fin, dim3(ceil_div(n_long, dev::TPB)), dim3(dev::TPB), 0, stream,
                           long_row_d.p, long_ptr_d.p, n_long, segpart_d.p, y);
    }
}
// rows with more than LONG_ROW entries -> segments of SPMV_SEG entries
inline void Solver::setup_long_rows(const std::vector<int>& rp) {
    std::vector<int> lo, hi, rows, ptr{0};
    for (int64_t r = 0; r < P.Q; ++r) {
        if (rp[r + 1] - rp[r] <= LONG_ROW) continue;
        rows.push_back((int)r);
        for (int k = rp[r]; k < rp[r + 1]; k += dev::SPMV_SEG) { lo.push_back(k); hi.push_back(std::min(k + dev::SPMV_SEG, rp[r + 1])); }
        ptr.push_back((int)lo.size());
    }
    n_seg = (int)lo.size(); n_long = (int)rows.size();
    if (n_seg == 0) return;
    seg_lo_d.alloc(n_seg); seg_hi_d.alloc(n_seg); long_row_d.alloc(n_long); long_ptr_d.alloc(n_long + 1); segpart_d.alloc(n_seg);
    seg_lo_d.upload(lo.data(), n_seg, stream); seg_hi_d.upload(hi.data(), n_seg, stream);
    long_row_d.upload(rows.data(), n_long, stream); long_ptr_d.upload(ptr.data(), n_long + 1, stream);
    PX_HIP(hipStreamSynchronize(stream));
}

// block-sharded solve: the coupling rows of M x hold this shard's partial sums; all-reduce them
// (RCCL on the device buffer, or through host memory) -- SURVEY.md section 8e
inline void Solver::reduce_coupling(double* Mx_dev) {
    const int nc = (int)coup_rows.size();
    hipLaunchKernelGGL(dev::k_gather_rows, dim3(ceil_div(nc, dev::TPB)), dim3(dev::TPB), 0, stream,
                       (const double*)Mx_dev, (const int*)coup_rows_d.p, nc, coup_buf_d.p);
    if (nccl) {
        // native: in-place sum over the shards on THIS stream -- no host synchronisation, no callback
        Rccl& rc = Rccl::get();
        collective_enqueued = true;
        rc.check(rc.AllReduce(coup_buf_d.p, coup_buf_d.p, (size_t)nc, ncclFloat64, ncclSum, nccl, stream), "ncclAllReduce");
        st.rccl_reductions++;
    } else if (reduce_vec_on_device) {
        PX_HIP(hipStreamSynchronize(stream));               // the collective runs on the caller's stream
        if (reduce_vec_fn(reduce_ctx, coup_buf_d.p, nc, 1) != 0) throw std::runtime_error("reduce_vec_fn failed");
    } else {
        coup_host.resize(nc);
        coup_buf_d.download(coup_host.data(), nc, stream);
        PX_HIP(hipStreamSynchronize(stream));
        reduce_vec_host(coup_host);
        coup_buf_d.upload(coup_host.data(), nc, stream);
    }
    hipLaunchKernelGGL(dev::k_scatter_rows, dim3(ceil_div(nc, dev::TPB)), dim3(dev::TPB), 0, stream,
                       Mx_dev, (const int*)coup_rows_d.p, nc, (const double*)coup_buf_d.p);
}

// psd_projection! (prox_operators.jl:33-66), one block: reads the packed block of xin,
// writes the projected block into xout (xin == xout on the dense path)
// the branch condition of psd_projection! (prox_operators.jl:46-49)
inline bool Solver::krylov_branch(int idx) const {
    return !opt.full_eig_decomp && target_rank[idx] <= opt.max_target_rank_krylov_eigs &&
           eig[idx].n > opt.min_size_krylov_eigs && (iter % opt.full_eig_freq) > opt.full_eig_len;
}
// may this block's Lanczos run ride in a batched launch (solver.hip.hpp lanczos_batch)?
inline bool Solver::batch_eligible(int idx, bool fuse) const {
    const EigWork& W = eig[idx];
    if (opt.block_batch == 0 || !krylov_branch(idx)) return false;
    if (opt.eigsolver == 1 || opt.psd_sign_engine == 1 || opt.lanczos_warm_start > 0 || opt.lanczos_cycle_kernel == 1 ||
        opt.krylovkit_eager)
        return false;
    // operator-form blocks keep their own path (their step kernels take per-block factor ranks)
    if (fuse && use_support && opt.lanczos_operator != 0 && W.fop_ok && (W.have_factors || W.x_prev_sparse)) return false;
    return std::max(2 * (int)target_rank[idx] + 1, (int)opt.eigsolver_min_lanczos) <= 63;
}
// psd_projection!'s loop over the blocks (prox_operators.jl:40-61): blocks of equal side whose Lanczos runs can
// be batched go through ONE launch per step (groups of up to LZB_MAX), the rest through run_blocks
// (one stream + host thread per block, or in sequence)
inline void Solver::project_blocks(const std::vector<int>& blocks, const double* xin, double* xout, bool fuse) {
    std::vector<int> rest;
    std::vector<std::vector<int>> groups;
    if (opt.block_batch != 0 && blocks.size() >= 2) {
        std::vector<int> cand;
        for (int idx : blocks) (batch_eligible(idx, fuse) ? cand : rest).push_back(idx);
        std::stable_sort(cand.begin(), cand.end(), [this](int a, int b) { return eig[a].n < eig[b].n; });
        size_t i0 = 0;
        while (i0 < cand.size()) {
            size_t i1 = i0;
            while (i1 < cand.size() && eig[cand[i1]].n == eig[cand[i0]].n) ++i1;
            // equal-side run [i0, i1): groups of up to LZB_MAX blocks; with the worker pool up, at least two groups of >= 2
            // blocks on request (options.block_batch_groups) so that they can overlap
            const size_t cnt = i1 - i0;
            if (cnt < 2) { rest.push_back(cand[i0]); i0 = i1; continue; }
            size_t ng = (cnt + dev::LZB_MAX - 1) / dev::LZB_MAX;
            const int want = opt.block_batch_groups < 0 ? 1 : (int)opt.block_batch_groups;   // (auto = 1: measured, see the header)
            if (parallel_blocks && want >= 2 && cnt >= 4) ng = std::max<size_t>(ng, std::min<size_t>((size_t)want, cnt / 2));
            for (size_t q = 0; q < ng; ++q) {
                const size_t a = i0 + q * cnt / ng, b = i0 + (q + 1) * cnt / ng;
                groups.emplace_back(cand.begin() + a, cand.begin() + b);
            }
            i0 = i1;
        }
        std::sort(rest.begin(), rest.end());
    } else {
        rest = blocks;
    }
    // groups run CONCURRENTLY when the block worker pool is up (one worker thread + the leader block's stream per group):
    // while one group's restart logic runs on the host the other group's cycle runs on the GPU (the step launches are
    // latency-bound: two groups side by side advance about as fast as one).  Per block nothing changes.
    if (batch_ctx.size() < groups.size()) {
        const size_t have = batch_ctx.size();
        batch_ctx.resize(groups.size());
        for (size_t q = have; q < groups.size(); ++q) batch_ctx[q].reset(new BatchCtx());
    }
    for (const std::vector<int>& g : groups) for (int idx : g) current_rank[idx] = 0;
    // (groups run side by side only ON REQUEST -- block_batch_groups >= 2; a model with more than LZB_MAX equal-side
    // blocks splits into several groups by necessity, and those run one after the other as in rounds 3-4: auto = sequential)
    const int want_groups = opt.block_batch_groups < 0 ? 1 : (int)opt.block_batch_groups;
    bool leaders_have_streams = parallel_blocks && groups.size() >= 2 && want_groups >= 2;
    for (const std::vector<int>& g : groups) leaders_have_streams = leaders_have_streams && eig[g[0]].stream != nullptr;
    if (leaders_have_streams) {
        std::vector<int> leaders;
        std::vector<int> slot_of(eig.size(), -1);
        for (size_t q = 0; q < groups.size(); ++q) { leaders.push_back(groups[q][0]); slot_of[groups[q][0]] = (int)q; }
        run_blocks(leaders, [this, xin, &groups, &slot_of](int leader) {
            const std::vector<int>& g = groups[slot_of[leader]];
            std::vector<int> nevs;
            for (int idx : g) nevs.push_back((int)target_rank[idx]);
            lanczos_batch(g, xin, nevs, slot_of[leader]);
        });
    } else {
        for (size_t q = 0; q < groups.size(); ++q) {
            std::vector<int> nevs;
            for (int idx : groups[q]) nevs.push_back((int)target_rank[idx]);
            lanczos_batch(groups[q], xin, nevs, (int)q);
        }
    }
    for (const std::vector<int>& g : groups)
        for (int idx : g) project_block(idx, xin, xout, fuse, true);
    if (!groups.empty()) merge_block_stats();
    if (!rest.empty()) run_blocks(rest, [this, xin, xout, fuse](int idx) { project_block(idx, xin, xout, fuse); });
}

inline void Solver::project_block(int idx, const double* xin, double* xout, bool fuse, bool lanczos_done) {
    EigWork& W = eig[idx];
    const double* xp = xin + P.blocks[idx].off;
    double* xo = xout + P.blocks[idx].off;
    current_rank[idx] = 0;
    const bool krylov = lanczos_done || krylov_branch(idx);
    // operator-form mat-vec: legal when this block's x_prev is known in factored form (or is
    // zero off the support) and the update is the sparse support update of this iteration
    W.use_fop = !lanczos_done && fuse && use_support && opt.lanczos_operator != 0 && W.fop_ok && krylov &&
                (W.have_factors || W.x_prev_sparse) && !(opt.krylovkit_eager && opt.eigsolver != 1);
    if (W.use_fop) {
        if (!W.have_factors) { W.F_r = 0; W.F_first = 0; }
        W.esv = esv_d.p + (W.have_factors ? 0 : ns);
    }
    if (!krylov) {
        // full_eig! (prox_operators.jl:111-126): every positive eigenpair.  In the IMPLICIT full-eig
        // regime (target_rank beyond max_target_rank_krylov_eigs: the reference falls back to LAPACK
        // because its Krylov wrapper is capped, prox_operators.jl:46-49) the positive part is
        // low-rank and known from the previous projection: it is computed by the Lanczos engine
        // (operator form allowed) instead of a dense O(n^3) eigensolver -- see full_eig_by_lanczos.
        // An explicit full_eig_decomp = true / periodic full_eig_freq request always gets the dense solver.
        const bool implicit = !opt.full_eig_decomp && (iter % opt.full_eig_freq) > opt.full_eig_len &&
                              W.n > opt.min_size_krylov_eigs;
        if (!implicit && opt.full_eig_lanczos != 1) {
            W.have_factors = false; W.x_prev_sparse = false; W.use_fop = false;
            full_eig_project(idx, xp, xo, fuse);
            return;
        }
        W.use_fop = fuse && use_support && opt.lanczos_operator != 0 && W.fop_ok && (W.have_factors || W.x_prev_sparse);
        if (W.use_fop) {
            if (!W.have_factors) { W.F_r = 0; W.F_first = 0; }
            W.esv = esv_d.p + (W.have_factors ? 0 : ns);
        }
        if (full_eig_by_lanczos(idx, xp, xo, fuse)) return;
        W.have_factors = false; W.x_prev_sparse = false; W.use_fop = false;
        full_eig_project(idx, xp, xo, fuse);
        return;
    }
    const bool used_fop = W.use_fop;
    const int nev = (int)target_rank[idx];
    if (!lanczos_done && !krylovdim_fits(nev)) {          // Krylov dimension beyond the step kernels: dense eigensolver, same projection
        W.use_fop = false;
        truncated_project_dense(idx, xp, xo, fuse, nev);
        return;
    }
    if (!lanczos_done) {                      // (a batched run has already filled W.vals / W.Z)
        if (exact_projection_by_sign(idx, xp, xo, fuse, nev)) return;
        const double t_kry = opt.psd_sign_engine == 1 ? now_s() : 0.0;
        lanczos(W, xp, nev);
        if (opt.psd_sign_engine == 1 && W.converged) {
            const double ms = (now_s() - t_kry) * 1e3;
            W.kry_ms = W.kry_ms < 0 ? ms : 0.75 * W.kry_ms + 0.25 * ms;
        }
    }
    W.use_fop = false;
    if (used_fop) W.lst.fop_projections++;
    if (!W.converged) {                       // prox_operators.jl:55-57
        W.lst.krylov_fallbacks++;
        W.have_factors = false; W.x_prev_sparse = false;
        full_eig_project(idx, xp, xo, fuse);
        return;
    }
    double mn = W.vals[0];
    for (double v : W.vals) mn = std::min(mn, v);
    min_eig[idx] = mn;                        // prox_operators.jl:95 / :74
    int first = 0, npos = 0;
    if (opt.eigsolver == 1) {                 // ascending arc.d, all nev used (prox_operators.jl:78-85)
        for (int i = 0; i < nev; ++i) if (W.vals[i] > 0.0) ++npos;
        first = nev - npos;
    } else {                                  // descending, min(target_rank, converged) used (:99-106)
        const int k = std::min(nev, W.converged_eigs);
        for (int i = 0; i < k; ++i) if (W.vals[i] > 0.0) ++npos;
    }
    current_rank[idx] += npos;
    W.last_npos = npos;                       // (lower bound: estimate for a later full_eig!-by-Lanczos)
    if (npos > 0) W.lam.upload(W.vals.data() + first, npos, stream);
    launch_reconstruct(W, W.Z.p + (size_t)first * W.npad, W.npad, W.lam.p, npos, xo,
                       fuse ? xp : nullptr, fuse ? idx : -1);
    W.recon_r += npos;
    if (W.fop_ok) {                           // x_new = Z[:, first..] diag(lam) Z': keep the factors
        std::swap(W.Z.p, W.F.p);
        W.F_first = first; W.F_r = npos;
        std::swap(W.lam.p, W.Flam.p);            // the eigenvalues just uploaded become the next projection's factors
        std::swap(W.lam.n, W.Flam.n);
        W.have_factors = true; W.x_prev_sparse = false;
    }
    if (W.sign_check_pending) verify_sign_engine(idx, xo);
}

// psd_sign_engine = 1: the Krylov branch computes the top target_rank eigenpairs and keeps the positive ones
// (prox_operators.jl:89-109).  Whenever fewer than target_rank eigenvalues are positive that IS the exact
// projection, min_eig (the smallest returned value) is <= 0, and the two decisions min_eig feeds
// (pdhg.jl:272 `> tol_psd`, residuals.jl:93 `< tol_psd`) are settled -- so the sign-function projection
// (sign_project.hip.hpp) may stand in for the Lanczos engine when it is the cheaper way to the same matrix:
// hub-row / clustered spectra that cost hundreds of mat-vecs per projection (maxG51: 541), many mid-size
// blocks (MIMO).  It is chosen per block from measured wall-clock averages of both engines, only when the
// previous projection was not truncated, and VERIFIED afterwards: #{lambda > 0} >= target_rank means the
// reference would have truncated, and the projection is redone by the Lanczos engine.
inline double sign_cost_model_ms(int ld) {                   // measured on MI355X (profiles/r02c_sign_projection_summary.md)
    static const int L[] = {128, 512, 1024, 1536, 2048, 3072, 4096};
    static const double T[] = {0.30, 0.60, 2.05, 4.8, 11.3, 31.2, 71.2};
    if (ld <= L[0]) return T[0];
    for (int i = 1; i < 7; ++i)
        if (ld <= L[i]) return T[i - 1] + (T[i] - T[i - 1]) * (double)(ld - L[i - 1]) / (L[i] - L[i - 1]);
    return T[6] * std::pow((double)ld / 4096.0, 3.0);
}
inline bool Solver::exact_projection_by_sign(int idx, const double* xp, double* xo, bool fuse, int nev) {
    if (opt.psd_sign_engine != 1) return false;
    EigWork& W = eig[idx];
    W.sign_check_pending = false;                                         // (a verification whose Lanczos half fell back)
    if (W.n < 33 || W.n > 16384) return false;
    if (W.kry_ms < 0.0 || W.last_npos < 0) return false;                  // no Lanczos measurement of this block yet
    if (W.last_npos >= nev) return false;                                 // truncation was active last time: the reference's engine decides
    if (W.sign_backoff > 0) { --W.sign_backoff; return false; }           // after a rejected attempt
    const double est = W.sign_ms >= 0.0 ? W.sign_ms : sign_cost_model_ms(W.nt * dev::TILE);
    if (W.kry_ms < 1.5 * est) { W.sign_streak = 0; return false; }
    if (W.sign_streak >= 256) { W.sign_streak = 0; return false; }        // re-measure the Lanczos engine now and then
    if (W.sign_disabled) return false;
    const bool inplace = xp == xo;
    if (W.sign_verified_left <= 0) {
        // verification round (first use, then every 64th projection): BOTH engines run on this input; the Lanczos
        // result is the one used, and the engine may serve the next 64 projections only if the two agree.
        // Single-vector Lanczos returns one eigenvector per DISTINCT eigenvalue: where the iterate has repeated
        // positive eigenvalues (MIMO's first iterates) the reference's projection is not the exact one, the
        // comparison fails, and the block stays with the reference's engine for the rest of the solve.
        if (W.sg_out.n < (size_t)W.N) W.sg_out.alloc(W.N);
        const long long cr = current_rank[idx];
        const double me = min_eig[idx];
        const int lnp = W.last_npos;
        if (full_eig_by_sign(idx, xp, W.sg_out.p, false, true)) W.sign_check_pending = current_rank[idx] < nev;
        current_rank[idx] = cr; min_eig[idx] = me; W.last_npos = lnp;
        return false;
    }
    double* out = xo;
    if (inplace) {                                                        // keep the input: the check below may reject
        if (W.sg_out.n < (size_t)W.N) W.sg_out.alloc(W.N);
        out = W.sg_out.p;
    }
    const double t0 = now_s();
    if (!full_eig_by_sign(idx, xp, out, fuse, true)) return false;
    const double ms = (now_s() - t0) * 1e3;
    W.sign_ms = W.sign_ms < 0 ? ms : 0.75 * W.sign_ms + 0.25 * ms;
    const int npos = (int)current_rank[idx];
    if (npos >= nev) {                                                    // truncation would be active: not the same projection
        W.lst.sign_engine_rejected++;
        current_rank[idx] = 0;
        W.sign_backoff = 32;
        return false;
    }
    if (inplace) PX_HIP(hipMemcpyAsync(xo, out, (size_t)W.N * sizeof(double), hipMemcpyDeviceToDevice, stream));
    W.lst.sign_engine_projections++;
    W.sign_streak++;
    W.sign_verified_left--;
    min_eig[idx] = 0.0;                 // the reference's value is lambda_min of the returned pairs, <= 0: same decisions
    W.have_factors = false; W.x_prev_sparse = false; W.use_fop = false;
    return true;
}

// second half of a verification round: xo holds the Lanczos engine's projection, W.sg_out the sign function's
inline void Solver::verify_sign_engine(int idx, const double* xo) {
    EigWork& W = eig[idx];
    W.sign_check_pending = false;
    if (W.sg_cmp.n < 2) W.sg_cmp.alloc(2);
    W.sg_cmp.zero(stream);
    hipLaunchKernelGGL(dev::k_maxdiff, dim3(grid_for(W.N)), dim3(dev::TPB), 0, stream, xo, (const double*)W.sg_out.p,
                       (long long)W.N, W.sg_cmp.p);
    double h[2] = {0.0, 0.0};
    W.sg_cmp.download(h, 2, stream);
    PX_HIP(hipStreamSynchronize(stream));
    W.lst.sign_engine_checks++;
    if (h[0] <= 1e-8 * std::max(h[1], 1e-300)) {
        W.sign_verified_left = 64;
    } else {
        W.sign_disabled = true;
        W.lst.sign_engine_mismatches++;
    }
}

// full_eig! needs X+ = sum over lambda_i > 0 of lambda_i v_i v_i' -- every POSITIVE eigenpair, not
// the spectrum (prox_operators.jl:115-124), current_rank = #{lambda_i > tol_psd}, min_eig = 0.  The
// reference calls LAPACK dsyevr because it has no partial solver with an unknown count; on this path
// (target_rank beyond max_target_rank_krylov_eigs) the iterate is typically low-rank, so the positive
// part is the top of the spectrum: run the Lanczos engine for g = (positives last time) + slack
// largest pairs; if the smallest of them is <= 0 (and all g converged to krylovkit_tol) every
// positive eigenvalue is among them.  Otherwise enlarge g once, then fall back to the dense solver.
// rocSOLVER dsyevd takes 139 ms at n = 4000 (1.5 TF/s); this takes the cost of a Krylov projection.
inline bool Solver::full_eig_by_lanczos(int idx, const double* xp, double* xo, bool fuse) {
    EigWork& W = eig[idx];
    if (opt.full_eig_lanczos == 0 || opt.eigsolver == 1 || W.n <= opt.min_size_krylov_eigs) return false;
    if (W.last_npos < 0) return false;                       // no estimate yet: dense eigensolver first
    if (W.fel_disabled) return false;                        // a verification failed earlier in this solve
    const int maxnev = std::min((W.cap - 2) / 2, W.n - 1);   // krylovdim = 2 nev + 1 <= cap - 1
    int g = W.last_npos + std::max(3, W.last_npos / 8);
    if (8 * W.last_npos > W.n) return false;                 // many positive pairs: the dense solver is the cheaper tool
    const bool used_fop = W.use_fop;
    // Verification (options.full_eig_lanczos_verify): this engine is the library's own algorithm, and single-vector
    // Lanczos returns ONE eigenvector per distinct eigenvalue -- a repeated positive eigenvalue, or a start vector
    // deficient in an eigendirection, would silently drop a copy from X+.  On the first call of a block and every
    // k-th after it the reference's engine (full_eig_project: sign function / dsyevd) ALSO projects the same input
    // (into a scratch buffer, BEFORE the reconstruction overwrites an in-place input); the results are compared
    // below and a mismatch hands the block back to the dense engine for the rest of the solve.
    const int vk = opt.full_eig_lanczos_verify < 0 ? 256 : opt.full_eig_lanczos_verify;
    bool verify = vk > 0 && (W.fel_served % vk) == 0;
    long long ref_rank = 0;
    int ref_npos = 0;
    bool ref_sign = false;
    if (verify) {
        if (W.sg_out.n < (size_t)W.N) W.sg_out.alloc(W.N);
        const long long cr = current_rank[idx];
        const double me = min_eig[idx];
        const int lnp = W.last_npos;
        const bool hf = W.have_factors, xs = W.x_prev_sparse, uf = W.use_fop;
        const long long fe0 = W.lst.full_eigs, fs0 = W.lst.full_eigs_sign;
        full_eig_project(idx, xp, W.sg_out.p, false);
        ref_rank = current_rank[idx]; ref_npos = W.last_npos; ref_sign = W.lst.full_eigs_sign > fs0;
        current_rank[idx] = cr; min_eig[idx] = me; W.last_npos = lnp;
        W.have_factors = hf; W.x_prev_sparse = xs; W.use_fop = uf;
        W.lst.full_eigs = fe0; W.lst.full_eigs_sign = fs0;   // (the check is not a full_eig! call of the solve)
    }
    auto use_reference_result = [&]() {
        W.fel_disabled = true;
        W.lst.full_eigs_lanczos_mismatches++;
        W.have_factors = false; W.x_prev_sparse = false; W.use_fop = false;
        if (xp != xo) { full_eig_project(idx, xp, xo, fuse); return; }   // input intact: the ordinary dense call
        // in-place call: the input is gone, the scratch buffer holds the dense engine's projection of it
        PX_HIP(hipMemcpyAsync(xo, W.sg_out.p, (size_t)W.N * sizeof(double), hipMemcpyDeviceToDevice, stream));
        W.lst.full_eigs++;
        if (ref_sign) W.lst.full_eigs_sign++;
        current_rank[idx] = ref_rank; min_eig[idx] = 0.0; W.last_npos = ref_npos;
    };
    for (int attempt = 0; attempt < 2; ++attempt) {
        if (g > maxnev || g < 1) break;
        lanczos(W, xp, g, true);
        if (!W.converged) {                                   // more positives than g, or no convergence
            if (attempt == 0 && g < maxnev) { g = std::min(2 * g + 2, maxnev); continue; }
            break;
        }
        const int npos = W.count;
        W.fel_served++;
        // guard: between consecutive projections of this regime the number of positive eigenvalues
        // moves by a few; a collapse means the Krylov space was fooled (e.g. decoupled coordinates
        // resolved long before the small positive pairs): let the dense solver decide
        if (npos + 2 + W.last_npos / 8 < W.last_npos) break;
        // per-call certificate (Solver::lanczos_certificate): nothing positive may be left outside the returned pairs
        const int cert_m = opt.full_eig_lanczos_certify < 0 ? 10 : opt.full_eig_lanczos_certify;
        if (cert_m >= 2) {
            double theta = 0.0, cscale = 0.0;
            if (lanczos_certificate(W, xp, npos, cert_m, theta, cscale)) {
                const double posres = lanczos_posres();
                if (!(theta <= posres * cscale)) {
                    // a positive direction the run did not see: the dense engine projects this input (intact: the
                    // reconstruction has not run yet); three failures leave the block to the dense engine for good
                    W.lst.full_eigs_lanczos_cert_failed++;
                    if (++W.fel_cert_fails >= 3) W.fel_disabled = true;
                    if (debug) std::fprintf(stderr, "[proxsdp] certificate failed: block %d iteration %lld theta %.3e scale %.3e npos %d\n",
                                            idx, (long long)iter, theta, cscale, npos);
                    break;
                }
                W.lst.full_eigs_lanczos_certified++;
            }
        }
        int rank = 0;
        for (int i = 0; i < npos; ++i) if (W.vals[i] > opt.tol_psd) ++rank;
        // (values are descending: the positive ones are the first npos)
        W.use_fop = false;
        if (used_fop) W.lst.fop_projections++;
        W.lst.full_eigs++; W.lst.full_eigs_lanczos++;
        current_rank[idx] = rank;
        min_eig[idx] = 0.0;                                  // prox_operators.jl:114
        W.last_npos = npos;
        if (npos > 0) W.lam.upload(W.vals.data(), npos, stream);
        launch_reconstruct(W, W.Z.p, W.npad, W.lam.p, npos, xo, fuse ? xp : nullptr, fuse ? idx : -1);
        W.recon_r += npos;
        if (W.fop_ok) {
            std::swap(W.Z.p, W.F.p);
            W.F_first = 0; W.F_r = npos;
            std::swap(W.lam.p, W.Flam.p);
            std::swap(W.lam.n, W.Flam.n);
            W.have_factors = true; W.x_prev_sparse = false;
        }
        if (verify) {
            if (W.sg_cmp.n < 2) W.sg_cmp.alloc(2);
            W.sg_cmp.zero(stream);
            hipLaunchKernelGGL(dev::k_maxdiff, dim3(grid_for(W.N)), dim3(dev::TPB), 0, stream, (const double*)xo,
                               (const double*)W.sg_out.p, (long long)W.N, W.sg_cmp.p);
            double h[2] = {0.0, 0.0};
            W.sg_cmp.download(h, 2, stream);
            PX_HIP(hipStreamSynchronize(stream));
            W.lst.full_eigs_lanczos_checks++;
            if (!(h[0] <= 1e-8 * std::max(h[1], 1e-300))) {
                W.lst.full_eigs--; W.lst.full_eigs_lanczos--;
                use_reference_result();
            }
        }
        return true;
    }
    W.use_fop = false;
    return false;
}

inline void Solver::psd_projection(double* x) {
    std::fill(min_eig.begin(), min_eig.end(), 0.0);
    if (!one_blocks.empty()) {
        hipLaunchKernelGGL(dev::k_clamp_scalars, dim3(ceil_div(one_blocks.size(), 256)), dim3(256), 0, stream,
                           x, one_off.p, (int)one_blocks.size(), one_min.p);
        for (int idx : one_blocks) current_rank[idx] = 0;
    }
    if (!small_blocks.empty()) {
        project_small_blocks(x);
        project_blocks(large_blocks, x, x, false);
    } else {
        project_blocks(big_blocks, x, x, false);
    }
}

// all small PSD blocks in one launch (kernels.hip.hpp k_small_psd_project); ranks come back with the
// iteration's next synchronisation (pinned buffer)
inline void Solver::project_small_blocks(double* x) {
    harvest_small_ranks();
    const int nb = (int)small_blocks.size();
    // ranks, positive counts and the schedule-test outcomes [nb each] go STRAIGHT into pinned host memory (no copy launch)
    // (small models only, as the scalars of the linesearch: zero_copy_small; otherwise through the device buffer and one copy)
    const bool zc = zero_copy_small();
    int* rk_host = zc ? reinterpret_cast<int*>(small_rank_host.p) : small_rank.p;
    if (zc) for (int q = 0; q < nb; ++q) rk_host[2 * nb + q] = 0;          // (the previous launch's values were harvested above)
    else PX_HIP(hipMemsetAsync(small_rank.p + 2 * nb, 0, (size_t)nb * sizeof(int), stream));
    bool any_jacobi = false;
    int jac_maxn = 0;
    for (int idx : small_blocks) if (P.blocks[idx].n <= small_jacobi_max) { any_jacobi = true; jac_maxn = std::max(jac_maxn, P.blocks[idx].n); }
    if (any_jacobi) {
        const int ld = jac_maxn | 1;
        const size_t lds = ((size_t)2 * jac_maxn * ld + 64) * sizeof(double) + 64 * sizeof(int);
        hipLaunchKernelGGL(dev::k_small_psd_project, dim3(nb), dim3(dev::TPB), lds, stream,
                           x, (const long long*)small_off.p, (const int*)small_side.p, opt.tol_psd, rk_host, rk_host + nb,
                           2, small_jacobi_max);
    }
    if (small_sign_maxn > 0) {
        // shortened, tested schedule as Solver::full_eig_by_sign: start at row 8 (sign_start_row overrides), fall back inside the kernel
        static const dev::SignSchedule sched;
        int jmax = 0;
        for (int j = 1; j + 2 < dev::SIGN_STEPS; ++j) if (std::sqrt(4e-13 * (double)dev::SS_MAXN) / sched.gain[j] <= 1e-10) jmax = j;   // (the test's tau at the largest side)
        const int j0 = std::max(0, std::min(opt.sign_start_row < 0 ? 8 : (int)opt.sign_start_row, jmax));
        int rfail = 0;
        for (int k = 0; k < dev::SIGN_STEPS; ++k) if (sched.l[k] <= 1e-10 * sched.gain[j0]) rfail = k;
        hipLaunchKernelGGL(dev::k_small_sign_project, dim3(nb), dim3(dev::SS_TPB), dev::small_sign_lds_bytes(small_sign_maxn), stream,
                           x, (const long long*)small_off.p, (const int*)small_side.p, small_jacobi_max + 1, dev::SS_MAXN,
                           rk_host, rk_host + nb, j0, rfail, j0 > 0 ? rk_host + 2 * nb : (int*)nullptr);
        st.full_eigs_sign += nb; st.sign_products += 57LL * nb;      // (counted per block below for the Jacobi ones)
        for (int idx : small_blocks) if (P.blocks[idx].n <= small_jacobi_max) { st.full_eigs_sign--; st.sign_products -= 57; }
    }
    if (!zc) PX_HIP(hipMemcpyAsync(small_rank_host.p, small_rank.p, (size_t)3 * nb * sizeof(int), hipMemcpyDeviceToHost, stream));
    small_pending = true;
    st.full_eigs += nb; st.batched_small_eigs += nb;
    for (int idx : small_blocks) { current_rank[idx] = 0; min_eig[idx] = 0.0; }
}
// (called after a stream synchronisation has happened since the launch: every iteration ends with one)
inline void Solver::harvest_small_ranks() {
    if (!small_pending) return;
    small_pending = false;
    const int* r = reinterpret_cast<const int*>(small_rank_host.p);
    for (size_t q = 0; q < small_blocks.size(); ++q) {
        if (r[q] < 0) throw HipError("sign-function projection of a small block produced non-finite values");
        current_rank[small_blocks[q]] = r[q];
    }
    // the shortened schedule's tests made inside k_small_sign_project (cumulative counters behind the ranks)
    const int* flag = r + 2 * small_blocks.size();            // per block: 1 = test passed, 2 = failed (fall-back rows ran), 0 = no test
    for (size_t q = 0; q < small_blocks.size(); ++q) { st.sign_short_pass += flag[q] == 1; st.sign_short_fail += flag[q] == 2; }
}

// primal_step! (pdhg.jl:611-637)
inline void Solver::primal_step_dev() {
    if (use_support) {
        // x_k (buffer xc) becomes the matrix to project IN PLACE on the support; the projection
        // is written to the other buffer, so off the support buffer xc still holds x_k = x_old
        double* xcur = xbuf[xc].p;
        double* xnew = xbuf[1 - xc].p;
        hipLaunchKernelGGL(dev::k_primal_update_S, dim3(ceil_div(std::max(ns, 1), dev::TPB)), dim3(dev::TPB), 0, stream,
                           xcur, supp_d.p, MtyS_cur.p, cS_d.p, primal_step, xsave_d.p, ns, esv_d.p);
        std::fill(min_eig.begin(), min_eig.end(), 0.0);
        double t0 = now_s();
        // block-sharded solve: a shard whose projection fails (e.g. non-finite input) must still join this
        // iteration's collectives, or its peers wait in them for ever: the error is kept, a flag travels with the
        // iteration's scalar record (linesearch_residual_support), and every shard aborts after that reduce
        if (sharded()) {
            try {
                if (opt.debug_fail_iteration > 0 && iter == opt.debug_fail_iteration)
                    throw std::runtime_error("injected projection failure (options.debug_fail_iteration)");
                project_blocks(big_blocks, xcur, xnew, true);
            } catch (...) { shard_error = std::current_exception(); }
        } else {
            project_blocks(big_blocks, xcur, xnew, true);
        }
        st.t_psd += now_s() - t0;
        if (P.sdplen < P.n)
            hipLaunchKernelGGL(dev::k_tail_copy_res, dim3(n_res_wg - tile_base.back()), dim3(dev::TPB), 0, stream,
                               xcur, xnew, (long long)P.sdplen, (long long)P.n, mask_d.p,
                               respart_d.p + tile_base.back(), rstride);
        spmv(xnew, Mxbuf[1 - mxc].p);
        if (!coup_rows.empty()) reduce_coupling(Mxbuf[1 - mxc].p);
        return;
    }
    const double* xi = xbuf[xc].p;
    double* xo = xbuf[1 - xc].p;
    hipLaunchKernelGGL(dev::k_primal_update, dim3(grid_for(P.n)), dim3(dev::TPB), 0, stream,
                       xo, xi, Mtybuf[mtyc].p, c_d.p, primal_step, (long long)P.n);
    if (!P.blocks.empty()) {
        double t0 = now_s();
        psd_projection(xo);
        st.t_psd += now_s() - t0;
    }
    if (!P.socs.empty())
        hipLaunchKernelGGL(dev::k_soc_project, dim3((int)P.socs.size()), dim3(dev::TPB), 0, stream,
                           xo, soc_off.p, soc_len.p);
    spmv(xo, Mxbuf[1 - mxc].p);
}

// linesearch! (pdhg.jl:532-582)
inline int Solver::linesearch() {
    primal_step = primal_step * std::sqrt(1.0 + theta);
    const int gq = std::min(PSTRIDE, grid_for(std::max<int64_t>(P.Q, 1)));
    const int gx = std::min(PSTRIDE, grid_for(P.n));
    int trials = 0;
    for (int i = 0; i < opt.max_linsearch_steps; ++i) {
        ++trials;
        theta = primal_step / primal_step_old;
        const double bt = beta * primal_step;
        hipLaunchKernelGGL(dev::k_dual_trial, dim3(gq), dim3(dev::TPB), 0, stream,
                           ybuf[yc].p, Mxbuf[1 - mxc].p, Mxbuf[mxc].p, bh_d.p, (int)P.p, (int)P.Q, bt, theta,
                           ybuf[1 - yc].p, part.p, 1);
        hipLaunchKernelGGL(dev::k_spmv_csc_norm, dim3(gx), dim3(dev::TPB), 0, stream,
                           csc_ptr.p, csc_row.p, csc_val.p, ybuf[1 - yc].p, Mtybuf[1 - mtyc].p, Mtybuf[mtyc].p,
                           (long long)P.n, part.p + PSTRIDE, 1);
        hipLaunchKernelGGL(dev::k_combine, dim3(1), dim3(dev::TPB), 0, stream,
                           part.p, PSTRIDE, PSTRIDE, 2, 0u, scal.p);
        // the residual / gap reductions of THIS candidate ride behind its trial (pure reductions over the candidate's
        // y and M'y): when it is the accepted one -- the common case -- residual_and_gap finds its nine scalars in the
        // same read-back and the iteration has one synchronisation less; a rejected candidate's are ignored
        enqueue_residual(primal_step, beta * primal_step);
        if (hscal_pin.p == nullptr) hscal_pin.alloc(64);
        PX_HIP(hipMemcpyAsync(hscal_pin.p, scal.p, 11 * sizeof(double), hipMemcpyDeviceToHost, stream));
        wait_stream();
        std::copy(hscal_pin.p, hscal_pin.p + 11, hscal.begin());
        residual_ready = true;
        const double y_norm = std::sqrt(hscal[0]), Mty_norm = std::sqrt(hscal[1]);
        if (debug && iter <= 5 && trials <= 6)
            std::fprintf(stderr, "[dbg] it %lld trial %d tau %.6e theta %.6e y_norm %.6e Mty_norm %.6e\n",
                         iter, trials, primal_step, theta, y_norm, Mty_norm);
        if (std::sqrt(beta) * primal_step * Mty_norm <= opt.delta * y_norm) break;
        primal_step *= opt.linsearch_decay;
        residual_ready = false;                  // (also when the loop ends on its trial limit: the reference then
                                                 //  continues with the DECAYED step and the last candidate's y)
    }
    primal_step_old = primal_step;
    dual_step = beta * primal_step;
    st.linesearch_trials += trials;
    return trials;
}

// dual_step! (pdhg.jl:584-609): the same kernels with theta = 1, bt = dual_step
inline void Solver::dual_step_plain() {
    const int gq = std::min(PSTRIDE, grid_for(std::max<int64_t>(P.Q, 1)));
    const int gx = std::min(PSTRIDE, grid_for(P.n));
    hipLaunchKernelGGL(dev::k_dual_trial, dim3(gq), dim3(dev::TPB), 0, stream,
                       ybuf[yc].p, Mxbuf[1 - mxc].p, Mxbuf[mxc].p, bh_d.p, (int)P.p, (int)P.Q, dual_step, 1.0,
                       ybuf[1 - yc].p, part.p, 0);
    if (P.dense())
        dense_mtv(1, ybuf[1 - yc].p, 0, true, Mtybuf[1 - mtyc].p, 0, Mtybuf[mtyc].p, nullptr, part.p + PSTRIDE, 0, false);
    else
    hipLaunchKernelGGL(dev::k_spmv_csc_norm, dim3(gx), dim3(dev::TPB), 0, stream,
                       csc_ptr.p, csc_row.p, csc_val.p, ybuf[1 - yc].p, Mtybuf[1 - mtyc].p, Mtybuf[mtyc].p,
                       (long long)P.n, part.p + PSTRIDE, 0);
    primal_step_old = primal_step;
    st.linesearch_trials += 1;
}

// compute_residual! + compute_gap! (residuals.jl:2-71), then the *_old rotation
// (:65-68) as index flips instead of four vector copies
inline void Solver::enqueue_residual(double pstep, double dstep) {
    const int gq = std::min(PSTRIDE, grid_for(std::max<int64_t>(P.Q, 1)));
    const int gx = std::min(PSTRIDE, grid_for(P.n));
    const double xold_coef = (iter == 1 && opt.advanced_initialization) ? 0.0 : 1.0;   // x_old = 0 at k = 1
    hipLaunchKernelGGL(dev::k_residual_x, dim3(gx), dim3(dev::TPB), 0, stream,
                       xbuf[1 - xc].p, xbuf[xc].p, xold_coef, Mtybuf[1 - mtyc].p, Mtybuf[mtyc].p, c_d.p,
                       pstep, (long long)P.n, part.p + 2 * PSTRIDE);
    // k_residual_x writes 3 quantities with stride gridDim; re-stride into the common layout
    // by launching with exactly PSTRIDE-strided output: handled by passing gridDim == gx and
    // combining with stride gx (see k_combine call below).
    hipLaunchKernelGGL(dev::k_residual_y, dim3(gq), dim3(dev::TPB), 0, stream,
                       ybuf[1 - yc].p, ybuf[yc].p, Mxbuf[1 - mxc].p, Mxbuf[mxc].p, bh_d.p, (int)P.p, (int)P.Q,
                       dstep, part.p + 5 * PSTRIDE);
    hipLaunchKernelGGL(dev::k_combine, dim3(1), dim3(dev::TPB), 0, stream,
                       part.p + 2 * PSTRIDE, gx, gx, 3, 0x3u, scal.p + 2);
    hipLaunchKernelGGL(dev::k_combine, dim3(1), dim3(dev::TPB), 0, stream,
                       part.p + 5 * PSTRIDE, gq, gq, 6, 0xFu, scal.p + 5);
}

inline void Solver::residual_and_gap() {
    if (!residual_ready) {
        enqueue_residual(primal_step, dual_step);
        if (hscal_pin.p == nullptr) hscal_pin.alloc(64);
        PX_HIP(hipMemcpyAsync(hscal_pin.p + 2, scal.p + 2, 9 * sizeof(double), hipMemcpyDeviceToHost, stream));
        wait_stream();
        std::copy(hscal_pin.p + 2, hscal_pin.p + 11, hscal.begin() + 2);
    }
    residual_ready = false;
    const double* s = hscal.data() + 2;
    if (debug && iter <= 5)
        std::fprintf(stderr, "[dbg] it %lld res: %.6e %.6e cx %.6e | %.6e %.6e eq %.6e in %.6e by %.6e hy %.6e\n",
                     iter, s[0], s[1], s[2], s[3], s[4], s[5], s[6], s[7], s[8]);
    const double pres = std::sqrt(g_n) * s[0] / std::max({s[1], g_norm_b, g_norm_h, 1.0});
    const double dres = std::sqrt(g_Q) * s[3] / std::max({s[4], g_norm_c, 1.0});
    h_pres.at(iter) = pres;
    h_dres.at(iter) = dres;
    h_comb.at(iter) = std::max(pres, dres);
    if (P.p > 0) equa_feasibility = s[5] / (1.0 + g_norm_b);
    if (P.m > 0) ineq_feasibility = s[6] / (1.0 + g_norm_h);
    h_feas.at(iter) = std::max(equa_feasibility, ineq_feasibility);
    const double po = s[2];
    double d_o = 0.0;
    if (P.p > 0) d_o -= s[7];
    if (P.m > 0) d_o -= s[8];
    h_pobj.at(iter) = po;
    h_dobj.at(iter) = d_o;
    h_gap.at(iter) = std::fabs(po - d_o) / (1.0 + std::fabs(po) + std::fabs(d_o));
    xc = 1 - xc; mtyc = 1 - mtyc; yc = 1 - yc; mxc = 1 - mxc;
}

// convergedrank (residuals.jl:88-101)
inline bool Solver::convergedrank() const {
    for (size_t idx = 0; idx < P.blocks.size(); ++idx) {
        if (!(P.blocks[idx].n < opt.min_size_krylov_eigs ||
              target_rank[idx] > opt.max_target_rank_krylov_eigs || min_eig[idx] < opt.tol_psd))
            return false;
    }
    return true;
}

// soc_convergence (residuals.jl:73-86)
inline bool Solver::soc_convergence() {
    if (P.socs.empty()) return true;
    hipLaunchKernelGGL(dev::k_soc_gap, dim3((int)P.socs.size()), dim3(dev::TPB), 0, stream,
                       xbuf[xc].p, soc_off.p, soc_len.p, soc_gap_d.p);
    std::vector<double> g(P.socs.size());
    soc_gap_d.download(g.data(), g.size(), stream);
    PX_HIP(hipStreamSynchronize(stream));
    for (double v : g) if (v >= opt.tol_soc) return false;
    return true;
}

// certificate_parameters (pdhg.jl:670-676)
inline void Solver::certificate_parameters() {
    certificate_search_min_iter = iter + 2 * opt.convergence_window + iter / 5 + 1000;
    certificate_search = true;
    time_limit *= 1.1;
    max_iter_local = max_iter_local + max_iter_local / 10;
}

// the rank-update rule shared by pdhg.jl:270-280 and :294-301
inline void Solver::bump_rank(int idx) {
    if (current_rank[idx] + opt.rank_slack >= target_rank[idx]) {
        if (min_eig[idx] > opt.tol_psd) {
            long long side = P.blocks[idx].n;
            if (opt.rank_increment == 0)
                target_rank[idx] = std::min<long long>(opt.rank_increment_factor * target_rank[idx], side);
            else
                target_rank[idx] = std::min<long long>(opt.rank_increment_factor + target_rank[idx], side);
        }
    }
}

// get_duals + dual_feas + cone_feas (pdhg.jl:678-732) on host vectors; the
// dual-cone eigenvalues come from the device (rocSOLVER, values only).
inline double Solver::dual_feas_host(const std::vector<double>& y, const std::vector<double>& cvec,
                                     std::vector<double>* dual_eq, std::vector<double>* dual_in,
                                     std::vector<double>* dual_cone_out) {
    std::vector<double> dc(cvec);
    if (P.dense()) {                                          // c + M'y with the caller's (unscaled) M
        DevBuf<double> yd(std::max<int64_t>(P.Q, 1)), cd(P.n), od(P.n);
        yd.upload(y.data(), P.Q, stream); cd.upload(cvec.data(), P.n, stream);
        dense_mtv(1, yd.p, 0, false, od.p, 0, nullptr, cd.p, nullptr, 0, false);
        od.download(dc.data(), P.n, stream);
        PX_HIP(hipStreamSynchronize(stream));
    }
    for (int64_t k = 0; k < P.n; ++k) {
        double acc = 0.0;
        for (int64_t q = P.colptr[k]; q < P.colptr[k + 1]; ++q) acc += P.val_orig[q] * y[P.rowidx[q]];
        dc[k] += acc;
        if (P.offdiag[k]) dc[k] /= 2.0;                       // fix_diag_scaling(dual_cone, cones, 2.0)
    }
    double ineq_viol = 0.0;
    if (P.m > 0) {
        double mn = y[P.p];
        for (int64_t i = 0; i < P.m; ++i) mn = std::min(mn, y[P.p + i]);
        ineq_viol = -std::min(0.0, mn);
    }
    double sdp_viol = 0.0;
    for (size_t idx = 0; idx < P.blocks.size(); ++idx) {
        const BlockInfo& B = P.blocks[idx];
        if (B.n == 1) {
            sdp_viol = std::max(sdp_viol, -std::min(0.0, dc[B.off]));
        } else {
            EigWork& W = eig[idx];
            // psd_vec_to_square(v, a, cones, sqrt(2)): off-diagonals / sqrt(2), i.e. smat(dc).
            // The reference takes minimum(eigen!(...)) of the full spectrum (pdhg.jl:685, one
            // O(n^3) dsyevr inside SolveTimeSec); only lambda_min is used, so for Lanczos-sized
            // blocks it is computed as -lambda_max(-smat(dc)) with the projection's own Lanczos
            // (a few hundred mat-vecs), falling back to the dense eigensolver if that does not
            // converge.  This also keeps rocSOLVER's 3-5 s one-off code-object load out of
            // solves that never take the full_eig! path.
            DevBuf<double> tmp(B.N);
            double mn = 0.0;
            bool have_mn = false;
            if (B.n > opt.min_size_krylov_eigs && opt.eigsolver != 1 && W.cap >= 26 && krylovdim_fits(1)) {
                std::vector<double> neg(dc.begin() + B.off, dc.begin() + B.off + B.N);
                for (double& v : neg) v = -v;
                tmp.upload(neg.data(), B.N, stream);
                const proxsdp_stats keep_st = W.lst;
                const long long keep_mv = W.mv_iter;
                const int keep_prev = W.prev_numiter, keep_num = W.numiter;
                W.use_fop = false;
                lanczos(W, tmp.p, 1);
                if (W.converged && W.converged_eigs >= 1 && !W.vals.empty()) { mn = -W.vals[0]; have_mn = true; }
                const long long mv = W.lst.lanczos_matvecs - keep_st.lanczos_matvecs;
                W.lst = keep_st;                               // exit-path work is not a projection
                st.exit_matvecs += mv;
                W.mv_iter = keep_mv;
                W.prev_numiter = keep_prev; W.numiter = keep_num;
                PX_HIP(hipStreamSynchronize(stream));          // `neg` goes out of scope
            }
            if (!have_mn) {
                tmp.upload(dc.data() + B.off, B.N, stream);
                std::vector<double> D;
                full_eig_values(W, tmp.p, dev::INV_SQRT2, false, D);
                mn = D[0];
                for (double v : D) mn = std::min(mn, v);
            }
            sdp_viol = std::max(sdp_viol, -std::min(0.0, mn));
        }
    }
    for (const SocInfo& S : P.socs) {
        double ss = 0.0;
        for (int i = 1; i < S.len; ++i) ss += dc[S.off + i] * dc[S.off + i];
        sdp_viol = std::max(sdp_viol, -std::min(0.0, dc[S.off] - std::sqrt(ss)));
    }
    double zero_viol = 0.0;
    for (int64_t k = P.conelen; k < P.n; ++k) zero_viol = std::max(zero_viol, std::fabs(dc[k]));
    if (dual_eq) dual_eq->assign(y.begin(), y.begin() + P.p);
    if (dual_in) dual_in->assign(y.begin() + P.p, y.end());
    if (dual_cone_out) *dual_cone_out = std::move(dc);
    return std::max({sdp_viol, ineq_viol, zero_viol});
}

// cache_solution (pdhg.jl:745-787).  NB the reference rescales pair.x IN PLACE
// (fix_diag_scaling on pair.x, :749); when a certificate search continues after
// this snapshot the iterate stays rescaled -- reproduced here on the device.
inline void Solver::cache_solution(const std::vector<double>& cvec) {
    double t0 = now_s();
    const double inv = 1.0 / std::sqrt(2.0);
    for (size_t idx = 0; idx < P.blocks.size(); ++idx) {
        const BlockInfo& B = P.blocks[idx];
        if (B.n < 2) continue;
        const int nt = ceil_div(B.n, dev::TILE);
        hipLaunchKernelGGL(dev::k_scale_offdiag, dim3(nt * (nt + 1) / 2), dim3(dev::TPB), 0, stream,
                           xbuf[xc].p + B.off, B.n, inv);
    }
    if (P.equilibrated) {                                    // pair.x = D*pair.x; pair.y = E*pair.y (pdhg.jl:751-755)
        if (Ddiag_d.n == 0) {
            Ddiag_d.alloc(P.n); Ediag_d.alloc(std::max<int64_t>(P.Q, 1));
            Ddiag_d.upload(P.Ddiag.data(), P.n, stream); Ediag_d.upload(P.Ediag.data(), P.Q, stream);
        }
        hipLaunchKernelGGL(dev::k_scale_by, dim3(grid_for(P.n)), dim3(dev::TPB), 0, stream, xbuf[xc].p, Ddiag_d.p, (long long)P.n);
        hipLaunchKernelGGL(dev::k_scale_by, dim3(grid_for(std::max<int64_t>(P.Q, 1))), dim3(dev::TPB), 0, stream,
                           ybuf[yc].p, Ediag_d.p, (long long)P.Q);
    }
    std::vector<double> x(P.n), y(P.Q);
    xbuf[xc].download(x.data(), P.n, stream);
    ybuf[yc].download(y.data(), P.Q, stream);
    PX_HIP(hipStreamSynchronize(stream));
    std::vector<double> slack(P.Q, 0.0);
    if (P.dense() && P.p > 0) {                              // A x with the caller's (unscaled) dense A
        DevBuf<double> sd(P.p);
        dense_mv(xbuf[xc].p, sd.p, false);
        sd.download(slack.data(), P.p, stream);
        PX_HIP(hipStreamSynchronize(stream));
    }
    for (int64_t k = 0; k < P.n; ++k) {
        const double xk = x[k];
        for (int64_t q = P.colptr[k]; q < P.colptr[k + 1]; ++q) slack[P.rowidx[q]] += P.val_orig[q] * xk;
    }
    if (!coup_rows.empty()) {                                 // M x of a coupling row: sum of the shards' partials
        std::vector<double> part_(coup_rows.size());
        for (size_t k = 0; k < coup_rows.size(); ++k) part_[k] = slack[coup_rows[k]];
        reduce_vec_host(part_);
        for (size_t k = 0; k < coup_rows.size(); ++k) slack[coup_rows[k]] = part_[k];
    }
    std::vector<double> deq, din, dcone;
    double dfeas = dual_feas_host(y, cvec, &deq, &din, &dcone);
    harvest_small_ranks();
    long long fr = 0;
    for (long long r : current_rank) fr += r;
    if (sharded()) {                         // dual feasibility and rank over all shards
        std::vector<double> sums = {(double)fr}, maxs = {dfeas};
        reduce(sums, maxs);
        fr = (long long)std::llround(sums[0]);
        dfeas = maxs[0];
    }
    res.status = stop_reason;
    merge_block_stats();
    // (a caller who forces equilibration gets the reference's aliased -- not contractive -- scaling iteration by default: say so)
    std::snprintf(res.status_string, sizeof(res.status_string), "%s%s%s", stop_reason_string.c_str(),
                  st.dense_truncated_projections > 0 ? " [Krylov dimension > 255: dense eigensolver served those projections]" : "",
                  (P.equilibrated && opt.equilibration_reference_aliasing) ? " [equilibration: the reference's aliased scaling, equilibration_reference_aliasing = 1]" : "");
    if (res.primal)    for (int64_t i = 0; i < P.n; ++i) res.primal[i] = x[P.inv[i]];
    if (res.dual_cone) for (int64_t i = 0; i < P.n; ++i) res.dual_cone[i] = dcone[P.inv[i]];
    if (res.dual_eq)   for (int64_t i = 0; i < P.p; ++i) res.dual_eq[i] = deq[i];
    if (res.dual_in)   for (int64_t i = 0; i < P.m; ++i) res.dual_in[i] = din[i];
    if (res.slack_eq)  for (int64_t i = 0; i < P.p; ++i) res.slack_eq[i] = slack[i] - P.b_orig[i];
    if (res.slack_in)  for (int64_t i = 0; i < P.m; ++i) res.slack_in[i] = slack[P.p + i] - P.h_orig[i];
    res.primal_residual = equa_feasibility;
    res.dual_residual = ineq_feasibility;
    res.objval = h_pobj.at(iter);
    res.dual_objval = h_dobj.at(iter);
    res.gap = h_gap.at(iter);
    res.iter = iter;
    res.final_rank = (int32_t)fr;
    res.primal_feasible_user_tol = h_feas.at(iter) <= opt.tol_feasibility;
    res.dual_feasible_user_tol = dfeas <= opt.tol_feasibility_dual;
    res.dual_feasibility = dfeas;
    res.certificate_found = certificate_found;
    res.result_count = 1;
    res.time = now_s() - time0;
    have_snapshot = true;
    // x was rescaled IN PLACE above (the reference quirk, pdhg.jl:749-755).  When a certificate
    // search continues from here, the operator-form factors F diag(Flam) F' still describe the
    // UNscaled iterate: drop them so the next projection reads the (rescaled) packed buffer, as the
    // fused residual and the dense / full-eig paths do.
    for (EigWork& W : eig) { W.have_factors = false; W.x_prev_sparse = false; }
    st.exit_time += now_s() - t0;
}

// ---- dense constraint matrix (proxsdp_problem.M_dense; kernels.hip.hpp "Dense constraint matrix")
inline void Solver::setup_dense() {
    if (!P.dense()) return;
    const size_t cnt = (size_t)P.p * (size_t)P.n;
    if (P.Mdense_on_device) {
        Md = P.Mdense;                                   // borrowed, read-only
    } else {
        Md_own.alloc(std::max<size_t>(cnt, 1));
        const size_t chunk = (size_t)1 << 27;            // 1 GiB of doubles per copy
        for (size_t o = 0; o < cnt; o += chunk)
            PX_HIP(hipMemcpy(Md_own.p + o, P.Mdense + o, std::min(chunk, cnt - o) * sizeof(double), hipMemcpyHostToDevice));
        Md = Md_own.p;
    }
    offdiag_d.alloc(P.n);
    offdiag_d.upload(P.offdiag.data(), P.n, stream);
    // M x: row groups x column slices; enough workgroups to fill 256 CUs several times over
    const int rg = ceil_div((int)std::max<int64_t>(P.p, 1), DMV_ROWS);
    dmv_slices = std::max(1, std::min(64, ceil_div(2048, rg)));
    const long long chunks = ((long long)P.n + dev::TPB * DMV_UNR - 1) / (dev::TPB * DMV_UNR);
    dmv_slices = (int)std::min<long long>(dmv_slices, std::max<long long>(chunks, 1));
    dmv_qpad = (int)std::max<int64_t>(P.p, 1);
    dmv_part.alloc((size_t)dmv_slices * dmv_qpad);
    Mtycand_d.alloc((size_t)3 * P.n);
    ycand_d.alloc((size_t)4 * std::max<int64_t>(P.Q, 1));
    bpart.alloc((size_t)4 * 2 * PSTRIDE); bpart.zero(stream);
    bscal.alloc(64); bscal.zero(stream);
    hbscal.assign(64, 0.0);
    // ||M||_F of the column-scaled matrix (pdhg.jl:121 after norm_scaling)
    const int gx = std::min(PSTRIDE, grid_for(P.n));
    hipLaunchKernelGGL(dev::k_dense_frob, dim3(gx), dim3(dev::TPB), 0, stream,
                       Md, (long long)P.n, (int)P.p, (long long)P.n, offdiag_d.p, std::sqrt(2.0) / 2.0, part.p);
    hipLaunchKernelGGL(dev::k_combine, dim3(1), dim3(dev::TPB), 0, stream, part.p, PSTRIDE, gx, 1, 0u, scal.p);
    double ss = 0.0;
    PX_HIP(hipMemcpyAsync(&ss, scal.p, sizeof(double), hipMemcpyDeviceToHost, stream));
    PX_HIP(hipStreamSynchronize(stream));
    g_frob = std::sqrt(ss + P.frob * P.frob);            // + the sparse rows (G)
}

// every pass over the dense A is bracketed by events (8 ms each at the BASELINE size: the
// bracketing is free); harvested after the iteration's last synchronisation
inline void Solver::dense_ev_begin() {
    if (dense_ev_used == dense_ev.size()) {
        hipEvent_t a, b;
        PX_HIP(hipEventCreate(&a)); PX_HIP(hipEventCreate(&b));
        dense_ev.emplace_back(a, b);
    }
    PX_HIP(hipEventRecord(dense_ev[dense_ev_used].first, stream));
}
inline void Solver::dense_ev_end() {
    PX_HIP(hipEventRecord(dense_ev[dense_ev_used].second, stream));
    ++dense_ev_used;
}
inline void Solver::dense_ev_harvest() {
    for (size_t i = 0; i < dense_ev_used; ++i) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, dense_ev[i].first, dense_ev[i].second) == hipSuccess) st.dense_ms += ms;
    }
    dense_ev_used = 0;
}

// y = M (s o x)  (scaled = the solver's M) or M x (the caller's M, exit path)
inline void Solver::dense_mv(const double* x, double* y, bool scaled) {
    if (P.p == 0) return;
    const int rg = ceil_div((int)P.p, DMV_ROWS);
    dense_ev_begin();
    auto lmv = [&](auto kern) {
        hipLaunchKernelGGL(kern, dim3(rg, dmv_slices), dim3(dev::TPB), 0, stream,
                           Md, (long long)P.n, (int)P.p, (long long)P.n, x, scaled ? offdiag_d.p : nullptr,
                           std::sqrt(2.0) / 2.0, dmv_part.p, dmv_qpad);
    };
    // shape sweep at n=2000, m=4000 (tools/gpurun_dense_tune.py): <4,4> 6.41, <8,4> 6.55, <4,8> 6.20,
    // <2,8> 6.20, <8,2> 6.51 TB/s
    lmv(dev::k_dense_mv<DMV_ROWS, DMV_UNR>);
    hipLaunchKernelGGL(dev::k_dense_mv_fin, dim3(ceil_div((int)P.p, dev::TPB)), dim3(dev::TPB), 0, stream,
                       dmv_part.p, dmv_qpad, dmv_slices, (int)P.p, y);
    dense_ev_end();
    st.dense_passes += 1;
}

// OUT_c = s o (M' Y_c) [+ addc], c < nc <= 3, one pass over M
inline void Solver::dense_mtv(int nc, const double* Y, long long ystride, bool scaled, double* OUT, long long ostride,
                              const double* old, const double* addc, double* normpart, long long cstride, bool addback) {
    int gx = std::min(PSTRIDE, grid_for(P.n));
    if (gx >= 8) gx &= ~7;                               // multiple of 8: XCD-aware chunk order in the kernel
    const unsigned char* od = scaled ? offdiag_d.p : nullptr;
    const double sc = std::sqrt(2.0) / 2.0;
    auto launch = [&](auto kern) {
        hipLaunchKernelGGL(kern, dim3(gx), dim3(dev::TPB), 0, stream, Md, (long long)P.n, (int)P.p, (long long)P.n,
                           Y, ystride, od, sc, OUT, ostride, old, addc, normpart, cstride,
                           scaled && P.nnz > 0 ? csc_ptr.p : nullptr, csc_row.p, csc_val.p, addback ? 1 : 0);
    };
    dense_ev_begin();
    if (nc == 1) launch(dev::k_dense_mtv<1, 1, 8>);
    else if (nc == 2) launch(dev::k_dense_mtv<2, 1, 8>);
    else launch(dev::k_dense_mtv<3, 1, 8>);      // <3,1,16> 6.41, <3,2,8> 6.39, <3,4,4> 6.47, <3,2,4> 6.37 TB/s: flat
    dense_ev_end();
    st.dense_passes += 1;
}

// linesearch! (pdhg.jl:532-582) with a dense M: the candidates tau, 3/4 tau, (3/4)^2 tau are
// evaluated by ONE pass over M (8 Q n bytes) and one synchronisation; the host takes the first
// candidate the reference's loop would have accepted.
inline int Solver::linesearch_dense() {
    constexpr int NC = 3;
    const int gq = std::min(PSTRIDE, grid_for(std::max<int64_t>(P.Q, 1)));
    const int gx = std::min(PSTRIDE, grid_for(P.n));
    const long long cstride = 2LL * PSTRIDE;
    const long long ystride = std::max<int64_t>(P.Q, 1);
    primal_step = primal_step * std::sqrt(1.0 + theta);
    int trials = 0;
    bool accepted = false;
    while (!accepted && trials < opt.max_linsearch_steps) {
        dev::TrialBatch tb{};
        double tau_c = primal_step;
        int nc = 0;
        for (; nc < NC && trials + nc < opt.max_linsearch_steps; ++nc) {
            tb.tau[nc] = tau_c;
            tb.theta[nc] = tau_c / primal_step_old;
            tb.bt[nc] = beta * tau_c;
            tb.sigma[nc] = beta * tau_c;
            tau_c *= opt.linsearch_decay;
        }
        tb.nc = nc;
        hipLaunchKernelGGL(dev::k_dual_trial_batch, dim3(gq, nc), dim3(dev::TPB), 0, stream,
                           ybuf[yc].p, Mxbuf[1 - mxc].p, Mxbuf[mxc].p, bh_d.p, (int)P.p, (int)P.Q, tb,
                           ycand_d.p, ystride, bpart.p, cstride);
        dense_mtv(nc, ycand_d.p, ystride, true, Mtycand_d.p, (long long)P.n, Mtybuf[mtyc].p, nullptr,
                  bpart.p + PSTRIDE, cstride, true);
        hipLaunchKernelGGL(dev::k_combine_multi, dim3(nc * 2), dim3(dev::TPB), 0, stream,
                           (const double*)bpart.p, PSTRIDE, std::max(gq, gx), 0ull, bscal.p, nc * 2,
                           (const double*)nullptr, 0, 0, (double*)nullptr);
        PX_HIP(hipMemcpyAsync(hbscal.data(), bscal.p, NC * 2 * sizeof(double), hipMemcpyDeviceToHost, stream));
        PX_HIP(hipStreamSynchronize(stream));
        for (int c = 0; c < nc; ++c) {
            ++trials;
            primal_step = tb.tau[c];
            theta = tb.theta[c];
            const double y_norm = std::sqrt(hbscal[2 * c]), Mty_norm = std::sqrt(hbscal[2 * c + 1]);
            const bool ok = std::sqrt(beta) * primal_step * Mty_norm <= opt.delta * y_norm;
            const bool last = trials >= opt.max_linsearch_steps;
            if (ok || last) {
                if (!ok) primal_step *= opt.linsearch_decay;     // reference quirk: decayed once more, trial kept
                accepted = true;
                PX_HIP(hipMemcpyAsync(ybuf[1 - yc].p, ycand_d.p + (size_t)c * ystride, (size_t)P.Q * 8,
                                      hipMemcpyDeviceToDevice, stream));
                PX_HIP(hipMemcpyAsync(Mtybuf[1 - mtyc].p, Mtycand_d.p + (size_t)c * P.n, (size_t)P.n * 8,
                                      hipMemcpyDeviceToDevice, stream));
                break;
            }
            primal_step = tb.tau[c] * opt.linsearch_decay;
        }
    }
    primal_step_old = primal_step;
    dual_step = beta * primal_step;
    st.linesearch_trials += trials;
    return trials;
}

// ---- support-aware path: setup and the batched linesearch + residual
inline void Solver::setup_support() {
    use_support = false;
    // (without linesearch the support path runs on request -- support_path = 1 -- and inside a block-sharded solve, which is built on it)
    if (opt.support_path == 0 || (!opt.line_search_flag && opt.support_path < 0)) return;
    if (!P.socs.empty() || !one_blocks.empty() || P.blocks.empty()) return;
    std::vector<int> supp;
    for (int64_t k = 0; k < P.n; ++k)
        if (P.colptr[k + 1] > P.colptr[k] || P.c[k] != 0.0) supp.push_back((int)k);
    if (opt.support_path < 0 && (8 * (int64_t)supp.size() > P.n || P.n < 4096)) return;   // auto: only when it pays
    ns = (int)supp.size();
    std::vector<unsigned> mask((size_t)(P.n + 31) / 32 + 1, 0u);
    std::vector<double> cS(std::max(ns, 1), 0.0);
    for (int s = 0; s < ns; ++s) { mask[supp[s] >> 5] |= 1u << (supp[s] & 31); cS[s] = P.c[supp[s]]; }
    supp_d.alloc(std::max(ns, 1)); mask_d.alloc(mask.size()); cS_d.alloc(std::max(ns, 1));
    xsave_d.alloc(std::max(ns, 1)); MtyS_cur.alloc(std::max(ns, 1)); MtyS_cand.alloc((size_t)4 * std::max(ns, 1));
    ycand_d.alloc((size_t)4 * std::max<int64_t>(P.Q, 1));
    supp_d.upload(supp.data(), ns, stream); mask_d.upload(mask.data(), mask.size(), stream);
    cS_d.upload(cS.data(), ns, stream);
    MtyS_cur.zero(stream); MtyS_cand.zero(stream); xsave_d.zero(stream); ycand_d.zero(stream);
    // residual partial slots: one per reconstruction tile of every block + the tail workgroups
    tile_base.clear();
    int base = 0;
    for (const BlockInfo& B : P.blocks) {
        tile_base.push_back(base);
        const int nt = ceil_div(B.n, dev::TILE);
        base += 8 * ceil_div(nt * (nt + 1) / 2, 8);      // reconstruction grid (padded to 8 XCDs)
    }
    tile_base.push_back(base);
    if (P.sdplen < P.n) base += std::min(256, ceil_div(P.n - P.sdplen, dev::TPB));
    n_res_wg = base;
    rstride = base;
    respart_d.alloc((size_t)2 * std::max(base, 1)); respart_d.zero(stream);
    bpart.alloc((size_t)4 * 11 * PSTRIDE); bpart.zero(stream);
    bscal.alloc(64); bscal.zero(stream);
    hbscal.assign(64, 0.0);
    // operator-form mat-vec: the support update as a symmetric sparse matrix per block (ELL)
    esv_d.alloc((size_t)2 * std::max(ns, 1)); esv_d.zero(stream);
    if (opt.lanczos_operator != 0) {
        size_t s0 = 0;
        for (size_t idx = 0; idx < P.blocks.size(); ++idx) {
            const BlockInfo& B = P.blocks[idx];
            EigWork& W = eig[idx];
            size_t s1 = s0;
            while (s1 < supp.size() && supp[s1] < B.off + B.N) ++s1;     // supp is ascending
            if (B.n < 2 || W.npad == 0) { s0 = s1; continue; }
            std::vector<int> cnt(W.npad, 0);
            std::vector<std::array<int, 3>> ent;                 // row, col, s
            ent.reserve(2 * (s1 - s0));
            for (size_t sidx = s0; sidx < s1; ++sidx) {
                const int64_t k = supp[sidx] - B.off;
                int64_t j = (int64_t)((std::sqrt(8.0 * (double)k + 1.0) - 1.0) / 2.0);
                while ((j + 1) * (j + 2) / 2 <= k) ++j;
                while (j * (j + 1) / 2 > k) --j;
                const int i = (int)(k - j * (j + 1) / 2);
                ent.push_back({i, (int)j, (int)sidx}); cnt[i]++;
                if (i != (int)j) { ent.push_back({(int)j, i, (int)sidx}); cnt[j]++; }
            }
            s0 = s1;
            int wmax = 1;
            for (int c : cnt) wmax = std::max(wmax, c);
            // ELL part of at most 32 entries per row; the rest of a wider (hub) row goes to an
            // overflow list that the row's workgroup reduces cooperatively
            const int w = std::min(wmax, 32);
            // measured (tools/gpurun_opcmp.py): below n ~ 2000 the two operators cost the same
            // (latency-bound), and a hub row costs the operator form ~15 us per mat-vec at
            // n = 1000 (maxG51) -- more than streaming a packed triangle of up to ~100 MB.  With
            // hub rows the operator form is therefore only taken for large blocks.
            if (wmax > w && B.n < 6000 && opt.lanczos_operator < 0) continue;
            std::vector<int> col((size_t)w * W.npad), sx((size_t)w * W.npad, -1);
            for (int k = 0; k < w; ++k) for (int i = 0; i < W.npad; ++i) col[(size_t)k * W.npad + i] = i;
            std::vector<std::vector<std::array<int, 2>>> over(W.npad);
            std::fill(cnt.begin(), cnt.end(), 0);
            for (const auto& e : ent) {
                const int kk = cnt[e[0]]++;
                if (kk < w) { const size_t at = (size_t)kk * W.npad + e[0]; col[at] = e[1]; sx[at] = e[2]; }
                else over[e[0]].push_back({e[1], e[2]});
            }
            W.ell_w = w;
            W.ell_col.alloc(col.size()); W.ell_sidx.alloc(sx.size());
            W.ell_col.upload(col.data(), col.size(), stream); W.ell_sidx.upload(sx.data(), sx.size(), stream);
            W.ov = dev::EllOverflow{};
            if (wmax > w) {
                std::vector<int> wptr(W.nt + 1, 0), wrow, wlo, whi, ocol, osx;
                for (int g = 0; g < W.nt; ++g) {
                    for (int rl = 0; rl < dev::LZ_ROWS; ++rl) {
                        const int row = g * dev::LZ_ROWS + rl;
                        if (row >= W.npad || over[row].empty()) continue;
                        wrow.push_back(rl); wlo.push_back((int)ocol.size());
                        for (const auto& e : over[row]) { ocol.push_back(e[0]); osx.push_back(e[1]); }
                        whi.push_back((int)ocol.size());
                    }
                    wptr[g + 1] = (int)wrow.size();
                }
                W.wr_ptr.alloc(wptr.size()); W.wr_row.alloc(wrow.size()); W.wr_lo.alloc(wlo.size()); W.wr_hi.alloc(whi.size());
                W.ov_col.alloc(ocol.size()); W.ov_sidx.alloc(osx.size());
                W.wr_ptr.upload(wptr.data(), wptr.size(), stream); W.wr_row.upload(wrow.data(), wrow.size(), stream);
                W.wr_lo.upload(wlo.data(), wlo.size(), stream); W.wr_hi.upload(whi.data(), whi.size(), stream);
                W.ov_col.upload(ocol.data(), ocol.size(), stream); W.ov_sidx.upload(osx.data(), osx.size(), stream);
                W.ov = dev::EllOverflow{W.wr_ptr.p, W.wr_row.p, W.wr_lo.p, W.wr_hi.p, W.ov_col.p, W.ov_sidx.p};
                PX_HIP(hipStreamSynchronize(stream));             // host vectors go out of scope
            }
            W.F.alloc((size_t)W.npad * W.cap); W.F.zero(stream);
            W.Flam.alloc(std::max(W.n, dev::MAXK)); W.Flam.zero(stream);
            W.tpart.alloc((size_t)dev::MAXK * W.pld); W.tpart.zero(stream);
            W.ebuf.alloc(W.npad); W.ebuf.zero(stream);
            W.apartf.alloc(W.pld); W.apartf.zero(stream);
            // persistent cycle kernel: granule buffers [64 workgroups][CY_CMAX][2], zero = never-valid tag
            W.xg1.alloc((size_t)32 * dev::CY_CMAX); W.xg1.zero(stream);
            W.xg2.alloc((size_t)32 * (dev::CY_CMAX + 128)); W.xg2.zero(stream);           // + the R rows of w'
            W.xf1.alloc(32); W.xf1.zero(stream); W.xf2.alloc(32); W.xf2.zero(stream);     // flags: zero = never-valid epoch
            W.cy_err.alloc(1); W.cy_err.zero(stream);
            W.warm_part.alloc(ceil_div(W.npad, dev::TPB)); W.warm_part.zero(stream);
            PX_HIP(hipStreamSynchronize(stream));                 // host vectors go out of scope
            W.fop_ok = true;
        }
    }
    PX_HIP(hipStreamSynchronize(stream));
    use_support = true;
}

// linesearch! (pdhg.jl:532-582) + compute_residual! + compute_gap! (residuals.jl) on the
// support path: up to 3 consecutive step-size candidates tau, 0.75 tau, 0.75^2 tau are
// evaluated by one batch of small kernels (the dense terms were produced by the fused
// reconstruction), the host reads all scalars with ONE synchronisation and takes the first
// candidate the reference's loop would have accepted.
inline int Solver::linesearch_residual_support() {
    constexpr int NC = 3;
    const int gq = std::min(PSTRIDE, grid_for(std::max<int64_t>(P.Q, 1)));
    const int gs = std::min(PSTRIDE, grid_for(std::max(ns, 1)));
    const long long cstride = 11LL * PSTRIDE;
    const long long ystride = std::max<int64_t>(P.Q, 1), mstride = std::max(ns, 1);
    const double xold_coef = (iter == 1 && opt.advanced_initialization) ? 0.0 : 1.0;
    // line_search_flag = false (round 6; the sharded loop needs this path): ONE candidate, dual_step! (pdhg.jl:584-609) -- y+ = y +
    // sigma (2 Mx - Mx_old) with the solver's own dual_step, accepted as it is; primal_step, dual_step and theta are left alone
    const bool ls = opt.line_search_flag;
    if (ls) primal_step = primal_step * std::sqrt(1.0 + theta);
    int trials = 0;
    bool accepted = false;
    const double* s_acc = nullptr;
    // (the off-support residual maxima of this iteration, independent of the candidate, are
    // reduced by the two extra workgroups of the batch's final combine)
    // block-sharded solve: every shard evaluated the same candidates on its own blocks/rows: combine
    auto reduce_candidates = [&](int nc) {
        if (!sharded()) return;
        std::vector<double> sums, maxs;
        for (int c = 0; c < NC; ++c) {
            const double* sc = hbscal.data() + 11 * c;
            for (int q : {0, 1, 4, 9, 10}) sums.push_back(c < nc ? sc[q] : 0.0);
            for (int q : {2, 3, 5, 6, 7, 8}) maxs.push_back(c < nc ? sc[q] : 0.0);
        }
        maxs.push_back(hbscal[NC * 11]); maxs.push_back(hbscal[NC * 11 + 1]);
        maxs.push_back(convergedrank() ? 0.0 : 1.0);          // any shard not rank-converged
        bool below = false;
        for (size_t idx = 0; idx < P.blocks.size(); ++idx) below = below || target_rank[idx] < P.blocks[idx].n;
        maxs.push_back(below ? 1.0 : 0.0);                    // any block with target_rank < side
        maxs.push_back(now_s() - time0);                      // one clock for the limits
        maxs.push_back(shard_error ? 1.0 : 0.0);              // a shard failed in this iteration's projection
        if (shard_error)                                      // (its own partials may be garbage: keep them finite)
            for (double& v : sums) if (!(v == v)) v = 0.0;
        reduce(sums, maxs);
        if (maxs.back() > 0.5) {
            if (shard_error) { std::exception_ptr e = shard_error; shard_error = nullptr; std::rethrow_exception(e); }
            throw std::runtime_error("another shard of the block-sharded solve failed in this iteration");
        }
        size_t si = 0, mi = 0;
        for (int c = 0; c < NC; ++c) {
            double* sc = hbscal.data() + 11 * c;
            for (int q : {0, 1, 4, 9, 10}) sc[q] = sums[si++];
            for (int q : {2, 3, 5, 6, 7, 8}) sc[q] = maxs[mi++];
        }
        hbscal[NC * 11] = maxs[mi++]; hbscal[NC * 11 + 1] = maxs[mi++];
        g_not_converged_rank = maxs[mi++] > 0.5;
        g_any_below_full = maxs[mi++] > 0.5;
        g_elapsed = maxs[mi++];
    };
    while (!accepted && (!ls || trials < opt.max_linsearch_steps)) {
        dev::TrialBatch tb{};
        double tau_c = primal_step;
        int nc = 0;
        for (; ls && nc < NC && trials + nc < opt.max_linsearch_steps; ++nc) {
            tb.tau[nc] = tau_c;
            tb.theta[nc] = tau_c / primal_step_old;
            tb.bt[nc] = beta * tau_c;
            tb.sigma[nc] = beta * tau_c;
            tau_c *= opt.linsearch_decay;
        }
        if (!ls) { nc = 1; tb.tau[0] = primal_step; tb.theta[0] = 1.0; tb.bt[0] = dual_step; tb.sigma[0] = dual_step; tb.plain = 1; }
        tb.nc = nc;
        hipLaunchKernelGGL(dev::k_dual_trial_batch, dim3(gq, nc), dim3(dev::TPB), 0, stream,
                           ybuf[yc].p, Mxbuf[1 - mxc].p, Mxbuf[mxc].p, bh_d.p, (int)P.p, (int)P.Q, tb,
                           ycand_d.p, ystride, bpart.p, cstride, (const double*)roww_d.p);
        hipLaunchKernelGGL(dev::k_spmvT_S_batch, dim3(gs, nc), dim3(dev::TPB), 0, stream,
                           csc_ptr.p, csc_row.p, csc_val.p, supp_d.p, ns, ycand_d.p, ystride,
                           MtyS_cand.p, mstride, MtyS_cur.p, bpart.p + PSTRIDE, cstride, tb.plain);
        hipLaunchKernelGGL(dev::k_residual_xy_batch, dim3(std::max(gs, gq), nc, 2), dim3(dev::TPB), 0, stream,
                           xbuf[1 - xc].p, supp_d.p, ns, xsave_d.p, xold_coef, MtyS_cand.p, mstride, MtyS_cur.p,
                           cS_d.p, gs,
                           ycand_d.p, ystride, ybuf[yc].p, Mxbuf[1 - mxc].p, Mxbuf[mxc].p, bh_d.p, (int)P.p, (int)P.Q, gq,
                           tb, bpart.p, PSTRIDE, cstride, (const double*)roww_d.p);
        // per candidate: q0,q1 sums | q2,q3 max, q4 sum | q5..q8 max, q9,q10 sum
        unsigned long long ismax = 0;
        for (int c = 0; c < nc; ++c) ismax |= 0x1ECull << (11 * c);      // bits 2,3,5,6,7,8
        // (measured, tools/_ab in round 5: letting this kernel write its scalars straight into pinned host memory -- as the small-model
        // path does -- costs the rank-63 iteration 3 %: a kernel that stores to host memory ends with a system-scope release, and
        // behind the reconstruction that means writing 64 MB of dirty L2 lines back first)
        hipLaunchKernelGGL(dev::k_combine_multi, dim3(nc * 11 + 2), dim3(dev::TPB), 0, stream,
                           (const double*)bpart.p, PSTRIDE, std::max(gq, gs), ismax, bscal.p, nc * 11,
                           (const double*)respart_d.p, rstride, n_res_wg, bscal.p + NC * 11);
        PX_HIP(hipMemcpyAsync(hbscal.data(), bscal.p, (NC * 11 + 2) * sizeof(double), hipMemcpyDeviceToHost, stream));
        wait_stream();
        reduce_candidates(nc);
        if (!ls) {
            trials = 1; accepted = true; s_acc = hbscal.data();
            hipLaunchKernelGGL(dev::k_copy2, dim3(grid_for((long long)P.Q + ns)), dim3(dev::TPB), 0, stream,
                               ybuf[1 - yc].p, (const double*)ycand_d.p, (long long)P.Q,
                               MtyS_cur.p, (const double*)MtyS_cand.p, (long long)ns);
            break;
        }
        for (int c = 0; c < nc; ++c) {
            ++trials;
            const double* sc = hbscal.data() + 11 * c;
            primal_step = tb.tau[c];
            theta = tb.theta[c];
            const double y_norm = std::sqrt(sc[0]), Mty_norm = std::sqrt(sc[1]);
            const bool ok = std::sqrt(beta) * primal_step * Mty_norm <= opt.delta * y_norm;
            const bool last = trials >= opt.max_linsearch_steps;
            if (ok || last) {
                accepted = true;
                s_acc = sc;
                if (!ok) {
                    // reference quirk (pdhg.jl:545-569): max_linsearch_steps exhausted -> the step is
                    // decayed once more while the last trial's y / Mty are kept, and compute_residual!
                    // / compute_gap! then run with THAT primal_step and dual_step.  Re-evaluate the
                    // residual scalars of candidate c with the final steps (rare path: one more batch).
                    primal_step *= opt.linsearch_decay;
                    dev::TrialBatch t1{};
                    t1.nc = 1; t1.tau[0] = primal_step; t1.theta[0] = theta; t1.bt[0] = beta * primal_step;
                    t1.sigma[0] = beta * primal_step;
                    hipLaunchKernelGGL(dev::k_residual_xy_batch, dim3(std::max(gs, gq), 1, 2), dim3(dev::TPB), 0, stream,
                                       xbuf[1 - xc].p, supp_d.p, ns, xsave_d.p, xold_coef,
                                       MtyS_cand.p + (size_t)c * mstride, mstride, MtyS_cur.p, cS_d.p, gs,
                                       ycand_d.p + (size_t)c * ystride, ystride, ybuf[yc].p, Mxbuf[1 - mxc].p, Mxbuf[mxc].p,
                                       bh_d.p, (int)P.p, (int)P.Q, gq, t1, bpart.p, PSTRIDE, cstride, (const double*)roww_d.p);
                    hipLaunchKernelGGL(dev::k_combine_multi, dim3(11 + 2), dim3(dev::TPB), 0, stream,
                                       (const double*)bpart.p, PSTRIDE, std::max(gq, gs), 0x1ECull, bscal.p, 11,
                                       (const double*)respart_d.p, rstride, n_res_wg, bscal.p + NC * 11);
                    PX_HIP(hipMemcpyAsync(hbscal.data(), bscal.p, (NC * 11 + 2) * sizeof(double), hipMemcpyDeviceToHost, stream));
                    PX_HIP(hipStreamSynchronize(stream));
                    reduce_candidates(1);
                    s_acc = hbscal.data();
                }
                // y <- y_c, Mty <- Mty_c  (one launch)
                hipLaunchKernelGGL(dev::k_copy2, dim3(grid_for((long long)P.Q + ns)), dim3(dev::TPB), 0, stream,
                                   ybuf[1 - yc].p, (const double*)(ycand_d.p + (size_t)c * ystride), (long long)P.Q,
                                   MtyS_cur.p, (const double*)(MtyS_cand.p + (size_t)c * mstride), (long long)ns);
                break;
            }
            primal_step = tb.tau[c] * opt.linsearch_decay;
        }
    }
    primal_step_old = primal_step;
    if (ls) dual_step = beta * primal_step;
    st.linesearch_trials += trials;
    // ---- residuals and gap from the accepted candidate's scalars
    const double tr0 = now_s();
    const double m0 = std::max(s_acc[2], hbscal[NC * 11]);
    const double m1 = std::max(s_acc[3], hbscal[NC * 11 + 1]);
    const double pres = std::sqrt(g_n) * m0 / std::max({m1, g_norm_b, g_norm_h, 1.0});
    const double dres = std::sqrt(g_Q) * s_acc[5] / std::max({s_acc[6], g_norm_c, 1.0});
    h_pres.at(iter) = pres;
    h_dres.at(iter) = dres;
    h_comb.at(iter) = std::max(pres, dres);
    if (g_p > 0) equa_feasibility = s_acc[7] / (1.0 + g_norm_b);
    if (g_m > 0) ineq_feasibility = s_acc[8] / (1.0 + g_norm_h);
    h_feas.at(iter) = std::max(equa_feasibility, ineq_feasibility);
    const double po = s_acc[4];
    double d_o = 0.0;
    if (g_p > 0) d_o -= s_acc[9];
    if (g_m > 0) d_o -= s_acc[10];
    h_pobj.at(iter) = po;
    h_dobj.at(iter) = d_o;
    h_gap.at(iter) = std::fabs(po - d_o) / (1.0 + std::fabs(po) + std::fabs(d_o));
    xc = 1 - xc; yc = 1 - yc; mxc = 1 - mxc;
    last_resid_s = now_s() - tr0;
    return trials;
}

// linesearch! + compute_residual! + compute_gap! on the GENERAL path (no support set, sparse M): the structure of
// linesearch_residual_support with full-vector kernels -- up to 3 consecutive candidates per batch, ONE read-back,
// the first candidate the reference's loop would have accepted wins.  Per candidate the arithmetic is that of
// linesearch() / residual_and_gap() (same kernels' bodies, same partial layout and combine order): identical scalars.
inline int Solver::linesearch_residual_general() {
    constexpr int NC = 3;
    const int gq = std::min(PSTRIDE, grid_for(std::max<int64_t>(P.Q, 1)));
    const int gx = std::min(PSTRIDE, grid_for(P.n));
    const long long cstride = 11LL * PSTRIDE;
    const long long ystride = std::max<int64_t>(P.Q, 1), mstride = P.n;
    const double xold_coef = (iter == 1 && opt.advanced_initialization) ? 0.0 : 1.0;
    if (Mtycand_d.n < (size_t)NC * P.n) {
        Mtycand_d.alloc((size_t)NC * P.n);
        ycand_d.alloc((size_t)4 * std::max<int64_t>(P.Q, 1));
        bpart.alloc((size_t)4 * 11 * PSTRIDE); bpart.zero(stream);
        bscal.alloc(64); bscal.zero(stream);
        hbscal.assign(64, 0.0);
        if (hscal_pin.p == nullptr) hscal_pin.alloc(64);
    }
    primal_step = primal_step * std::sqrt(1.0 + theta);
    int trials = 0;
    bool accepted = false;
    const double* s_acc = nullptr;
    auto residual_batch = [&](const dev::TrialBatch& tb, int nc, int c0) {
        hipLaunchKernelGGL(dev::k_residual_xy_full_batch, dim3(std::max(gx, gq), nc, 2), dim3(dev::TPB), 0, stream,
                           xbuf[1 - xc].p, xbuf[xc].p, xold_coef, Mtycand_d.p + (size_t)c0 * mstride, mstride, Mtybuf[mtyc].p,
                           c_d.p, (long long)P.n, gx,
                           ycand_d.p + (size_t)c0 * ystride, ystride, ybuf[yc].p, Mxbuf[1 - mxc].p, Mxbuf[mxc].p,
                           bh_d.p, (int)P.p, (int)P.Q, gq, tb, bpart.p, PSTRIDE, cstride);
    };
    auto read_back = [&](int nc) {
        unsigned long long ismax = 0;
        for (int c = 0; c < nc; ++c) ismax |= 0x1ECull << (11 * c);      // bits 2,3,5,6,7,8 of every candidate
        // small models: the scalars go STRAIGHT into pinned host memory (no copy launch: 6 us of a 60 us iteration); larger ones keep
        // the copy (a kernel that stores to host memory ends with a system-scope release of everything the iteration left dirty)
        hipLaunchKernelGGL(dev::k_combine_multi, dim3(nc * 11), dim3(dev::TPB), 0, stream,
                           (const double*)bpart.p, PSTRIDE, std::max(gq, gx), ismax, zero_copy_small() ? hscal_pin.p : bscal.p, nc * 11,
                           (const double*)nullptr, 0, 0, (double*)nullptr);
        if (!zero_copy_small()) PX_HIP(hipMemcpyAsync(hscal_pin.p, bscal.p, NC * 11 * sizeof(double), hipMemcpyDeviceToHost, stream));
        wait_stream();
        std::copy(hscal_pin.p, hscal_pin.p + NC * 11, hbscal.begin());
    };
    while (!accepted && trials < opt.max_linsearch_steps) {
        dev::TrialBatch tb{};
        double tau_c = primal_step;
        int nc = 0;
        for (; nc < NC && trials + nc < opt.max_linsearch_steps; ++nc) {
            tb.tau[nc] = tau_c;
            tb.theta[nc] = tau_c / primal_step_old;
            tb.bt[nc] = beta * tau_c;
            tb.sigma[nc] = beta * tau_c;
            tau_c *= opt.linsearch_decay;
        }
        tb.nc = nc;
        hipLaunchKernelGGL(dev::k_dual_trial_batch, dim3(gq, nc), dim3(dev::TPB), 0, stream,
                           ybuf[yc].p, Mxbuf[1 - mxc].p, Mxbuf[mxc].p, bh_d.p, (int)P.p, (int)P.Q, tb,
                           ycand_d.p, ystride, bpart.p, cstride, (const double*)nullptr);
        hipLaunchKernelGGL(dev::k_spmv_csc_norm_batch, dim3(gx, nc), dim3(dev::TPB), 0, stream,
                           csc_ptr.p, csc_row.p, csc_val.p, ycand_d.p, ystride, Mtycand_d.p, mstride, Mtybuf[mtyc].p,
                           (long long)P.n, bpart.p + PSTRIDE, cstride);
        residual_batch(tb, nc, 0);
        read_back(nc);
        for (int c = 0; c < nc; ++c) {
            ++trials;
            const double* sc = hbscal.data() + 11 * c;
            primal_step = tb.tau[c];
            theta = tb.theta[c];
            const double y_norm = std::sqrt(sc[0]), Mty_norm = std::sqrt(sc[1]);
            if (debug && iter <= 5 && trials <= 6)
                std::fprintf(stderr, "[dbg] it %lld trial %d tau %.6e theta %.6e y_norm %.6e Mty_norm %.6e\n",
                             iter, trials, primal_step, theta, y_norm, Mty_norm);
            const bool ok = std::sqrt(beta) * primal_step * Mty_norm <= opt.delta * y_norm;
            const bool last = trials >= opt.max_linsearch_steps;
            if (ok || last) {
                accepted = true;
                s_acc = sc;
                if (!ok) {
                    // reference quirk (pdhg.jl:545-569): trial limit reached -> the step is decayed once more while the
                    // last trial's y / Mty are kept; the residual scalars are re-evaluated with THAT step (rare path)
                    primal_step *= opt.linsearch_decay;
                    dev::TrialBatch t1{};
                    t1.nc = 1; t1.tau[0] = primal_step; t1.theta[0] = theta; t1.bt[0] = beta * primal_step;
                    t1.sigma[0] = beta * primal_step;
                    residual_batch(t1, 1, c);
                    read_back(1);
                    // (candidate c's two norms, slots 0 and 1, are not needed any more)
                    s_acc = hbscal.data();
                }
                hipLaunchKernelGGL(dev::k_copy2, dim3(grid_for((long long)P.Q + P.n)), dim3(dev::TPB), 0, stream,
                                   ybuf[1 - yc].p, (const double*)(ycand_d.p + (size_t)c * ystride), (long long)P.Q,
                                   Mtybuf[1 - mtyc].p, (const double*)(Mtycand_d.p + (size_t)c * mstride), (long long)P.n);
                break;
            }
            primal_step = tb.tau[c] * opt.linsearch_decay;
        }
    }
    primal_step_old = primal_step;
    dual_step = beta * primal_step;
    st.linesearch_trials += trials;
    // ---- residuals and gap from the accepted candidate's scalars (residual_and_gap)
    const double tr0 = now_s();
    const double* s = s_acc + 2;
    if (debug && iter <= 5)
        std::fprintf(stderr, "[dbg] it %lld res: %.6e %.6e cx %.6e | %.6e %.6e eq %.6e in %.6e by %.6e hy %.6e\n",
                     iter, s[0], s[1], s[2], s[3], s[4], s[5], s[6], s[7], s[8]);
    const double pres = std::sqrt(g_n) * s[0] / std::max({s[1], g_norm_b, g_norm_h, 1.0});
    const double dres = std::sqrt(g_Q) * s[3] / std::max({s[4], g_norm_c, 1.0});
    h_pres.at(iter) = pres;
    h_dres.at(iter) = dres;
    h_comb.at(iter) = std::max(pres, dres);
    if (P.p > 0) equa_feasibility = s[5] / (1.0 + g_norm_b);
    if (P.m > 0) ineq_feasibility = s[6] / (1.0 + g_norm_h);
    h_feas.at(iter) = std::max(equa_feasibility, ineq_feasibility);
    const double po = s[2];
    double d_o = 0.0;
    if (P.p > 0) d_o -= s[7];
    if (P.m > 0) d_o -= s[8];
    h_pobj.at(iter) = po;
    h_dobj.at(iter) = d_o;
    h_gap.at(iter) = std::fabs(po - d_o) / (1.0 + std::fabs(po) + std::fabs(d_o));
    xc = 1 - xc; mtyc = 1 - mtyc; yc = 1 - yc; mxc = 1 - mxc;
    last_resid_s = now_s() - tr0;
    return trials;
}

// ---- hooks for the kernel-level test entry points
inline void Solver::test_project(int idx, double* xp, int tr) {
    target_rank.assign(1, tr); current_rank.assign(1, 0); min_eig.assign(1, 0.0);
    iter = 1;
    project_block(idx, xp - P.blocks[idx].off, xp - P.blocks[idx].off, false);
}
inline void Solver::test_spmv(bool transpose, const double* in, double* out) {
    setup_device();
    std::vector<int> rp(P.Q + 1), cp(P.n + 1);
    for (int64_t i = 0; i <= P.Q; ++i) rp[i] = (int)P.rowptr[i];
    for (int64_t i = 0; i <= P.n; ++i) cp[i] = (int)P.colptr[i];
    const int64_t nz = std::max<int64_t>(P.nnz, 1);
    csr_ptr.alloc(P.Q + 1); csr_col.alloc(nz); csr_val.alloc(nz);
    csc_ptr.alloc(P.n + 1); csc_row.alloc(nz); csc_val.alloc(nz);
    csr_ptr.upload(rp.data(), P.Q + 1, stream); csc_ptr.upload(cp.data(), P.n + 1, stream);
    csr_col.upload(P.colidx.data(), P.nnz, stream); csr_val.upload(P.rval.data(), P.nnz, stream);
    csc_row.upload(P.rowidx.data(), P.nnz, stream); csc_val.upload(P.val.data(), P.nnz, stream);
    csr_wave = P.Q > 0 && (double)P.nnz / (double)P.Q > 8.0;
    setup_long_rows(rp);
    DevBuf<double> xin(std::max<int64_t>(transpose ? P.Q : P.n, 1)), xout(std::max<int64_t>(transpose ? P.n : P.Q, 1));
    xin.upload(in, transpose ? P.Q : P.n, stream);
    if (transpose)
        hipLaunchKernelGGL(dev::k_spmv_csc, dim3(grid_for(P.n)), dim3(dev::TPB), 0, stream,
                           csc_ptr.p, csc_row.p, csc_val.p, xin.p, xout.p, (long long)P.n);
    else
        spmv(xin.p, xout.p);
    xout.download(out, transpose ? P.n : P.Q, stream);
    PX_HIP(hipStreamSynchronize(stream));
}

// ---- state seam (include/proxsdp_hip.h proxsdp_state)
inline void Solver::check_state_shape(const proxsdp_state& s, const char* what) const {
    const std::string w(what);
    if (s.struct_size != (int64_t)sizeof(proxsdp_state)) throw std::invalid_argument(w + " state: struct_size mismatch");
    if (s.n != P.n || s.Q != P.Q || s.n_psd != (int64_t)P.blocks.size())
        throw std::invalid_argument(w + " state: n / Q / n_psd do not match the problem");
    if (s.hist_len != 2 * (int64_t)opt.convergence_window)
        throw std::invalid_argument(w + " state: hist_len must be 2 * convergence_window");
    if (s.iteration < 1) throw std::invalid_argument(w + " state: iteration must be >= 1");
    if (!s.x || !s.Mty || !s.hist || (P.Q > 0 && (!s.y || !s.Mx)) ||
        (!P.blocks.empty() && (!s.target_rank || !s.current_rank || !s.min_eig)))
        throw std::invalid_argument(w + " state: NULL array");
    if (sharded() || P.equilibrated) throw std::domain_error("state capture / resume: not with a block-sharded solve or equilibration");
}

// continue from resume_state: called after the "Init" section built every buffer; overwrites the initial point
inline void Solver::apply_resume() {
    const proxsdp_state& s = *resume_state;
    check_state_shape(s, "resume");
    xbuf[xc].upload(s.x, P.n, stream);
    ybuf[yc].upload(s.y, P.Q, stream);
    Mxbuf[mxc].upload(s.Mx, P.Q, stream);
    Mtybuf[mtyc].upload(s.Mty, P.n, stream);
    std::vector<double> mtyS;
    if (use_support) {                                   // the support path carries M'y on the support only
        std::vector<int> supp(ns);
        supp_d.download(supp.data(), ns, stream);
        PX_HIP(hipStreamSynchronize(stream));
        mtyS.resize(std::max(ns, 1), 0.0);
        for (int q = 0; q < ns; ++q) mtyS[q] = s.Mty[supp[q]];
        MtyS_cur.upload(mtyS.data(), ns, stream);
    }
    PX_HIP(hipStreamSynchronize(stream));
    primal_step = s.scal[0]; primal_step_old = s.scal[1]; dual_step = s.scal[2];
    beta = s.scal[3]; theta = s.scal[4]; adapt_level = s.scal[5];
    equa_feasibility = s.scal[6]; ineq_feasibility = s.scal[7]; dual_feasibility = s.scal[8];
    rank_update = (int)s.ints[0]; update_cont = (int)s.ints[1]; ada_count = (int)s.ints[2];
    for (size_t idx = 0; idx < P.blocks.size(); ++idx) {
        target_rank[idx] = std::min<long long>(std::max<long long>(s.target_rank[idx], 1), P.blocks[idx].n);
        current_rank[idx] = s.current_rank[idx];
        min_eig[idx] = s.min_eig[idx];
    }
    CircularVector* H[PROXSDP_STATE_NHIST] = {&h_gap, &h_pobj, &h_dobj, &h_feas, &h_pres, &h_dres, &h_comb};
    for (int q = 0; q < PROXSDP_STATE_NHIST; ++q)
        std::copy(s.hist + (size_t)q * s.hist_len, s.hist + (size_t)(q + 1) * s.hist_len, H[q]->v.begin());
    iter = s.iteration;
    // the iterate is a general matrix now: neither zero off the support nor known in factored form
    for (EigWork& W : eig) { W.have_factors = false; W.x_prev_sparse = false; }
}

// write capture_state: the stream is idle (every iteration ends with a synchronisation)
inline void Solver::write_capture() {
    proxsdp_state& s = *capture_state;
    if (certificate_search) return;          // (the seam does not carry a certificate search: ints[3] stays 0, the solve goes on)
    PX_HIP(hipStreamSynchronize(stream));
    harvest_small_ranks();
    xbuf[xc].download(s.x, P.n, stream);
    ybuf[yc].download(s.y, P.Q, stream);
    Mxbuf[mxc].download(s.Mx, P.Q, stream);
    if (use_support) {
        std::vector<int> supp(ns);
        std::vector<double> mtyS(std::max(ns, 1), 0.0);
        supp_d.download(supp.data(), ns, stream);
        MtyS_cur.download(mtyS.data(), ns, stream);
        PX_HIP(hipStreamSynchronize(stream));
        std::fill(s.Mty, s.Mty + P.n, 0.0);
        for (int q = 0; q < ns; ++q) s.Mty[supp[q]] = mtyS[q];
    } else {
        Mtybuf[mtyc].download(s.Mty, P.n, stream);
    }
    PX_HIP(hipStreamSynchronize(stream));
    std::fill(std::begin(s.scal), std::end(s.scal), 0.0);
    std::fill(std::begin(s.ints), std::end(s.ints), 0);
    s.scal[0] = primal_step; s.scal[1] = primal_step_old; s.scal[2] = dual_step;
    s.scal[3] = beta; s.scal[4] = theta; s.scal[5] = adapt_level;
    s.scal[6] = equa_feasibility; s.scal[7] = ineq_feasibility; s.scal[8] = dual_feasibility;
    s.ints[0] = rank_update; s.ints[1] = update_cont; s.ints[2] = ada_count;
    for (size_t idx = 0; idx < P.blocks.size(); ++idx) {
        s.target_rank[idx] = target_rank[idx]; s.current_rank[idx] = current_rank[idx]; s.min_eig[idx] = min_eig[idx];
    }
    CircularVector* H[PROXSDP_STATE_NHIST] = {&h_gap, &h_pobj, &h_dobj, &h_feas, &h_pres, &h_dres, &h_comb};
    for (int q = 0; q < PROXSDP_STATE_NHIST; ++q) std::copy(H[q]->v.begin(), H[q]->v.end(), s.hist + (size_t)q * s.hist_len);
    s.ints[3] = 1;
}

// chambolle_pock (pdhg.jl:1-530)
inline void Solver::run() {
    const double t_init0 = now_s();
    if (!opt.approx_norm && (P.dense() || sharded()))
        throw std::domain_error("approx_norm=false with a dense A or a block-sharded solve is not implemented");
    if (P.n <= 0) throw std::invalid_argument("problem has no variables");
    if (opt.convergence_window <= 0) throw std::invalid_argument("convergence_window must be positive");
    // (the reference's `for i in 1:max_linsearch_steps` simply runs no trial then and keeps stale norms; the batched
    // candidates here need at least one)
    if (opt.line_search_flag && opt.max_linsearch_steps < 1) throw std::invalid_argument("max_linsearch_steps must be >= 1");
    // fault injection is a TEST switch: it only works in a process that asks for it by environment as well
    if (opt.debug_fail_iteration > 0) {
        const char* e = std::getenv("PROXSDP_HIP_FAULT_INJECTION");
        if (!(e && e[0] == '1')) throw std::invalid_argument("debug_fail_iteration needs PROXSDP_HIP_FAULT_INJECTION=1 in the environment (test switch)");
    }
    theta = opt.initial_theta; adapt_level = opt.initial_adapt_level; beta = opt.initial_beta;
    const int window = opt.convergence_window;
    const size_t nb = P.blocks.size();
    target_rank.assign(nb, 2); current_rank.assign(nb, 2); min_eig.assign(nb, 0.0);
    for (size_t idx = 0; idx < nb; ++idx)               // 2 in the reference (pdhg.jl:19-20)
        target_rank[idx] = std::min<long long>(std::max(opt.initial_target_rank, 1), P.blocks[idx].n);
    time_limit = opt.time_limit;
    {   // global constants (sums over shards of a block-sharded solve; local values otherwise)
        double pw = (double)P.p, mw = (double)P.m, nb2 = P.norm_b * P.norm_b, nh2 = P.norm_h * P.norm_h;
        for (size_t k = 0; k < coup_rows.size(); ++k) {          // a coupling row counts on its owner only
            if (coup_owned[k]) continue;
            const int r = coup_rows[k];
            if (r < P.p) { pw -= 1.0; nb2 -= P.b_orig[r] * P.b_orig[r]; }
            else { mw -= 1.0; nh2 -= P.h_orig[r - P.p] * P.h_orig[r - P.p]; }
        }
        std::vector<double> sums = {(double)P.n, pw, mw, std::max(nb2, 0.0), std::max(nh2, 0.0),
                                    P.norm_c * P.norm_c, P.frob * P.frob};
        std::vector<double> maxs = {(nb > 0 || !P.socs.empty()) ? 1.0 : 0.0};
        reduce(sums, maxs);
        g_n = sums[0]; g_p = sums[1]; g_m = sums[2]; g_Q = g_p + g_m;
        g_norm_b = sharded() ? std::sqrt(sums[3]) : P.norm_b;
        g_norm_h = sharded() ? std::sqrt(sums[4]) : P.norm_h;
        g_norm_c = sharded() ? std::sqrt(sums[5]) : P.norm_c;
        g_frob = sharded() ? std::sqrt(sums[6]) : P.frob;
        g_conic = maxs[0] > 0.5;
    }
    // (a block-sharded solve always runs the support-aware batched path: the library-only knob support_path is overridden below;
    //  round 6: check_dual_feas and line_search_flag = false are served there too)
    if (opt.max_iter <= 0) max_iter_local = g_conic ? opt.max_iter_conic : opt.max_iter_lp;
    else max_iter_local = opt.max_iter;
    ada_count = 0;
    h_gap.init(2 * window); h_pobj.init(2 * window); h_dobj.init(2 * window); h_feas.init(2 * window);
    h_pres.init(2 * window); h_dres.init(2 * window); h_comb.init(2 * window);
    b_host = P.b; h_host = P.h; c_host = P.c;

    // ---- device state ("Init", pdhg.jl:54-142)
    setup_device();
    bool big_block = false;
    for (const BlockInfo& B : P.blocks) big_block = big_block || B.n >= 256;
    // opt-in only: measured on MI355X/ROCm 7.2, loading rocSOLVER's code objects from a second
    // thread stalls this thread's kernel launches (a 0.3 s solve took 8 s), so by default the
    // exit path pays the one-off ~3 s initialisation itself in a cold process
    if (big_block && opt.rocsolver_warmup == 1) start_rocsolver_warmup();
    if (opt.host_eig_threads > 0) QlPool::get().ensure(opt.host_eig_threads);
    for (int k = 0; k < 2; ++k) {
        xbuf[k].alloc(P.n); Mtybuf[k].alloc(P.n);
        ybuf[k].alloc(std::max<int64_t>(P.Q, 1)); Mxbuf[k].alloc(std::max<int64_t>(P.Q, 1));
        xbuf[k].zero(stream); Mtybuf[k].zero(stream); ybuf[k].zero(stream); Mxbuf[k].zero(stream);
    }
    c_d.alloc(P.n); c_d.upload(P.c.data(), P.n, stream);
    bh_d.alloc(std::max<int64_t>(P.Q, 1));
    {
        std::vector<double> bh(P.Q);
        std::copy(P.b.begin(), P.b.end(), bh.begin());
        std::copy(P.h.begin(), P.h.end(), bh.begin() + P.p);
        bh_d.upload(bh.data(), P.Q, stream);
        PX_HIP(hipStreamSynchronize(stream));
    }
    if (!coup_rows.empty()) {
        coup_rows_d.alloc(coup_rows.size()); coup_buf_d.alloc(coup_rows.size()); roww_d.alloc(std::max<int64_t>(P.Q, 1));
        std::vector<double> rw(P.Q, 1.0);
        for (size_t k = 0; k < coup_rows.size(); ++k) if (!coup_owned[k]) rw[coup_rows[k]] = 0.0;
        coup_rows_d.upload(coup_rows.data(), coup_rows.size(), stream);
        roww_d.upload(rw.data(), P.Q, stream);
        PX_HIP(hipStreamSynchronize(stream));
    }
    part.alloc((size_t)NQ * PSTRIDE); part.zero(stream);
    scal.alloc(NQ); scal.zero(stream);
    hscal.assign(NQ, 0.0);
    {   // sparse operator, both orientations, int32 indices
        std::vector<int> rp(P.Q + 1), cp(P.n + 1);
        // (prepare() guarantees n, Q, nnz < 2^31: the casts below cannot truncate)
        for (int64_t i = 0; i <= P.Q; ++i) rp[i] = (int)P.rowptr[i];
        for (int64_t i = 0; i <= P.n; ++i) cp[i] = (int)P.colptr[i];
        csr_ptr.alloc(P.Q + 1); csr_col.alloc(std::max<int64_t>(P.nnz, 1)); csr_val.alloc(std::max<int64_t>(P.nnz, 1));
        csc_ptr.alloc(P.n + 1); csc_row.alloc(std::max<int64_t>(P.nnz, 1)); csc_val.alloc(std::max<int64_t>(P.nnz, 1));
        csr_ptr.upload(rp.data(), P.Q + 1, stream); csc_ptr.upload(cp.data(), P.n + 1, stream);
        csr_col.upload(P.colidx.data(), P.nnz, stream); csr_val.upload(P.rval.data(), P.nnz, stream);
        csc_row.upload(P.rowidx.data(), P.nnz, stream); csc_val.upload(P.val.data(), P.nnz, stream);
        PX_HIP(hipStreamSynchronize(stream));
        csr_wave = P.Q > 0 && (double)P.nnz / (double)P.Q > 8.0;
        setup_long_rows(rp);
    }
    eig.resize(nb);
    // small blocks by the sign function in one launch: auto (and small_block_batch = 2) unless the tiled sign projection is
    // switched off or the solve asks for tolerances below that engine's 1e-10-of-the-scale floor (as full_eig_by_sign)
    const bool small_sign = opt.small_block_batch == 2 ||
        (opt.small_block_batch < 0 && opt.full_eig_sign != 0 &&
         std::min({opt.tol_gap, opt.tol_feasibility, opt.tol_primal, opt.tol_dual}) >= 1e-8);
    small_jacobi_max = small_sign ? (opt.small_block_batch == 2 ? 1 : 2) : 64;   // (measured, tools/gpurun_jacobi_vs_sign.py: from side 3 on the sign kernel wins)
    {
        std::vector<long long> offs;
        const double* ur = user_resid;
        for (size_t idx = 0; idx < nb; ++idx) {
            const BlockInfo& B = P.blocks[idx];
            if (B.n == 1) { one_blocks.push_back((int)idx); offs.push_back(B.off); if (ur) ur += 1; continue; }
            EigWork& W = eig[idx];
            big_blocks.push_back((int)idx);
            // side 2..64 and never on the Krylov path: ONE launch for all of them (dense vector path): batched Jacobi
            // (auto: side 2 only -- measured: 80 / 140 / 370 us per iteration at side 3 / 8 / 16 where the sign kernel stays at 60; at side 22 the
            // single-workgroup Jacobi takes 950 us, at side 50 5.9 ms against
            // 1.1 ms for one rocSOLVER call; two blocks of side 10 and 5: 1.4x faster batched; seven of side 2: 6.4x) and,
            // round 5, the one-workgroup LDS-resident sign projection (small_sign.hip.hpp) for sides 3 .. 64
            if (opt.small_block_batch != 0 && B.n <= (small_sign ? std::max(small_jacobi_max, std::min(64, small_sign_cap)) : (opt.small_block_batch > 0 ? 64 : 32)) &&
                B.n <= opt.min_size_krylov_eigs && !sharded())
                small_blocks.push_back((int)idx);
            else
                large_blocks.push_back((int)idx);
            // Krylov workspace: the largest target rank the Krylov path may see, and room for
            // full_eig!-by-Lanczos (up to 94 pairs) on blocks that can take it
            int max_nev = std::min<int>(std::max<int>(opt.max_target_rank_krylov_eigs, 2), B.n);
            if (opt.full_eig_lanczos != 0 && B.n > opt.min_size_krylov_eigs && B.n >= 400)
                max_nev = std::max(max_nev, std::min(94, B.n / 4));
            alloc_eigwork(W, B.n, max_nev);
            W.resid_host.resize(W.npad, 0.0);
            if (ur) { std::copy(ur, ur + B.n, W.resid_host.begin()); ur += B.n; }
            else start_vector(B.n, (uint64_t)opt.eigsolver_resid_seed,
                              opt.eigsolver == 1 ? opt.arpack_resid_init : opt.krylovkit_resid_init,
                              W.resid_host.data());
            double nr = norm2(W.resid_host.data(), B.n);        // KrylovKit: v = x0 / norm(x0)
            if (!(nr > 0.0)) throw std::invalid_argument("Lanczos start vector has zero norm");
            for (int i = 0; i < B.n; ++i) W.resid_host[i] /= nr;
            W.resid.upload(W.resid_host.data(), W.npad, stream);
        }
        if (small_blocks.size() < 2 && opt.small_block_batch < 0 && !small_sign) {   // auto, Jacobi only: a batch needs several blocks
            large_blocks = big_blocks; small_blocks.clear();
        }
        // several eigensolver-sized blocks: project them concurrently, one worker thread and one
        // stream per block (the per-block Lanczos chains are launch-latency-bound, so they overlap
        // almost perfectly: MIMO n=512 x 8 on one GPU).  options.block_threads = 0 disables.
        {
            int nthreads = opt.block_threads < 0 ? 8 : std::min(opt.block_threads, 64);
            // (blocks handled by the batched small-block kernel need no worker / stream of their own)
            const std::vector<int>& pool_blocks = small_blocks.empty() ? big_blocks : large_blocks;
            nthreads = std::min<int>(nthreads, (int)pool_blocks.size());
            if (nthreads >= 2) {
                for (int idx : pool_blocks) {
                    PX_HIP(hipStreamCreate(&eig[idx].stream));
                    PX_HIP(hipEventCreateWithFlags(&eig[idx].done, hipEventDisableTiming));
                }
                PX_HIP(hipEventCreateWithFlags(&ev_main, hipEventDisableTiming));
                PX_HIP(hipStreamSynchronize(stream));           // workspace zero-fills precede any block-stream work
                start_workers(nthreads);
                parallel_blocks = true;
            }
        }
        if (!small_blocks.empty()) {
            std::vector<long long> so; std::vector<int> ss;
            for (int idx : small_blocks) { so.push_back(P.blocks[idx].off); ss.push_back(P.blocks[idx].n); small_maxn = std::max(small_maxn, P.blocks[idx].n); }
            small_off.alloc(so.size()); small_side.alloc(ss.size()); small_rank.alloc(3 * ss.size() + 2); small_rank.zero(stream);   // [rank | npos] per block, then the sign kernel's cumulative [pass, fail] counters
            small_off.upload(so.data(), so.size(), stream); small_side.upload(ss.data(), ss.size(), stream);
            small_rank_host.alloc(2 * ss.size() + 2);      // (ints: rank | positive count | schedule-test outcome per block)
            small_sign_maxn = 0;
            int jac_maxn = 0;
            for (int sd : ss) { if (sd > small_jacobi_max) small_sign_maxn = std::max(small_sign_maxn, sd); else jac_maxn = std::max(jac_maxn, sd); }
            const size_t lds = ((size_t)2 * jac_maxn * (jac_maxn | 1) + 64) * sizeof(double) + 64 * sizeof(int);
            if (lds > 48 * 1024)
                PX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(dev::k_small_psd_project),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            if (small_sign_maxn > 0 && dev::small_sign_lds_bytes(small_sign_maxn) > 48 * 1024)
                PX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(dev::k_small_sign_project),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)dev::small_sign_lds_bytes(small_sign_maxn)));
            PX_HIP(hipStreamSynchronize(stream));
        }
        if (!offs.empty()) {
            one_off.alloc(offs.size()); one_min.alloc(offs.size());
            one_off.upload(offs.data(), offs.size(), stream);
            PX_HIP(hipStreamSynchronize(stream));
        }
    }
    if (!P.socs.empty()) {
        std::vector<long long> so; std::vector<int> sl;
        for (const SocInfo& S : P.socs) { so.push_back(S.off); sl.push_back(S.len); }
        soc_off.alloc(so.size()); soc_len.alloc(sl.size()); soc_gap_d.alloc(so.size());
        soc_off.upload(so.data(), so.size(), stream); soc_len.upload(sl.data(), sl.size(), stream);
        PX_HIP(hipStreamSynchronize(stream));
    }
    if (sharded()) opt.support_path = 1;                 // the sharded loop is built on the batched path
    if (P.dense()) opt.support_path = 0;                 // every column of a dense M is in the support
    setup_support();
    setup_dense();
    if (sharded() && !use_support)
        throw std::domain_error("block-sharded solve needs the support-aware path (no SOC / 1x1 cones)");
    double spectral_norm = g_frob;                       // LinearAlgebra.norm(M), pdhg.jl:121
    if (!opt.approx_norm)                                // Arpack.svds(M, nsv=1), pdhg.jl:108-119
        spectral_norm = spectral_norm_host(P, [](int K, double* T, double* d) { return symeig_dense(K, T, d, true); });
    if (spectral_norm < 1e-10) spectral_norm = 1.0;
    primal_step = 1.0 / spectral_norm;
    primal_step_old = primal_step;
    dual_step = primal_step;
    if (opt.advanced_initialization) {                   // x = tau*c (pdhg.jl:138-142)
        std::vector<double> x0(P.n);
        for (int64_t i = 0; i < P.n; ++i) x0[i] = primal_step * P.c[i];
        xbuf[xc].upload(x0.data(), P.n, stream);
        PX_HIP(hipStreamSynchronize(stream));
    }
    PX_HIP(hipStreamSynchronize(stream));
    long long k_first = 1;
    if (resume_state) { apply_resume(); k_first = resume_state->iteration + 1; }
    if (capture_state) { check_state_shape(*capture_state, "capture"); capture_state->ints[3] = 0; }
    st.init_time = now_s() - t_init0;

    auto snapshot = [&]() { cache_solution(P.c_orig); };
    auto cert_infeas = [&]() {                           // certificate_infeasibility (pdhg.jl:655-668)
        std::fill(c_host.begin(), c_host.end(), 0.0);
        c_d.zero(stream);
        if (use_support) cS_d.zero(stream);
        certificate_parameters();
    };
    auto cert_dual_infeas = [&]() {                      // certificate_dual_infeasibility (pdhg.jl:639-653)
        std::fill(b_host.begin(), b_host.end(), 0.0);
        std::fill(h_host.begin(), h_host.end(), 0.0);
        bh_d.zero(stream);
        certificate_parameters();
    };
    int n_snap = 0;

    // ---- "CP loop" (pdhg.jl:145-484)
    const double t_loop0 = now_s();
    const long long kmax = 2 * max_iter_local;
    for (long long k = k_first; k <= kmax; ++k) {
        if (capture_state && k == capture_state->iteration + 1 && k > k_first) write_capture();
        iter = k;
        lz_matvec_iter = 0; recon_r_iter = 0;
        const double tp0 = now_s();
        primal_step_dev();
        st.t_primal += now_s() - tp0;                    // "primal" section of pdhg.jl:150 (includes t_psd)
        const double tl0 = now_s();
        if (use_support) {
            // (fused paths: the residual / gap REDUCTIONS ride in the candidates' batch and its one read-back -- they are
            // part of t_linesearch; t_residual counts what is left of compute_residual! / compute_gap!: the host scalars)
            last_trials = linesearch_residual_support();
            st.t_linesearch += now_s() - tl0 - last_resid_s; st.t_residual += last_resid_s;
        } else {
            if (opt.line_search_flag && !P.dense() && !sharded() && opt.general_batch != 0) {
                last_trials = linesearch_residual_general();       // trials + residual + gap in one batch and one read-back
                st.t_linesearch += now_s() - tl0 - last_resid_s; st.t_residual += last_resid_s;
            } else {
            if (opt.line_search_flag) last_trials = P.dense() ? linesearch_dense() : linesearch();
            else { dual_step_plain(); last_trials = 1; }
            const double tl1 = now_s();
            residual_and_gap();
            st.t_linesearch += tl1 - tl0;
            st.t_residual += now_s() - tl1;
            }
        }
        {   // algorithmic bytes of this iteration
            const double t = (double)last_trials;
            double bb = 8.0 * (double)P.n * (11.0 + 3.0 * t) + 12.0 * (double)P.nnz * (1.0 + t) +
                        8.0 * (double)P.Q * (8.0 + 6.0 * t);
            if (P.dense()) {                             // M x + one M'y pass per candidate batch
                dense_ev_harvest();
                bb += 8.0 * (double)P.p * (double)P.n * (double)(st.dense_passes - dense_passes_seen);
                dense_passes_seen = st.dense_passes;
            }
            for (size_t idx = 0; idx < nb; ++idx) {
                if (P.blocks[idx].n < 2) continue;
                // per-block L and r are accumulated over blocks in lz_matvec_iter / recon_r_iter;
                // single-block instances (all BASELINE configs) make this exact
                (void)idx;
            }
            if (nb > 0) {
                const BlockInfo& B0 = P.blocks[0];
                bb += (8.0 * (double)B0.N + 16.0 * (double)B0.n) * (double)lz_matvec_iter +
                      8.0 * (double)B0.n * (double)recon_r_iter;
            }
            st.algorithmic_bytes += bb;
        }
        if (res.trace && res.trace_rows < opt.trace_capacity) {
            double* row = res.trace + (size_t)res.trace_rows * PROXSDP_TRACE_COLS;
            row[0] = (double)k; row[1] = h_pobj.at(k); row[2] = h_dobj.at(k); row[3] = h_gap.at(k);
            row[4] = h_feas.at(k); row[5] = h_pres.at(k); row[6] = h_dres.at(k); row[7] = primal_step;
            row[8] = beta; row[9] = theta; row[10] = nb ? (double)target_rank[0] : 0.0; row[11] = (double)last_trials;
            row[12] = now_s() - t_loop0; row[13] = (double)lz_matvec_iter;
            res.trace_rows++;
        }
        if (opt.check_dual_feas && k % opt.check_dual_feas_freq == 0) {
            std::vector<double> y(P.Q);
            ybuf[yc].download(y.data(), P.Q, stream);
            PX_HIP(hipStreamSynchronize(stream));
            std::vector<double> cc(P.c_orig);
            if (stop_reason == 6) std::fill(cc.begin(), cc.end(), 0.0);
            dual_feasibility = dual_feas_host(y, cc, nullptr, nullptr, nullptr);
            if (sharded()) {                             // every shard tests its own columns: the model's value is the largest
                std::vector<double> sums, maxs = {dual_feasibility};
                reduce(sums, maxs);
                dual_feasibility = maxs[0];
            }
        }
        if (opt.log_verbose && opt.log_freq > 0 && k % opt.log_freq == 0)
            std::printf("|%9lld| %+.4e %+.4e %.2e %.2e %.2e %.2e %4lld %8.2f\n", k, h_pobj.at(k), h_dobj.at(k),
                        h_gap.at(k), h_feas.at(k), h_pres.at(k), h_dres.at(k), nb ? target_rank[0] : 0,
                        now_s() - time0);
        if (iter < certificate_search_min_iter) continue;

        if (opt.certificate_search && certificate_search) {          // pdhg.jl:184-244
            if (stop_reason == 6) {
                if (h_dobj.at(k) > opt.certificate_obj_tol) {
                    std::vector<double> y(P.Q);
                    ybuf[yc].download(y.data(), P.Q, stream);
                    PX_HIP(hipStreamSynchronize(stream));
                    std::vector<double> zc(P.n, 0.0);
                    dual_feasibility = dual_feas_host(y, zc, nullptr, nullptr, nullptr);
                    if (sharded()) {
                        std::vector<double> sums, maxs = {dual_feasibility};
                        reduce(sums, maxs);
                        dual_feasibility = maxs[0];
                    }
                    if (dual_feasibility < opt.tol_feasibility_dual) {
                        certificate_found = true;
                        stop_reason_string += " [Dual ray found]";
                        break;
                    }
                }
            } else {
                if (h_pobj.at(k) < -opt.certificate_obj_tol && h_feas.at(iter) < opt.tol_feasibility) {
                    certificate_found = true;
                    stop_reason_string += " [Primal ray found]";
                    break;
                }
            }
            if ((h_pobj.at(k) < -opt.certificate_fail_tol && h_dobj.at(k) < -opt.certificate_fail_tol &&
                 h_feas.at(iter) < -opt.certificate_fail_tol) || std::isnan(h_comb.at(k))) {
                stop_reason_string += " [Failed to find certificate]";
                break;
            }
        }

        // ---- convergence / rank update / divergence / adaptive steps (pdhg.jl:246-332)
        rank_update += 1;
        if (h_gap.at(iter) <= opt.tol_gap && h_feas.at(iter) <= opt.tol_feasibility &&
            (!opt.check_dual_feas || dual_feasibility < opt.tol_feasibility_dual)) {
            if ((sharded() ? !g_not_converged_rank : convergedrank()) && soc_convergence() && iter > opt.min_iter) {
                if (!certificate_search) {
                    stop_reason = 1;
                    stop_reason_string = "Optimal solution found";
                } else {
                    stop_reason_string += " [Failed to find certificate - type 2]";
                }
                break;
            } else if (rank_update > window) {
                update_cont += 1;
                if (update_cont > 0) {
                    for (size_t idx = 0; idx < nb; ++idx) bump_rank((int)idx);
                    rank_update = 0; update_cont = 0;
                }
            }
        } else if (k > window && h_comb.at(k - window) < h_comb.at(k) && rank_update > window) {
            update_cont += 1;
            if (update_cont > opt.divergence_min_update) {
                if (sharded() && g_any_below_full) { rank_update = 0; update_cont = 0; }
                for (size_t idx = 0; idx < nb; ++idx) {
                    if (target_rank[idx] < P.blocks[idx].n) { rank_update = 0; update_cont = 0; }
                    bump_rank((int)idx);
                }
            }
        } else if (h_pres.at(k) > opt.tol_primal && h_dres.at(k) < opt.tol_dual && k > window) {
            if (++ada_count > opt.adapt_window) {
                ada_count = 0;
                if (opt.line_search_flag) { beta *= (1.0 - adapt_level); primal_step /= std::sqrt(1.0 - adapt_level); }
                else { primal_step /= (1.0 - adapt_level); dual_step *= (1.0 - adapt_level); }
                adapt_level *= opt.adapt_decay;
            }
        } else if (h_pres.at(k) < opt.tol_primal && h_dres.at(k) > opt.tol_dual && k > window) {
            if (++ada_count > opt.adapt_window) {
                ada_count = 0;
                if (opt.line_search_flag) { beta /= (1.0 - adapt_level); primal_step *= std::sqrt(1.0 - adapt_level); }
                else { primal_step *= (1.0 - adapt_level); dual_step /= (1.0 - adapt_level); }
                adapt_level *= opt.adapt_decay;
            }
        }

        // ---- iteration / time limits (pdhg.jl:334-382)
        const double elapsed = sharded() ? g_elapsed : now_s() - time0;
        if (iter >= max_iter_local || elapsed >= time_limit) {
            if (iter > opt.min_iter_time_infeas && h_gap.max_abs_diff() < opt.infeas_stable_gap_tol &&
                h_gap.at(k) > opt.infeas_limit_gap_tol) {
                if (h_feas.at(iter) <= opt.tol_feasibility / 100) {
                    stop_reason = 5;
                    stop_reason_string = "Problem declared unbounded due to lack of improvement";
                    if (opt.certificate_search && !certificate_search) { cert_dual_infeas(); snapshot(); ++n_snap; }
                    else if (opt.certificate_search && certificate_search) {}
                    else break;
                } else if (h_feas.at(iter) > opt.infeas_feasibility_tol) {
                    stop_reason = 6;
                    stop_reason_string = "Problem declared infeasible due to lack of improvement";
                    if (opt.certificate_search && !certificate_search) { cert_infeas(); snapshot(); ++n_snap; }
                    else if (opt.certificate_search && certificate_search) {}
                    else break;
                }
            } else if (iter >= max_iter_local) {
                stop_reason = 3;
                stop_reason_string = "Iteration limit of " + std::to_string(max_iter_local) + " was hit";
            } else {
                stop_reason = 2;
                stop_reason_string = "Time limit hit, limit: " + std::to_string(time_limit) +
                                     " time: " + std::to_string(now_s() - time0);
            }
            if (iter >= max_iter_local || elapsed >= time_limit) break;
        }
        if (opt.certificate_search && certificate_search) continue;

        // ---- objective blow-up / stalls (pdhg.jl:389-483)
        if ((iter > opt.min_iter_max_obj && h_dobj.at(k) > opt.max_obj) || std::isnan(h_dobj.at(k))) {
            stop_reason = 6;
            stop_reason_string = "Infeasible: |Dual objective| = " + std::to_string(h_dobj.at(k)) +
                                 " > maximum allowed = " + std::to_string(opt.max_obj);
            if (opt.certificate_search && !certificate_search) { cert_infeas(); snapshot(); ++n_snap; }
            else break;
        }
        if ((iter > opt.min_iter_max_obj && h_pobj.at(k) < -opt.max_obj) || std::isnan(h_pobj.at(k))) {
            stop_reason = 5;
            stop_reason_string = "Unbounded: |Primal objective| = " + std::to_string(h_pobj.at(k)) +
                                 " > maximum allowed = " + std::to_string(opt.max_obj);
            if (opt.certificate_search && !certificate_search) { cert_dual_infeas(); snapshot(); ++n_snap; }
            else break;
        }
        if (iter > opt.min_iter_max_obj && h_gap.at(k) > opt.infeas_limit_gap_tol &&
            h_feas.at(iter) > opt.infeas_feasibility_tol &&
            h_feas.max_abs_diff() < opt.infeas_stable_feasibility_tol) {
            stop_reason = 6;
            stop_reason_string = "Infeasible: feasibility stalled at " + std::to_string(h_feas.at(iter));
            if (opt.certificate_search && !certificate_search) { cert_infeas(); snapshot(); ++n_snap; }
            else break;
        }
        if (iter > opt.min_iter_max_obj && h_gap.at(k) > 1 - opt.infeas_gap_tol &&
            h_gap.max_abs_diff() < opt.infeas_stable_gap_tol) {
            if (std::fabs(h_dobj.at(k)) > std::fabs(h_pobj.at(k)) && h_feas.at(iter) > opt.infeas_feasibility_tol) {
                stop_reason = 6;
                stop_reason_string = "Infeasible: duality gap stalled at 100 % with |Dual objective| >> |Primal objective|";
                if (opt.certificate_search && !certificate_search) { cert_infeas(); snapshot(); ++n_snap; }
                else break;
            } else if (std::fabs(h_pobj.at(k)) > std::fabs(h_dobj.at(k)) && h_feas.at(iter) <= opt.tol_feasibility) {
                stop_reason = 5;
                stop_reason_string = "Unbounded: duality gap stalled at 100 % with |Dual objective| << |Primal objective|";
                if (opt.certificate_search && !certificate_search) { cert_dual_infeas(); snapshot(); ++n_snap; }
                else break;
            }
        }
    }
    PX_HIP(hipStreamSynchronize(stream));
    if (capture_state && capture_state->ints[3] == 0 && iter == capture_state->iteration) write_capture();
    for (EigWork& W : eig) harvest_full_eig_events(W);
    merge_block_stats();
    if (cy_dbg.n) {
        long long t[16];
        cy_dbg.download(t, 16, stream);
        PX_HIP(hipStreamSynchronize(stream));
        const double sN = (double)std::max<long long>(st.cycle_steps, 1);
        std::fprintf(stderr, "[cycle ticks/step @100MHz] closeA+opA %.1f |B %.1f |ei,vk,av+C %.1f | X1 %.1f | t,alpha,coef %.1f |H %.1f | w' dots %.1f |I+wp+put+fold %.1f |K %.1f | X2 %.1f (steps %lld)\n",
                     t[0] / sN, t[1] / sN, t[2] / sN, t[3] / sN, t[4] / sN, t[5] / sN, t[6] / sN, t[7] / sN, t[8] / sN, t[9] / sN,
                     (long long)st.cycle_steps);
        std::fprintf(stderr, "[cycle] XCC id mask of the active workgroups: 0x%llx; shader clock during the cycles: %.0f MHz\n",
                     (unsigned long long)t[15], t[10] > 0 ? 100.0 * (double)t[11] / (double)t[10] : 0.0);
    }
    st.loop_time = now_s() - t_loop0;
    if (warm.joinable()) warm.join();

    // ---- results (pdhg.jl:486-529)
    if (opt.certificate_search && certificate_search) {
        if (certificate_found) {
            std::vector<double> cc(P.c_orig);
            if (stop_reason == 6) std::fill(cc.begin(), cc.end(), 0.0);
            cache_solution(cc);
        }
    } else {
        cache_solution(P.c_orig);
    }
    (void)n_snap;
    if (debug && dbg_lz[5] > 0)
        std::fprintf(stderr, "[proxsdp] single-block Lanczos: %.0f cycles; per cycle: wait for the GPU %.1f us | after-cycle host logic (eigensolve, "
                     "convergence, restart rotation staging) %.1f us; per projection: results (Ritz coefficients + rotation staging) %.1f us\n",
                     dbg_lz[5], 1e6 * dbg_lz[0] / dbg_lz[5], 1e6 * dbg_lz[1] / dbg_lz[5], 1e6 * dbg_lz[3] / std::max(1.0, (double)st.lanczos_calls));
    if (b1_dbg.p != nullptr) {
        long long t[8] = {0};
        b1_dbg.download(t, 8, stream);
        PX_HIP(hipStreamSynchronize(stream));
        if (t[4] > 0)
            std::fprintf(stderr, "[proxsdp] k_lz_block1: %lld launches, %lld steps; per launch: prologue %.2f us | step loop %.2f us (%.2f us per step) | "
                         "epilogue %.2f us\n", t[4], t[3], 0.01 * t[0] / t[4], 0.01 * t[1] / t[4], 0.01 * t[1] / std::max(1LL, t[3]), 0.01 * t[2] / t[4]);
    }
    if (debug && dbg_batch[4] > 0)
        std::fprintf(stderr, "[dbg] batched Lanczos: %.0f cycles; per cycle enqueue %.1f us, wait %.1f us, restart logic %.1f us, flush %.1f us\n",
                     dbg_batch[4], 1e6 * dbg_batch[0] / dbg_batch[4], 1e6 * dbg_batch[1] / dbg_batch[4],
                     1e6 * dbg_batch[2] / dbg_batch[4], 1e6 * dbg_batch[3] / dbg_batch[4]);
    res.stats = st;
}

}  // namespace proxsdp
