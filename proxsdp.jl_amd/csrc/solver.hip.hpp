// Device-resident PDHG solve: the replacement for chambolle_pock
// (/root/reference/src/pdhg.jl:1-530) and everything below it.
//
// Host role: the scalar control logic of pdhg.jl:166-483 (a few dozen flops per
// iteration), the K x K Rayleigh-quotient eigenproblem of the thick-restart
// Lanczos, and kernel launches.  Every vector of length Nx or Q, the sparse
// operator in both orientations and the Lanczos basis live in HBM for the whole
// solve; per iteration the host reads back a handful of scalars.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <array>
#include <memory>
#include <mutex>
#include <condition_variable>
#include <functional>
#include <exception>
#include <rocblas/rocblas.h>
#include <rocsolver/rocsolver.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "host_util.hpp"
#include "kernels.hip.hpp"
#include "lanczos_cycle.hip.hpp"
#include "lanczos_block1.hip.hpp"
#include "small_sign.hip.hpp"
#include "sign_project.hip.hpp"
#include "rccl_dl.hpp"
#include "prep.hpp"

namespace proxsdp {

struct HipError : std::runtime_error { using std::runtime_error::runtime_error; };

#define PX_HIP(call)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess)                                                                \
            throw ::proxsdp::HipError(std::string(#call) + ": " + hipGetErrorString(e_) + " (" +       \
                           __FILE__ + ":" + std::to_string(__LINE__) + ")");                \
    } while (0)
#define PX_ROC(call)                                                                         \
    do {                                                                                     \
        rocblas_status s_ = (call);                                                          \
        if (s_ != rocblas_status_success)                                                    \
            throw ::proxsdp::HipError(std::string(#call) + ": rocblas status " + std::to_string((int)s_)); \
    } while (0)

template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    DevBuf() = default;
    explicit DevBuf(size_t count) { alloc(count); }
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
    DevBuf& operator=(DevBuf&& o) noexcept {
        if (this != &o) { release(); p = o.p; n = o.n; o.p = nullptr; o.n = 0; }
        return *this;
    }
    ~DevBuf() { release(); }
    void alloc(size_t count) {
        release();
        n = count;
        if (count == 0) return;
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&p), count * sizeof(T));
        if (e != hipSuccess) { p = nullptr; throw std::bad_alloc(); }
    }
    void release() { if (p) { (void)hipFree(p); p = nullptr; } n = 0; }
    void zero(hipStream_t s) { if (n) PX_HIP(hipMemsetAsync(p, 0, n * sizeof(T), s)); }
    void upload(const T* h, size_t count, hipStream_t s) {
        if (count) PX_HIP(hipMemcpyAsync(p, h, count * sizeof(T), hipMemcpyHostToDevice, s));
    }
    void download(T* h, size_t count, hipStream_t s) const {
        if (count) PX_HIP(hipMemcpyAsync(h, p, count * sizeof(T), hipMemcpyDeviceToHost, s));
    }
};

// pinned host memory (fast, truly asynchronous small read-backs)
struct PinnedBuf {
    double* p = nullptr;
    PinnedBuf() = default;
    PinnedBuf(const PinnedBuf&) = delete;
    PinnedBuf& operator=(const PinnedBuf&) = delete;
    PinnedBuf(PinnedBuf&& o) noexcept : p(o.p) { o.p = nullptr; }
    PinnedBuf& operator=(PinnedBuf&& o) noexcept { if (this != &o) { release(); p = o.p; o.p = nullptr; } return *this; }
    ~PinnedBuf() { release(); }
    void alloc(size_t count) {
        release();
        if (hipHostMalloc((void**)&p, count * sizeof(double), hipHostMallocDefault) != hipSuccess) { p = nullptr; throw std::bad_alloc(); }
    }
    void release() { if (p) { (void)hipHostFree(p); p = nullptr; } }
};

static inline double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
static inline int ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }
static inline int grid_for(long long n) {          // memory-bound passes: cap the grid, grid-stride the rest
    long long g = (n + dev::TPB - 1) / dev::TPB;
    return (int)std::max<long long>(1, std::min<long long>(g, 2048));
}

// structs.jl:2-30
struct CircularVector {
    std::vector<double> v;
    int l = 0;
    void init(int len) { v.assign(len, 0.0); l = len; }
    double& at(long long i) { return v[(size_t)(((i - 1) % l + l) % l)]; }
    double max_abs_diff() const {
        double val = 0.0;
        for (int i = 0; i < l; ++i) val = std::max(val, std::fabs(v[i] - v[(i + l - 1) % l]));
        return val;
    }
};

// ------------------------------------------------------------------ per-block Lanczos workspace
struct EigEvents {                  // optional profiling of the dominant kernel
    std::vector<hipEvent_t> e0, e1;
    size_t used = 0;
};

// The solver's stream.  Block workers (one host thread per PSD block while the blocks are
// projected concurrently) override it for their thread with the block's own stream, so every
// launch / copy / synchronisation inside the per-block code lands on that stream.
struct StreamRef {
    hipStream_t main = nullptr;
    static inline thread_local hipStream_t tl = nullptr;
    operator hipStream_t() const { return tl ? tl : main; }
};

// Host state of ONE thick-restart Lanczos run (one PSD block): what KrylovKit keeps between the restarts of an
// eigsolve call.  Split out of Solver::lanczos so that the single-block driver and the batched multi-block driver
// (lanczos_batch: one launch per step for several blocks) share the restart logic line by line.
struct LzRun {
    int nev = 0, krylovdim = 0, ld = 0;
    bool arpack = false, positive_part = false;
    double tol = 0.0, step_tol = 0.0;
    long long maxiter = 0;
    std::vector<double> T, Tw, D, U, f, al, be, Qa, da, ea;
    int howmany = 0, numiter = 1, converged = 0, K = 0, kfirst = 0, pos_count = -1, m_arrow = 0;
    bool pos_fail = false, presymv = false;
    double betaK = 0.0;
    // split + rank-one merge (host_eig_merge.hpp): first part solved under the GPU's cycle.  The solver object lives
    // in the block's workspace (its ~0.5 MB of tables are reused across projections, not re-allocated per call)
    SplitEig* splitp = nullptr;
    SplitEig& split_ref() { return *splitp; }
    bool split_ready = false;         // split.first() succeeded for the cycle in progress
    bool merge_active = false;        // D / f of the last eigensolve came from the merge: U holds no vectors yet
    std::vector<double> erow;
};

struct EigWork {
    int n = 0, nt = 0, npad = 0, nwg = 0, cap = 0, pld = 0;   // cap = columns of V (krylovdim_max + 1)
    int64_t N = 0;
    DevBuf<double> V, Z;            // npad x cap each (V: Krylov basis, Z: rotation target / Ritz vectors)
    DevBuf<double> w, Ppart, hpart1, hpart2, hsum1, hred, U, lam, resid, Apart, arrow;
    DevBuf<double> resid2;           // second, independent start vector: certificate run of a Lanczos-served full_eig!
    int napart = 0;
    PinnedBuf arrow_host;
    const double* arrow_p = nullptr;   // where the kernels read the arrow part: behind U after a restart upload
    // alphas[MAXK] | betas[MAXK] | LanczosCtl in ONE device record, read back with one copy
    DevBuf<double> rec;
    double* alphas_p = nullptr; double* betas_p = nullptr; dev::LanczosCtl* ctl_p = nullptr;
    PinnedBuf rec_pinned;           // pinned mirror of `rec`
    double* rec_host = nullptr;
    static constexpr size_t REC_DOUBLES = 2 * dev::MAXK + sizeof(dev::LanczosCtl) / sizeof(double);
    static constexpr size_t USTAGE_DOUBLES = (size_t)dev::MAXK * dev::MAXK + 2 * dev::MAXK;
    // full-eig fallback
    DevBuf<double> A, D, E;
    DevBuf<rocblas_int> info;
    std::vector<double> resid_host;
    PinnedBuf Ustage[2];            // pinned: the upload is a truly asynchronous copy (a pageable source makes the runtime stage and wait)
    int ustage_next = 0;
    // results of the last call
    std::vector<double> vals;
    int count = 0, converged_eigs = 0, numiter = 0, prev_numiter = 1;
    bool converged = false;
    // operator-form mat-vec (kernels.hip.hpp "Operator-form mat-vec")
    DevBuf<double> F, Flam, tpart, ebuf, apartf;   // (lam / Flam swap roles every projection)   // F: npad x cap, the previous projection's Ritz vectors (swapped with Z)
    DevBuf<int> ell_col, ell_sidx;                 // E in ELL form, [k * npad + row]
    DevBuf<int> wr_ptr, wr_row, wr_lo, wr_hi, ov_col, ov_sidx;   // entries beyond the ELL width (hub rows)
    dev::EllOverflow ov{};
    int ell_w = 0, F_first = 0, F_r = 0;
    bool fop_ok = false;                           // structures built (support path, narrow rows)
    bool have_factors = false;                     // x_prev of this block is F[:, F_first .. +F_r) diag(Flam) F'
    bool x_prev_sparse = true;                     // x_prev is zero off the support (initial iterate)
    bool use_fop = false;                          // the projection in progress uses the operator form
    int last_npos = -1;                            // positive eigenvalues found by the last full_eig! of this block
    // early read-back of the recurrence coefficients (host_eig_merge): side stream + events + pinned mirror
    hipStream_t side = nullptr;
    hipEvent_t ev_mid = nullptr, ev_early = nullptr;
    PinnedBuf rec_early;
    SplitEig split;
    int batch_slot = 0;                            // position of this block in the batched run in progress
    // one-workgroup cycle kernel (lanczos_block1.hip.hpp): the restart rotation is deferred into the next cycle's launch
    bool b1_defer = false;                         // Solver::rotate stages U and returns (set around lz_after_cycle)
    int b1_rot_K = 0;                              // > 0: a rotation is pending, U (K x kfirst, compact) at b1_rot_U (pinned staging)
    const double* b1_rot_U = nullptr;
    LzRun lzrun;                                   // host state of the run in progress (buffers reused across projections)
    long long fel_served = 0;                      // full_eig! calls of this block served by the Lanczos engine
    bool fel_disabled = false;                     // ... switched off after a failed verification (full_eig_lanczos_verify)
    int fel_cert_fails = 0;                        // failed certificates of this block (full_eig_lanczos_certify)
    // persistent Lanczos cycle kernel (lanczos_cycle.hip.hpp): granule buffers, epoch counter, error word
    DevBuf<double> xg1, xg2, warm_part;
    DevBuf<unsigned> xf1, xf2;
    DevBuf<int> cy_err;
    PinnedBuf cy_err_host;                         // pinned mirror of cy_err (first 4 bytes)
    unsigned cy_epoch = 1;
    bool cy_disabled = false;
    hipEvent_t cye[2] = {nullptr, nullptr};
    bool cye_pending = false;
    const double* esv = nullptr;                   // support values of E for the projection in progress
    // per-block execution context: counters are merged into the solver's after the projections
    hipStream_t stream = nullptr;                  // own stream (concurrent block projections)
    hipEvent_t done = nullptr;
    proxsdp_stats lst{};
    long long mv_iter = 0, recon_r = 0;
    EigEvents ev, evo;                             // profiled mat-vec / orthogonalisation launches
    hipEvent_t fe[3] = {nullptr, nullptr, nullptr};   // full_eig!: before solver | after solver | after reconstruction
    bool fe_pending = false;
    // full_eig! by the matrix sign function (sign_project.hip.hpp): A | X | X' | Y | Q, ld x ld each
    DevBuf<double> sgA, sgX, sgX2, sgY, sgQ, sg_part, sg_part2, sg_sc;
    PinnedBuf sg_host;
    int sg_ld = 0;
    int sg_npart = 0;
    int sg_row = -1, sg_ok = 16, sg_idle = 0, sg_hold = 0;   // shortened schedule of the sign iteration (full_eig_by_sign)
    hipEvent_t sg_ev = nullptr;
    bool sg_small = false;                         // products on 32 x 32 tiles (blocks up to side 3072)
    int sg_nt48 = 0;                               // > 0: products on 48 x 48 tiles, this many per side (chosen by makespan)
    bool sg_pending = false;                       // fe[] hold a sign projection's events (all of it is "solver")
    // cost-based engine choice on the Krylov branch (psd_sign_engine): wall-clock averages of this block's
    // Lanczos and sign projections, projections since the last Lanczos probe, scratch for in-place calls
    double kry_ms = -1.0, sign_ms = -1.0;
    int sign_streak = 0, sign_backoff = 0;
    int sign_verified_left = 0;                    // projections the engine may still serve before it is checked again
    bool sign_disabled = false, sign_check_pending = false;
    DevBuf<double> sg_cmp;                         // max |difference|, max |value| of a verification
    DevBuf<double> sg_out;
};


// requests of one batched rotation launch (lanczos_batch): filled through Solver::rotate while a sink is installed
struct RotSink {
    // one slot per block of the batch: filled by that block's restart logic (possibly on a helper thread), packed into
    // the contiguous pinned staging buffer by flush_rotations on the calling thread
    struct Req { EigWork* W = nullptr; const double* V = nullptr; double* out = nullptr; int K = 0, ncols = 0, copy_src = -1,
                 copy_dst = 0, nextra = 0; bool valid = false; std::vector<double> data; };
    std::array<Req, dev::LZB_MAX> slot;
};

class Solver {
public:
    // time0 is stamped BEFORE prepare(): the reference's clock starts at the top of chambolle_pock
    // (pdhg.jl:13), so preprocessing and equilibration count towards Result.time and time_limit
    Solver(const proxsdp_problem& prob, const proxsdp_options& opt_in, proxsdp_result& res_out)
        : opt(opt_in), res(res_out), time0(now_s()), P(prepare(prob, &opt_in)) {
        user_resid = prob.eig_resid;
        reduce_fn = prob.reduce_fn;
        reduce_ctx = prob.reduce_ctx;
        if (prob.nccl_comm != nullptr) {
            // native path: the library issues the collectives itself on its own stream (rccl_dl.hpp)
            Rccl& rc = Rccl::get();
            rc.require();
            nccl = static_cast<ncclComm_t>(prob.nccl_comm);
            rc.check(rc.CommCount(nccl, &nccl_world), "ncclCommCount");
            rc.check(rc.CommUserRank(nccl, &nccl_rank), "ncclCommUserRank");
            reduce_fn = nullptr;                     // (ignored when a communicator is given)
        }
        if (prob.n_coupling > 0) {
            if ((!prob.reduce_fn || !prob.reduce_vec_fn) && nccl == nullptr)
                throw std::invalid_argument("coupling rows need nccl_comm, or reduce_fn and reduce_vec_fn");
            if (!prob.coupling_rows || !prob.coupling_owned)
                throw std::invalid_argument("coupling rows need coupling_rows and coupling_owned");
            reduce_vec_fn = nccl ? nullptr : prob.reduce_vec_fn;
            reduce_vec_on_device = nccl ? true : prob.reduce_vec_on_device != 0;
            for (int64_t k = 0; k < prob.n_coupling; ++k) {
                const int64_t r = prob.coupling_rows[k];
                if (r < 0 || r >= prob.p + prob.m) throw std::invalid_argument("coupling row out of range");
                coup_rows.push_back((int)r);
                coup_owned.push_back(prob.coupling_owned[k] != 0);
            }
        }
    }
    // engine-only instance for the kernel-level test entry points (no problem data)
    Solver(const proxsdp_options& opt_in, proxsdp_result& res_out) : opt(opt_in), res(res_out) {
        time0 = now_s();
    }
    ~Solver() {
        if (warm.joinable()) warm.join();
        stop_workers();
        for (EigWork& W : eig) {
            for (auto e : W.ev.e0) (void)hipEventDestroy(e);
            for (auto e : W.ev.e1) (void)hipEventDestroy(e);
            for (auto e : W.evo.e0) (void)hipEventDestroy(e);
            for (auto e : W.evo.e1) (void)hipEventDestroy(e);
            for (auto e : W.fe) if (e) (void)hipEventDestroy(e);
            for (auto e : W.cye) if (e) (void)hipEventDestroy(e);
            if (W.done) (void)hipEventDestroy(W.done);
            if (W.stream) (void)hipStreamDestroy(W.stream);
            if (W.ev_mid) (void)hipEventDestroy(W.ev_mid);
            if (W.ev_early) (void)hipEventDestroy(W.ev_early);
            if (W.sg_ev) (void)hipEventDestroy(W.sg_ev);
            if (W.side) (void)hipStreamDestroy(W.side);
        }
        if (ev_main) (void)hipEventDestroy(ev_main);
        for (auto& pr : dense_ev) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
        if (blas) (void)rocblas_destroy_handle(blas);
        if (stream.main) (void)hipStreamDestroy(stream.main);
    }
    void run();

    // pieces also used by the kernel-level test entry points
    void setup_device();
    void alloc_eigwork(EigWork& W, int n, int max_nev);
    void lanczos(EigWork& W, const double* xp, int nev, bool positive_part = false);
    void lz_launch_step(EigWork& W, const double* xp, int k, int kfirst, double step_tol, bool& presymv);
    bool lanczos_certificate(EigWork& W, const double* xp, int npos, int msteps, double& theta_max, double& scale);
    void lanczos_batch(const std::vector<int>& blocks, const double* xbase, const std::vector<int>& nevs, int ctx_slot = 0);
    // batched rotations: U of every block staged in ONE pinned buffer, one upload, one launch (grid.z = block)
    double dbg_batch[5] = {0, 0, 0, 0, 0};         // debug: enqueue | wait | restart logic | flush seconds, cycles
    double dbg_lz[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // debug (PROXSDP_HIP_DEBUG): single-block run -- wait for the cycle | second() +
                                                   // merge | last row + convergence | Ritz coefficients | staging + upload + launch | cycles
    // (thread-local: several batched runs may be in progress at once, one per group of blocks, each on its own worker thread)
    static inline thread_local RotSink* rot_sink = nullptr;
    // per-GROUP state of a batched run (round 5: equal-side blocks are split into groups that run CONCURRENTLY, each on its own
    // stream and host thread -- one group's restart logic on the host overlaps the other group's cycle on the GPU)
    struct BatchCtx {
        DevBuf<double> U;
        PinnedBuf U_host, rec_host;
        hipEvent_t ev = nullptr;                   // end of a batched cycle's record copies (the speculated mat-vecs run behind it)
        std::unique_ptr<SpinPool> pool;            // helper threads for the per-block restart logic
        ~BatchCtx() { if (ev) (void)hipEventDestroy(ev); }
    };
    std::vector<std::unique_ptr<BatchCtx>> batch_ctx;
    std::mutex batch_stats_mu;
    // helper threads for the rank-one merge of the K x K eigensolve (secular roots, Gu-Eisenstat weights, eigenvector columns:
    // independent per root / column, so the results do not depend on who computes them).  They spin only while a projection
    // with krylovdim >= 64 is in progress (armed at its start, disarmed at its end).
    std::unique_ptr<SpinPool> merge_pool;
    ParFor merge_par;
    int merge_helpers() const { return opt.host_merge_threads < 0 ? 3 : std::min(15, (int)opt.host_merge_threads); }
    static constexpr size_t LZB_USTRIDE = 64 * 64 + 2 * dev::MAXK;
    void flush_rotations(RotSink& sink, BatchCtx& C);
    bool lz_init(EigWork& W, struct LzRun& R, int nev, bool positive_part);
    void lz_prepare_arrow(struct LzRun& R);
    bool lz_after_cycle(EigWork& W, struct LzRun& R, bool speculated);
    void lz_finish_run(EigWork& W, struct LzRun& R);
    void lanczos_eager(EigWork& W, const double* xp, int nev);
    void lz_merge_vectors(EigWork& W, struct LzRun& R, int ncols);
    bool lz_split_first(EigWork& W, struct LzRun& R, int k1);
    void full_eig_values(EigWork& W, const double* xp, double offscale, bool vectors, std::vector<double>& Dhost);
    void harvest_full_eig_events(EigWork& W);
    bool cycle_plan(const EigWork& W, int krylovdim, int& R, int& G, bool& f_in_lds) const;
    void launch_cycle(EigWork& W, int kfirst, int krylovdim, double tol, int R, int G, bool f_in_lds);
    int cycle_lds_cap = 0;                        // dynamic LDS granted to k_lz_cycle (setup_device)
    int block1_lds_cap = 0;                       // dynamic LDS granted to k_lz_block1 (setup_device)
    DevBuf<long long> b1_dbg;                     // PROXSDP_HIP_DEBUG_B1: tick sums of k_lz_block1 (prologue | loop | epilogue | steps | launches)
    bool block1_plan(const EigWork& W, int krylovdim, bool split_wanted) const;
    void launch_block1(EigWork& W, const double* xp, int kfirst, int krylovdim, double tol);
    bool sg48_ok = false;                         // 72 KiB of dynamic LDS granted to k_sym_gemm48 (setup_device)
    DevBuf<long long> cy_dbg;                     // PROXSDP_HIP_DEBUG_CYCLE: per-phase tick sums
    void launch_symv(EigWork& W, const double* xp, const double* v, bool use_ctl);
    void launch_symv_finish(EigWork& W, const double* xp, int kclose, double tol, bool use_carry);
    void launch_reconstruct(EigWork& W, const double* Z, int ldz, const double* lam, int r, double* xp_out,
                            const double* xp_old = nullptr, int blk = -1);
    void rotate(EigWork& W, int K, const std::vector<double>& U, int ldu, int ncols, double* out, int copy_src, int copy_dst,
                const double* extra, int nextra);

    // test hooks (capi.hip)
    void test_project(int idx, double* xp, int tr);
    void test_full_eig(const double* xin, double* xout) { current_rank.assign(1, 0); min_eig.assign(1, 0.0); full_eig_project(0, xin, xout, false); }
    long long test_rank() const { return current_rank.empty() ? 0 : current_rank[0]; }
    double test_min_eig() const { return min_eig.empty() ? 0.0 : min_eig[0]; }
    void test_spmv(bool transpose, const double* in, double* out);

    bool debug = std::getenv("PROXSDP_HIP_DEBUG") != nullptr;
    proxsdp_options opt;
    proxsdp_result& res;
    double time0 = 0;               // (declared before P: initialised first)
    Prep P;
    StreamRef stream;
    rocblas_handle blas = nullptr;
    std::vector<EigWork> eig;
    EigEvents ev;
    proxsdp_stats st{};
    const double* user_resid = nullptr;
    // block-sharded solve: scalar all-reduce across shards (include/proxsdp_hip.h)
    int (*reduce_fn)(void*, double*, int32_t, double*, int32_t) = nullptr;
    void* reduce_ctx = nullptr;
    bool sharded() const { return reduce_fn != nullptr || nccl != nullptr; }
    std::exception_ptr shard_error;                 // this shard's projection failed in the current iteration (see primal_step_dev)
    // native RCCL path (proxsdp_problem.nccl_comm): one all-gather of the packed scalar record per reduce, combined
    // on the host in rank order (the same bits on every rank); all-reduce of the coupling buffer on the stream
    ncclComm_t nccl = nullptr;
    int nccl_world = 1, nccl_rank = 0;
    DevBuf<double> nccl_send, nccl_recv, nccl_tmp;
    PinnedBuf nccl_host;
    size_t nccl_cap = 0;
    void reduce_native(std::vector<double>& sums, std::vector<double>& maxs);
    // coupling rows of a block-sharded solve (include/proxsdp_hip.h): partial M x summed over the shards
    int (*reduce_vec_fn)(void*, double*, int64_t, int32_t) = nullptr;
    bool reduce_vec_on_device = false;
    std::vector<int> coup_rows;
    std::vector<char> coup_owned;
    DevBuf<int> coup_rows_d;
    DevBuf<double> coup_buf_d, roww_d;
    std::vector<double> coup_host;
    void reduce_coupling(double* Mx_dev);          // Mx[coupling rows] <- sum over shards
    // wait for the solver's (or the calling block thread's) stream at a read-back on the critical path
    void wait_stream() {
        hipStream_t s = stream;
        if (nccl) { wait_collective(s); return; }
        if (opt.host_wait_spin == 0) { PX_HIP(hipStreamSynchronize(s)); return; }
        for (;;) {
            const hipError_t e = hipStreamQuery(s);
            if (e == hipSuccess) return;
            if (e != hipErrorNotReady) PX_HIP(e);
#if defined(__x86_64__)
            _mm_pause();
#endif
        }
    }
    // Native RCCL path: every host wait that may sit behind a collective is BOUNDED (a peer that left the
    // solve -- an exception between two collectives, a dead process -- used to leave this rank in
    // hipStreamSynchronize for ever).  After PROXSDP_HIP_COLLECTIVE_TIMEOUT_S seconds (default 300) without progress
    // this rank aborts its communicator (ncclCommAbort: its own pending collective kernels stop) and fails the solve.
    double collective_timeout_s = -1.0;
    void wait_collective(hipStream_t s) {
        if (collective_timeout_s < 0.0) {
            const char* e = std::getenv("PROXSDP_HIP_COLLECTIVE_TIMEOUT_S");
            collective_timeout_s = (e && std::atof(e) > 0.0) ? std::atof(e) : 300.0;
        }
        const double t0 = now_s();
        for (unsigned spins = 0;; ++spins) {
            const hipError_t e = hipStreamQuery(s);
            if (e == hipSuccess) return;
            if (e != hipErrorNotReady) PX_HIP(e);
            if ((spins & 0xfff) == 0xfff && now_s() - t0 > collective_timeout_s) {
                abort_comm();
                throw std::runtime_error("block-sharded solve: a collective did not complete within " +
                                         std::to_string((int)collective_timeout_s) + " s (did another shard leave the solve?)");
            }
#if defined(__x86_64__)
            _mm_pause();
#endif
        }
    }
    // local abort of the native communicator (it is unusable afterwards; the caller destroys it)
    void abort_comm() {
        if (!nccl) return;
        Rccl& rc = Rccl::get();
        if (rc.ok() && rc.CommAbort) (void)rc.CommAbort(nccl);
        nccl_aborted = true;
    }
    bool nccl_aborted = false;
    bool collective_enqueued = false;     // a collective of this solve may be pending on the stream (abort only then)
    // state seam (proxsdp_hip_solve_ex): continue from / write out the iterate at an iteration boundary
    const proxsdp_state* resume_state = nullptr;
    proxsdp_state* capture_state = nullptr;
    void check_state_shape(const proxsdp_state& s, const char* what) const;
    void apply_resume();
    void write_capture();
    int ada_count = 0;                    // pdhg.jl:306-332 (a local of chambolle_pock)
    double last_resid_s = 0.0;            // host part of compute_residual! / compute_gap! of the last fused linesearch call
    void wait_event(hipEvent_t ev) {
        if (opt.host_wait_spin == 0) { PX_HIP(hipEventSynchronize(ev)); return; }
        for (;;) {
            const hipError_t e = hipEventQuery(ev);
            if (e == hipSuccess) return;
            if (e != hipErrorNotReady) PX_HIP(e);
#if defined(__x86_64__)
            _mm_pause();
#endif
        }
    }
    void reduce_vec_host(std::vector<double>& v) {
        if (v.empty()) return;
        if (nccl) {                                  // (exit path: slacks of the coupling rows)
            Rccl& rc = Rccl::get();
            if (nccl_tmp.n < v.size()) nccl_tmp.alloc(v.size());
            nccl_tmp.upload(v.data(), v.size(), stream);
            collective_enqueued = true;
            rc.check(rc.AllReduce(nccl_tmp.p, nccl_tmp.p, v.size(), ncclFloat64, ncclSum, nccl, stream), "ncclAllReduce");
            nccl_tmp.download(v.data(), v.size(), stream);
            wait_collective(stream);
            st.rccl_reductions++;
            return;
        }
        if (reduce_vec_fn(reduce_ctx, v.data(), (int64_t)v.size(), 0) != 0) throw std::runtime_error("reduce_vec_fn failed");
    }
    void reduce(std::vector<double>& sums, std::vector<double>& maxs) {
        if (nccl) { reduce_native(sums, maxs); return; }
        if (!reduce_fn) return;
        if (reduce_fn(reduce_ctx, sums.data(), (int32_t)sums.size(), maxs.data(), (int32_t)maxs.size()) != 0)
            throw std::runtime_error("reduce_fn failed");
    }
    // global (all-shard) problem constants; equal to the local ones when not sharded
    double g_n = 0, g_Q = 0, g_p = 0, g_m = 0, g_norm_b = 0, g_norm_h = 0, g_norm_c = 0, g_frob = 0;
    bool g_conic = false, g_not_converged_rank = false, g_any_below_full = false;
    double g_elapsed = 0;
    std::thread warm;               // loads rocSOLVER's code objects while the loop runs
    // concurrent block projections: one worker thread per PSD block (up to 8), each driving its
    // block's Lanczos on the block's own stream
    std::mutex blas_mutex;
    std::vector<std::thread> workers;
    std::mutex pool_mu;
    std::condition_variable pool_cv, pool_done_cv;
    std::vector<int> pool_queue;               // block indices waiting for a worker
    int pool_pending = 0;
    bool pool_stop = false;
    std::exception_ptr pool_error;
    std::function<void(int)> pool_job;
    hipEvent_t ev_main = nullptr;
    bool parallel_blocks = false;
    void start_workers(int nthreads);
    void stop_workers();
    void run_blocks(const std::vector<int>& blocks, const std::function<void(int)>& job);
    void merge_block_stats();
    void start_rocsolver_warmup();

private:
    // device state
    DevBuf<double> xbuf[2], Mtybuf[2], ybuf[2], Mxbuf[2], c_d, bh_d, part, scal;
    DevBuf<int> csr_ptr, csr_col, csc_ptr, csc_row;
    DevBuf<double> csr_val, csc_val;
    DevBuf<long long> one_off, soc_off;
    DevBuf<int> soc_len;
    DevBuf<double> one_min, soc_gap_d;
    int xc = 0, mtyc = 0, yc = 0, mxc = 0;     // index of the "current" buffer of each ping-pong pair
    std::vector<int> one_blocks;               // indices of 1x1 PSD blocks
    std::vector<int> big_blocks;               // indices of the blocks that are projected by an eigensolver
    // blocks of side 2..64 below min_size_krylov_eigs: one batched Jacobi launch on the dense vector path
    std::vector<int> small_blocks, large_blocks;
    DevBuf<long long> small_off;
    DevBuf<int> small_side, small_rank;        // small_rank: [rank | npos] per block
    PinnedBuf small_rank_host;
    int small_maxn = 0;
    int small_jacobi_max = 64;                 // small blocks up to this side: batched Jacobi; above: k_small_sign_project
    int small_sign_maxn = 0;                   // largest side served by k_small_sign_project (0: none)
    int small_sign_cap = 0;                    // largest side whose dynamic LDS the device grants (setup_device; 64 on gfx950)
    // per-iteration read-backs written by the kernels straight into pinned host memory: only where the iteration leaves little
    // dirty data behind (a kernel that stores to host memory ends with a system-scope release)
    bool zero_copy_small() const { return P.n <= (1 << 16); }
    bool small_pending = false;
    void project_small_blocks(double* x);
    void harvest_small_ranks();
    std::vector<double> hscal;
    bool csr_wave = false;
    int rotate_lds_cap = 60 * 1024;           // dynamic LDS granted to k_lz_rotate (setup_device)
    int rotate_mfma_lds_cap = 0;              // ... to k_lz_rotate_mfma
    // rows longer than LONG_ROW entries: segmented SpMV (kernels.hip.hpp k_spmv_csr_seg)
    static constexpr int LONG_ROW = 8192;
    DevBuf<int> seg_lo_d, seg_hi_d, long_row_d, long_ptr_d;
    DevBuf<double> segpart_d;
    int n_seg = 0, n_long = 0;
    void setup_long_rows(const std::vector<int>& rp);

    // scalar state (Params, structs.jl:159-192)
    double theta = 1, beta = 1, adapt_level = 0.9, primal_step = 0, primal_step_old = 0, dual_step = 0;
    long long iter = 0;
    int rank_update = 0, update_cont = 0, stop_reason = 0;
    std::string stop_reason_string = "Not optimized";
    std::vector<long long> target_rank, current_rank;
    std::vector<double> min_eig;
    double dual_feasibility = -1.0;
    bool certificate_search = false, certificate_found = false;
    long long certificate_search_min_iter = 0;
    long long max_iter_local = 0;
    double time_limit = 0;
    int last_trials = 0;
    long long lz_matvec_iter = 0; long long recon_r_iter = 0;
    // histories (Residuals, structs.jl:100-125)
    CircularVector h_gap, h_pobj, h_dobj, h_feas, h_pres, h_dres, h_comb;
    double equa_feasibility = 0, ineq_feasibility = 0;

    void primal_step_dev();
    void psd_projection(double* x);
    void project_block(int idx, const double* xin, double* xout, bool fuse, bool lanczos_done = false);
    void project_blocks(const std::vector<int>& blocks, const double* xin, double* xout, bool fuse);
    bool krylov_branch(int idx) const;
    bool batch_eligible(int idx, bool fuse) const;
    void setup_support();
    int  linesearch_residual_support();
    int  linesearch_residual_general();
    void setup_dense();
    int  linesearch_dense();
    void dense_mv(const double* x, double* y, bool scaled);
    void dense_mtv(int nc, const double* Y, long long ystride, bool scaled, double* OUT, long long ostride,
                   const double* old, const double* addc, double* normpart, long long cstride, bool addback);
    void full_eig_project(int idx, const double* xp_in, double* xp_out, bool fuse);
    bool full_eig_by_lanczos(int idx, const double* xp_in, double* xp_out, bool fuse);
    bool full_eig_by_sign(int idx, const double* xp_in, double* xp_out, bool fuse, bool force = false);
    void truncated_project_dense(int idx, const double* xp_in, double* xp_out, bool fuse, int nev);
    // acceptance threshold of the Lanczos-served full_eig! (options.full_eig_lanczos_posres, default 1e-6 of the spectral
    // scale), never looser than 1 % of the tightest tolerance the user asked for (a user at tolerance 1e-7
    // must not be handed a projection that may miss a positive eigenvalue of 1e-6)
    double lanczos_posres() const {
        const double base = opt.full_eig_lanczos_posres > 0.0 ? opt.full_eig_lanczos_posres : 1e-6;
        const double tmin = std::min({opt.tol_gap, opt.tol_feasibility, opt.tol_primal, opt.tol_dual});
        return tmin > 0.0 ? std::min(base, 1e-2 * tmin) : base;
    }
    bool krylovdim_fits(int nev) const { return std::max(2 * nev + 1, (int)opt.eigsolver_min_lanczos) <= dev::MAXK - 1; }
    bool exact_projection_by_sign(int idx, const double* xp, double* xo, bool fuse, int nev);
    void verify_sign_engine(int idx, const double* xo);
    template <int EPI, bool FUSE>
    void sym_gemm(EigWork& W, const double* Pm, const double* Qm, double* T, const double* Y, double ca, double cb,
                  double cc, const double* dsc, double* part, double* xp_out, const double* xp_old, int blk);
    void spmv(const double* x, double* y);
    void spmv_sparse(const double* x, double* y);
    int  linesearch();
    void dual_step_plain();
    void residual_and_gap();
    bool convergedrank() const;
    bool soc_convergence();
    double dual_feas_host(const std::vector<double>& y, const std::vector<double>& cvec,
                          std::vector<double>* dual_eq, std::vector<double>* dual_in, std::vector<double>* dual_cone);
    void cache_solution(const std::vector<double>& cvec);
    void certificate_parameters();
    void bump_rank(int idx);
    std::vector<double> b_host, h_host, c_host;   // current (possibly zeroed by a certificate search)
    bool have_snapshot = false;
    // support-aware vector passes
    bool use_support = false;
    int ns = 0, rstride = 0, n_res_wg = 0;
    std::vector<int> tile_base;                 // first residual-partial slot of each PSD block
    DevBuf<int> supp_d;
    DevBuf<unsigned> mask_d;
    DevBuf<double> cS_d, xsave_d, MtyS_cur, MtyS_cand, ycand_d, respart_d, bpart, bscal;
    DevBuf<double> Ediag_d, Ddiag_d; // equilibration diagonals (exit path)
    DevBuf<double> esv_d;            // [2][ns]: support values of E (update only | whole entry), k_primal_update_S
    // dense constraint matrix (proxsdp_problem.M_dense): borrowed device pointer or own upload
    const double* Md = nullptr;
    DevBuf<double> Md_own, dmv_part, Mtycand_d;
    DevBuf<unsigned char> offdiag_d;
    int dmv_slices = 1, dmv_qpad = 0;
    static constexpr int DMV_ROWS = 8, DMV_UNR = 4;       // k_dense_mv shape (rows per workgroup, strips in flight)
    long long dense_passes_seen = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> dense_ev;   // one pair per pass of the current iteration
    size_t dense_ev_used = 0;
    void dense_ev_begin();
    void dense_ev_end();
    void dense_ev_harvest();
    std::vector<double> hbscal;
    PinnedBuf hscal_pin;            // read-back of the general path's scalars (linesearch norms | residual / gap sums)
    bool residual_ready = false;    // the accepted linesearch candidate's residual scalars are already in hscal
    void enqueue_residual(double pstep, double dstep);
};

// ------------------------------------------------------------------ setup
inline void Solver::alloc_eigwork(EigWork& W, int n, int max_nev) {
    W.n = n;
    W.N = (int64_t)n * (n + 1) / 2;
    W.nt = ceil_div(n, dev::TILE);
    W.npad = W.nt * dev::TILE;
    W.nwg = ceil_div(n, dev::TPB);
    // the step kernels hold up to dev::MAXK - 1 = 255 basis columns; a projection whose Krylov dimension
    // max(2 target_rank + 1, eigsolver_min_lanczos) is larger (options.jl:76,88 accept any value) is served by the dense
    // eigensolver instead (Solver::truncated_project_dense): the workspace is sized for what the kernels can run
    int kd = std::min(std::max(2 * max_nev + 1, (int)opt.eigsolver_min_lanczos), dev::MAXK - 1);
    W.cap = kd + 1;
    W.V.alloc((size_t)W.npad * W.cap);
    W.Z.alloc((size_t)W.npad * W.cap);
    W.w.alloc(W.npad);
    W.Ppart.alloc((size_t)W.nt * W.npad);
    W.pld = ceil_div(W.nt, dev::WAVE) * dev::WAVE;   // partial dots: [MAXK][pld], padding stays zero
    W.hpart1.alloc((size_t)W.pld * dev::MAXK);
    W.hpart2.alloc((size_t)W.pld * dev::MAXK);
    W.hpart1.zero(stream); W.hpart2.zero(stream);
    W.hsum1.alloc(dev::MAXK);
    W.hred.alloc(dev::MAXK); W.hred.zero(stream);
    W.napart = 8 * ceil_div(W.nt * (W.nt + 1) / 2, 8);
    W.Apart.alloc(W.napart); W.Apart.zero(stream);      // padding tiles never write: stay zero
    W.arrow.alloc(2 * dev::MAXK); W.arrow.zero(stream);   // arrow: f | D (see k_lz_orth)
    W.arrow_host.alloc(2 * dev::MAXK);
    W.rec.alloc(EigWork::REC_DOUBLES);
    W.alphas_p = W.rec.p; W.betas_p = W.rec.p + dev::MAXK;
    W.ctl_p = reinterpret_cast<dev::LanczosCtl*>(W.rec.p + 2 * dev::MAXK);
    W.rec_pinned.alloc(EigWork::REC_DOUBLES);
    W.rec_host = W.rec_pinned.p;
    W.U.alloc(EigWork::USTAGE_DOUBLES);
    W.Ustage[0].alloc(EigWork::USTAGE_DOUBLES); W.Ustage[1].alloc(EigWork::USTAGE_DOUBLES);
    W.arrow_p = W.arrow.p;
    W.lam.alloc(std::max(n, dev::MAXK));
    W.resid.alloc(W.npad);
    W.V.zero(stream); W.Z.zero(stream); W.w.zero(stream);
    W.hsum1.zero(stream); W.rec.zero(stream);
    W.resid.zero(stream);
}

inline void Solver::setup_device() {
    int ndev = 0;
    PX_HIP(hipGetDeviceCount(&ndev));
    if (ndev <= 0) throw HipError("no HIP device available");
    if (opt.device_id < 0 || opt.device_id >= ndev) throw std::invalid_argument("device_id out of range");
    PX_HIP(hipSetDevice(opt.device_id));
    PX_HIP(hipStreamCreate(&stream.main));
    // gfx950: 160 KiB of LDS per CU.  The restart rotation at K = 127, keep = 78 needs 145 KiB (U tile + V tile): with 144 KiB
    // it fell into two launches (54 us on average, profiles/r03b); take the largest grant the runtime accepts
    rotate_lds_cap = 60 * 1024;
    for (int kb : {160, 156, 152, 144}) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(dev::k_lz_rotate),
                                hipFuncAttributeMaxDynamicSharedMemorySize, kb * 1024) == hipSuccess) { rotate_lds_cap = kb * 1024; break; }
        (void)hipGetLastError();
    }
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dev::k_lzb_rotate), hipFuncAttributeMaxDynamicSharedMemorySize, rotate_lds_cap);
    for (int kb : {156, 152, 144, 128, 96, 64}) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(dev::k_lz_rotate_mfma),
                                hipFuncAttributeMaxDynamicSharedMemorySize, kb * 1024) == hipSuccess) { rotate_mfma_lds_cap = kb * 1024; break; }
        (void)hipGetLastError();
    }
    sg48_ok = true;
    for (const void* f : {reinterpret_cast<const void*>(dev::k_sym_gemm48<dev::SG_PLAIN, 4>), reinterpret_cast<const void*>(dev::k_sym_gemm48<dev::SG_POLY, 4>)})
        sg48_ok = sg48_ok && hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dev::sg48_lds_bytes(4)) == hipSuccess;
    if (!sg48_ok) (void)hipGetLastError();
    if (std::getenv("PROXSDP_HIP_DEBUG_CYCLE") != nullptr) { cy_dbg.alloc(16); cy_dbg.zero(stream); }
    cycle_lds_cap = 0;
    for (int kb : {160, 156, 152, 144, 128, 96, 64}) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(dev::k_lz_cycle<128>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, kb * 1024) == hipSuccess &&
            hipFuncSetAttribute(reinterpret_cast<const void*>(dev::k_lz_cycle<64>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, kb * 1024) == hipSuccess) {
            cycle_lds_cap = kb * 1024;
            break;
        }
        (void)hipGetLastError();
    }
    // one-launch small-block projection: 4 matrices of side x (side + 2) doubles in LDS (132 KiB at side 64).  Probed here so
    // that a device / partition that does not grant it sends the larger of those blocks to the tiled engines instead of
    // failing the solve
    small_sign_cap = 0;
    for (int sd : {64, 56, 48, 40, 32, 24, 16}) {
        const size_t bytes = dev::small_sign_lds_bytes(sd);
        if (bytes <= 48 * 1024 || hipFuncSetAttribute(reinterpret_cast<const void*>(dev::k_small_sign_project),
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess) {
            small_sign_cap = sd;
            break;
        }
        (void)hipGetLastError();
    }
    if (std::getenv("PROXSDP_HIP_DEBUG_B1") != nullptr) { b1_dbg.alloc(8); b1_dbg.zero(stream); }
    block1_lds_cap = 0;
    for (int kb : {160, 144, 128, 112, 96}) {
        bool ok = true;
        for (const void* f : {reinterpret_cast<const void*>(dev::k_lz_block1<1, false>), reinterpret_cast<const void*>(dev::k_lz_block1<2, false>),
                              reinterpret_cast<const void*>(dev::k_lz_block1<1, true>), reinterpret_cast<const void*>(dev::k_lz_block1<2, true>)})
            ok = ok && hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, kb * 1024) == hipSuccess;
        if (ok) { block1_lds_cap = kb * 1024; break; }
        (void)hipGetLastError();
    }
    PX_ROC(rocblas_create_handle(&blas));
    PX_ROC(rocblas_set_stream(blas, stream));
}

// one-workgroup cycle kernel for medium blocks (lanczos_block1.hip.hpp): the step kernels' arithmetic bit for bit, one launch
// per Lanczos cycle.  lanczos_cycle_kernel: 2 = on, -1 (auto) = on where it applies, 0 / 1 = off.
inline bool Solver::block1_plan(const EigWork& W, int krylovdim, bool split_wanted) const {
    if (!(opt.lanczos_cycle_kernel == 2 || opt.lanczos_cycle_kernel < 0) || block1_lds_cap == 0) return false;
    if (W.nt > dev::B1_MAXNT || krylovdim > dev::B1_KMAX || split_wanted || rot_sink != nullptr) return false;
    // auto: one row group per virtual workgroup (side <= 256).  Measured (profiles/r06_medium_blocks.md): one CU issues ~700
    // instructions per wave and step for 16 waves -- 4.6-5.3 us per step against the step kernels' ~12 at side <= 192 -- but
    // with two row groups per wave (side 257 .. 512) it is 8-12 us per step: no gain, so those sides stay with the step kernels
    if (opt.lanczos_cycle_kernel < 0 && W.nt > dev::B1_NV) return false;
    if (W.use_fop && (W.F_r > 4 * dev::B1_NPV || W.ov.wr_ptr != nullptr || W.ell_w < 1)) return false;
    // the packed triangle must be resident in LDS (one CU cannot stream it from L2 once per step fast enough: measured,
    // profiles/r06_medium_blocks.md); the operator form's E goes there too when it fits, else it is read through L2
    const dev::B1Lds L = dev::b1_lds_plan(W.nt, W.npad, W.use_fop, W.use_fop ? 0 : W.N, 0);
    return (size_t)L.total * sizeof(double) <= (size_t)block1_lds_cap;
}

inline void Solver::launch_block1(EigWork& W, const double* xp, int kfirst, int krylovdim, double tol) {
    dev::Block1Args a{};
    a.xp = xp; a.n = W.n; a.nt = W.nt; a.npad = W.npad;
    a.ldv = W.npad; a.kfirst = kfirst; a.kd = krylovdim; a.tol = tol;
    a.Vout = W.V.p;
    if (W.b1_rot_K > 0) {                        // restart rotation deferred by Solver::rotate: old basis in Z (the pointers were swapped)
        a.Vin = W.Z.p; a.U = W.b1_rot_U; a.rotK = W.b1_rot_K;
        W.b1_rot_K = 0;
    } else {
        a.Vin = W.V.p; a.U = nullptr; a.rotK = 0;
    }
    bool ell_lds = false;
    if (W.use_fop) {
        a.Vp = W.F.p + (size_t)W.F_first * W.npad; a.lam = W.Flam.p; a.rp = W.F_r;
        a.ell_col = W.ell_col.p; a.ell_sidx = W.ell_sidx.p; a.ell_w = W.ell_w; a.esv = W.esv;
        const dev::B1Lds Le = dev::b1_lds_plan(W.nt, W.npad, true, 0, W.ell_w);
        ell_lds = (size_t)Le.total * sizeof(double) <= (size_t)block1_lds_cap;
    }
    a.ell_in_lds = ell_lds ? 1 : 0;
    a.arrow = W.arrow_p;
    a.alphas = W.alphas_p; a.betas = W.betas_p; a.ctl = W.ctl_p;
    a.dbg = b1_dbg.p;
    const dev::B1Lds L = dev::b1_lds_plan(W.nt, W.npad, W.use_fop, W.use_fop ? 0 : W.N, ell_lds ? W.ell_w : 0);
    const size_t lds = (size_t)L.total * sizeof(double);
    const bool two = W.nt > dev::B1_NV;
    auto kern = W.use_fop ? (two ? dev::k_lz_block1<2, true> : dev::k_lz_block1<1, true>)
                          : (two ? dev::k_lz_block1<2, false> : dev::k_lz_block1<1, false>);
    const bool prof = opt.profile_symv_every > 0;
    if (prof) {
        if (W.cye[0] == nullptr) { PX_HIP(hipEventCreate(&W.cye[0])); PX_HIP(hipEventCreate(&W.cye[1])); }
        hipExtLaunchKernelGGL(kern, dim3(1), dim3(dev::B1_TPB), lds, stream, W.cye[0], W.cye[1], 0, a);
        W.cye_pending = true;
    } else {
        hipLaunchKernelGGL(kern, dim3(1), dim3(dev::B1_TPB), lds, stream, a);
    }
    W.lst.cycle_launches++;
    W.lst.cycle_steps += krylovdim - kfirst;
    W.lst.symv_launches += krylovdim - kfirst;
    W.lst.symv_bytes += (double)(krylovdim - kfirst) * (8.0 * (double)W.N + 16.0 * (double)W.n);
}

// persistent cycle kernel: rows per workgroup, grid and whether the previous factors fit in LDS
inline bool Solver::cycle_plan(const EigWork& W, int krylovdim, int& R, int& G, bool& f_in_lds) const {
    if (std::getenv("PROXSDP_HIP_DEBUG_CYCLE") != nullptr)
        std::fprintf(stderr, "[cycle_plan] knob %d disabled %d cap %d par %d fop %d ov %d xg %zu kd %d Fr %d ellw %d npad %d\n",
                     (int)opt.lanczos_cycle_kernel, (int)W.cy_disabled, cycle_lds_cap, (int)parallel_blocks, (int)W.use_fop,
                     (int)(W.ov.wr_ptr != nullptr), W.xg1.n, krylovdim, W.F_r, W.ell_w, W.npad);
    // auto (-1) = off: measured 10.9 us per Lanczos step against 14.1 us for the step kernels at n = 4000,
    // K = 25 -- +5 % iterations/s, not worth an assumption about workgroup placement by default
    if (opt.lanczos_cycle_kernel != 1 || W.cy_disabled || cycle_lds_cap == 0 || parallel_blocks) return false;
    if (!W.use_fop || W.ov.wr_ptr != nullptr || W.xg1.n == 0) return false;
    if (krylovdim + 2 > dev::CY_CMAX - 2 || W.F_r > 126) return false;
    const size_t cap = (size_t)cycle_lds_cap - 512;
    for (int r : {128, 64}) {
        const int S = dev::NWAVE / (r / 64);
        if (W.ell_w > dev::CY_NE * S) continue;
        const int g = ceil_div(W.npad, r);
        if (g > 32) continue;                               // one XCD: 32 CUs, one workgroup each
        for (bool fl : {true, false}) {
            if (dev::cycle_lds_bytes(krylovdim, W.F_r, r, g, fl) <= cap) { R = r; G = g; f_in_lds = fl; return true; }
        }
    }
    return false;
}

inline void Solver::launch_cycle(EigWork& W, int kfirst, int krylovdim, double tol, int R, int G, bool f_in_lds) {
    dev::CycleArgs a{};
    a.V = W.V.p; a.ldv = W.npad; a.n = W.n; a.npad = W.npad;
    a.kfirst = kfirst; a.kd = krylovdim; a.tol = tol;
    a.Vp = W.F.p + (size_t)W.F_first * W.npad; a.rp = W.F_r; a.lam = W.Flam.p;
    a.ell_col = W.ell_col.p; a.ell_sidx = W.ell_sidx.p; a.ell_w = W.ell_w; a.esv = W.esv;
    a.arrow = W.arrow_p;
    a.alphas = W.alphas_p; a.betas = W.betas_p; a.ctl = W.ctl_p;
    a.x1 = W.xg1.p; a.x2 = W.xg2.p; a.xs1 = dev::CY_CMAX; a.xs2 = dev::CY_CMAX + 128;
    a.f1 = W.xf1.p; a.f2 = W.xf2.p;
    a.epoch0 = W.cy_epoch;
    a.R = R; a.G = G; a.f_in_lds = f_in_lds ? 1 : 0;
    a.err = W.cy_err.p;
    a.dbg = cy_dbg.p;
    W.cy_epoch += 2u * (unsigned)(krylovdim - kfirst) + 8u;
    if (W.cy_epoch > 0xF0000000u) {                        // tags would wrap: start over on zeroed buffers
        W.xf1.zero(stream); W.xf2.zero(stream); W.cy_epoch = 1; a.epoch0 = 1;
        W.cy_epoch += 2u * (unsigned)(krylovdim - kfirst) + 8u;
    }
    const size_t lds = dev::cycle_lds_bytes(krylovdim, W.F_r, R, G, f_in_lds);
    const bool prof = opt.profile_symv_every > 0;
    if (prof) {
        if (W.cye[0] == nullptr) { PX_HIP(hipEventCreate(&W.cye[0])); PX_HIP(hipEventCreate(&W.cye[1])); }
        if (R == 128) hipExtLaunchKernelGGL(dev::k_lz_cycle<128>, dim3(8 * G), dim3(dev::TPB), lds, stream, W.cye[0], W.cye[1], 0, a);
        else hipExtLaunchKernelGGL(dev::k_lz_cycle<64>, dim3(8 * G), dim3(dev::TPB), lds, stream, W.cye[0], W.cye[1], 0, a);
        W.cye_pending = true;
    } else {
        if (R == 128) hipLaunchKernelGGL(dev::k_lz_cycle<128>, dim3(8 * G), dim3(dev::TPB), lds, stream, a);
        else hipLaunchKernelGGL(dev::k_lz_cycle<64>, dim3(8 * G), dim3(dev::TPB), lds, stream, a);
    }
    W.lst.cycle_launches++;
    W.lst.cycle_steps += krylovdim - kfirst;
    W.lst.symv_launches += krylovdim - kfirst;
    W.lst.symv_bytes += (double)(krylovdim - kfirst) * (8.0 * (double)W.N + 16.0 * (double)W.n);
}

// rocSOLVER's first dsyevd in a process pays ~3 s of code-object loading.  The exit path
// (cone_feas, pdhg.jl:685) always needs one.  Optional (PROXSDP_HIP_WARMUP=1): a dummy dsyevd
// on a private handle/stream in a background thread -- OFF by default because the loader
// serialises against the solver thread's launches.
inline void Solver::start_rocsolver_warmup() {
    const int dev_id = opt.device_id;
    warm = std::thread([dev_id]() {
        if (hipSetDevice(dev_id) != hipSuccess) return;
        hipStream_t s = nullptr;
        rocblas_handle h = nullptr;
        if (hipStreamCreate(&s) != hipSuccess) return;
        if (rocblas_create_handle(&h) == rocblas_status_success) {
            (void)rocblas_set_stream(h, s);
            double* buf = nullptr;
            rocblas_int* info = nullptr;
            const int n = 96;                           // large enough to take the blocked path
            if (hipMalloc((void**)&buf, sizeof(double) * (n * n + 2 * n)) == hipSuccess &&
                hipMalloc((void**)&info, sizeof(rocblas_int)) == hipSuccess) {
                (void)hipMemsetAsync(buf, 0, sizeof(double) * (n * n + 2 * n), s);
                (void)rocsolver_dsyevd(h, rocblas_evect_original, rocblas_fill_upper, n, buf, n,
                                       buf + n * n, buf + n * n + n, info);
                (void)rocsolver_dsyevd(h, rocblas_evect_none, rocblas_fill_upper, n, buf, n,
                                       buf + n * n, buf + n * n + n, info);
                (void)hipStreamSynchronize(s);
            }
            if (buf) (void)hipFree(buf);
            if (info) (void)hipFree(info);
            (void)rocblas_destroy_handle(h);
        }
        (void)hipStreamDestroy(s);
    });
}

// ------------------------------------------------------------------ kernels launch helpers
// launch helper: profiled launches carry their own start/stop events (the dispatch's
// timestamps, i.e. the kernel alone, as rocprofv3 reports it)
template <typename K, typename... Args>
static inline void launch_prof(bool prof, hipEvent_t e0, hipEvent_t e1, K kern, dim3 grid, hipStream_t stream, Args... args) {
    if (prof) hipExtLaunchKernelGGL(kern, grid, dim3(dev::TPB), 0, stream, e0, e1, 0, args...);
    else hipLaunchKernelGGL(kern, grid, dim3(dev::TPB), 0, stream, args...);
}

inline void Solver::launch_symv(EigWork& W, const double* xp, const double* v, bool use_ctl) {
    const int ntile = 8 * ceil_div(W.nt * (W.nt + 1) / 2, 8);     // one workgroup per 64x64 tile, padded to 8 XCDs
    bool prof = opt.profile_symv_every > 0 && (W.lst.symv_launches % opt.profile_symv_every) == 0;
    size_t slot = 0;
    if (prof) {
        if (W.ev.used == W.ev.e0.size()) {
            hipEvent_t a, b;
            PX_HIP(hipEventCreate(&a)); PX_HIP(hipEventCreate(&b));
            W.ev.e0.push_back(a); W.ev.e1.push_back(b);
        }
        slot = W.ev.used++;
    }
    // profiled launches carry their own start/stop events (the dispatch's timestamps, i.e. the
    // kernel alone, as rocprofv3 reports it -- not the gaps to the neighbouring launches)
    hipEvent_t e0 = prof ? W.ev.e0[slot] : nullptr, e1 = prof ? W.ev.e1[slot] : nullptr;
    if (W.use_fop) {
        auto kern = (W.F_r <= 64) ? dev::k_fop<1> : dev::k_fop<2>;
        launch_prof(prof, e0, e1, kern, dim3(W.nt), stream,
                    v, (const double*)(W.F.p + (size_t)W.F_first * W.npad), W.npad, W.F_r,
                    (const int*)W.ell_col.p, (const int*)W.ell_sidx.p, W.ell_w, W.npad, W.esv, W.tpart.p, W.pld,
                    W.ebuf.p, W.apartf.p, (const dev::LanczosCtl*)(use_ctl ? W.ctl_p : nullptr), W.ov);
    } else {
        launch_prof(prof, e0, e1, dev::k_symv_packed, dim3(ntile), stream,
                    xp, W.n, W.nt, W.npad, v, W.Ppart.p, (const dev::LanczosCtl*)(use_ctl ? W.ctl_p : nullptr), W.Apart.p);
    }
    W.lst.symv_launches++;
    W.lst.symv_bytes += 8.0 * (double)W.N + 16.0 * (double)W.n;
}

// partial-dot buffer written by the orthogonalisation of step k
inline double* lz_hpart(EigWork& W, int k) {
    return (k & 1) ? W.hpart2.p : W.hpart1.p;
}

// k_symv_finish: closes Lanczos step `kclose` and runs the mat-vec of step kclose+1 on w'
inline void Solver::launch_symv_finish(EigWork& W, const double* xp, int kclose, double tol, bool use_carry) {
    const int ntile = 8 * ceil_div(W.nt * (W.nt + 1) / 2, 8);
    bool prof = opt.profile_symv_every > 0 && (W.lst.symv_launches % opt.profile_symv_every) == 0;
    size_t slot = 0;
    if (prof) {
        if (W.ev.used == W.ev.e0.size()) {
            hipEvent_t a, b;
            PX_HIP(hipEventCreate(&a)); PX_HIP(hipEventCreate(&b));
            W.ev.e0.push_back(a); W.ev.e1.push_back(b);
        }
        slot = W.ev.used++;
    }
    hipEvent_t e0 = prof ? W.ev.e0[slot] : nullptr, e1 = prof ? W.ev.e1[slot] : nullptr;
    if (W.use_fop) {
        const int nchf = (kclose + 1 <= 64) ? 1 : (kclose + 1 <= 128) ? 2 : (kclose + 1 <= 192) ? 3 : 4;
        auto kern = (W.F_r <= 64) ? (nchf == 1 ? dev::k_fop_finish<1, 1> : nchf == 2 ? dev::k_fop_finish<1, 2> : nchf == 3 ? dev::k_fop_finish<1, 3> : dev::k_fop_finish<1, 4>)
                                  : (nchf == 1 ? dev::k_fop_finish<2, 1> : nchf == 2 ? dev::k_fop_finish<2, 2> : nchf == 3 ? dev::k_fop_finish<2, 3> : dev::k_fop_finish<2, 4>);
        launch_prof(prof, e0, e1, kern, dim3(2 * W.nt), stream,
                    (const double*)W.w.p, W.V.p, W.npad, kclose, (const double*)lz_hpart(W, kclose), W.pld,
                    (const double*)W.hsum1.p, W.alphas_p, W.betas_p, W.ctl_p, tol, use_carry ? 1 : 0, W.nt,
                    (const double*)(W.F.p + (size_t)W.F_first * W.npad), W.F_r,
                    (const int*)W.ell_col.p, (const int*)W.ell_sidx.p, W.ell_w, W.npad, W.esv, W.tpart.p, W.ebuf.p,
                    W.apartf.p, W.hred.p, W.ov);
    } else {
        const int nchf = (kclose + 1 <= 64) ? 1 : (kclose + 1 <= 128) ? 2 : (kclose + 1 <= 192) ? 3 : 4;
        auto ksf = nchf == 1 ? dev::k_symv_finish<1> : nchf == 2 ? dev::k_symv_finish<2> : nchf == 3 ? dev::k_symv_finish<3> : dev::k_symv_finish<4>;
        launch_prof(prof, e0, e1, ksf, dim3(W.nt + ntile), stream,
                    xp, W.n, W.nt, W.npad, W.Ppart.p, (const double*)W.w.p, W.V.p, W.npad, kclose,
                    (const double*)lz_hpart(W, kclose), W.pld, (const double*)W.hsum1.p, W.alphas_p, W.betas_p, W.ctl_p,
                    tol, use_carry ? 1 : 0, W.Apart.p, W.hred.p);
    }
    W.lst.symv_launches++;
    W.lst.symv_bytes += 8.0 * (double)W.N + 16.0 * (double)W.n;
}

inline void Solver::launch_reconstruct(EigWork& W, const double* Z, int ldz, const double* lam, int r, double* xp_out,
                                       const double* xp_old, int blk) {
    const int ntile = 8 * ceil_div(W.nt * (W.nt + 1) / 2, 8);     // padded to the 8 XCDs (xcd_tile)
    // fp64 MFMA SYRK from rank 16 on in auto mode (measured cross-over at n = 4000: equal at 12-16,
    // 1.17x at 26, 1.32x at 63, 1.7x at n/2; docs/DESIGN_history_r1_r3.md section 5)
    const bool mfma = opt.reconstruct_mfma == 1 || (opt.reconstruct_mfma < 0 && r >= 16);
    if (mfma) {
        W.lst.mfma_reconstructions++;
        if (use_support && xp_old != nullptr && blk >= 0)
            hipLaunchKernelGGL(dev::k_reconstruct_mfma<true>, dim3(ntile), dim3(dev::TPB), 0, stream,
                               Z, ldz, lam, r, W.n, xp_out, xp_old, mask_d.p, (long long)P.blocks[blk].off,
                               respart_d.p + tile_base[blk], rstride);
        else
            hipLaunchKernelGGL(dev::k_reconstruct_mfma<false>, dim3(ntile), dim3(dev::TPB), 0, stream,
                               Z, ldz, lam, r, W.n, xp_out, (const double*)nullptr, (const unsigned*)nullptr, 0LL,
                               (double*)nullptr, 0);
        return;
    }
    if (use_support && xp_old != nullptr && blk >= 0)
        hipLaunchKernelGGL(dev::k_reconstruct_packed<true>, dim3(ntile), dim3(dev::TPB), 0, stream,
                           Z, ldz, lam, r, W.n, xp_out, xp_old, mask_d.p, (long long)P.blocks[blk].off,
                           respart_d.p + tile_base[blk], rstride);
    else
        hipLaunchKernelGGL(dev::k_reconstruct_packed<false>, dim3(ntile), dim3(dev::TPB), 0, stream,
                           Z, ldz, lam, r, W.n, xp_out, (const double*)nullptr, (const unsigned*)nullptr, 0LL,
                           (double*)nullptr, 0);
}

inline void Solver::rotate(EigWork& W, int K, const std::vector<double>& U, int ldu, int ncols,
                           double* out, int copy_src, int copy_dst, const double* extra, int nextra) {
    // U: host column-major (ldu x >=ncols); upload the K x ncols part compactly from a
    // staging buffer that alternates between two slots: two rotations can be in flight
    // between host synchronisations (restart rotation, then the final Ritz-vector one)
    if (rot_sink != nullptr) {
        // batched run: stage the compact K x ncols part (+ the arrow) for ONE upload and ONE launch per cycle
        RotSink::Req& q = rot_sink->slot[W.batch_slot];
        q.W = &W; q.V = W.V.p; q.out = out; q.K = K; q.ncols = ncols; q.copy_src = copy_src; q.copy_dst = copy_dst;
        q.nextra = std::max(nextra, 0); q.valid = true;
        q.data.resize((size_t)K * std::max(ncols, 0) + (size_t)q.nextra);
        for (int c = 0; c < ncols; ++c)
            for (int j = 0; j < K; ++j) q.data[(size_t)c * K + j] = U[(size_t)c * ldu + j];
        for (int t = 0; t < q.nextra; ++t) q.data[(size_t)K * std::max(ncols, 0) + t] = extra[t];
        return;
    }
    double* tmp = W.Ustage[W.ustage_next].p;
    W.ustage_next ^= 1;
    // `extra` (the arrow part f | D of the restarted Rayleigh quotient) rides behind U in the same
    // host-to-device copy; the kernels find it at W.arrow_p
    if ((size_t)K * std::max(ncols, 1) + (size_t)std::max(nextra, 0) > EigWork::USTAGE_DOUBLES)
        throw std::logic_error("rotation staging buffer too small");
    for (int c = 0; c < ncols; ++c)
        for (int j = 0; j < K; ++j) tmp[(size_t)c * K + j] = U[(size_t)c * ldu + j];
    for (int q = 0; q < nextra; ++q) tmp[(size_t)K * ncols + q] = extra[q];
    if (W.b1_defer && nextra > 0 && K < 32 && copy_src == K && copy_dst == ncols && out == W.Z.p) {
        // the next cycle's launch (k_lz_block1) rotates in its prologue, reading U and the arrow part from this pinned buffer
        W.b1_rot_K = K; W.b1_rot_U = tmp;
        W.arrow_p = tmp + (size_t)K * ncols;
        return;
    }
    W.U.upload(tmp, (size_t)K * ncols + (size_t)std::max(nextra, 0), stream);
    if (nextra > 0) W.arrow_p = W.U.p + (size_t)K * ncols;
    // fp64 MFMA form from K = 32 / 16 columns on (skinny GEMM); grid = row tiles x groups of 16 columns; LDS: V tile + one U group
    {
        const int Kp = (K + 3) & ~3;
        const size_t lds_m = ((size_t)Kp * dev::RM_LDV + (size_t)dev::RM_CG * (Kp + 2)) * sizeof(double);
        if (K >= 32 && ncols >= 16 && (int)lds_m <= rotate_mfma_lds_cap) {
            hipLaunchKernelGGL(dev::k_lz_rotate_mfma, dim3(W.nt, ceil_div(ncols, dev::RM_CG)), dim3(dev::TPB), lds_m, stream,
                               (const double*)W.V.p, W.npad, K, (const double*)W.U.p, ncols, out, W.npad, copy_src, copy_dst);
            return;
        }
    }
    // dynamic LDS: U chunk (K x cn) + V tile (K x 65).  gfx950 has 160 KiB of LDS per CU: the
    // kernel is allowed 144 KiB (setup_device), so a whole restart rotation is ONE launch up to
    // K = 127 (with a 60 KiB cap the V tile alone no longer fitted at K >= 116 and the loop
    // degenerated into one launch per column: 141 launches per iteration at target rank 63)
    const int vbytes = K * (dev::LZ_ROWS + 1) * 8;
    const int maxcols = std::max(1, (rotate_lds_cap - vbytes) / (8 * K));
    int c0 = 0;
    do {
        const int cn = std::max(0, std::min(maxcols, ncols - c0));
        const bool last = (c0 + cn >= ncols);
        if (cn > 0 || (last && copy_src >= 0))
            hipLaunchKernelGGL(dev::k_lz_rotate, dim3(W.nt), dim3(dev::TPB), (size_t)K * cn * 8 + vbytes, stream,
                               W.V.p, W.npad, W.n, K, W.U.p + (size_t)c0 * K, K, cn,
                               out + (size_t)c0 * W.npad, W.npad, last ? copy_src : -1, copy_dst - c0);
        c0 += std::max(cn, 1);
    } while (c0 < ncols);
}

// ------------------------------------------------------------------ Lanczos (KrylovKit eigsolve)
// eigsolver.jl:798-823 -> KrylovKit.eigsolve(A, resid, nev, :LR, Lanczos(orth, krylovdim, maxiter, tol))
// with A = Symmetric(smat(xp)).  eigsolver == 1 selects ARPACK's acceptance rule
// (dsaupd: |resid_i| <= tol*max(eps^(2/3), |theta_i|) for all nev wanted values,
// eigsolver.jl:668-746) on the same thick-restart engine.
// positive_part (full_eig! served by this engine, pdhg_loop.hip.hpp full_eig_by_lanczos): the wanted
// set is "every eigenpair with lambda > 0", at most nev of them.  A cycle ends the run when all
// positive Ritz pairs are converged to tol AND the first non-positive Ritz pair j is itself resolved
// to 1e-7 of the spectral scale (the solver's own tol_psd is 1e-7; measured on the n = 4000 default
// solve: 1e-9 / 1e-7 / 1e-5 give the same 8643 iterations and the same objective to 1e-13, in
// 21.6 / 18.4 / 12.9 s): a Krylov space that has resolved pair j has (generically) resolved
// everything above it, which is the same reliance every Lanczos acceptance rule makes.  (A residual
// merely smaller than |theta_j| is NOT enough: an unconverged Ritz value is a mixture and can sit
// below zero while small positive eigenvalues are still unresolved -- measured on gpp500-1.)
// Returns the j positive pairs (count = j, possibly 0, converged = true), or converged = false when
// more than nev Ritz values are positive / maxiter is hit.
// parameters of the run and the per-call reset of W; false = the call ends at once (dsaupd argument errors)
inline bool Solver::lz_init(EigWork& W, LzRun& R, int nev, bool positive_part) {
    R.nev = nev; R.positive_part = positive_part;
    R.arpack = (opt.eigsolver == 1);
    int krylovdim = std::max(2 * nev + 1, (int)opt.eigsolver_min_lanczos);
    // positive-part mode is the library's own algorithm (not KrylovKit's call): a larger Krylov space
    // resolves the bulk-edge pairs that decide it with fewer restarts (options.full_eig_lanczos_kdim10)
    if (positive_part) {
        const int mult10 = opt.full_eig_lanczos_kdim10 > 0 ? opt.full_eig_lanczos_kdim10 : 30;
        krylovdim = std::min(W.cap - 1, std::max(krylovdim, nev * mult10 / 10 + 8));
    }
    if (krylovdim + 1 > W.cap) throw std::invalid_argument("Lanczos workspace too small for the requested rank");
    R.krylovdim = krylovdim;
    R.tol = R.arpack ? opt.arpack_tol : opt.krylovkit_tol;
    R.maxiter = R.arpack ? (long long)opt.arpack_max_iter : (long long)opt.krylovkit_max_iter;
    W.prev_numiter = std::max(W.numiter, 1);
    W.converged = false; W.count = 0; W.converged_eigs = 0; W.numiter = 0; W.vals.clear();
    W.lst.lanczos_calls++;
    if (R.arpack && (!(0 < nev && nev < W.n) || krylovdim > W.n)) return false;   // dsaupd info=-1/-3 -> error -> fallback
    R.step_tol = R.arpack ? 0.0 : R.tol;          // invariant-subspace test inside the recurrence
    R.ld = krylovdim + 1;
    R.T.assign((size_t)R.ld * R.ld, 0.0); R.D.assign(R.ld, 0.0); R.f.assign(R.ld, 0.0);
    R.al.assign(R.ld, 0.0); R.be.assign(R.ld, 0.0);
    R.howmany = nev; R.numiter = 1; R.converged = 0; R.K = 0; R.kfirst = 0; R.pos_count = -1;
    R.pos_fail = false; R.presymv = false; R.betaK = 0.0;
    R.split_ready = false; R.merge_active = false;
    R.splitp = &W.split;
    return true;
}

// Ritz-vector coefficients U[:, 0..ncols) (descending order) of the last eigensolve when it was done by the merge
inline void Solver::lz_merge_vectors(EigWork& W, LzRun& R, int ncols) {
    const double t0 = now_s();
    const int K = R.K;
    std::vector<int> cols(std::max(ncols, 1));
    for (int c = 0; c < ncols; ++c) cols[c] = K - 1 - c;              // evals are ascending
    R.U.assign((size_t)K * std::max(ncols, 1), 0.0);
    if (ncols > 0) R.split_ref().M.vectors(cols.data(), ncols, R.U.data());
    W.lst.host_eig_time += now_s() - t0;
}

// while the GPU works through the enqueued steps: reduce the arrow part [diag(D) f; f' .] of this cycle's
// Rayleigh quotient (known since the restart) to tridiagonal form, so that only the QL sweep is left on
// the critical path once alpha/beta arrive (host_util.hpp symeig_tridiag_from; K = 53: 108 -> 54 us,
// K = 127: 1330 -> 544 us)
inline void Solver::lz_prepare_arrow(LzRun& R) {
    R.m_arrow = R.kfirst;
    if (R.m_arrow <= 0) return;
    const int n1 = R.m_arrow + 1, ld = R.ld;
    R.Qa.assign((size_t)n1 * n1, 0.0); R.da.assign(n1, 0.0); R.ea.assign(n1, 0.0);
    for (int j = 0; j < R.m_arrow; ++j) {
        R.Qa[(size_t)j * n1 + j] = R.T[(size_t)j * ld + j];
        R.Qa[(size_t)j * n1 + R.m_arrow] = R.Qa[(size_t)R.m_arrow * n1 + j] = R.T[(size_t)j * ld + R.m_arrow];
    }
    householder_tridiag(n1, R.Qa.data(), R.da.data(), R.ea.data());
}

// After a cycle's read-back (W.rec_host valid): Rayleigh quotient, K x K eigensolve, convergence.  Returns true
// when the run goes on: the restart rotation has been enqueued and R.kfirst / R.T describe the next cycle;
// false when the run is over (lz_finish_run produces the results).  `speculated`: the first mat-vec of the
// next cycle was already enqueued before the read-back.
inline bool Solver::lz_after_cycle(EigWork& W, LzRun& R, bool speculated) {
    const int krylovdim = R.krylovdim, ld = R.ld, kfirst = R.kfirst, nev = R.nev;
    const double tol = R.tol;
    std::vector<double>& T = R.T; std::vector<double>& D = R.D; std::vector<double>& U = R.U;
    std::vector<double>& f = R.f; std::vector<double>& al = R.al; std::vector<double>& be = R.be;
    dev::LanczosCtl hctl{};
    std::copy(W.rec_host, W.rec_host + krylovdim, al.begin());
    std::copy(W.rec_host + dev::MAXK, W.rec_host + dev::MAXK + krylovdim, be.begin());
    std::memcpy(&hctl, W.rec_host + 2 * dev::MAXK, sizeof(hctl));
    const int Kend = hctl.stop ? hctl.kstop : krylovdim;
    // launches after the stop flag are no-ops; count the mat-vecs that did work
    {
        long long skipped = (long long)(krylovdim - Kend);
        W.lst.lanczos_matvecs += (Kend - kfirst);
        W.lst.symv_launches -= skipped;
        W.lst.symv_bytes -= skipped * (8.0 * (double)W.N + 16.0 * (double)W.n);
        W.mv_iter += (Kend - kfirst);
    }
    for (int k = kfirst; k < Kend; ++k) {
        T[(size_t)k * ld + k] = al[k];
        if (k + 1 < Kend) { T[(size_t)k * ld + k + 1] = be[k]; T[(size_t)(k + 1) * ld + k] = be[k]; }
    }
    const int K = R.K = Kend;
    const double betaK = R.betaK = be[K - 1];
    if (!(betaK == betaK)) {                         // NaN guard: treat as not converged
        R.converged = 0; R.pos_fail = true; R.howmany = 0;
        return false;
    }
    if (betaK <= tol && K < R.howmany && !R.arpack) R.howmany = K;
    R.merge_active = false;
    if (K > 1 && R.split_ready && K == krylovdim) {
        // split + rank-one merge: T1' was decomposed while the GPU ran the cycle; here the tail and the merge
        const double te0 = now_s();
        if (R.split_ref().second(K, al.data(), be.data()) == 0) {
            R.erow.resize(K);
            R.split_ref().M.row_of_vectors(K - 1, R.erow.data());
            for (int c = 0; c < K; ++c) {                // :LR -> descending
                D[c] = R.split_ref().M.evals[K - 1 - c];
                f[c] = betaK * R.erow[K - 1 - c];
            }
            R.merge_active = true;
            W.lst.host_eigs++; W.lst.host_eig_merges++;
        }
        W.lst.host_eig_time += now_s() - te0;
    }
    R.split_ready = false;
    if (R.merge_active) {
        // (Ritz-vector coefficients are formed below / in lz_finish_run, for the columns actually needed)
    } else if (K == 1) {
        D[0] = T[0]; U.assign(1, 1.0); f[0] = betaK;
    } else {
        R.Tw.assign((size_t)K * K, 0.0);
        for (int c = 0; c < K; ++c)
            for (int r = 0; r < K; ++r) R.Tw[(size_t)c * K + r] = T[(size_t)c * ld + r];
        std::vector<double> Dasc(K);
        const double te0 = now_s();
        if (R.m_arrow > 0 && K > R.m_arrow)
            symeig_tridiag_from(K, R.m_arrow, R.Qa.data(), R.da.data(), R.ea.data(), al.data(), be.data(), R.Tw.data(), Dasc.data());
        else
            symeig_dense(K, R.Tw.data(), Dasc.data(), kfirst == 0);       // (also the fallback of a failed merge: Qa was not prepared)
        W.lst.host_eig_time += now_s() - te0; W.lst.host_eigs++;
        U.assign((size_t)K * K, 0.0);
        for (int c = 0; c < K; ++c) {                // :LR -> descending
            D[c] = Dasc[K - 1 - c];
            for (int r = 0; r < K; ++r) U[(size_t)c * K + r] = R.Tw[(size_t)(K - 1 - c) * K + r];
            f[c] = betaK * U[(size_t)c * K + (K - 1)];
        }
    }
    int converged = 0;
    if (!R.arpack) {
        // (positive-part mode, the library's own algorithm: optionally a residual RELATIVE to the spectral scale,
        // options.full_eig_lanczos_tol; KrylovKit's rule -- absolute krylovkit_tol -- everywhere else)
        double tol_c = tol;
        if (R.positive_part && opt.full_eig_lanczos_tol > 0.0)
            tol_c = std::max(tol, opt.full_eig_lanczos_tol * std::max(std::fabs(D[0]), std::fabs(D[K - 1])));
        while (converged < K && std::fabs(f[converged]) <= tol_c) ++converged;
    } else {
        const double eps23 = std::pow(2.220446049250313e-16 / 2.0, 2.0 / 3.0);
        int want = std::min(nev, K);
        for (int i = 0; i < want; ++i)
            if (std::fabs(f[i]) <= tol * std::max(eps23, std::fabs(D[i]))) ++converged;
    }
    R.converged = converged;
    if (R.positive_part) {
        int j = 0;
        while (j < K && D[j] > 0.0) ++j;
        if (j > nev) { R.pos_fail = true; return false; }      // more positive pairs than the workspace was sized for
        const double scale = std::max(std::fabs(D[0]), std::fabs(D[K - 1]));
        // the deciding pair: the first STRICTLY negative Ritz value (an exactly-zero pair with zero
        // residual is what a decoupled, unused coordinate of the block looks like: it is found at
        // once and says nothing about the pairs around it)
        int jn = j;
        while (jn < K && D[jn] >= -1e-12 * scale) ++jn;
        const double posres = lanczos_posres();
        if (converged >= j && (jn == K || std::fabs(f[jn]) <= std::max(tol, posres * scale))) { R.pos_count = j; return false; }
        if (K < krylovdim || R.numiter == R.maxiter) { R.pos_fail = true; return false; }
    } else {
        if (converged >= R.howmany) return false;
        if (K < krylovdim) return false;                 // invariant subspace without convergence (arpack rule)
    }
    if (R.numiter == R.maxiter) return false;
    // (positive-part mode, measured in round 4 on the default-options n = 4000 solve, tools/gpurun_r04_pp.sh: Krylov
    // dimension 4g / 5g instead of 3g + 8 and a kept part of positives + 25 % / 40 % of the rest all land within
    // -3 % .. +11 % of this rule's 12.4 s with the same 8651 iterations -- the rule stays)
    const int keep = R.arpack ? std::min(krylovdim - 1, nev + std::max(1, (krylovdim - nev) / 2))
                              : (3 * krylovdim + 2 * converged) / 5;
    // arrow part of the restarted T for k_lz_orth: f (couplings of v_K with the kept Ritz
    // vectors) and D (their Ritz values)
    for (int j = 0; j < keep; ++j) { W.arrow_host.p[j] = f[j]; W.arrow_host.p[dev::MAXK + j] = D[j]; }
    if (R.merge_active) lz_merge_vectors(W, R, keep);
    rotate(W, K, U, K, keep, W.Z.p, K, keep, W.arrow_host.p, 2 * dev::MAXK);   // Z[:, :keep] = V U[:, :keep]; Z[:, keep] = V[:, K]
    std::swap(W.V.p, W.Z.p);
    std::fill(T.begin(), T.end(), 0.0);
    for (int j = 0; j < keep; ++j) {
        T[(size_t)j * ld + j] = D[j];
        T[(size_t)j * ld + keep] = f[j];
        T[(size_t)keep * ld + j] = f[j];
    }
    R.kfirst = keep;
    R.presymv = speculated;
    ++R.numiter;
    W.lst.lanczos_restarts++;
    return true;
}

// results of the run (KrylovKit's return values / dseupd's), Ritz vectors into W.Z
inline void Solver::lz_finish_run(EigWork& W, LzRun& R) {
    const int K = R.K, nev = R.nev;
    W.numiter = R.numiter;
    if (R.positive_part) {
        if (R.pos_fail || R.pos_count < 0) { W.converged = false; return; }
        W.count = R.pos_count; W.converged_eigs = R.pos_count; W.converged = true;
        W.vals.assign(R.D.begin(), R.D.begin() + R.pos_count);
        if (R.pos_count > 0 && R.merge_active) lz_merge_vectors(W, R, R.pos_count);
        if (R.pos_count > 0) rotate(W, K, R.U, K, R.pos_count, W.Z.p, -1, 0, nullptr, 0);
        return;
    }
    if (R.pos_fail) { W.converged = false; return; }     // NaN guard
    if (R.arpack) {
        // _saupd!/_seupd! (eigsolver.jl:668-746): converged only when all nev pairs are
        W.converged_eigs = R.converged;
        if (R.converged < nev) { W.converged = false; return; }
        W.count = nev;
        W.vals.assign(R.D.begin(), R.D.begin() + nev);
        std::reverse(W.vals.begin(), W.vals.end());      // arc.d is ascending
        std::vector<double> Ur((size_t)K * nev);
        for (int c = 0; c < nev; ++c)
            for (int r = 0; r < K; ++r) Ur[(size_t)c * K + r] = R.U[(size_t)(nev - 1 - c) * K + r];
        rotate(W, K, Ur, K, nev, W.Z.p, -1, 0, nullptr, 0);
        W.converged = true;
        return;
    }
    int howmany = R.howmany;
    if (R.converged > howmany) howmany = R.converged;
    W.count = howmany;
    W.vals.assign(R.D.begin(), R.D.begin() + howmany);
    W.converged_eigs = R.converged;
    W.converged = (R.converged != 0);                    // eigsolver.jl:816-818
    if (R.merge_active) lz_merge_vectors(W, R, howmany);
    rotate(W, K, R.U, K, howmany, W.Z.p, -1, 0, nullptr, 0);          // Ritz vectors B*v
}

// KrylovKit's `eager = true` (options.jl:112, eigsolver.jl:809): the Rayleigh quotient is decomposed and the convergence
// test made after EVERY expansion step once the basis holds `howmany` vectors, not only when it is full -- a projection
// ends as soon as its wanted pairs have converged.  Off by default in the reference; here every such step is a
// mini-cycle of the same kernels: mat-vec on the exact v_k, recurrence + full re-orthogonalisation (`first` form of
// k_lz_orth with the coupling beta_{k-1} e_{k-1} as its "arrow"), step closing, read-back, K x K eigensolve on the
// host.  Three launches and one synchronisation per step: a faithful, not a fast, mode.  Packed-triangle operator.
inline void Solver::lanczos_eager(EigWork& W, const double* xp, int nev) {
    LzRun& R = W.lzrun;
    W.use_fop = false;
    if (!lz_init(W, R, nev, false)) return;
    if (R.arpack) throw std::invalid_argument("krylovkit_eager applies to eigsolver = 2 (KrylovKit) only");
    const int kd = R.krylovdim, ld = R.ld;
    const double tol = R.tol;
    hipLaunchKernelGGL(dev::k_lz_begin, dim3(ceil_div(W.npad, dev::TPB)), dim3(dev::TPB), 0, stream,
                       W.V.p, (const double*)W.resid.p, W.npad, W.ctl_p);
    std::vector<double>& T = R.T; std::vector<double>& D = R.D; std::vector<double>& U = R.U; std::vector<double>& f = R.f;
    std::vector<double> fstep(2 * dev::MAXK, 0.0), Dasc;
    int K = 0, kfirst = 0, converged = 0;
    double beta = 0.0;
    auto launch_orth = [&](int k, int keep) {
        const int nch = (k + 1 <= 64) ? 1 : (k + 1 <= 128) ? 2 : (k + 1 <= 192) ? 3 : 4;
        double* hp[2] = {W.hpart1.p, W.hpart2.p};
        dev::FopArgs fo{};
        auto go = [&](auto kern) {
            hipLaunchKernelGGL(kern, dim3(W.nt), dim3(dev::TPB), 0, stream,
                               (const double*)W.Ppart.p, W.nt, W.npad, (const double*)W.V.p, W.npad, k, W.w.p,
                               (const double*)W.hred.p, hp[k & 1], W.pld, W.hsum1.p, (const dev::LanczosCtl*)W.ctl_p,
                               (const double*)W.alphas_p, (const double*)W.betas_p, (const double*)W.Apart.p, W.napart, 1,
                               (const double*)W.arrow_p, keep, fo);
        };
        if (nch == 1) go(dev::k_lz_orth<1, 0>); else if (nch == 2) go(dev::k_lz_orth<2, 0>);
        else if (nch == 3) go(dev::k_lz_orth<3, 0>); else go(dev::k_lz_orth<4, 0>);
    };
    while (true) {
        const int k = K;                                  // expand: v_k is an exact, normalised basis vector
        launch_symv(W, xp, W.V.p + (size_t)k * W.npad, true);
        int keep = kfirst;
        if (k > kfirst) {                                 // ordinary step: w -= beta_{k-1} v_{k-1}
            std::fill(fstep.begin(), fstep.begin() + k, 0.0);
            fstep[k - 1] = beta;
            W.arrow.upload(fstep.data(), (size_t)k, stream);
            W.arrow_p = W.arrow.p;
            keep = k;
        }
        launch_orth(k, keep);
        auto klf = (k + 1 <= 64) ? dev::k_lz_finish<1> : (k + 1 <= 128) ? dev::k_lz_finish<2> : (k + 1 <= 192) ? dev::k_lz_finish<3> : dev::k_lz_finish<4>;
        hipLaunchKernelGGL(klf, dim3(W.nt), dim3(dev::TPB), 0, stream, W.w.p, W.n, W.V.p, W.npad, k, lz_hpart(W, k), W.pld,
                           W.hsum1.p, W.alphas_p, W.betas_p, W.ctl_p, 0.0, 0, W.hred.p);
        PX_HIP(hipMemcpyAsync(W.rec_host, W.rec.p, EigWork::REC_DOUBLES * sizeof(double), hipMemcpyDeviceToHost, stream));
        PX_HIP(hipStreamSynchronize(stream));
        const double al_k = W.rec_host[k];
        beta = W.rec_host[dev::MAXK + k];
        W.lst.lanczos_matvecs++; W.mv_iter++;
        if (!(beta == beta) || !(al_k == al_k)) { R.pos_fail = true; R.K = K; break; }      // NaN guard
        T[(size_t)k * ld + k] = al_k;
        if (k > kfirst) { T[(size_t)(k - 1) * ld + k] = T[(size_t)k * ld + (k - 1)] = W.rec_host[dev::MAXK + k - 1]; }
        K = k + 1;
        if (beta <= tol && K < R.howmany) R.howmany = K;
        if (K == kd || beta <= tol || K >= R.howmany) {
            if (K == 1) {
                D[0] = T[0]; U.assign(1, 1.0); f[0] = beta;
            } else {
                R.Tw.assign((size_t)K * K, 0.0);
                for (int c = 0; c < K; ++c)
                    for (int r = 0; r < K; ++r) R.Tw[(size_t)c * K + r] = T[(size_t)c * ld + r];
                Dasc.resize(K);
                const double te0 = now_s();
                symeig_dense(K, R.Tw.data(), Dasc.data(), false);
                W.lst.host_eig_time += now_s() - te0; W.lst.host_eigs++;
                U.assign((size_t)K * K, 0.0);
                for (int c = 0; c < K; ++c) {
                    D[c] = Dasc[K - 1 - c];
                    for (int r = 0; r < K; ++r) U[(size_t)c * K + r] = R.Tw[(size_t)(K - 1 - c) * K + r];
                    f[c] = beta * U[(size_t)c * K + (K - 1)];
                }
            }
            converged = 0;
            while (converged < K && std::fabs(f[converged]) <= tol) ++converged;
            R.K = K; R.converged = converged;
            if (converged >= R.howmany) break;
        }
        if (K < kd) {
            if (beta <= tol) { R.K = K; break; }          // invariant subspace (howmany was reduced above: not reached normally)
            continue;
        }
        if (R.numiter == R.maxiter) break;
        // thick restart, as lz_after_cycle
        const int keepn = (3 * kd + 2 * converged) / 5;
        for (int j = 0; j < keepn; ++j) { W.arrow_host.p[j] = f[j]; W.arrow_host.p[dev::MAXK + j] = D[j]; }
        rotate(W, K, U, K, keepn, W.Z.p, K, keepn, W.arrow_host.p, 2 * dev::MAXK);
        std::swap(W.V.p, W.Z.p);
        std::fill(T.begin(), T.end(), 0.0);
        for (int j = 0; j < keepn; ++j) {
            T[(size_t)j * ld + j] = D[j];
            T[(size_t)j * ld + keepn] = f[j];
            T[(size_t)keepn * ld + j] = f[j];
        }
        kfirst = keepn; K = keepn;
        ++R.numiter;
        W.lst.lanczos_restarts++;
    }
    lz_finish_run(W, R);
}

// first phase of the split eigensolve, while the GPU runs the rest of the cycle: wait for the early copy of the
// recurrence coefficients (side stream), decompose T1' = [arrow / first half] - |b| e e'
inline bool Solver::lz_split_first(EigWork& W, LzRun& R, int k1) {
    R.split_ready = false;
    R.m_arrow = 0;                                   // (a failed merge falls back to the dense QL path)
    if (hipEventSynchronize(W.ev_early) != hipSuccess) return false;
    const double t0 = now_s();
    const double* al_e = W.rec_early.p;
    const double* be_e = W.rec_early.p + dev::MAXK;
    dev::LanczosCtl c{};
    std::memcpy(&c, W.rec_early.p + 2 * dev::MAXK, sizeof(c));
    const int m = R.kfirst;
    bool ok = !c.stop;                               // an invariant subspace ends the cycle early: general path
    for (int j = m; ok && j < k1; ++j) ok = (al_e[j] == al_e[j]) && (be_e[j] == be_e[j]) && be_e[j] > 0.0;
    if (ok) {
        std::vector<double> Dk(std::max(m, 1)), fk(std::max(m, 1));
        for (int j = 0; j < m; ++j) { Dk[j] = R.T[(size_t)j * R.ld + j]; fk[j] = R.T[(size_t)j * R.ld + m]; }
        ok = R.split_ref().first(k1, m, Dk.data(), fk.data(), al_e, be_e) == 0;
    }
    R.split_ready = ok;
    W.lst.host_eig_overlap_time += now_s() - t0;
    return ok;
}

// the launches of Lanczos step k of a cycle that started at basis column kfirst: the mat-vec (on the exact v_k at the
// start of a cycle, otherwise fused with the closing of step k-1 and run on the uncorrected w') and the
// recurrence / re-orthogonalisation kernel
inline void Solver::lz_launch_step(EigWork& W, const double* xp, int k, int kfirst, double step_tol, bool& presymv) {
            if (k == kfirst) {
                if (!presymv) launch_symv(W, xp, W.V.p + (size_t)k * W.npad, true);   // v_k is ready (start)
                else { W.lst.symv_launches++; W.lst.symv_bytes += 8.0 * (double)W.N + 16.0 * (double)W.n; }
                presymv = false;               // after a restart the mat-vec of v_keep is already in Ppart
            } else {
                // close step k-1 and run the mat-vec of step k in one launch
                launch_symv_finish(W, xp, k - 1, step_tol, k - 1 > kfirst);
            }
            // recurrence, predicted correction and the measured full pass in one launch;
            // the partial-dot buffers alternate by step parity (read k-1, write k)
            double* hp[2] = {W.hpart1.p, W.hpart2.p};
            dev::FopArgs fo{};
            if (W.use_fop) {
                fo.Vp = W.F.p + (size_t)W.F_first * W.npad; fo.lam = W.Flam.p; fo.rp = W.F_r;
                fo.tpart = W.tpart.p; fo.ebuf = W.ebuf.p; fo.apart = W.apartf.p;
            }
            const int nch = (k + 1 <= 64) ? 1 : (k + 1 <= 128) ? 2 : (k + 1 <= 192) ? 3 : 4;
            const int nchp = !W.use_fop ? 0 : (W.F_r <= 64 ? 1 : 2);
            const bool prof_o = opt.profile_symv_every > 0 && (W.lst.symv_launches % opt.profile_symv_every) == 1;
            size_t oslot = 0;
            if (prof_o) {
                if (W.evo.used == W.evo.e0.size()) {
                    hipEvent_t a, b;
                    PX_HIP(hipEventCreate(&a)); PX_HIP(hipEventCreate(&b));
                    W.evo.e0.push_back(a); W.evo.e1.push_back(b);
                }
                oslot = W.evo.used++;
            }
            auto launch_orth = [&](auto kern) {
                launch_prof(prof_o, prof_o ? W.evo.e0[oslot] : nullptr, prof_o ? W.evo.e1[oslot] : nullptr, kern,
                                   dim3(W.nt), stream,
                                   (const double*)W.Ppart.p, W.nt, W.npad, (const double*)W.V.p, W.npad, k, W.w.p,
                                   (const double*)W.hred.p, hp[k & 1], W.pld, W.hsum1.p,
                                   (const dev::LanczosCtl*)W.ctl_p, (const double*)W.alphas_p, (const double*)W.betas_p,
                                   (const double*)W.Apart.p, W.napart, k == kfirst ? 1 : 0, (const double*)W.arrow_p,
                                   kfirst, fo);
            };
            switch (nch * 3 + nchp) {
                case 3: launch_orth(dev::k_lz_orth<1, 0>); break;
                case 4: launch_orth(dev::k_lz_orth<1, 1>); break;
                case 5: launch_orth(dev::k_lz_orth<1, 2>); break;
                case 6: launch_orth(dev::k_lz_orth<2, 0>); break;
                case 7: launch_orth(dev::k_lz_orth<2, 1>); break;
                case 8: launch_orth(dev::k_lz_orth<2, 2>); break;
                case 9: launch_orth(dev::k_lz_orth<3, 0>); break;
                case 10: launch_orth(dev::k_lz_orth<3, 1>); break;
                case 11: launch_orth(dev::k_lz_orth<3, 2>); break;
                case 12: launch_orth(dev::k_lz_orth<4, 0>); break;
                case 13: launch_orth(dev::k_lz_orth<4, 1>); break;
                default: launch_orth(dev::k_lz_orth<4, 2>); break;
            }
}

// Certificate of a Lanczos-served full_eig! (options.full_eig_lanczos_certify).  Single-vector Lanczos
// returns one eigenvector per DISTINCT eigenvalue the start vector has a component along: a repeated positive eigenvalue or
// a deficient start vector silently drops positive eigenpairs from X+.  After the run has converged with npos positive
// pairs (Ritz vectors in W.Z, values in W.vals) a SECOND, independent pseudo-random vector is orthogonalised against them
// and run through `msteps` steps of the same recurrence with the Ritz vectors as the locked part of the basis (a thick
// restart whose coupling row is zero): Lanczos on the operator deflated by the returned pairs.  Everything that is left
// must be <= 0; the largest Ritz value of the msteps x msteps tridiagonal is a LOWER bound of the largest remaining
// eigenvalue and converges to an isolated one within a few steps.  Returns false when the check could not run (workspace
// too small, non-finite coefficients); theta_max / scale are what the caller tests.
inline bool Solver::lanczos_certificate(EigWork& W, const double* xp, int npos, int msteps, double& theta_max, double& scale) {
    if (npos + msteps + 1 > W.cap || npos + msteps + 1 > dev::KLD || msteps < 2) return false;
    if (W.resid2.n == 0) {
        std::vector<double> r2(W.npad, 0.0);
        start_vector(W.n, (uint64_t)opt.eigsolver_resid_seed ^ 0x9e3779b97f4a7c15ull, 3, r2.data());   // (normalised)
        W.resid2.alloc(W.npad);
        W.resid2.upload(r2.data(), W.npad, stream);
        PX_HIP(hipStreamSynchronize(stream));                       // (r2 goes out of scope)
    }
    std::swap(W.V.p, W.Z.p);                                        // the basis buffer holds the Ritz vectors in columns [0, npos)
    auto nch_of = [](int cols) { return cols <= 64 ? 1 : cols <= 128 ? 2 : cols <= 192 ? 3 : 4; };
    if (npos == 0) {
        hipLaunchKernelGGL(dev::k_lz_begin, dim3(ceil_div(W.npad, dev::TPB)), dim3(dev::TPB), 0, stream,
                           W.V.p, (const double*)W.resid2.p, W.npad, W.ctl_p);
    } else {
        // v_npos = (r - Z Z'r) / |.|: measured dots, then the ordinary step closing
        PX_HIP(hipMemcpyAsync(W.w.p, W.resid2.p, (size_t)W.npad * sizeof(double), hipMemcpyDeviceToDevice, stream));
        const int kc = npos - 1, nch = nch_of(npos);
        auto km = nch == 1 ? dev::k_lz_measure<1> : nch == 2 ? dev::k_lz_measure<2> : nch == 3 ? dev::k_lz_measure<3> : dev::k_lz_measure<4>;
        hipLaunchKernelGGL(km, dim3(W.nt), dim3(dev::TPB), 0, stream, (const double*)W.w.p, (const double*)W.V.p, W.npad, kc,
                           lz_hpart(W, kc), W.pld, W.ctl_p);
        auto kf = nch == 1 ? dev::k_lz_finish<1> : nch == 2 ? dev::k_lz_finish<2> : nch == 3 ? dev::k_lz_finish<3> : dev::k_lz_finish<4>;
        hipLaunchKernelGGL(kf, dim3(W.nt), dim3(dev::TPB), 0, stream, (const double*)W.w.p, W.n, W.V.p, W.npad, kc,
                           (const double*)lz_hpart(W, kc), W.pld, (const double*)W.hsum1.p, W.alphas_p, W.betas_p, W.ctl_p, 0.0, 0, W.hred.p);
    }
    // locked part: coupling row f = 0, Ritz values D (the prediction of k_lz_orth multiplies the measured junk by them)
    for (int j = 0; j < npos; ++j) { W.arrow_host.p[j] = 0.0; W.arrow_host.p[dev::MAXK + j] = W.vals[j]; }
    if (npos > 0) {
        PX_HIP(hipMemcpyAsync(W.arrow.p, W.arrow_host.p, (size_t)npos * sizeof(double), hipMemcpyHostToDevice, stream));
        PX_HIP(hipMemcpyAsync(W.arrow.p + dev::MAXK, W.arrow_host.p + dev::MAXK, (size_t)npos * sizeof(double), hipMemcpyHostToDevice, stream));
    }
    W.arrow_p = W.arrow.p;
    const int kend = npos + msteps;
    bool presymv = false;
    for (int k = npos; k < kend; ++k) lz_launch_step(W, xp, k, npos, 0.0, presymv);
    {
        const int nch = nch_of(kend);
        auto kf = nch == 1 ? dev::k_lz_finish<1> : nch == 2 ? dev::k_lz_finish<2> : nch == 3 ? dev::k_lz_finish<3> : dev::k_lz_finish<4>;
        hipLaunchKernelGGL(kf, dim3(W.nt), dim3(dev::TPB), 0, stream, (const double*)W.w.p, W.n, W.V.p, W.npad, kend - 1,
                           (const double*)lz_hpart(W, kend - 1), W.pld, (const double*)W.hsum1.p, W.alphas_p, W.betas_p, W.ctl_p, 0.0,
                           (kend - 1 > npos) ? 1 : 0, W.hred.p);
    }
    PX_HIP(hipMemcpyAsync(W.rec_host, W.rec.p, EigWork::REC_DOUBLES * sizeof(double), hipMemcpyDeviceToHost, stream));
    wait_stream();
    std::swap(W.V.p, W.Z.p);
    W.evo.used = 0; W.ev.used = 0;                                  // (profile events recorded by lz_launch_step: not a projection's)
    W.lst.lanczos_matvecs += msteps; W.mv_iter += msteps; W.lst.cert_matvecs += msteps;
    // the deflated recurrence may end early (beta <= 0 exactly: what is left of the operator acts as a multiple of the
    // identity on the complement -- the later launches are no-ops and their record entries are stale values of the main
    // run): only the steps that ran form the tridiagonal
    dev::LanczosCtl hctl{};
    std::memcpy(&hctl, W.rec_host + 2 * dev::MAXK, sizeof(hctl));
    if (hctl.stop) msteps = std::max(0, std::min(msteps, hctl.kstop - npos));
    if (msteps < 1) return false;
    const double* al = W.rec_host + npos;
    const double* be = W.rec_host + dev::MAXK + npos;
    std::vector<double> T((size_t)msteps * msteps, 0.0), d(msteps, 0.0);
    for (int j = 0; j < msteps; ++j) {
        if (!(al[j] == al[j]) || !(j + 1 < msteps ? be[j] == be[j] : true)) return false;
        T[(size_t)j * msteps + j] = al[j];
        if (j + 1 < msteps) T[(size_t)j * msteps + j + 1] = T[(size_t)(j + 1) * msteps + j] = be[j];
    }
    if (msteps == 1) d[0] = al[0];
    else if (symeig_dense(msteps, T.data(), d.data(), true) != 0) return false;
    theta_max = d[msteps - 1];
    scale = std::max({std::fabs(d[0]), std::fabs(theta_max), npos > 0 ? std::fabs(W.vals[0]) : 0.0});
    return true;
}

inline void Solver::lanczos(EigWork& W, const double* xp, int nev, bool positive_part) {
    if (opt.krylovkit_eager && opt.eigsolver != 1 && !positive_part) { lanczos_eager(W, xp, nev); return; }
    LzRun& R = W.lzrun;
    if (!lz_init(W, R, nev, positive_part)) return;
    const int krylovdim = R.krylovdim;
    const double step_tol = R.step_tol;
    if (krylovdim >= 96) QlPool::get().arm();            // host eigensolve helpers wake up under the first Lanczos cycle
    // start from the previous projection's Ritz vectors: on the Krylov branch only on request (library-only knob; the
    // reference starts every projection from the same fixed vector, krylovkit_reset_resid = false); in positive-part mode
    // (full_eig! served by this engine: the library's own algorithm, the start vector is its own choice) by default
    // -- same X+ to krylovkit_tol, fewer restarts (lanczos_warm_start = -1 switches it off there too)
    const bool warm_start = positive_part ? (opt.lanczos_warm_start >= 0) : (opt.lanczos_warm_start > 0);
    if (warm_start && W.fop_ok && W.have_factors && W.F_r > 0 && W.tpart.n > 0) {
        const int nb = ceil_div(W.npad, dev::TPB);
        hipLaunchKernelGGL(dev::k_lz_warm_sum, dim3(nb), dim3(dev::TPB), 0, stream,
                           W.V.p, (const double*)(W.F.p + (size_t)W.F_first * W.npad), W.npad, W.F_r,
                           (const double*)W.resid.p, W.npad, W.warm_part.p,
                           (const double*)(positive_part && W.F_r <= dev::TPB ? W.Flam.p : nullptr),
                           positive_part ? opt.full_eig_lanczos_warm_pow : 0.0);
        hipLaunchKernelGGL(dev::k_lz_warm_scale, dim3(nb), dim3(dev::TPB), 0, stream,
                           W.V.p, W.npad, (const double*)W.warm_part.p, nb, W.ctl_p);
        W.lst.warm_starts++;
    } else
    hipLaunchKernelGGL(dev::k_lz_begin, dim3(ceil_div(W.npad, dev::TPB)), dim3(dev::TPB), 0, stream,
                       W.V.p, (const double*)W.resid.p, W.npad, W.ctl_p);

    int cyR = 0, cyG = 0;
    bool cyF = false;
    const bool cyc = cycle_plan(W, krylovdim, cyR, cyG, cyF);
    int cy_err_host = 0;
    if (cyc && W.cy_err_host.p == nullptr) { W.cy_err_host.alloc(1); W.cy_err_host.p[0] = 0.0; }
    // split + rank-one merge of the K x K eigensolve (options.host_eig_merge): not for the dsaupd rule (its
    // wanted set is re-ordered afterwards) and not under the persistent cycle kernel
    const int merge_from = opt.host_eig_merge == 0 ? (1 << 30) : (opt.host_eig_merge == 1 ? 24 : 64);
    // one-workgroup cycle kernel (medium blocks): where the split eigensolve would want a mid-cycle read-back the step
    // kernels stay in charge, so that either engine hands the host the same work
    const bool blk1 = !cyc && block1_plan(W, krylovdim, !R.arpack && krylovdim >= merge_from);
    const bool use_split = !cyc && !blk1 && !R.arpack && krylovdim >= merge_from;
    struct PoolGuard { SpinPool* p; ~PoolGuard() { if (p) p->disarm(); } } pool_guard{nullptr};
    W.split.M.par = nullptr;
    if (use_split && merge_helpers() > 0 && StreamRef::tl == nullptr) {      // (not from a block worker thread: one pool per solver)
        if (!merge_pool) {
            merge_pool.reset(new SpinPool(merge_helpers()));
            merge_par = [this](int n, const std::function<void(int)>& body) { merge_pool->run(n, body); };
        }
        merge_pool->arm();
        pool_guard.p = merge_pool.get();
        W.split.M.par = &merge_par;
        W.split.M.nchunk = merge_helpers() + 1;
    }
    if (use_split && W.side == nullptr) {
        PX_HIP(hipStreamCreateWithFlags(&W.side, hipStreamNonBlocking));
        PX_HIP(hipEventCreateWithFlags(&W.ev_mid, hipEventDisableTiming));
        PX_HIP(hipEventCreateWithFlags(&W.ev_early, hipEventDisableTiming));
        W.rec_early.alloc(EigWork::REC_DOUBLES);
    }
    while (true) {
        const int kfirst = R.kfirst;
        // split point: the arrow with its hub after a restart, the first half of the tridiagonal in the first cycle;
        // the coefficients of step k1 - 1 exist once the launch of step k1 has closed it
        // (first cycle: T1 takes 72 % of the steps -- its QL, ~0.45 ms (k1/127)^3, still ends before the remaining
        // steps do, and the tail left for the critical path shrinks to a quarter of the basis: (0.28)^3 of the QL work)
        const int k1 = (kfirst == 0) ? (18 * krylovdim) / 25 : kfirst + 1;
        const bool split_cycle = use_split && k1 >= 1 && k1 <= krylovdim - 2;
        if (cyc) {
            launch_cycle(W, kfirst, krylovdim, step_tol, cyR, cyG, cyF);
            PX_HIP(hipMemcpyAsync(W.cy_err_host.p, W.cy_err.p, sizeof(int), hipMemcpyDeviceToHost, stream));
        } else if (blk1) {
            launch_block1(W, xp, kfirst, krylovdim, step_tol);
            R.presymv = false;
        } else {
        for (int k = kfirst; k < krylovdim; ++k) {
            lz_launch_step(W, xp, k, kfirst, step_tol, R.presymv);
            if (split_cycle && k == k1) {
                // alphas / betas up to step k1 - 1 are final: copy them out on the side stream while the cycle runs on
                PX_HIP(hipEventRecord(W.ev_mid, stream));
                PX_HIP(hipStreamWaitEvent(W.side, W.ev_mid, 0));
                PX_HIP(hipMemcpyAsync(W.rec_early.p, W.rec.p, EigWork::REC_DOUBLES * sizeof(double), hipMemcpyDeviceToHost, W.side));
                PX_HIP(hipEventRecord(W.ev_early, W.side));
            }
        }
        auto klf = (krylovdim <= 64) ? dev::k_lz_finish<1> : (krylovdim <= 128) ? dev::k_lz_finish<2> : (krylovdim <= 192) ? dev::k_lz_finish<3> : dev::k_lz_finish<4>;
        hipLaunchKernelGGL(klf, dim3(W.nt), dim3(dev::TPB), 0, stream,
                           W.w.p, W.n, W.V.p, W.npad, krylovdim - 1, lz_hpart(W, krylovdim - 1), W.pld, W.hsum1.p,
                           W.alphas_p, W.betas_p, W.ctl_p, step_tol, (krylovdim - 1 > kfirst) ? 1 : 0, W.hred.p);
        }
        // the first mat-vec of a possible next cycle only needs v_K = V[:,krylovdim], which is
        // final now: enqueue it before the host round trip so the GPU works during the K x K
        // eigensolve (wasted only when this cycle turns out to be the last one)
        // -- speculated only when this block's previous projection needed a restart too
        const bool speculate = !cyc && !blk1 && (W.prev_numiter > 1 || R.numiter > 1);
        if (speculate) {
            launch_symv(W, xp, W.V.p + (size_t)krylovdim * W.npad, true);
            W.lst.symv_launches--; W.lst.symv_bytes -= 8.0 * (double)W.N + 16.0 * (double)W.n;   // counted when used
        }
        PX_HIP(hipMemcpyAsync(W.rec_host, W.rec.p, EigWork::REC_DOUBLES * sizeof(double), hipMemcpyDeviceToHost, stream));
        // host work under the GPU's cycle: the first part of the split eigensolve, or the arrow reduction of the QL path
        if (!(split_cycle && lz_split_first(W, R, k1))) lz_prepare_arrow(R);
        const double tdbg0 = debug ? now_s() : 0.0;
        wait_stream();
        if (debug) { dbg_lz[0] += now_s() - tdbg0; dbg_lz[5] += 1.0; }
        if (W.cye_pending) {
            W.cye_pending = false;
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, W.cye[0], W.cye[1]) == hipSuccess) W.lst.cycle_ms += ms;
        }
        if (cyc) std::memcpy(&cy_err_host, W.cy_err_host.p, sizeof(int));
        if (cyc && cy_err_host != 0) {
            // a spin inside the persistent kernel timed out (its workgroups were not all resident):
            // switch this block to the step kernels for the rest of the solve and redo the projection
            W.cy_disabled = true;
            W.cy_err.zero(stream);
            W.lst.lanczos_calls--;
            lanczos(W, xp, nev, positive_part);
            return;
        }
        if (W.ev.used) {                                   // harvest profiled symv launches
            for (size_t s = 0; s < W.ev.used; ++s) {
                float ms = 0.f;
                if (hipEventElapsedTime(&ms, W.ev.e0[s], W.ev.e1[s]) == hipSuccess) {
                    W.lst.symv_profiled_ms += ms; W.lst.symv_profiled++;
                }
            }
            W.ev.used = 0;
        }
        for (size_t s = 0; s < W.evo.used; ++s) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, W.evo.e0[s], W.evo.e1[s]) == hipSuccess) {
                W.lst.orth_profiled_ms += ms; W.lst.orth_profiled++;
            }
        }
        W.evo.used = 0;
        const double tdbg1 = debug ? now_s() : 0.0;
        W.b1_defer = blk1;
        const bool go_on = lz_after_cycle(W, R, speculate);
        W.b1_defer = false;
        if (debug) dbg_lz[1] += now_s() - tdbg1;
        if (!go_on) break;
    }
    const double tdbg2 = debug ? now_s() : 0.0;
    lz_finish_run(W, R);
    if (debug) dbg_lz[3] += now_s() - tdbg2;
}

inline void Solver::flush_rotations(RotSink& S, BatchCtx& C) {
    dev::LzRotBatch B{};
    size_t used = 0, lds = 0;
    int nt = 0;
    for (RotSink::Req& r : S.slot) {
        if (!r.valid) continue;
        r.valid = false;
        const size_t len = r.data.size();
        if (used + len > (size_t)dev::LZB_MAX * LZB_USTRIDE) throw std::logic_error("flush_rotations: staging buffer too small");
        std::memcpy(C.U_host.p + used, r.data.data(), len * sizeof(double));
        B.r[B.nb++] = dev::LzRot{r.V, C.U.p + used, r.out, r.K, r.ncols, r.copy_src, r.copy_dst};
        if (r.nextra > 0) r.W->arrow_p = C.U.p + used + (size_t)r.K * std::max(r.ncols, 0);
        lds = std::max(lds, ((size_t)r.K * std::max(r.ncols, 0) + (size_t)r.K * (dev::LZ_ROWS + 1)) * sizeof(double));
        B.npad = r.W->npad; nt = r.W->nt;
        used += (len + 7) & ~(size_t)7;
    }
    if (B.nb == 0) return;
    if ((int)lds > rotate_lds_cap) throw std::logic_error("flush_rotations: LDS budget exceeded");
    PX_HIP(hipMemcpyAsync(C.U.p, C.U_host.p, used * sizeof(double), hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(dev::k_lzb_rotate, dim3(nt, 1, B.nb), dim3(dev::TPB), lds, stream, B);
}

// KrylovKit eigsolve of SEVERAL blocks of equal side at once (kernels.hip.hpp "BATCHED Lanczos step"): one
// launch per step for all of them (grid.z = block) on the solver's stream, per-block host restart logic
// (lz_after_cycle) between the cycles.  Per block the arithmetic, the mat-vec count and the restart count are
// those of lanczos(); blocks that have converged drop out of the launches.  Preconditions (checked by
// batch_eligible): plain KrylovKit mode, packed-triangle operator, krylovdim <= 63.
inline void Solver::lanczos_batch(const std::vector<int>& blocks, const double* xbase, const std::vector<int>& nevs, int ctx_slot) {
    const int nb = (int)blocks.size();
    if (ctx_slot < 0 || ctx_slot >= (int)batch_ctx.size() || !batch_ctx[ctx_slot]) throw std::logic_error("lanczos_batch: no context for this group");
    BatchCtx& C = *batch_ctx[ctx_slot];
    const bool concurrent = StreamRef::tl != nullptr;        // running on a group worker thread: shared counters go through a lock
    long long steps_local = 0;
    if (nb < 1 || nb > dev::LZB_MAX) throw std::invalid_argument("lanczos_batch: 1..LZB_MAX blocks");
    auto Rq = [&](int q) -> LzRun& { return eig[blocks[q]].lzrun; };     // host state of block q's run (lives in its workspace)
    std::vector<char> live(nb, 0), ran(nb, 0);
    EigWork& W0 = eig[blocks[0]];
    dev::LzBatch B{};
    B.n = W0.n; B.nt = W0.nt; B.npad = W0.npad; B.pld = W0.pld; B.napart = W0.napart; B.nb = nb;
    for (int q = 0; q < nb; ++q) {
        EigWork& W = eig[blocks[q]];
        W.use_fop = false;
        W.batch_slot = q;
        live[q] = ran[q] = lz_init(W, Rq(q), nevs[q], false) ? 1 : 0;
        if (Rq(q).krylovdim > 63) throw std::invalid_argument("lanczos_batch: krylovdim > 63");
        B.tol = Rq(q).step_tol;
    }
    auto fill = [&](int q) -> dev::LzBlk& {
        EigWork& W = eig[blocks[q]];
        dev::LzBlk& b = B.b[q];
        b.xp = xbase + P.blocks[blocks[q]].off;
        b.Ppart = W.Ppart.p; b.wbuf = W.w.p; b.V = W.V.p; b.hpart1 = W.hpart1.p; b.hpart2 = W.hpart2.p;
        b.hsum = W.hsum1.p; b.alphas = W.alphas_p; b.betas = W.betas_p; b.ctl = W.ctl_p; b.Apart = W.Apart.p;
        b.hred = W.hred.p; b.arrow = W.arrow_p; b.resid = W.resid.p;
        return b;
    };
    const int ntile = 8 * ceil_div(W0.nt * (W0.nt + 1) / 2, 8);
    if (C.U.n == 0) {
        C.U.alloc(dev::LZB_MAX * LZB_USTRIDE); C.U_host.alloc(dev::LZB_MAX * LZB_USTRIDE);
        C.rec_host.alloc(dev::LZB_MAX * EigWork::REC_DOUBLES);
    }
    RotSink sink;
    // helper threads for the restart logic (options.block_threads: -1 auto = one per block up to 8, 0 = none): they spin
    // only while this projection is in progress
    const int nhelp = std::min(nb, opt.block_threads < 0 ? 8 : (int)opt.block_threads) - 1;
    if (nhelp >= 1 && (!C.pool || C.pool->helpers() < nhelp)) C.pool.reset(new SpinPool(nhelp));
    SpinPool* pool = (nhelp >= 1) ? C.pool.get() : nullptr;
    struct SinkGuard { SpinPool* p; ~SinkGuard() { Solver::rot_sink = nullptr; if (p) p->disarm(); } } guard{pool};
    rot_sink = &sink;
    if (pool) pool->arm();
    for (int q = 0; q < nb; ++q) fill(q).mode = live[q] ? 1 : 0;
    hipLaunchKernelGGL(dev::k_lzb_begin, dim3(ceil_div(W0.npad, dev::TPB), 1, nb), dim3(dev::TPB), 0, stream, B);
    const double mv_bytes = 8.0 * (double)W0.N + 16.0 * (double)W0.n;
    long long nlaunch = 0, prof_blocks = 0;
    double tp0 = debug ? now_s() : 0.0;                  // PROXSDP_HIP_DEBUG: host-time split of the batched run
    auto lap = [&](double& acc) { if (debug && !concurrent) { const double t = now_s(); acc += t - tp0; tp0 = t; } };
    std::vector<char> presymv(nb, 0);                // the first mat-vec of the block's next cycle is already in Ppart (speculated)
    if (C.ev == nullptr) PX_HIP(hipEventCreateWithFlags(&C.ev, hipEventDisableTiming));
    double* const rec_out = C.rec_host.p;
    const int rec_n = (int)EigWork::REC_DOUBLES;
    while (true) {
        int tmax = 0, nlive = 0;
        for (int q = 0; q < nb; ++q)
            if (live[q]) { tmax = std::max(tmax, Rq(q).krylovdim - Rq(q).kfirst); ++nlive; }
        if (nlive == 0) break;
        for (int t = 0; t <= tmax; ++t) {
            bool any_orth = false, any_mv = false;
            for (int q = 0; q < nb; ++q) {
                dev::LzBlk& b = fill(q);               // (V / arrow pointers change at a restart)
                b.mode = 0;
                if (!live[q]) continue;
                const int k = Rq(q).kfirst + t, kd = Rq(q).krylovdim;
                b.k = k; b.keep = Rq(q).kfirst;
                b.mode = (t == 0) ? (presymv[q] ? 0 : 1) : (k < kd) ? 2 : (k == kd) ? 3 : 0;
                any_mv = any_mv || b.mode != 0;
                if (k < kd) {
                    any_orth = true;
                    eig[blocks[q]].lst.symv_launches++; eig[blocks[q]].lst.symv_bytes += mv_bytes;
                }
            }
            // every profile_symv_every-th batched mat-vec launch carries its own start/stop events
            const bool prof = any_mv && opt.profile_symv_every > 0 && t > 0 && t < tmax && (nlaunch++ % opt.profile_symv_every) == 0;
            if (prof) {
                if (W0.ev.used == W0.ev.e0.size()) {
                    hipEvent_t a, b2;
                    PX_HIP(hipEventCreate(&a)); PX_HIP(hipEventCreate(&b2));
                    W0.ev.e0.push_back(a); W0.ev.e1.push_back(b2);
                }
                const size_t slot = W0.ev.used++;
                hipExtLaunchKernelGGL(dev::k_lzb_mv, dim3(W0.nt + ntile, 1, nb), dim3(dev::TPB), 0, stream,
                                      W0.ev.e0[slot], W0.ev.e1[slot], 0, B, rec_out, rec_n);
                int act = 0;
                for (int q = 0; q < nb; ++q) act += (B.b[q].mode == 2) ? 1 : 0;
                prof_blocks += act;
            } else if (any_mv)          // (t = 0 with every live block's first mat-vec already speculated: nothing to launch)
            hipLaunchKernelGGL(dev::k_lzb_mv, dim3(W0.nt + ntile, 1, nb), dim3(dev::TPB), 0, stream, B, rec_out, rec_n);
            if (!any_orth) continue;
            for (int q = 0; q < nb; ++q) {
                dev::LzBlk& b = B.b[q];
                const bool act = live[q] && b.k < Rq(q).krylovdim;       // step b.k of the block's cycle exists
                b.mode = act ? 1 : 0;                                     // (k_lzb_orth: != 0 = active)
            }
            hipLaunchKernelGGL(dev::k_lzb_orth, dim3(W0.nt, 1, nb), dim3(dev::TPB), 0, stream, B);
            steps_local += nlive;
        }
        // every live block's record [alphas | betas | ctl] has been copied out by the closing workgroup 0 of its mode-3 launch
        // (straight into pinned host memory): the host waits for THAT point of the stream ...
        PX_HIP(hipEventRecord(C.ev, stream));
        // ... while the GPU already runs the first mat-vec of every block's NEXT cycle (on v_K = V[:, krylovdim], final since
        // the mode-3 closing; the restart rotation copies it to column `keep`, so the tiles' result stays valid) -- wasted
        // only for the blocks that turn out to have converged
        for (int q = 0; q < nb; ++q) {
            dev::LzBlk& b = fill(q);
            b.mode = live[q] ? 1 : 0;
            b.k = live[q] ? Rq(q).krylovdim : 0; b.keep = 0;
        }
        hipLaunchKernelGGL(dev::k_lzb_mv, dim3(W0.nt + ntile, 1, nb), dim3(dev::TPB), 0, stream, B, (double*)nullptr, 0);
        for (int q = 0; q < nb; ++q) presymv[q] = live[q];
        for (int q = 0; q < nb; ++q) if (live[q]) lz_prepare_arrow(Rq(q));
        lap(dbg_batch[0]);                               // enqueue (+ arrow reductions)
        wait_event(C.ev);
        lap(dbg_batch[1]);                               // waiting for the GPU
        for (int q = 0; q < nb; ++q)
            if (live[q]) std::memcpy(eig[blocks[q]].rec_host, C.rec_host.p + (size_t)q * EigWork::REC_DOUBLES, EigWork::REC_DOUBLES * sizeof(double));
        for (size_t sl = 0; sl < W0.ev.used; ++sl) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, W0.ev.e0[sl], W0.ev.e1[sl]) == hipSuccess) {
                W0.lst.symv_profiled_ms += ms; W0.lst.symv_profiled++;
            }
        }
        W0.ev.used = 0;
        // per-block restart logic (K x K eigensolve, convergence, rotation matrix): independent, on the helper threads
        std::exception_ptr err[dev::LZB_MAX] = {};
        auto job = [&](int q) {
            if (!live[q]) return;
            RotSink* const prev = Solver::rot_sink;          // (thread-local: a helper thread stages into THIS run's sink)
            Solver::rot_sink = &sink;
            try { live[q] = lz_after_cycle(eig[blocks[q]], Rq(q), false) ? 1 : 0; }
            catch (...) { err[q] = std::current_exception(); live[q] = 0; }
            Solver::rot_sink = prev;
        };
        if (pool) pool->run(nb, job);
        else for (int q = 0; q < nb; ++q) job(q);
        for (int q = 0; q < nb; ++q) if (err[q]) std::rethrow_exception(err[q]);
        lap(dbg_batch[2]);                               // restart logic
        flush_rotations(sink, C);                    // the restart rotations of this cycle: one upload, one launch
        lap(dbg_batch[3]);
        if (!concurrent) dbg_batch[4] += 1.0;
    }
    for (int q = 0; q < nb; ++q) if (ran[q]) lz_finish_run(eig[blocks[q]], Rq(q));
    flush_rotations(sink, C);                        // the Ritz vectors of every block
    {
        std::lock_guard<std::mutex> lk(batch_stats_mu);
        st.batched_block_steps += steps_local;
        st.batched_profiled_blocks += prof_blocks;   // blocks served by the event-bracketed launches (bytes = this x (8N + 16n))
    }
}

// eigen!(Symmetric(smat(xp))) through rocSOLVER dsyevd (ascending), the dense
// fallback (prox_operators.jl:113, pdhg.jl:685)
inline void Solver::full_eig_values(EigWork& W, const double* xp, double offscale, bool vectors,
                                    std::vector<double>& Dhost) {
    const int n = W.n;
    if (W.A.n < (size_t)n * n) {
        W.A.alloc((size_t)n * n); W.D.alloc(n); W.E.alloc(n); W.info.alloc(1);
    }
    const int ntile = W.nt * (W.nt + 1) / 2;
    hipLaunchKernelGGL(dev::k_unpack_upper, dim3(ntile), dim3(dev::TPB), 0, stream, xp, n, W.A.p, n, offscale);
    std::lock_guard<std::mutex> lk(blas_mutex);          // one rocBLAS handle: dense fallbacks run one at a time
    PX_ROC(rocblas_set_stream(blas, (hipStream_t)stream));
    PX_ROC(rocsolver_dsyevd(blas, vectors ? rocblas_evect_original : rocblas_evect_none, rocblas_fill_upper,
                            n, W.A.p, n, W.D.p, W.E.p, W.info.p));
    Dhost.resize(n);
    rocblas_int info = 0;
    W.D.download(Dhost.data(), n, stream);
    PX_HIP(hipMemcpyAsync(&info, W.info.p, sizeof(info), hipMemcpyDeviceToHost, stream));
    PX_HIP(hipStreamSynchronize(stream));
    if (info != 0) throw HipError("rocsolver_dsyevd did not converge (info=" + std::to_string(info) + ")");
}

// full_eig! (prox_operators.jl:111-126)
template <int EPI, bool FUSE>
inline void Solver::sym_gemm(EigWork& W, const double* Pm, const double* Qm, double* T, const double* Y, double ca,
                             double cb, double cc, const double* dsc, double* part, double* xp_out,
                             const double* xp_old, int blk) {
    W.lst.sign_products++;
    if constexpr (EPI != dev::SG_FINAL) {
        if (W.sg_nt48 > 0) {                                // 48 x 48 tiles: fewer, larger tiles where that shortens the busiest CU's queue
            const int grid48 = 8 * ceil_div(W.sg_nt48 * (W.sg_nt48 + 1) / 2, 8);
            hipLaunchKernelGGL((dev::k_sym_gemm48<EPI, 4>), dim3(grid48), dim3(dev::TPB), dev::sg48_lds_bytes(4), stream, Pm, Qm, W.sg_ld,
                               W.sg_nt48, T, Y, ca, cb, cc, dsc, part);
            return;
        }
        if (W.sg_small) {                                   // 32 x 32 tiles: small blocks need the workgroups
            const int nt32 = 2 * W.nt;
            const int grid32 = 8 * ceil_div(nt32 * (nt32 + 1) / 2, 8);
            hipLaunchKernelGGL((dev::k_sym_gemm32<EPI>), dim3(grid32), dim3(dev::TPB), 0, stream, Pm, Qm, W.sg_ld, nt32, T, Y,
                               ca, cb, cc, dsc, part);
            return;
        }
    }
    const int grid = 8 * ceil_div(W.nt * (W.nt + 1) / 2, 8);
    hipLaunchKernelGGL((dev::k_sym_gemm<EPI, FUSE>), dim3(grid), dim3(dev::TPB), 0, stream, Pm, Qm, W.sg_ld, W.nt, T, Y,
                       ca, cb, cc, dsc, part, W.n, xp_out, xp_old, FUSE ? (const unsigned*)mask_d.p : nullptr,
                       FUSE ? (long long)P.blocks[blk].off : 0LL, FUSE ? respart_d.p + tile_base[blk] : nullptr,
                       FUSE ? rstride : 0);
}

// full_eig! without an eigendecomposition: X+ = (X + X sign(X)) / 2, sign by an odd-polynomial
// iteration of fp64 MFMA products (sign_project.hip.hpp).  current_rank = #{lambda > 0} =
// (tr S + tr S^2) / 2 (the reference counts lambda > tol_psd: eigenvalues in (0, tol_psd] are the only
// difference, and the count only feeds Result.final_rank on this path: min_eig = 0 blocks bump_rank).
inline bool Solver::full_eig_by_sign(int idx, const double* xp_in, double* xp_out, bool fuse, bool force) {
    EigWork& W = eig[idx];
    const int n = W.n;
    if (!force) {
        if (opt.full_eig_sign == 0) return false;
        // auto: from side 33 on (below: the batched Jacobi kernel / rocSOLVER are faster), up to the side whose five
        // padded work matrices fit comfortably (side 16384: 10.7 GB; the cap was 4096 until round 4 -- maxG55 / maxG60,
        // sides 5000 / 7000, fell to rocSOLVER's dsyevd at ~0.2 % of the fp64 peak) and while HBM has room for them
        if (opt.full_eig_sign < 0 && (n < 33 || n > 16384)) return false;
        // auto: X+ carries an absolute error of up to ~1e-10 x the spectral scale on this path (l_0 of the
        // iteration): users who ask for tolerances near that floor get the LAPACK-accurate dense eigensolver
        if (opt.full_eig_sign < 0 && std::min({opt.tol_gap, opt.tol_feasibility, opt.tol_primal, opt.tol_dual}) < 1e-8) return false;
    }
    const int ld = W.nt * dev::TILE;
    const int ntile = W.nt * (W.nt + 1) / 2, grid = 8 * ceil_div(ntile, 8);
    if (W.sg_ld != ld) {
        // first use: the five work matrices must fit -- whoever asks (auto mode, full_eig_sign = 1, or the opt-in
        // psd_sign_engine with force = true: ADVICE r4); no room, or an allocation that fails anyway (several block
        // threads can pass the test together), DECLINES this engine instead of failing the solve
        static std::mutex sign_alloc_mu;
        std::lock_guard<std::mutex> lk(sign_alloc_mu);
        size_t free_b = 0, total_b = 0;
        const size_t need = (size_t)6 * (size_t)ld * (size_t)ld * sizeof(double);
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < need) return false;
        const size_t sz = (size_t)ld * ld;
        try {
            W.sgA.alloc(sz); W.sgX.alloc(sz); W.sgX2.alloc(sz); W.sgY.alloc(sz); W.sgQ.alloc(sz);
        } catch (const std::bad_alloc&) {
            W.sgA.release(); W.sgX.release(); W.sgX2.release(); W.sgY.release(); W.sgQ.release();
            (void)hipGetLastError();
            return false;
        }
        W.sgA.zero(stream);                                  // the padding stays zero: only entries < n are rewritten
        const int small_max = opt.sign_small_tile_max > 0 ? opt.sign_small_tile_max : 3072;   // measured: 32-tiles win up to n ~ 3500
        W.sg_small = ld <= small_max;
        const int nt32 = 2 * W.nt;
        W.sg_npart = W.sg_small ? 8 * ceil_div(nt32 * (nt32 + 1) / 2, 8) : grid;
        // tile shape by makespan: the busiest of the 256 CUs runs ceil(tiles / 256) tiles of T x T entries each.  n = 1000: 528 tiles of
        // 32 x 32 (3 x 1024) against 231 of 48 x 48 (1 x 2304); n = 800: 325 (2 x 1024) against 153 (2304) -- the 32-tiles stay.  Measured
        // (tools/gpurun_tile48_sizes.sh, ms per projection 32 / 48): 700: 0.44 / 0.44, 800: 0.77 / 0.83, 1000: 1.21 / 1.00, 1200: 1.57 / 2.0,
        // 1500: 2.9 / ~3.6, 2000 (a tie of the rule): 6.6 / 6.4, 2500: 12.9 / 13.2 -- the rule picks the faster one each time.
        // PROXSDP_HIP_SIGN_TILE48 = 0 / 1: never / whenever the tiles fit (measurement)
        W.sg_nt48 = 0;
        {
            const int nb48 = ceil_div(n, dev::SG_T48);
            const char* env = std::getenv("PROXSDP_HIP_SIGN_TILE48");
            const int knob = env != nullptr ? std::atoi(env) : -1;
            if (W.sg_small && sg48_ok && knob != 0 && nb48 * dev::SG_T48 <= ld) {
                const int cus = 256;
                const long long c32 = (long long)ceil_div(nt32 * (nt32 + 1) / 2, cus) * 32 * 32;
                const long long c48 = (long long)ceil_div(nb48 * (nb48 + 1) / 2, cus) * 48 * 48;
                if (knob == 1 || c48 <= c32) {
                    W.sg_nt48 = nb48;
                    W.sg_npart = 8 * ceil_div(nb48 * (nb48 + 1) / 2, 8);
                    // rows / columns [48 nb48, ld) are never written by these tiles and are read by every product's K loop
                    W.sgX.zero(stream); W.sgX2.zero(stream); W.sgY.zero(stream); W.sgQ.zero(stream);
                    // (measured and dropped: walking the tiles in S x S patches per XCD instead of column-major ranges -- half the
                    // operand blocks per XCD's L2 -- changes nothing, 0.95 - 0.98 ms per projection either way: the Infinity Cache
                    // serves the misses under the 24 loads in flight; eight k-steps per group instead of four: slower, 1.01 ms)
                }
            }
        }
        W.sg_part.alloc(W.sg_npart); W.sg_part2.alloc(W.sg_npart); W.sg_sc.alloc(16); W.sg_host.alloc(16);
        W.sg_sc.zero(stream);
        W.sg_ld = ld;
    }
    const bool prof = opt.profile_symv_every > 0;
    if (prof) {
        if (W.fe[0] == nullptr) for (auto& e : W.fe) PX_HIP(hipEventCreate(&e));
        harvest_full_eig_events(W);
        PX_HIP(hipEventRecord(W.fe[0], stream));
    }
    const bool fz = fuse && use_support;
    double* sc = W.sg_sc.p;
    hipLaunchKernelGGL(dev::k_unpack_sym, dim3(ntile), dim3(dev::TPB), 0, stream, xp_in, n, W.sgA.p, ld,
                       dev::INV_SQRT2, W.sg_part.p);
    hipLaunchKernelGGL(dev::k_sign_scalars, dim3(1), dim3(dev::TPB), 0, stream, (const double*)W.sg_part.p, ntile, 0, sc);
    // Y0 = A A / f^2 and its Frobenius norm g: s = f sqrt(g) >= ||A||_2
    sym_gemm<dev::SG_PLAIN, false>(W, W.sgA.p, W.sgA.p, W.sgY.p, nullptr, 0, 0, 0, sc + 0, W.sg_part.p, nullptr, nullptr, -1);
    hipLaunchKernelGGL(dev::k_sign_scalars, dim3(1), dim3(dev::TPB), 0, stream, (const double*)W.sg_part.p, W.sg_npart, 1, sc);
    double* X = W.sgX.p;
    double* Xn = W.sgX2.p;
    // rows [from, SIGN_STEPS) of the table; `first`: X does not exist yet (the step works on A and Y0)
    auto run_rows = [&](int from, bool first) {
        for (int k = from; k < dev::SIGN_STEPS; ++k) {
            const dev::SignStep& c = dev::SIGN_TABLE[k];
            const bool last = k + 1 == dev::SIGN_STEPS;
            if (first && k == from) {
                sym_gemm<dev::SG_POLY, false>(W, W.sgY.p, W.sgY.p, W.sgQ.p, W.sgY.p, c.a, c.b, c.c, sc + 8, nullptr, nullptr, nullptr, -1);
                sym_gemm<dev::SG_PLAIN, false>(W, W.sgA.p, W.sgQ.p, X, nullptr, 0, 0, 0, sc + 1, nullptr, nullptr, nullptr, -1);
            } else if (last && dev::SIGN_LAST_CUBIC) {
                // Q = (3 I - X X) / 2 straight from the product's epilogue (the Y term is switched off)
                sym_gemm<dev::SG_POLY, false>(W, X, X, W.sgQ.p, X, 1.5, 0.0, -0.5, nullptr, nullptr, nullptr, nullptr, -1);
                sym_gemm<dev::SG_PLAIN, false>(W, X, W.sgQ.p, Xn, nullptr, 0, 0, 0, nullptr, W.sg_part2.p, nullptr, nullptr, -1);
                std::swap(X, Xn);
            } else {
                sym_gemm<dev::SG_PLAIN, false>(W, X, X, W.sgY.p, nullptr, 0, 0, 0, nullptr, nullptr, nullptr, nullptr, -1);
                sym_gemm<dev::SG_POLY, false>(W, W.sgY.p, W.sgY.p, W.sgQ.p, W.sgY.p, c.a, c.b, c.c, nullptr, nullptr, nullptr, nullptr, -1);
                sym_gemm<dev::SG_PLAIN, false>(W, X, W.sgQ.p, Xn, nullptr, 0, 0, 0, nullptr, last ? W.sg_part2.p : nullptr,
                                               nullptr, nullptr, -1);
                std::swap(X, Xn);
            }
        }
    };
    // SHORTENED SCHEDULE.  The table is laid out for |eigenvalues| down to 1e-10 s; the matrices a solve projects rarely
    // come closer to singular than 1e-5 s (measured on the SDPLIB iterates: 1e-5 .. 1e-3).  So the iteration STARTS AT ROW
    // j > 0 (the rows from l_j on: every |eigenvalue| >= l_j s converges exactly as before: to 1 - 1e-13 after the last,
    // cubic, row) and the result is TESTED: with t_i the eigenvalues of the computed S,
    // ||S||_F^2 - ||S S||_F^2 = sum t_i^2 (1 - t_i^2) vanishes iff every t_i is in {0, +-1}; an eigenvalue that started
    // below l_j has been amplified by gain_j (the product of the rows' slopes at 0) and shows up as t^2 unless it started
    // below sqrt(tau) / gain_j <= 1e-10 s -- the size of eigenvalue the full table does not resolve either.  tau = 4e-13 n:
    // the converged eigenvalues contribute <= 2e-13 each (measured on SDPLIB iterates: the statistic stays below 6e-11 at
    // n = 501 / 1000 when it passes and above 1e-8 when it fails).  A failed test continues with the rows from
    // l_r <= 1e-10 gain_j on: same guarantee as the full table, 63 - 64 products instead of 57.
    auto launch_final = [&]() {
        if (fz) sym_gemm<dev::SG_FINAL, true>(W, W.sgA.p, X, nullptr, nullptr, 0, 0, 0, nullptr, W.sg_part.p, xp_out, xp_in, idx);
        else sym_gemm<dev::SG_FINAL, false>(W, W.sgA.p, X, nullptr, nullptr, 0, 0, 0, nullptr, W.sg_part.p, xp_out, nullptr, -1);
        if (prof) { PX_HIP(hipEventRecord(W.fe[1], stream)); PX_HIP(hipEventRecord(W.fe[2], stream)); W.fe_pending = true; }
    };
    static const dev::SignSchedule sched;
    const int row_opt = opt.sign_start_row;
    const double tau = 4e-13 * (double)n;
    int jmax = 0;
    for (int j = 1; j + 2 < dev::SIGN_STEPS; ++j) if (std::sqrt(tau) / sched.gain[j] <= 1e-10) jmax = j;
    // auto: per block, start at row 8 (the iterates of the SDPLIB solves fail its test once in 70 .. 400 projections);
    // a failure moves the next 8 projections two rows down (an eigenvalue crossing zero lingers for a few iterations),
    // a second failure soon after lowers the block's row for good (back up one row per 64 clean projections);
    // at row 0 (no test) the shortened schedule is tried again every 64 projections
    const int jtop = std::min(8, jmax);
    if (W.sg_row < 0) W.sg_row = jtop;
    if (row_opt < 0 && W.sg_row == 0 && jtop > 0 && ++W.sg_idle >= 64) { W.sg_row = std::min(6, jtop); W.sg_idle = 0; W.sg_ok = 16; }
    int j0 = row_opt < 0 ? (W.sg_hold > 0 ? std::max(0, W.sg_row - 2) : W.sg_row) : row_opt;
    j0 = std::max(0, std::min(j0, jmax));
    if (row_opt < 0 && j0 == 0 && W.sg_hold > 0) --W.sg_hold;          // (a hold at row 0 runs the full table, untested)
    bool resolved = false;
    run_rows(j0, true);
    if (j0 > 0) {
        // the test: ||S S||_F^2 needs one more product (its result is not used otherwise)
        sym_gemm<dev::SG_PLAIN, false>(W, X, X, W.sgY.p, nullptr, 0, 0, 0, nullptr, W.sg_part.p, nullptr, nullptr, -1);
        hipLaunchKernelGGL(dev::k_sign_check, dim3(3), dim3(dev::TPB), 0, stream, (const double*)W.sg_part2.p,
                           (const double*)W.sg_part.p, W.sg_npart, (const double*)X, ld, n, sc);
        PX_HIP(hipMemcpyAsync(W.sg_host.p, sc, 16 * sizeof(double), hipMemcpyDeviceToHost, stream));
        // the final product is enqueued BEFORE the host looks at the test (out of place only: a redo must find its input
        // intact): the GPU runs it while the host reads the three scalars and goes back to enqueueing the iteration
        const bool ahead = (const double*)xp_out != xp_in;
        if (ahead) {
            if (W.sg_ev == nullptr) PX_HIP(hipEventCreateWithFlags(&W.sg_ev, hipEventDisableTiming));
            PX_HIP(hipEventRecord(W.sg_ev, stream));
            launch_final();
            wait_event(W.sg_ev);
        } else {
            wait_stream();
        }
        const double m2 = W.sg_host.p[7], m4 = W.sg_host.p[11];
        resolved = std::isfinite(m2) && std::isfinite(m4) && std::fabs(m2 - m4) <= tau;
        if (debug) std::fprintf(stderr, "[dbg] sign short: block %d row %d m2 - m4 %.3e tau %.3e %s\n", idx, j0, m2 - m4, tau, resolved ? "pass" : "FAIL");
        if (resolved) {
            W.lst.sign_short_pass++;
            if (row_opt < 0) {
                if (W.sg_hold > 0) --W.sg_hold;
                if (++W.sg_ok >= 64 + 16 && W.sg_row < jtop) { W.sg_row++; W.sg_ok = 16; }
            }
            if (!ahead) launch_final();
        } else {
            W.lst.sign_short_fail++;
            if (row_opt < 0) {
                if (W.sg_ok < 16) W.sg_row = std::max(0, W.sg_row - 1);
                W.sg_hold = 8; W.sg_ok = 0;
            }
            int r = 0;
            for (int k = 0; k < dev::SIGN_STEPS; ++k) if (sched.l[k] <= 1e-10 * sched.gain[j0]) r = k;
            run_rows(r, false);
        }
    }
    if (!resolved) {
        launch_final();
        hipLaunchKernelGGL(dev::k_sign_scalars, dim3(1), dim3(dev::TPB), 0, stream, (const double*)W.sg_part2.p, W.sg_npart, 3, sc);
        hipLaunchKernelGGL(dev::k_sign_scalars, dim3(1), dim3(dev::TPB), 0, stream, (const double*)W.sg_part.p, grid, 2, sc);
        PX_HIP(hipMemcpyAsync(W.sg_host.p, sc, 16 * sizeof(double), hipMemcpyDeviceToHost, stream));
        wait_stream();
    }
    const double tr = W.sg_host.p[5], fro2 = W.sg_host.p[7];
    if (!std::isfinite(tr) || !std::isfinite(fro2)) throw HipError("sign-function projection produced non-finite values");
    const int npos = (int)std::llround(0.5 * (tr + fro2));
    if (!force) { W.lst.full_eigs++; W.lst.full_eigs_sign++; }
    current_rank[idx] = std::max(0, std::min(npos, n));
    min_eig[idx] = 0.0;                                      // prox_operators.jl:114
    W.last_npos = current_rank[idx];
    return true;
}

inline void Solver::full_eig_project(int idx, const double* xp_in, double* xp_out, bool fuse) {
    if (full_eig_by_sign(idx, xp_in, xp_out, fuse)) return;
    EigWork& W = eig[idx];
    std::vector<double> D;
    const bool prof = opt.profile_symv_every > 0;
    if (prof) {
        if (W.fe[0] == nullptr) for (auto& e : W.fe) PX_HIP(hipEventCreate(&e));
        harvest_full_eig_events(W);                       // previous call's events (completed: see the sync below)
        PX_HIP(hipEventRecord(W.fe[0], stream));
    }
    full_eig_values(W, xp_in, dev::INV_SQRT2, true, D);
    if (prof) PX_HIP(hipEventRecord(W.fe[1], stream));
    W.lst.full_eigs++;
    const int n = W.n;
    int npos = 0, rank = 0;
    for (int i = 0; i < n; ++i) { if (D[i] > 0.0) ++npos; if (D[i] > opt.tol_psd) ++rank; }
    current_rank[idx] = rank;
    min_eig[idx] = 0.0;
    W.last_npos = npos;
    // ascending order: the positive eigenpairs are the trailing npos columns
    launch_reconstruct(W, W.A.p + (size_t)(n - npos) * n, n, W.D.p + (n - npos), npos, xp_out,
                       fuse ? xp_in : nullptr, fuse ? idx : -1);
    if (prof) { PX_HIP(hipEventRecord(W.fe[2], stream)); W.fe_pending = true; }
    W.recon_r += npos;
}
// The Krylov branch (prox_operators.jl:68-109) for a Krylov dimension the step kernels cannot hold (> 255 columns:
// target_rank > 127, or eigsolver_min_lanczos > 255): the same truncated projection from the dense eigensolver -- the
// top `nev` eigenpairs of dsyevd are what a converged KrylovKit / ARPACK run returns, min_eig is the smallest of them
// (:74,:95), the positive ones among them rebuild the block (:78-85,:99-106).
inline void Solver::truncated_project_dense(int idx, const double* xp_in, double* xp_out, bool fuse, int nev) {
    EigWork& W = eig[idx];
    std::vector<double> D;
    full_eig_values(W, xp_in, dev::INV_SQRT2, true, D);
    const int n = W.n, k = std::min(nev, n);
    int npos = 0;
    for (int i = n - k; i < n; ++i) if (D[i] > 0.0) ++npos;       // ascending: the top k are the trailing k
    min_eig[idx] = D[n - k];
    current_rank[idx] += npos;
    W.last_npos = npos;
    W.converged = true; W.converged_eigs = k;
    launch_reconstruct(W, W.A.p + (size_t)(n - npos) * n, n, W.D.p + (n - npos), npos, xp_out,
                       fuse ? xp_in : nullptr, fuse ? idx : -1);
    W.recon_r += npos;
    W.have_factors = false; W.x_prev_sparse = false; W.use_fop = false;
    W.lst.dense_truncated_projections++;
}
inline void Solver::harvest_full_eig_events(EigWork& W) {
    if (!W.fe_pending) return;
    W.fe_pending = false;
    if (hipEventSynchronize(W.fe[2]) != hipSuccess) return;
    float a = 0.f, b = 0.f;
    if (hipEventElapsedTime(&a, W.fe[0], W.fe[1]) == hipSuccess) W.lst.full_eig_solver_ms += a;
    if (hipEventElapsedTime(&b, W.fe[1], W.fe[2]) == hipSuccess) W.lst.full_eig_recon_ms += b;
}

// ---- worker pool for concurrent block projections
inline void Solver::start_workers(int nthreads) {
    const int dev_id = opt.device_id;
    for (int t = 0; t < nthreads; ++t)
        workers.emplace_back([this, dev_id]() {
            (void)hipSetDevice(dev_id);
            for (;;) {
                int idx;
                {
                    std::unique_lock<std::mutex> lk(pool_mu);
                    pool_cv.wait(lk, [this]() { return pool_stop || !pool_queue.empty(); });
                    if (pool_stop) return;
                    idx = pool_queue.back();
                    pool_queue.pop_back();
                }
                try {
                    StreamRef::tl = eig[idx].stream;
                    PX_HIP(hipStreamWaitEvent(eig[idx].stream, ev_main, 0));
                    pool_job(idx);
                    PX_HIP(hipEventRecord(eig[idx].done, eig[idx].stream));
                } catch (...) {
                    std::lock_guard<std::mutex> lk(pool_mu);
                    if (!pool_error) pool_error = std::current_exception();
                }
                StreamRef::tl = nullptr;
                {
                    std::lock_guard<std::mutex> lk(pool_mu);
                    if (--pool_pending == 0) pool_done_cv.notify_all();
                }
            }
        });
}
inline void Solver::stop_workers() {
    {
        std::lock_guard<std::mutex> lk(pool_mu);
        pool_stop = true;
    }
    pool_cv.notify_all();
    for (std::thread& t : workers) if (t.joinable()) t.join();
    workers.clear();
}
// run job(idx) for every listed block: concurrently on the blocks' streams when the pool is
// up, otherwise in sequence on the solver's stream.  Returns after all host work is done and
// the solver's stream has been made to wait for the block streams.
inline void Solver::run_blocks(const std::vector<int>& blocks, const std::function<void(int)>& job) {
    bool have_streams = true;
    for (int idx : blocks) have_streams = have_streams && eig[idx].stream != nullptr;
    if (!parallel_blocks || blocks.size() < 2 || !have_streams) {
        for (int idx : blocks) job(idx);
        merge_block_stats();
        return;
    }
    PX_HIP(hipEventRecord(ev_main, stream.main));
    {
        std::lock_guard<std::mutex> lk(pool_mu);
        pool_job = job;
        pool_queue.assign(blocks.rbegin(), blocks.rend());
        pool_pending = (int)blocks.size();
        pool_error = nullptr;
    }
    pool_cv.notify_all();
    {
        std::unique_lock<std::mutex> lk(pool_mu);
        pool_done_cv.wait(lk, [this]() { return pool_pending == 0; });
    }
    if (pool_error) std::rethrow_exception(pool_error);
    for (int idx : blocks) PX_HIP(hipStreamWaitEvent(stream.main, eig[idx].done, 0));
    merge_block_stats();
}
inline void Solver::merge_block_stats() {
    // per-block counters (filled on the blocks' own threads) into the solver's; one line per field
#define PX_MERGE(f) st.f += a.f
    for (EigWork& W : eig) {
        proxsdp_stats& a = W.lst;
        PX_MERGE(lanczos_matvecs); PX_MERGE(lanczos_restarts); PX_MERGE(lanczos_calls);
        PX_MERGE(full_eigs); PX_MERGE(krylov_fallbacks);
        PX_MERGE(symv_launches); PX_MERGE(symv_bytes);
        PX_MERGE(symv_profiled); PX_MERGE(symv_profiled_ms);
        PX_MERGE(orth_profiled); PX_MERGE(orth_profiled_ms);
        PX_MERGE(full_eig_solver_ms); PX_MERGE(full_eig_recon_ms);
        PX_MERGE(full_eigs_lanczos); PX_MERGE(full_eigs_lanczos_checks); PX_MERGE(full_eigs_lanczos_mismatches);
        PX_MERGE(full_eigs_lanczos_certified); PX_MERGE(full_eigs_lanczos_cert_failed); PX_MERGE(cert_matvecs);
        PX_MERGE(dense_truncated_projections);
        PX_MERGE(full_eigs_sign); PX_MERGE(sign_products); PX_MERGE(sign_short_pass); PX_MERGE(sign_short_fail);
        PX_MERGE(sign_engine_projections); PX_MERGE(sign_engine_rejected);
        PX_MERGE(sign_engine_checks); PX_MERGE(sign_engine_mismatches);
        PX_MERGE(batched_block_steps);
        PX_MERGE(host_eigs); PX_MERGE(host_eig_time); PX_MERGE(host_eig_merges); PX_MERGE(host_eig_overlap_time);
        PX_MERGE(warm_starts); PX_MERGE(device_eigs); PX_MERGE(mfma_reconstructions);
        PX_MERGE(cycle_launches); PX_MERGE(cycle_steps); PX_MERGE(cycle_ms);
        PX_MERGE(fop_projections);
        a = proxsdp_stats{};
        lz_matvec_iter += W.mv_iter; recon_r_iter += W.recon_r;
        W.mv_iter = 0; W.recon_r = 0;
    }
#undef PX_MERGE
}

// scalar reduce of a block-sharded solve over RCCL: every rank's packed record [sums | maxs] is all-gathered on the
// solver's stream (one collective, <= 400 bytes per rank: latency-bound over xGMI) and combined on the host in
// rank order, so every rank computes the same bits and takes the same step-size / rank / termination decisions
inline void Solver::reduce_native(std::vector<double>& sums, std::vector<double>& maxs) {
    const size_t ns = sums.size(), nm = maxs.size(), n = ns + nm;
    if (n == 0) return;
    Rccl& rc = Rccl::get();
    const size_t W = (size_t)nccl_world;
    if (nccl_cap < n) {
        nccl_cap = std::max<size_t>(n, 64);
        nccl_send.alloc(nccl_cap); nccl_recv.alloc(nccl_cap * W); nccl_host.alloc(nccl_cap * (W + 1));
    }
    double* h = nccl_host.p;
    std::copy(sums.begin(), sums.end(), h);
    std::copy(maxs.begin(), maxs.end(), h + ns);
    PX_HIP(hipMemcpyAsync(nccl_send.p, h, n * sizeof(double), hipMemcpyHostToDevice, stream));
    collective_enqueued = true;
    rc.check(rc.AllGather(nccl_send.p, nccl_recv.p, n, ncclFloat64, nccl, stream), "ncclAllGather");
    double* all = h + nccl_cap;
    PX_HIP(hipMemcpyAsync(all, nccl_recv.p, n * W * sizeof(double), hipMemcpyDeviceToHost, stream));
    wait_collective(stream);
    for (size_t q = 0; q < ns; ++q) {
        double a = all[q];
        for (size_t r = 1; r < W; ++r) a += all[r * n + q];
        sums[q] = a;
    }
    for (size_t q = 0; q < nm; ++q) {
        double a = all[ns + q];
        for (size_t r = 1; r < W; ++r) a = std::max(a, all[r * n + ns + q]);
        maxs[q] = a;
    }
    st.rccl_reductions++;
}

}  // namespace proxsdp
