// Host-side one-shot preparation of a solve (no HIP calls here):
//   validate + copy the borrowed problem, preprocess! (variable order, cones
//   first -- /root/reference/src/scaling.jl:2-26), norm_scaling (x sqrt(2)/2 on
//   off-diagonal PSD columns of A, G and on c -- scaling.jl:28-58), M = vcat(A,G)
//   (pdhg.jl:104) in both orientations, ||M||_F (pdhg.jl:121), data norms
//   (pdhg.jl:14-16), cone layout (util.jl:2-16).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>
#include "../../include/proxsdp_hip.h"

namespace proxsdp {

struct BlockInfo { int n; int64_t N; int64_t off; };
struct SocInfo { int64_t off; int len; };

struct Prep {
    int64_t n = 0, p = 0, m = 0, Q = 0, nnz = 0;
    std::vector<int64_t> ord;          // new position k holds original variable ord[k]
    std::vector<int64_t> inv;          // sortperm(ord): original variable i sits at inv[i]
    // M = [A;G] with reordered columns, CSC, scaled (val) and unscaled (val_orig)
    std::vector<int64_t> colptr;       // n+1
    std::vector<int32_t> rowidx;       // nnz, rows of G offset by p
    std::vector<double> val, val_orig;
    // CSR of the scaled M
    std::vector<int64_t> rowptr;       // Q+1
    std::vector<int32_t> colidx;
    std::vector<double> rval;
    std::vector<double> b, h, c, c_orig;   // b, h: as used in the loop (row-scaled when equilibrated); c: reordered+scaled; c_orig: reordered, unscaled
    std::vector<double> b_orig, h_orig;    // the caller's b, h (exit path: slacks)
    // equilibration (pdhg.jl:64-92, equilibration.jl): M <- E M D, [b;h] <- E [b;h], c <- D c
    bool equilibrated = false;
    std::vector<double> Ediag, Ddiag;
    std::vector<uint8_t> offdiag;          // per new position: 1 if off-diagonal PSD entry
    std::vector<BlockInfo> blocks;
    std::vector<SocInfo> socs;
    int64_t sdplen = 0, conelen = 0;
    double norm_b = 0, norm_h = 0, norm_c = 0, frob = 0;
    // dense A (borrowed pointer, row-major p x n): the sparse members above then hold G only
    // and the solver adds the dense part to frob (one device pass)
    const double* Mdense = nullptr;
    bool Mdense_on_device = false;
    bool dense() const { return Mdense != nullptr; }
};

inline double norm2(const double* v, int64_t n) {
    double s = 0.0;
    for (int64_t i = 0; i < n; ++i) s += v[i] * v[i];
    return std::sqrt(s);
}

inline void check_csc(const proxsdp_csc& M, int64_t rows, int64_t cols, int base, const char* name) {
    if (M.nrows != rows || M.ncols != cols)
        throw std::invalid_argument(std::string(name) + ": shape mismatch");
    if (cols > 0 && M.colptr == nullptr) throw std::invalid_argument(std::string(name) + ": colptr is NULL");
    if (cols == 0) return;
    if (M.colptr[0] != base) throw std::invalid_argument(std::string(name) + ": colptr[0] != index_base");
    for (int64_t j = 0; j < cols; ++j)
        if (M.colptr[j + 1] < M.colptr[j]) throw std::invalid_argument(std::string(name) + ": colptr not monotone");
    int64_t nnz = M.colptr[cols] - base;
    if (nnz > 0 && (M.rowval == nullptr || M.nzval == nullptr))
        throw std::invalid_argument(std::string(name) + ": rowval/nzval is NULL");
    for (int64_t k = 0; k < nnz; ++k) {
        int64_t r = M.rowval[k] - base;
        if (r < 0 || r >= rows) throw std::invalid_argument(std::string(name) + ": row index out of range");
    }
}

// equilibrate! (equilibration.jl:1-72) on the reordered, unscaled M (CSC).  The reference replaces
// v by its mean in every iteration (:56-58), so D comes out as a multiple of the identity: kept.
// The reference's `E = Diagonal(u)` / `D = Diagonal(v)` (:16-17) wrap u and v WITHOUT copying, so
// `E.diag .= exp.(u)` / `D.diag .= exp.(v)` (:25-26) overwrite u by exp(u) and v by exp(v) at the top of
// every iteration, and gradient, step and running average continue from those values.
// options.equilibration_reference_aliasing = 1 (default) restates exactly that arithmetic; 0 = the
// iteration the code evidently intends (u, v kept; E = exp(u), D = exp(v)) -- rounds 1-4 behaviour.
// Same switch in oracle/pdhg.py:equilibrate.
inline void equilibrate_host(const Prep& R, const proxsdp_options& opt, std::vector<double>& Ed, std::vector<double>& Dd) {
    const int64_t nQ = R.Q, n = R.n;
    const double alpha = std::pow((double)n / (double)nQ, 0.25), beta = std::pow((double)nQ / (double)n, 0.25);
    const double alpha2 = alpha * alpha, beta2 = beta * beta, gamma = 0.1;
    std::vector<double> u(nQ, 0.0), v(n, 0.0), u_(nQ, 0.0), v_(n, 0.0), rn(nQ), cn(n);
    Ed.assign(nQ, 1.0); Dd.assign(n, 1.0);
    const bool alias = opt.equilibration_reference_aliasing != 0;
    for (int64_t it = 1; it <= opt.equilibration_iters; ++it) {
        for (int64_t r = 0; r < nQ; ++r) Ed[r] = std::exp(u[r]);
        for (int64_t k = 0; k < n; ++k) Dd[k] = std::exp(v[k]);
        if (alias) { u = Ed; v = Dd; }                    // E.diag === u, D.diag === v (equilibration.jl:16-17,25-26)
        std::fill(rn.begin(), rn.end(), 0.0);
        for (int64_t k = 0; k < n; ++k) {
            double cs = 0.0;
            for (int64_t q = R.colptr[k]; q < R.colptr[k + 1]; ++q) {
                const double m = (R.val_orig[q] * Dd[k]) * Ed[R.rowidx[q]];
                rn[R.rowidx[q]] += m * m;
                cs += m * m;
            }
            cn[k] = cs;
        }
        const double step = 2.0 / (gamma * ((double)it + 1.0));
        for (int64_t r = 0; r < nQ; ++r) {
            const double g = rn[r] - alpha2 + gamma * u[r];
            u[r] = std::min(opt.equilibration_ub, std::max(u[r] - step * g, opt.equilibration_lb));
        }
        double sv = 0.0;
        for (int64_t k = 0; k < n; ++k) { v[k] -= step * (cn[k] - beta2 + gamma * v[k]); sv += v[k]; }
        const double vm = std::min(opt.equilibration_ub, std::max(sv / (double)n, 0.0));
        for (int64_t k = 0; k < n; ++k) v[k] = vm;
        const double a = 2.0 / ((double)it + 2.0), b = (double)it / ((double)it + 2.0);
        for (int64_t r = 0; r < nQ; ++r) u_[r] = a * u[r] + b * u_[r];
        for (int64_t k = 0; k < n; ++k) v_[k] = a * v[k] + b * v_[k];
    }
    for (int64_t r = 0; r < nQ; ++r) Ed[r] = std::exp(u_[r]);
    for (int64_t k = 0; k < n; ++k) Dd[k] = std::exp(v_[k]);
}

inline Prep prepare(const proxsdp_problem& P, const proxsdp_options* opt = nullptr) {
    Prep R;
    const int base = P.index_base;
    if (base != 0 && base != 1) throw std::invalid_argument("index_base must be 0 or 1");
    if (P.n < 0 || P.p < 0 || P.m < 0) throw std::invalid_argument("negative dimension");
    if (P.n >= (int64_t)1 << 31) throw std::invalid_argument("n >= 2^31 not supported");
    // row indices, row pointers and the y-space kernels use int32 (nnz is range-checked below)
    if (P.p + P.m >= (int64_t)1 << 31) throw std::invalid_argument("p + m >= 2^31 not supported");
    R.n = P.n; R.p = P.p; R.m = P.m; R.Q = P.p + P.m;
    const bool dense = P.M_dense != nullptr;
    if (!dense) check_csc(P.A, P.p, P.n, base, "A");
    else if (P.reduce_fn != nullptr || P.nccl_comm != nullptr)
        throw std::invalid_argument("A_dense cannot be combined with a block-sharded solve (reduce_fn / nccl_comm)");
    check_csc(P.G, P.m, P.n, base, "G");
    if ((P.p > 0 && !P.b) || (P.m > 0 && !P.h) || (P.n > 0 && !P.c))
        throw std::invalid_argument("b, h or c is NULL");
    if (P.n_psd < 0 || P.n_soc < 0) throw std::invalid_argument("negative cone count");
    if (P.n_psd > 0 && (!P.psd_ptr || !P.psd_idx)) throw std::invalid_argument("psd_ptr/psd_idx is NULL");
    if (P.n_soc > 0 && (!P.soc_ptr || !P.soc_idx)) throw std::invalid_argument("soc_ptr/soc_idx is NULL");

    // ---- preprocess!: cones first (PSD blocks in cone order, then SOC), then the rest sorted
    std::vector<uint8_t> used(P.n, 0);
    R.ord.reserve(P.n);
    R.offdiag.assign(P.n, 0);
    int64_t pos = 0;
    for (int64_t k = 0; k < P.n_psd; ++k) {
        int64_t len = P.psd_ptr[k + 1] - P.psd_ptr[k];
        if (len <= 0) throw std::invalid_argument("empty PSD cone");
        int64_t side = (int64_t)((std::sqrt(8.0 * (double)len + 1.0) - 1.0) / 2.0);
        while (side * (side + 1) / 2 < len) ++side;
        while (side * (side + 1) / 2 > len) --side;
        if (side * (side + 1) / 2 != len) throw std::invalid_argument("PSD cone length is not triangular");
        if (side > 46340) throw std::invalid_argument("PSD side too large");
        R.blocks.push_back({(int)side, len, pos});
        int64_t q = P.psd_ptr[k];
        for (int64_t j = 0; j < side; ++j)
            for (int64_t i = 0; i <= j; ++i, ++q) {
                int64_t v = P.psd_idx[q] - base;
                if (v < 0 || v >= P.n || used[v]) throw std::invalid_argument("PSD cone variable out of range or repeated");
                used[v] = 1;
                R.ord.push_back(v);
                R.offdiag[pos++] = (i != j);
            }
    }
    R.sdplen = pos;
    for (int64_t k = 0; k < P.n_soc; ++k) {
        int64_t len = P.soc_ptr[k + 1] - P.soc_ptr[k];
        if (len <= 0) throw std::invalid_argument("empty SOC cone");
        R.socs.push_back({pos, (int)len});
        for (int64_t q = P.soc_ptr[k]; q < P.soc_ptr[k + 1]; ++q) {
            int64_t v = P.soc_idx[q] - base;
            if (v < 0 || v >= P.n || used[v]) throw std::invalid_argument("SOC variable out of range or repeated");
            used[v] = 1;
            R.ord.push_back(v);
            ++pos;
        }
    }
    R.conelen = pos;
    for (int64_t v = 0; v < P.n; ++v)
        if (!used[v]) R.ord.push_back(v);
    R.inv.assign(P.n, 0);
    for (int64_t k = 0; k < P.n; ++k) R.inv[R.ord[k]] = k;

    // ---- vectors
    R.b.assign(P.b, P.b + P.p);
    R.h.assign(P.h, P.h + P.m);
    R.b_orig = R.b; R.h_orig = R.h;
    R.norm_b = norm2(P.b, P.p);
    R.norm_h = norm2(P.h, P.m);
    R.norm_c = norm2(P.c, P.n);
    const double cte = std::sqrt(2.0) / 2.0;
    R.c.resize(P.n); R.c_orig.resize(P.n);
    for (int64_t k = 0; k < P.n; ++k) {
        double v = P.c[R.ord[k]];
        R.c_orig[k] = v;
        R.c[k] = R.offdiag[k] ? v * cte : v;
    }

    if (dense) {
        for (int64_t k = 0; k < P.n; ++k)
            if (R.ord[k] != k)
                throw std::invalid_argument("A_dense requires the variables in solver order (cone variables first, in cone order)");
        R.Mdense = P.M_dense;
        R.Mdense_on_device = P.M_dense_on_device != 0;
    }
    // ---- M = vcat(A, G) with reordered columns (a dense A stays out of the sparse structures)
    int64_t nnzA = (P.n > 0 && !dense) ? P.A.colptr[P.n] - base : 0;
    int64_t nnzG = P.n > 0 ? P.G.colptr[P.n] - base : 0;
    R.nnz = nnzA + nnzG;
    if (R.nnz >= ((int64_t)1 << 31) - 1) throw std::invalid_argument("nnz(M) >= 2^31 not supported by the sparse path");
    R.colptr.assign(P.n + 1, 0);
    R.rowidx.resize(R.nnz); R.val.resize(R.nnz); R.val_orig.resize(R.nnz);
    int64_t w = 0;
    for (int64_t k = 0; k < P.n; ++k) {
        const int64_t j = R.ord[k];
        R.colptr[k] = w;
        for (int64_t q = dense ? 0 : P.A.colptr[j] - base; q < (dense ? 0 : P.A.colptr[j + 1] - base); ++q, ++w) {
            R.rowidx[w] = (int32_t)(P.A.rowval[q] - base);
            R.val_orig[w] = P.A.nzval[q];
        }
        for (int64_t q = P.G.colptr[j] - base; q < P.G.colptr[j + 1] - base; ++q, ++w) {
            R.rowidx[w] = (int32_t)(P.G.rowval[q] - base + P.p);
            R.val_orig[w] = P.G.nzval[q];
        }
    }
    R.colptr[P.n] = w;
    // ---- diagonal preconditioning (pdhg.jl:64-92): only ever active when forced, or when the
    // smallest and largest entries of M (implicit zeros included) are within `equilibration_limit`
    bool equil = opt != nullptr && opt->equilibration != 0;
    if (equil) {
        double hi = 0.0, lo = 0.0;
        if (R.nnz > 0) {
            hi = lo = R.val_orig[0];
            for (double v : R.val_orig) { hi = std::max(hi, v); lo = std::min(lo, v); }
            if (R.nnz < R.Q * R.n) { hi = std::max(hi, 0.0); lo = std::min(lo, 0.0); }
        }
        if (hi == 0.0 || lo / hi <= opt->equilibration_limit) equil = false;
    }
    if (opt != nullptr && opt->equilibration_force) equil = true;
    if (equil) {
        if (dense || P.reduce_fn != nullptr || P.nccl_comm != nullptr) throw std::domain_error("equilibration with a dense A or a block-sharded solve is not implemented");
        if (R.Q == 0 || R.n == 0) throw std::invalid_argument("equilibration needs a non-empty M");
        equilibrate_host(R, *opt, R.Ediag, R.Ddiag);
        R.equilibrated = true;
        for (int64_t i = 0; i < R.p; ++i) R.b[i] *= R.Ediag[i];
        for (int64_t i = 0; i < R.m; ++i) R.h[i] *= R.Ediag[R.p + i];
        for (int64_t k = 0; k < P.n; ++k) R.c[k] *= R.Ddiag[k];       // (c already carries the sqrt(2)/2 factor: it commutes)
    }
    double ss = 0.0;
    for (int64_t k = 0; k < P.n; ++k) {
        const double sc = R.offdiag[k] ? cte : 1.0;
        for (int64_t q = R.colptr[k]; q < R.colptr[k + 1]; ++q) {
            double v = R.val_orig[q];
            if (equil) v = R.Ediag[R.rowidx[q]] * v * R.Ddiag[k];
            R.val[q] = v * sc;
            ss += R.val[q] * R.val[q];
        }
    }
    R.frob = std::sqrt(ss);

    // ---- CSR of the scaled M (counting sort; column order inside a row ascending)
    R.rowptr.assign(R.Q + 1, 0);
    for (int64_t q = 0; q < R.nnz; ++q) R.rowptr[R.rowidx[q] + 1]++;
    for (int64_t r = 0; r < R.Q; ++r) R.rowptr[r + 1] += R.rowptr[r];
    R.colidx.resize(R.nnz); R.rval.resize(R.nnz);
    std::vector<int64_t> cur(R.rowptr.begin(), R.rowptr.end() - 1);
    for (int64_t k = 0; k < P.n; ++k)
        for (int64_t q = R.colptr[k]; q < R.colptr[k + 1]; ++q) {
            int64_t d = cur[R.rowidx[q]]++;
            R.colidx[d] = (int32_t)k;
            R.rval[d] = R.val[q];
        }
    return R;
}

// sigma_max of the scaled M (approx_norm = false; the reference calls Arpack.svds(M, nsv=1),
// pdhg.jl:108-119): Lanczos with full re-orthogonalisation on M'M from a fixed start vector,
// stopped when the Ritz residual is below 1e-14 relative.  `symeig` = host_util's symeig_dense.
template <typename SymEig>
inline double spectral_norm_host(const Prep& R, SymEig symeig) {
    const int64_t n = R.n, Q = R.Q;
    if (n == 0 || Q == 0 || R.nnz == 0) return 0.0;
    const int kmax = (int)std::min<int64_t>(std::min(n, Q), 120);
    std::vector<std::vector<double>> V;
    std::vector<double> al, be, v(n, 1.0 / std::sqrt((double)n)), u(Q), w(n);
    double sigma2 = 0.0;
    for (int k = 0; k < kmax; ++k) {
        V.push_back(v);
        std::fill(u.begin(), u.end(), 0.0);
        for (int64_t c = 0; c < n; ++c)
            for (int64_t q = R.colptr[c]; q < R.colptr[c + 1]; ++q) u[R.rowidx[q]] += R.val[q] * v[c];
        for (int64_t c = 0; c < n; ++c) {
            double acc = 0.0;
            for (int64_t q = R.colptr[c]; q < R.colptr[c + 1]; ++q) acc += R.val[q] * u[R.rowidx[q]];
            w[c] = acc;
        }
        double a = 0.0;
        for (int64_t c = 0; c < n; ++c) a += w[c] * v[c];
        al.push_back(a);
        for (int pass = 0; pass < 2; ++pass)
            for (const std::vector<double>& vj : V) {
                double h = 0.0;
                for (int64_t c = 0; c < n; ++c) h += vj[c] * w[c];
                for (int64_t c = 0; c < n; ++c) w[c] -= h * vj[c];
            }
        double b = 0.0;
        for (int64_t c = 0; c < n; ++c) b += w[c] * w[c];
        b = std::sqrt(b);
        const int K = k + 1;
        std::vector<double> T((size_t)K * K, 0.0), d(K);
        for (int i = 0; i < K; ++i) { T[(size_t)i * K + i] = al[i]; if (i + 1 < K) T[(size_t)i * K + i + 1] = T[(size_t)(i + 1) * K + i] = be[i]; }
        symeig(K, T.data(), d.data());
        sigma2 = d[K - 1];
        const double resid = std::fabs(b * T[(size_t)(K - 1) * K + (K - 1)]);     // beta * last component of the top Ritz vector
        if (b <= 1e-14 * std::max(1.0, std::fabs(sigma2)) || resid <= 1e-14 * std::fabs(sigma2)) break;
        be.push_back(b);
        for (int64_t c = 0; c < n; ++c) v[c] = w[c] / b;
    }
    return std::sqrt(std::max(sigma2, 0.0));
}

}  // namespace proxsdp
