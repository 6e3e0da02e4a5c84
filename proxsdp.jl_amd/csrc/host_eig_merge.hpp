// Rank-one merge of two symmetric eigen-decompositions (one level of Cuppen's divide and conquer, with
// Gu-Eisenstat's stable eigenvectors) -- host only, no HIP.
//
// Why: the thick-restart Lanczos (KrylovKit's eigsolve, /root/reference/src/eigsolver.jl:802-812) needs the
// eigen-decomposition of its K x K Rayleigh quotient at the end of every cycle, with the GPU waiting.  The
// implicit-QL solver (host_util.hpp) is a scalar sqrt/divide chain of ~0.85 K^2 dependent rotations: 0.45 ms at
// K = 127, a quarter of the rank-63 iteration.  But most of that matrix is known long before the cycle ends:
//     T = [ T1  b e e' ;  b e e'  T2 ],    T1 = the arrow part of the restart plus the first new step
//                                               (or the first half of the tridiagonal in the first cycle)
// so T1' = T1 - |b| e e' is decomposed WHILE THE GPU RUNS THE REST OF THE CYCLE, and only T2' (the small tail)
// and the merge  T = Q (D + rho z z') Q'  are left on the critical path: a secular equation per eigenvalue
// (independent, vectorisable sums) and one small GEMM for the eigenvectors actually needed.
//
// Algorithm (LAPACK dlaed1/2/3/4 is the model; written from the published method, not transcribed):
//   1. z = Q' [e_k1 ; sgn(b) e_1] / sqrt 2, rho = 2 |b|;  poles d = merged, sorted eigenvalues of T1', T2'
//   2. deflation: rho |z_j| <= tol  ->  (d_j, q_j) is an eigenpair as it stands; two close poles are rotated
//      so that one of them gets z = 0 (Givens on the two eigenvector columns)
//   3. secular equation 1 + rho sum_j z_j^2 / (d_j - lambda) = 0 on the non-deflated poles: one root per
//      interval, found from the nearer pole as origin (so that d_j - lambda is formed without cancellation)
//      by rational interpolation on the two neighbouring poles ("middle way"), safeguarded by bisection
//   4. z is RECOMPUTED from the roots (Loewner / Gu-Eisenstat): the eigenvectors
//      s_i = (zhat_j / (d_j - lambda_i))_j are then orthogonal to working precision whatever the accuracy of
//      the individual roots
//   5. eigenvectors of T: Q[:, nondeflated] S[:, wanted] (block structure of Q kept: top / bottom row panels)
#pragma once
#include <algorithm>
#include <cmath>
#include <functional>
#include <limits>
#include <vector>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace proxsdp {

#if defined(__x86_64__)
// 8 rows x 4 columns of  U[r0+r.., c] = sum_jj Q[:, qcol[jj]] w_c[jj]  with the accumulators in registers.
// (FMA is switched on for this one routine: the rest of the host code is built with contraction off.)
__attribute__((target("avx2,fma")))
static inline void rank1_panel_8x4(const double* Qf, int K, const int* qcol, int kk, int row, const double* w0,
                                   const double* w1, const double* w2, const double* w3, double* u0, double* u1,
                                   double* u2, double* u3) {
    __m256d a00 = _mm256_setzero_pd(), a10 = a00, a01 = a00, a11 = a00, a02 = a00, a12 = a00, a03 = a00, a13 = a00;
    for (int jj = 0; jj < kk; ++jj) {
        const double* qc = Qf + (size_t)qcol[jj] * K + row;
        const __m256d q0 = _mm256_loadu_pd(qc), q1 = _mm256_loadu_pd(qc + 4);
        __m256d w = _mm256_broadcast_sd(w0 + jj);
        a00 = _mm256_fmadd_pd(q0, w, a00); a10 = _mm256_fmadd_pd(q1, w, a10);
        w = _mm256_broadcast_sd(w1 + jj);
        a01 = _mm256_fmadd_pd(q0, w, a01); a11 = _mm256_fmadd_pd(q1, w, a11);
        w = _mm256_broadcast_sd(w2 + jj);
        a02 = _mm256_fmadd_pd(q0, w, a02); a12 = _mm256_fmadd_pd(q1, w, a12);
        w = _mm256_broadcast_sd(w3 + jj);
        a03 = _mm256_fmadd_pd(q0, w, a03); a13 = _mm256_fmadd_pd(q1, w, a13);
    }
    if (u0) { _mm256_storeu_pd(u0, a00); _mm256_storeu_pd(u0 + 4, a10); }
    if (u1) { _mm256_storeu_pd(u1, a01); _mm256_storeu_pd(u1 + 4, a11); }
    if (u2) { _mm256_storeu_pd(u2, a02); _mm256_storeu_pd(u2 + 4, a12); }
    if (u3) { _mm256_storeu_pd(u3, a03); _mm256_storeu_pd(u3 + 4, a13); }
}
#endif

// par(n, body): run body(0) .. body(n-1), possibly on helper threads (the driver passes its spin pool); nullptr = serial.
// Every root / column is computed by the same code whichever thread takes it: results do not depend on the runner.
using ParFor = std::function<void(int, const std::function<void(int)>&)>;

struct Rank1Merge {
    const ParFor* par = nullptr;           // set by the caller before build() / vectors()
    int nchunk = 4;                        // independent chunks handed to the runner
    int K = 0, k1 = 0, k2 = 0, k = 0;      // k = number of non-deflated poles
    double rho = 0.0;
    std::vector<double> Qf;                // K x K column-major: eigenvectors of blkdiag(T1', T2') in pole order (+ deflation rotations)
    std::vector<double> d, z;              // poles (sorted ascending) and z in that order
    std::vector<char> ctype;               // 1 = rows [0,k1) only, 2 = rows [k1,K) only, 3 = full
    std::vector<int> nd, df;               // non-deflated / deflated pole indices
    std::vector<double> dn, zn;            // the non-deflated poles and weights
    std::vector<double> lam;               // roots (k), ascending
    std::vector<double> delta;             // k x k: delta[i*k + j] = dn[j] - lam[i], formed from the root's origin
    std::vector<double> S;                 // k x k: column i (at S[i*k]) = normalised eigenvector of D + rho z z'
    std::vector<double> evals;             // K eigenvalues of T, ascending
    std::vector<int> src;                  // evals[c] comes from root src[c] >= 0, or from deflated pole -(src[c]+1)
    int max_iter_seen = 0;

    // ---- secular function on the non-deflated poles, evaluated at lambda = dn[org] + tau
    // psi: poles 0..isplit, phi: poles isplit+1..k-1 (with their derivatives)
    inline void eval(int org, double tau, int isplit, double& psi, double& dpsi, double& phi, double& dphi) const {
        const double dorg = dn[org];
        double a = 0.0, da = 0.0, b = 0.0, db = 0.0;
        for (int j = 0; j <= isplit; ++j) {
            const double dl = (dn[j] - dorg) - tau;
            const double t = zn[j] / dl;
            a += zn[j] * t; da += t * t;
        }
        for (int j = isplit + 1; j < k; ++j) {
            const double dl = (dn[j] - dorg) - tau;
            const double t = zn[j] / dl;
            b += zn[j] * t; db += t * t;
        }
        psi = rho * a; dpsi = rho * da; phi = rho * b; dphi = rho * db;
    }

    // root i of the secular equation; returns origin and tau (lambda = dn[org] + tau)
    void solve_root(int i, int& org_out, double& tau_out, int& iters_out) const {
        const double eps = 2.220446049250313e-16;
        iters_out = 0;
        if (k == 1) { org_out = 0; tau_out = rho * zn[0] * zn[0]; return; }
        const bool last = (i == k - 1);
        const int ia = last ? k - 2 : i, ib = ia + 1;        // the two poles kept exactly by the interpolation
        const int isplit = ia;                               // psi = poles <= ia, phi = poles >= ib
        double lo, hi;                                       // bracket of tau (relative to the chosen origin)
        int org;
        double psi, dpsi, phi, dphi;
        if (!last) {
            const double gap = dn[i + 1] - dn[i];
            eval(i, 0.5 * gap, isplit, psi, dpsi, phi, dphi);
            const double fmid = 1.0 + psi + phi;
            if (fmid >= 0.0) { org = i; lo = 0.0; hi = 0.5 * gap; }          // root in (d_i, mid]
            else { org = i + 1; lo = -0.5 * gap; hi = 0.0; }                 // root in (mid, d_{i+1})
        } else {
            double zz = 0.0;
            for (int j = 0; j < k; ++j) zz += zn[j] * zn[j];
            org = k - 1; lo = 0.0; hi = rho * zz;                            // lambda_max <= d_max + rho |z|^2
        }
        double tau = 0.5 * (lo + hi);                         // start in the middle of the bracket
        double f = 0.0;
        int it = 0;
        for (; it < 80; ++it) {
            if (!(tau > lo && tau < hi)) tau = 0.5 * (lo + hi);
            // the open ends of the bracket are poles: never evaluate exactly there
            if (tau == lo || tau == hi) break;
            eval(org, tau, isplit, psi, dpsi, phi, dphi);
            f = 1.0 + psi + phi;
            const double erretm = 8.0 * (std::fabs(psi) + std::fabs(phi)) + 2.0 + std::fabs(tau) * (dpsi + dphi);
            if (std::fabs(f) <= eps * erretm) break;
            if (f < 0.0) lo = tau; else hi = tau;
            if (hi - lo <= 2.0 * eps * std::max(std::fabs(lo), std::fabs(hi))) { tau = 0.5 * (lo + hi); break; }
            // middle way: psi ~ r + s / (da - lambda), phi ~ R + S / (db - lambda), value and slope matched at tau
            const double Da = (dn[ia] - dn[org]) - tau, Db = (dn[ib] - dn[org]) - tau;
            // (last root: both kept poles lie to the left of it; phi is then the single pole ib, represented exactly)
            const double s = dpsi * Da * Da, Sb = dphi * Db * Db;
            const double c = f - dpsi * Da - dphi * Db;
            const double bq = c * (Da + Db) + s + Sb;
            const double w = Da * Db * f;
            double eta;
            if (c == 0.0) {
                eta = (bq != 0.0) ? w / bq : 0.0;
            } else {
                double disc = bq * bq - 4.0 * c * w;
                if (disc < 0.0) disc = 0.0;
                const double sq = std::sqrt(disc);
                const double q = (bq >= 0.0) ? 0.5 * (bq + sq) : 0.5 * (bq - sq);
                const double e1 = (q != 0.0) ? w / q : 0.0;          // small root
                const double e2 = q / c;                              // large root
                const double t1 = tau + e1, t2 = tau + e2;
                const bool in1 = (t1 > lo && t1 < hi), in2 = (t2 > lo && t2 < hi);
                if (in1 && in2) eta = (std::fabs(e1) <= std::fabs(e2)) ? e1 : e2;
                else if (in1) eta = e1;
                else if (in2) eta = e2;
                else eta = 0.5 * (lo + hi) - tau;
            }
            double tn = tau + eta;
            if (!(tn > lo && tn < hi) || !(tn == tn)) tn = 0.5 * (lo + hi);
            if (tn == tau) break;
            tau = tn;
        }
        iters_out = it;
        org_out = org; tau_out = tau;
    }

    // Build the merged problem.  Q1 (k1 x k1), Q2 (k2 x k2) column-major eigenvectors of T1', T2' (d1, d2 ascending);
    // beta = the coupling entry T[k1-1, k1] (T1' and T2' already carry the -|beta| corrections on their corner entries).
    void build(int k1_, int k2_, const double* Q1, const double* d1, const double* Q2, const double* d2, double beta) {
        const double eps = 2.220446049250313e-16;
        k1 = k1_; k2 = k2_; K = k1 + k2;
        const double sgn = beta < 0.0 ? -1.0 : 1.0;
        rho = 2.0 * std::fabs(beta);
        const double isq2 = 0.70710678118654752440;
        // ---- merge the two sorted pole lists
        d.assign(K, 0.0); z.assign(K, 0.0); ctype.assign(K, 0);
        Qf.assign((size_t)K * K, 0.0);
        {
            int a = 0, b = 0;
            for (int p = 0; p < K; ++p) {
                const bool takeA = (b >= k2) || (a < k1 && d1[a] <= d2[b]);
                double* col = Qf.data() + (size_t)p * K;
                if (takeA) {
                    d[p] = d1[a]; z[p] = Q1[(size_t)a * k1 + (k1 - 1)] * isq2; ctype[p] = 1;
                    for (int r = 0; r < k1; ++r) col[r] = Q1[(size_t)a * k1 + r];
                    ++a;
                } else {
                    d[p] = d2[b]; z[p] = sgn * Q2[(size_t)b * k2] * isq2; ctype[p] = 2;
                    for (int r = 0; r < k2; ++r) col[k1 + r] = Q2[(size_t)b * k2 + r];
                    ++b;
                }
            }
        }
        // ---- deflation
        double zmax = 0.0, dmax = 0.0;
        for (int p = 0; p < K; ++p) { zmax = std::max(zmax, std::fabs(z[p])); dmax = std::max(dmax, std::fabs(d[p])); }
        const double tol = 8.0 * eps * std::max(dmax, zmax);
        nd.clear(); df.clear();
        if (rho * zmax <= tol) {
            for (int p = 0; p < K; ++p) df.push_back(p);
        } else {
            int pj = -1;
            for (int j = 0; j < K; ++j) {
                if (rho * std::fabs(z[j]) <= tol) { df.push_back(j); continue; }
                if (pj < 0) { pj = j; continue; }
                double s = z[pj], c = z[j];
                const double tau = std::sqrt(c * c + s * s);
                const double t = d[j] - d[pj];
                c /= tau; s = -s / tau;
                if (std::fabs(t * c * s) <= tol) {
                    z[j] = tau; z[pj] = 0.0;
                    double* __restrict__ x = Qf.data() + (size_t)pj * K;
                    double* __restrict__ y = Qf.data() + (size_t)j * K;
                    for (int r = 0; r < K; ++r) {
                        const double xr = x[r], yr = y[r];
                        x[r] = c * xr + s * yr;
                        y[r] = c * yr - s * xr;
                    }
                    const double tt = d[pj] * c * c + d[j] * s * s;
                    d[j] = d[pj] * s * s + d[j] * c * c;
                    d[pj] = tt;
                    if (ctype[pj] != ctype[j]) ctype[pj] = ctype[j] = 3;
                    df.push_back(pj);
                    pj = j;
                } else {
                    nd.push_back(pj);
                    pj = j;
                }
            }
            if (pj >= 0) nd.push_back(pj);
        }
        k = (int)nd.size();
        dn.resize(k); zn.resize(k);
        for (int j = 0; j < k; ++j) { dn[j] = d[nd[j]]; zn[j] = z[nd[j]]; }
        // (a rotation can leave d[pj] marginally above the next pole: keep the non-deflated poles increasing)
        for (int j = 1; j < k; ++j)
            if (!(dn[j] > dn[j - 1])) dn[j] = std::nextafter(dn[j - 1], 1e300);
        // ---- secular equation: roots and the differences dn[j] - lam[i]
        lam.assign(k, 0.0);
        delta.assign((size_t)k * k, 0.0);
        max_iter_seen = 0;
        // the three O(k^2) phases below are independent per root / per pole: chunks of them go to the runner
        const int nc = (par != nullptr && k >= 32) ? std::max(1, nchunk) : 1;
        auto chunks = [&](const std::function<void(int, int)>& body) {          // body(lo, hi) over [0, k) in nc pieces
            if (nc == 1) { body(0, k); return; }
            const std::function<void(int)> one = [&](int c) { body((int)((long long)k * c / nc), (int)((long long)k * (c + 1) / nc)); };
            (*par)(nc, one);
        };
        std::vector<int> iters(std::max(k, 1), 0);
        chunks([&](int lo, int hi) {
            for (int i = lo; i < hi; ++i) {
                int org; double tau;
                solve_root(i, org, tau, iters[i]);
                lam[i] = dn[org] + tau;
                double* dl = delta.data() + (size_t)i * k;
                for (int j = 0; j < k; ++j) dl[j] = (dn[j] - dn[org]) - tau;
                // interlacing is what makes the Loewner formula below positive: enforce it against rounding
                if (dl[i] >= 0.0) dl[i] = -std::numeric_limits<double>::min();
                if (i + 1 < k && dl[i + 1] <= 0.0) dl[i + 1] = std::numeric_limits<double>::min();
            }
        });
        for (int i = 0; i < k; ++i) max_iter_seen = std::max(max_iter_seen, iters[i]);
        // ---- Gu-Eisenstat: the z for which lam are the EXACT eigenvalues
        std::vector<double> zhat(k, 0.0);
        chunks([&](int lo, int hi) {
            for (int j = lo; j < hi; ++j) {
                // prod_i (lam_i - dn_j) / prod_{i != j} (dn_i - dn_j), paired so that every ratio is positive
                double prod = -delta[(size_t)(k - 1) * k + j];                  // lam_{k-1} - dn_j > 0
                for (int i = 0; i < j; ++i) prod *= delta[(size_t)i * k + j] / (dn[j] - dn[i]);             // (dn_j - lam_i)/(dn_j - dn_i)
                for (int i = j; i < k - 1; ++i) prod *= delta[(size_t)i * k + j] / (dn[j] - dn[i + 1]);     // (dn_j - lam_i)/(dn_j - dn_{i+1}), both negative
                const double v = std::sqrt(std::fabs(prod) / rho);
                zhat[j] = zn[j] < 0.0 ? -v : v;
            }
        });
        // ---- eigenvectors of D + rho z z'
        S.assign((size_t)k * k, 0.0);
        chunks([&](int lo, int hi) {
            for (int i = lo; i < hi; ++i) {
                double* s = S.data() + (size_t)i * k;
                const double* dl = delta.data() + (size_t)i * k;
                double nn = 0.0;
                for (int j = 0; j < k; ++j) { s[j] = zhat[j] / dl[j]; nn += s[j] * s[j]; }
                const double inv = 1.0 / std::sqrt(nn);
                for (int j = 0; j < k; ++j) s[j] *= inv;
            }
        });
        // ---- all eigenvalues of T, ascending, with their sources
        std::vector<std::pair<double, int>> all;
        all.reserve(K);
        for (int i = 0; i < k; ++i) all.emplace_back(lam[i], i);
        for (int p : df) all.emplace_back(d[p], -(p + 1));
        std::stable_sort(all.begin(), all.end(), [](const std::pair<double, int>& a, const std::pair<double, int>& b) { return a.first < b.first; });
        evals.resize(K); src.resize(K);
        for (int c = 0; c < K; ++c) { evals[c] = all[c].first; src[c] = all[c].second; }
    }

    // entry `row` of every eigenvector of T (index = position in evals)
    void row_of_vectors(int row, double* out) const {
        std::vector<double> qrow(k);
        for (int j = 0; j < k; ++j) qrow[j] = Qf[(size_t)nd[j] * K + row];
        for (int c = 0; c < K; ++c) {
            if (src[c] >= 0) {
                const double* s = S.data() + (size_t)src[c] * k;
                double a = 0.0;
                for (int j = 0; j < k; ++j) a += qrow[j] * s[j];
                out[c] = a;
            } else {
                out[c] = Qf[(size_t)(-(src[c] + 1)) * K + row];
            }
        }
    }

    // eigenvectors of T for the listed positions of evals: U (K x ncols, column-major)
    void vectors(const int* cols, int ncols, double* U) const {
        // gather the non-deflated columns that touch the top / bottom row panel
        std::vector<int> top, bot;
        for (int j = 0; j < k; ++j) {
            const char t = ctype[nd[j]];
            if (t == 1 || t == 3) top.push_back(j);
            if (t == 2 || t == 3) bot.push_back(j);
        }
        // register-blocked panel product: 8 rows x 4 output columns per inner loop (accumulators stay in registers,
        // every column of Q is read once per 4 columns of U)
        auto panel = [&](const std::vector<int>& idx, int r0, int nr) {
            const int kk = (int)idx.size();
            if (kk == 0 || nr <= 0) return;
            std::vector<int> qcol(kk);
            for (int jj = 0; jj < kk; ++jj) qcol[jj] = nd[idx[jj]];
            const int nblk = (ncols + 3) / 4;
            // blocks of four output columns are independent: pieces of them go to the runner (each with its own scratch)
            auto blocks = [&](int b0, int b1) {
            std::vector<double> wbuf((size_t)4 * kk);
            for (int c0 = 4 * b0; c0 < std::min(ncols, 4 * b1); c0 += 4) {
                const int cb = std::min(4, ncols - c0);
                double* u[4] = {nullptr, nullptr, nullptr, nullptr};
                for (int q = 0; q < 4; ++q) {
                    double* wq = wbuf.data() + (size_t)q * kk;
                    if (q < cb) {
                        u[q] = U + (size_t)(c0 + q) * K + r0;
                        const int sc = src[cols[c0 + q]];
                        if (sc >= 0) {
                            const double* sp = S.data() + (size_t)sc * k;
                            for (int jj = 0; jj < kk; ++jj) wq[jj] = sp[idx[jj]];
                        } else {
                            for (int jj = 0; jj < kk; ++jj) wq[jj] = 0.0;
                        }
                    } else {
                        for (int jj = 0; jj < kk; ++jj) wq[jj] = 0.0;
                    }
                }
                const double* w0 = wbuf.data(); const double* w1 = w0 + kk; const double* w2 = w1 + kk; const double* w3 = w2 + kk;
                int r = 0;
#if defined(__x86_64__)
                // (the panel kernel needs AVX2 + FMA: checked once at run time, the scalar loop below serves any other host)
                static const bool have_fma = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma");
                for (; have_fma && r + 8 <= nr; r += 8)
                    rank1_panel_8x4(Qf.data(), K, qcol.data(), kk, r0 + r, w0, w1, w2, w3, u[0] ? u[0] + r : nullptr,
                                    u[1] ? u[1] + r : nullptr, u[2] ? u[2] + r : nullptr, u[3] ? u[3] + r : nullptr);
#endif
                for (; r < nr; ++r) {
                    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
                    for (int jj = 0; jj < kk; ++jj) {
                        const double qv = Qf[(size_t)qcol[jj] * K + r0 + r];
                        a0 += qv * w0[jj]; a1 += qv * w1[jj]; a2 += qv * w2[jj]; a3 += qv * w3[jj];
                    }
                    if (u[0]) u[0][r] = a0;
                    if (u[1]) u[1][r] = a1;
                    if (u[2]) u[2][r] = a2;
                    if (u[3]) u[3][r] = a3;
                }
            }
            };
            const int nc = (par != nullptr && nblk >= 8) ? std::max(1, nchunk) : 1;
            if (nc == 1) blocks(0, nblk);
            else {
                const std::function<void(int)> one = [&](int c) { blocks((int)((long long)nblk * c / nc), (int)((long long)nblk * (c + 1) / nc)); };
                (*par)(nc, one);
            }
        };
        std::fill(U, U + (size_t)K * ncols, 0.0);
        panel(top, 0, k1);
        panel(bot, k1, k2);
        for (int c = 0; c < ncols; ++c) {
            const int sc = src[cols[c]];
            if (sc >= 0) continue;
            const double* qc = Qf.data() + (size_t)(-(sc + 1)) * K;
            double* u = U + (size_t)c * K;
            for (int r = 0; r < K; ++r) u[r] = qc[r];
        }
    }
};

}  // namespace proxsdp
