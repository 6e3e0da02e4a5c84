// Host-only helpers of libproxsdp_hip (no HIP calls in this header).
//
//  * start_vector     : replacement for eigsolver_update_resid!
//                       (/root/reference/src/eigsolver.jl:397-411); Julia's
//                       MersenneTwister stream cannot be reproduced outside Julia,
//                       so library and oracle (oracle/eig.py:start_vector) share a
//                       counter-based generator, bit-identical on both sides.
//  * symeig_dense     : eigen-decomposition of the K x K Rayleigh quotient of the
//                       thick-restart Lanczos (KrylovKit does this with LAPACK on
//                       the host as well): Householder tridiagonalisation followed
//                       by implicit-shift QL.
//  * option table     : name -> field map for RawOptimizerAttribute semantics
//                       (/root/reference/src/MOI_wrapper.jl:84-103).
#pragma once
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <cstddef>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>
#include <algorithm>
#if defined(__x86_64__)
#include <immintrin.h>
#endif
#include "../../include/proxsdp_hip.h"
#include "host_eig_merge.hpp"

namespace proxsdp {

// ------------------------------------------------------------------ start vector
inline uint64_t splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
inline double uniform53(uint64_t seed, uint64_t counter) {
    uint64_t key = seed ^ ((counter + 1ull) * 0xD1342543DE82EF95ull);
    return (double)(splitmix64(key) >> 11) * (1.0 / 9007199254740992.0);
}
// init 3: normalised Irwin-Hall(12)-6 "normal"; 2: uniform; 1: ones; else zeros
inline void start_vector(int64_t n, uint64_t seed, int init, double* out) {
    if (init == 3) {
        double ss = 0.0;
        for (int64_t i = 0; i < n; ++i) {
            double z = 0.0;
            for (int k = 0; k < 12; ++k) z = z + uniform53(seed, (uint64_t)i * 12ull + (uint64_t)k);
            z = z - 6.0;
            out[i] = z;
            ss = ss + z * z;                 // sequential, as np.cumsum in the oracle
        }
        double nrm = std::sqrt(ss);
        for (int64_t i = 0; i < n; ++i) out[i] = out[i] / nrm;
    } else if (init == 2) {
        for (int64_t i = 0; i < n; ++i) out[i] = uniform53(seed, (uint64_t)i * 12ull);
    } else if (init == 1) {
        for (int64_t i = 0; i < n; ++i) out[i] = 1.0;
    } else {
        for (int64_t i = 0; i < n; ++i) out[i] = 0.0;
    }
}

// ------------------------------------------------------------------ small symmetric eig
// a: column-major n x n symmetric (full storage); on exit columns are orthonormal
// eigenvectors, d ascending eigenvalues.  Returns 0, or 1 if QL failed to converge.
// `tridiagonal`: a is already tridiagonal (the first Lanczos cycle of every projection): the
// Householder reduction (about 40 % of the work) is skipped.
// (Measured: AVX-512 clones of this routine are slower than the AVX2 build at K ~ 50; half of
// the QL time is the scalar rotation set-up, hence plain sqrt instead of hypot below -- the
// Rayleigh-quotient entries are O(|X|), far from the overflow range hypot guards against.)
// Phase 1: Householder reduction of a (column-major, full symmetric storage) to tridiagonal form;
// on exit a holds the accumulated orthogonal transform Q (a_in = Q T Q'), d the diagonal and
// e[i] the coupling between i-1 and i (e[0] = 0).  The reflectors only ever act on the leading
// coordinates of a row, so for an arrow matrix [D f; f' c] the last coordinate is left alone:
// Q = blkdiag(Q~, 1) and d[n-1] = c -- which lets the Lanczos driver reduce the arrow part of a
// restarted Rayleigh quotient before the rest of it exists (see symeig_tridiag_from).
inline void householder_tridiag(int n, double* a, double* d, double* e) {
    auto V = [&](int i, int j) -> double& { return a[(size_t)j * n + i]; };
    if (n == 1) { d[0] = a[0]; e[0] = 0.0; a[0] = 1.0; return; }
    // ---- Householder reduction to tridiagonal form (accumulating the transform)
    for (int j = 0; j < n; ++j) d[j] = V(n - 1, j);
    for (int i = n - 1; i > 0; --i) {
        double scale = 0.0, h = 0.0;
        for (int k = 0; k < i; ++k) scale += std::fabs(d[k]);
        if (scale == 0.0) {
            e[i] = d[i - 1];
            for (int j = 0; j < i; ++j) { d[j] = V(i - 1, j); V(i, j) = 0.0; V(j, i) = 0.0; }
        } else {
            for (int k = 0; k < i; ++k) { d[k] /= scale; h += d[k] * d[k]; }
            double f = d[i - 1];
            double g = std::sqrt(h);
            if (f > 0) g = -g;
            e[i] = scale * g;
            h -= f * g;
            d[i - 1] = f - g;
            for (int j = 0; j < i; ++j) e[j] = 0.0;
            for (int j = 0; j < i; ++j) {
                f = d[j];
                V(j, i) = f;
                g = e[j] + V(j, j) * f;
                for (int k = j + 1; k <= i - 1; ++k) { g += V(k, j) * d[k]; e[k] += V(k, j) * f; }
                e[j] = g;
            }
            f = 0.0;
            for (int j = 0; j < i; ++j) { e[j] /= h; f += e[j] * d[j]; }
            double hh = f / (h + h);
            for (int j = 0; j < i; ++j) e[j] -= hh * d[j];
            for (int j = 0; j < i; ++j) {
                f = d[j]; g = e[j];
                for (int k = j; k <= i - 1; ++k) V(k, j) -= (f * e[k] + g * d[k]);
                d[j] = V(i - 1, j);
                V(i, j) = 0.0;
            }
        }
        d[i] = h;
    }
    for (int i = 0; i < n - 1; ++i) {
        V(n - 1, i) = V(i, i);
        V(i, i) = 1.0;
        double h = d[i + 1];
        if (h != 0.0) {
            for (int k = 0; k <= i; ++k) d[k] = V(k, i + 1) / h;
            for (int j = 0; j <= i; ++j) {
                double g = 0.0;
                for (int k = 0; k <= i; ++k) g += V(k, i + 1) * V(k, j);
                for (int k = 0; k <= i; ++k) V(k, j) -= g * d[k];
            }
        }
        for (int k = 0; k <= i; ++k) V(k, i + 1) = 0.0;
    }
    for (int j = 0; j < n; ++j) { d[j] = V(n - 1, j); V(n - 1, j) = 0.0; }
    V(n - 1, n - 1) = 1.0;
    e[0] = 0.0;
}

// ---- helper threads for the eigenvector accumulation of large Rayleigh quotients -----------------
// At K = 127 (target rank 63) one QL eigensolve is ~0.41 ms of which the scalar rotation recurrence
// (sqrt / divide chain on d, e) is ~0.18 ms and applying the ~14 000 rotations to the K x K eigenvector
// matrix the rest.  The recurrence never reads the eigenvector matrix, so the calling thread runs it
// ALONE, logging every sweep's rotations, while a few helper threads replay the log on disjoint ROW
// SLICES of the matrix as it is produced (rows are independent under column rotations).  Every entry
// sees exactly the arithmetic of the serial loop: bit-identical results.
// Round 2, first version: helpers slept on a condition variable between eigensolves and the caller slept
// until they were done -- MEASURED ON THE MI355X BOX (256 hardware threads): 6-9x SLOWER than serial
// (3.7-5.1 ms instead of 0.56 ms per iteration), because idle cores sit in deep C-states and every wake-up
// costs up to a millisecond.  Second version (this one) never sleeps on the critical path: the pool is ARMED
// when a projection with a large Krylov dimension starts, armed helpers spin on the job counter and stay
// hot for 20 ms after the last job, the caller spins on the completion counter.  MEASURED AGAIN on the same
// box (tools/gpurun_eigthreads.py): still 5x slower (K = 127 standalone 0.97 -> 4.6 / 4.9 / 4.9 / 6.8 ms
// with 1 / 2 / 4 / 8 helpers; rank-63 window 359 -> 92-162 it/s) -- streaming ~3500 freshly written cache
// lines of rotation log from the producing core to cores on other CCDs / the other socket costs more than
// the 0.23 ms of arithmetic it spreads.  PROXSDP_HIP_EIG_THREADS therefore defaults to 0 (serial); the code
// stays because it is bit-identical and tested, for hosts where the cores share a cache.
struct QlSweep { int lo, hi, off; };                     // rotations i = hi-1 .. lo, (c, s) at log[off + (hi-1-i)]
struct QlJob {
    int n = 0;
    double* a = nullptr;
    const QlSweep* sweeps = nullptr;
    const double* cs = nullptr;                          // interleaved c, s
    std::atomic<int> ready{0};                           // sweeps published so far
    std::atomic<int> total{-1};                          // number of sweeps, once known
    int parts = 1;                                       // row slices, one per helper
};
inline void ql_replay(const QlJob& J, int r0, int r1) {
    const int n = J.n;
    double* a = J.a;
    int done = 0;
    for (;;) {
        const int tot = J.total.load(std::memory_order_acquire);
        int avail = J.ready.load(std::memory_order_acquire);
        if (done >= avail) {
            if (tot >= 0 && done >= tot) return;
#if defined(__x86_64__)
            _mm_pause();
#endif
            continue;
        }
        for (; done < avail; ++done) {
            const QlSweep& S = J.sweeps[done];
            const double* cs = J.cs + 2 * (size_t)S.off;
            for (int i = S.hi - 1, q = 0; i >= S.lo; --i, ++q) {
                const double c = cs[2 * q], s = cs[2 * q + 1];
                double* __restrict__ ci = a + (size_t)i * n;
                double* __restrict__ ci1 = a + (size_t)(i + 1) * n;
                for (int k = r0; k < r1; ++k) {
                    const double h = ci1[k];
                    ci1[k] = s * ci[k] + c * h;
                    ci[k] = c * ci[k] - s * h;
                }
            }
        }
    }
}
class QlPool {
public:
    static QlPool& get() { static QlPool p; return p; }
    int helpers() const { return nth_.load(std::memory_order_acquire); }
    // at least t helper threads (explicit request of a caller, e.g. the bit-identity test)
    void ensure(int t) {
        if (!busy_.try_lock()) return;
        t = std::min(t, 16);
        // (no job can be in flight while busy_ is held: a new helper starts from the current job counter)
        // th_ has capacity 16 from construction (never reallocates); readers only look at nth_
        while ((int)th_.size() < t) {
            const int i = (int)th_.size();
            const long long g0 = gen_.load(std::memory_order_acquire);
            th_.emplace_back([this, i, g0]() { loop(i, g0); });
            nth_.store((int)th_.size(), std::memory_order_release);
        }
        busy_.unlock();
    }
    // helpers leave their condition variable and spin for the next 20 ms (called when a projection with a
    // large Krylov dimension starts, and by start())
    void arm() {
        if (helpers() == 0) return;
        armed_until_.store(now_ns() + 20000000LL, std::memory_order_release);
        if (sleepers_.load(std::memory_order_acquire) > 0) {
            std::lock_guard<std::mutex> lk(mu_);
            cv_.notify_all();
        }
    }
    // run job J on the helpers: every row of the matrix belongs to one helper's slice
    bool start(QlJob* J) {
        if (helpers() == 0 || !busy_.try_lock()) return false;
        J->parts = helpers();                    // (stable while busy_ is held: ensure() takes the same lock)
        done_.store(0, std::memory_order_relaxed);
        job_.store(J, std::memory_order_release);
        gen_.fetch_add(1, std::memory_order_acq_rel);
        arm();
        return true;
    }
    void finish() {
        const int t = helpers();
        while (done_.load(std::memory_order_acquire) < t) {
#if defined(__x86_64__)
            _mm_pause();
#endif
        }
        job_.store(nullptr, std::memory_order_release);
        busy_.unlock();
    }
private:
    static long long now_ns() {
        return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
    }
    QlPool() { th_.reserve(16); }          // helpers are created by ensure() (options.host_eig_threads), at most 16
    ~QlPool() {
        { std::lock_guard<std::mutex> lk(mu_); stop_.store(true); }
        cv_.notify_all();
        for (auto& t : th_) if (t.joinable()) t.join();
    }
    void loop(int idx, long long seen) {
        unsigned spins = 0;
        for (;;) {
            const long long g = gen_.load(std::memory_order_acquire);
            if (g != seen) {
                seen = g;
                QlJob* J = job_.load(std::memory_order_acquire);
                if (J != nullptr && idx < J->parts) {
                    const int parts = J->parts, n = J->n;
                    ql_replay(*J, (int)((long long)n * idx / parts), (int)((long long)n * (idx + 1) / parts));
                }
                done_.fetch_add(1, std::memory_order_acq_rel);
                continue;
            }
            if (stop_.load(std::memory_order_relaxed)) return;
#if defined(__x86_64__)
            _mm_pause();
#endif
            if ((++spins & 1023u) == 0 && now_ns() > armed_until_.load(std::memory_order_acquire)) {
                std::unique_lock<std::mutex> lk(mu_);
                sleepers_.fetch_add(1, std::memory_order_acq_rel);
                cv_.wait(lk, [&]() {
                    return stop_.load() || gen_.load(std::memory_order_acquire) != seen ||
                           now_ns() <= armed_until_.load(std::memory_order_acquire);
                });
                sleepers_.fetch_sub(1, std::memory_order_acq_rel);
                if (stop_.load()) return;
            }
        }
    }
    std::vector<std::thread> th_;
    std::mutex mu_, busy_;
    std::condition_variable cv_;
    std::atomic<QlJob*> job_{nullptr};
    std::atomic<long long> gen_{0}, armed_until_{0};
    std::atomic<int> done_{0}, sleepers_{0}, nth_{0};
    std::atomic<bool> stop_{false};
};

// Phase 2: implicit-shift QL on the tridiagonal (d, e as left by phase 1), accumulating the
// rotations into the n x n matrix a (which must hold the transform so far: Q, or the identity).
// On exit columns of a are orthonormal eigenvectors, d ascending.  Returns 0, or 1 if QL failed.
// threads: -1 = the pool's choice (helpers from n >= 96), 0 = serial.
inline int ql_implicit(int n, double* a, double* d, double* e_in, int threads = -1) {
    auto V = [&](int i, int j) -> double& { return a[(size_t)j * n + i]; };
    std::vector<double> e(e_in, e_in + n);
    for (int i = 1; i < n; ++i) e[i - 1] = e[i];
    e[n - 1] = 0.0;
    double f = 0.0, tst1 = 0.0;
    const double eps = 2.220446049250313e-16;
    int rc = 0;
    // ---- optional helpers: the recurrence below only logs its rotations
    QlJob job;
    static thread_local std::vector<QlSweep> sweeps;            // reused across calls (no 1 MB malloc per eigensolve)
    static thread_local std::vector<double> cslog;
    cslog.clear();
    bool logged = false;
    if (threads != 0 && n >= 96) {
        QlPool& pool = QlPool::get();
        if (threads > 0) pool.ensure(threads);
        if (pool.helpers() > 0) {
            if (sweeps.size() < (size_t)200 * n + 8) sweeps.resize((size_t)200 * n + 8);   // <= 200 QL iterations per eigenvalue
            cslog.reserve((size_t)8 * n * n);                    // ~0.85 n^2 rotations expected (2 doubles each)
            job.n = n; job.a = a; job.sweeps = sweeps.data();
            logged = true;
        }
    }
    int nsweep = 0;
    bool started = false;
    for (int l = 0; l < n; ++l) {
        tst1 = std::max(tst1, std::fabs(d[l]) + std::fabs(e[l]));
        int m = l;
        while (m < n) { if (std::fabs(e[m]) <= eps * tst1) break; ++m; }
        if (m >= n) m = n - 1;
        if (m > l) {
            int iter = 0;
            do {
                if (++iter > 200) { rc = 1; break; }
                double g = d[l];
                double p = (d[l + 1] - g) / (2.0 * e[l]);
                double r = std::sqrt(p * p + 1.0);
                if (p < 0) r = -r;
                d[l] = e[l] / (p + r);
                d[l + 1] = e[l] * (p + r);
                double dl1 = d[l + 1];
                double h = g - d[l];
                for (int i = l + 2; i < n; ++i) d[i] -= h;
                f += h;
                p = d[m];
                double c = 1.0, c2 = c, c3 = c, el1 = e[l + 1], s = 0.0, s2 = 0.0;
                const size_t off = cslog.size() / 2;
                if (logged && cslog.size() + 2 * (size_t)(m - l) > cslog.capacity()) {
                    // (never expected: 4 n^2 rotations reserved; finish serially if it happens)
                    if (started) { job.total.store(nsweep, std::memory_order_release); QlPool::get().finish(); started = false; }
                    else { job.total.store(nsweep, std::memory_order_release); ql_replay(job, 0, n); }
                    logged = false;
                }
                for (int i = m - 1; i >= l; --i) {
                    c3 = c2; c2 = c; s2 = s;
                    g = c * e[i];
                    h = c * p;
                    r = std::sqrt(p * p + e[i] * e[i]);
                    e[i + 1] = s * r;
                    s = e[i] / r;
                    c = p / r;
                    p = c * d[i] - s * g;
                    d[i + 1] = h + s * (c * g + s * d[i]);
                    if (logged) { cslog.push_back(c); cslog.push_back(s); }
                    else {
                        // (explicit no-alias column pointers: this loop must vectorise -- measured 2.2x on the
                        // whole eigensolve at K = 127 against a version the compiler left scalar)
                        double* __restrict__ ci = a + (size_t)i * n;
                        double* __restrict__ ci1 = a + (size_t)(i + 1) * n;
                        const double cc = c, ss = s;
                        for (int k = 0; k < n; ++k) {
                            const double hk = ci1[k];
                            ci1[k] = ss * ci[k] + cc * hk;
                            ci[k] = cc * ci[k] - ss * hk;
                        }
                    }
                }
                if (logged) {
                    sweeps[nsweep] = QlSweep{l, m, (int)off};
                    ++nsweep;
                    job.cs = cslog.data();                       // (capacity reserved: the pointer never moves)
                    job.ready.store(nsweep, std::memory_order_release);
                    if (!started) started = QlPool::get().start(&job);
                }
                p = -s * s2 * c3 * el1 * e[l] / dl1;
                e[l] = s * p;
                d[l] = c * p;
            } while (std::fabs(e[l]) > eps * tst1);
        }
        d[l] = d[l] + f;
        e[l] = 0.0;
    }
    if (logged) {
        job.total.store(nsweep, std::memory_order_release);
        if (started) {
            QlPool::get().finish();                                    // (spins: the helpers trail the recurrence closely)
        } else {
            ql_replay(job, 0, n);                                      // pool busy (another solver thread): serial replay
        }
    }
    // ---- sort ascending
    for (int i = 0; i < n - 1; ++i) {
        int k = i; double p = d[i];
        for (int j = i + 1; j < n; ++j) if (d[j] < p) { k = j; p = d[j]; }
        if (k != i) {
            d[k] = d[i]; d[i] = p;
            for (int j = 0; j < n; ++j) std::swap(V(j, i), V(j, k));
        }
    }
    return rc;
}

inline int symeig_dense(int n, double* a, double* d, bool tridiagonal = false, int threads = -1) {
    if (n <= 0) return 0;
    if (n == 1) { d[0] = a[0]; a[0] = 1.0; return 0; }
    std::vector<double> e(n, 0.0);
    auto V = [&](int i, int j) -> double& { return a[(size_t)j * n + i]; };
    if (tridiagonal) {
        for (int j = 0; j < n; ++j) d[j] = V(j, j);
        for (int j = 1; j < n; ++j) e[j] = V(j, j - 1);
        std::fill(a, a + (size_t)n * n, 0.0);
        for (int j = 0; j < n; ++j) V(j, j) = 1.0;
    } else {
        householder_tridiag(n, a, d, e.data());
    }
    return ql_implicit(n, a, d, e.data(), threads);
}

// Eigen-decomposition of the restarted Rayleigh quotient
//     T = [ diag(D)  f   0 ;  f'  alpha_m  beta_m e1' ;  0  beta_m e1  tridiag(alpha, beta) ]   (K x K, m = keep)
// from a PRE-REDUCED arrow part: Qa ((m+1) x (m+1), column-major), da, ea = householder_tridiag of
// [diag(D) f; f' 0], computed while the GPU was still running the Lanczos steps of the cycle.
// al[m..K), be[m..K-1) are the recurrence coefficients of the new steps.  U (K x K) receives the
// eigenvectors, d the ascending eigenvalues.
inline int symeig_tridiag_from(int K, int m, const double* Qa, const double* da, const double* ea,
                               const double* al, const double* be, double* U, double* d) {
    std::vector<double> e(K, 0.0);
    std::fill(U, U + (size_t)K * K, 0.0);
    for (int c = 0; c <= m; ++c)
        for (int r = 0; r <= m; ++r) U[(size_t)c * K + r] = Qa[(size_t)c * (m + 1) + r];
    for (int j = m + 1; j < K; ++j) U[(size_t)j * K + j] = 1.0;
    for (int j = 0; j < m; ++j) d[j] = da[j];
    for (int j = m; j < K; ++j) d[j] = al[j];
    for (int j = 1; j <= m; ++j) e[j] = ea[j];
    for (int j = m + 1; j < K; ++j) e[j] = be[j - 1];
    return ql_implicit(K, U, d, e.data());
}

// ------------------------------------------------------------------ spin pool
// A few helper threads for SHORT independent host jobs on the critical path (the per-block restart logic of a batched
// multi-block Lanczos run: 8 K x K eigensolves of ~10 us each per cycle).  Waking a sleeping thread costs more than such a
// job, so the helpers sleep only while the pool is DISARMED; armed (for the duration of one batched projection) they spin
// on a generation counter, the caller takes part in the work, and completion is a spin on a counter.
class SpinPool {
public:
    explicit SpinPool(int nthreads) {
        for (int t = 0; t < nthreads; ++t) th_.emplace_back([this]() { loop(); });
    }
    ~SpinPool() {
        { std::lock_guard<std::mutex> lk(mu_); stop_.store(true); }
        cv_.notify_all();
        for (auto& t : th_) if (t.joinable()) t.join();
    }
    SpinPool(const SpinPool&) = delete;
    SpinPool& operator=(const SpinPool&) = delete;
    int helpers() const { return (int)th_.size(); }
    void arm() {
        { std::lock_guard<std::mutex> lk(mu_); armed_.store(true, std::memory_order_release); }
        cv_.notify_all();
    }
    void disarm() { armed_.store(false, std::memory_order_release); }
    // fn(i) for every i in [0, n), on the helpers and the calling thread; returns when all are done.
    // fn must not throw (callers catch inside and report through their own state).
    // The job index is handed out through ONE 64-bit word [generation | next index] by compare-and-swap: a helper
    // that was descheduled inside the previous generation's loop cannot take an index of this one by accident (with a
    // separate generation counter and index counter it could: a stale fetch_add executed a job twice and let run()
    // return early -- ADVICE r3), and the job description it reads is the one published with that word.
    template <typename F>
    void run(int n, F&& fn) {
        if (n <= 0) return;
        std::function<void(int)> f = fn;
        job_.store(&f, std::memory_order_relaxed);
        njobs_.store(n, std::memory_order_relaxed);
        done_.store(0, std::memory_order_relaxed);
        // the ticket carries 32 bits of the generation: keep the counter itself in that range (after 2^32
        // run() calls `t >> 32` could never equal a 64-bit g again) and never hand out generation 0 (the helpers' "nothing seen")
        gen_ = (gen_ + 1) & 0xffffffffull;
        if (gen_ == 0) gen_ = 1;
        const uint64_t g = gen_;
        ticket_.store(g << 32, std::memory_order_release);
        work(g);
        while (done_.load(std::memory_order_acquire) < n) {
#if defined(__x86_64__)
            _mm_pause();
#endif
        }
    }
private:
    void work(uint64_t g) {
        for (;;) {
            uint64_t t = ticket_.load(std::memory_order_acquire);
            if ((t >> 32) != g) return;                                  // another generation's word: not ours to touch
            const int i = (int)(t & 0xffffffffu);
            if (i >= njobs_.load(std::memory_order_relaxed)) return;
            if (!ticket_.compare_exchange_weak(t, t + 1, std::memory_order_acq_rel, std::memory_order_acquire)) continue;
            (*job_.load(std::memory_order_relaxed))(i);                  // (run() cannot return before done_ counts this job)
            done_.fetch_add(1, std::memory_order_acq_rel);
        }
    }
    void loop() {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&]() { return stop_.load() || armed_.load(std::memory_order_acquire); });
                if (stop_.load()) return;
            }
            while (armed_.load(std::memory_order_acquire) && !stop_.load(std::memory_order_relaxed)) {
                const uint64_t g = ticket_.load(std::memory_order_acquire) >> 32;
                if (g != seen) { seen = g; work(g); }
                else {
#if defined(__x86_64__)
                    _mm_pause();
#endif
                }
            }
        }
    }
    std::vector<std::thread> th_;
    std::mutex mu_;
    std::condition_variable cv_;
    std::atomic<bool> armed_{false}, stop_{false};
    uint64_t gen_ = 0;                                   // (only the thread that calls run() touches it)
    std::atomic<uint64_t> ticket_{0};                    // [generation : 32 | next job index : 32]
    std::atomic<int> done_{0}, njobs_{0};
    std::atomic<const std::function<void(int)>*> job_{nullptr};
};

// ------------------------------------------------------------------ split + rank-one merge (host_eig_merge.hpp)
// The same K x K Rayleigh quotient T as symeig_tridiag_from (m = 0: plain tridiagonal), decomposed in two phases:
//   first(): T1' = T[0:k1, 0:k1] - |b| e e'  (b = T[k1-1, k1]) -- needs only the arrow part and the first k1 - m
//            recurrence coefficients: the Lanczos driver calls it while the GPU is still running the cycle
//   second(): T2' = T[k1:K, k1:K] - |b| e e' (tridiagonal tail) and the merge -- the part left on the critical path
// k1 = m + 1 after a restart (the arrow with its hub), any 1 <= k1 < K in the first cycle.
struct SplitEig {
    Rank1Merge M;
    std::vector<double> Q1, d1, Q2, d2, e2;
    int k1 = 0, K = 0;
    double beta = 0.0;
    bool have_first = false;
    int first(int k1_, int m, const double* D, const double* f, const double* al, const double* be) {
        k1 = k1_; have_first = false;
        if (k1 < 1 || (m > 0 && k1 != m + 1)) return 1;
        beta = be[k1 - 1];
        if (!(beta == beta) || beta == 0.0) return 1;
        Q1.assign((size_t)k1 * k1, 0.0); d1.assign(k1, 0.0);
        int rc;
        if (m == 0) {
            for (int j = 0; j < k1; ++j) {
                Q1[(size_t)j * k1 + j] = al[j];
                if (j + 1 < k1) { Q1[(size_t)j * k1 + j + 1] = be[j]; Q1[(size_t)(j + 1) * k1 + j] = be[j]; }
            }
            Q1[(size_t)(k1 - 1) * k1 + (k1 - 1)] -= std::fabs(beta);
            rc = symeig_dense(k1, Q1.data(), d1.data(), true, 0);
        } else {
            for (int j = 0; j < m; ++j) {
                Q1[(size_t)j * k1 + j] = D[j];
                Q1[(size_t)j * k1 + m] = f[j]; Q1[(size_t)m * k1 + j] = f[j];
            }
            Q1[(size_t)m * k1 + m] = al[m] - std::fabs(beta);
            rc = symeig_dense(k1, Q1.data(), d1.data(), false, 0);
        }
        have_first = (rc == 0);
        return rc;
    }
    int second(int K_, const double* al, const double* be) {
        K = K_;
        const int k2 = K - k1;
        if (!have_first || k2 < 1) return 1;
        Q2.assign((size_t)k2 * k2, 0.0); d2.assign(k2, 0.0);
        for (int j = 0; j < k2; ++j) {
            Q2[(size_t)j * k2 + j] = al[k1 + j];
            if (j + 1 < k2) { Q2[(size_t)j * k2 + j + 1] = be[k1 + j]; Q2[(size_t)(j + 1) * k2 + j] = be[k1 + j]; }
        }
        Q2[0] -= std::fabs(beta);
        if (symeig_dense(k2, Q2.data(), d2.data(), true, 0) != 0) return 1;
        M.build(k1, k2, Q1.data(), d1.data(), Q2.data(), d2.data(), beta);
        return 0;
    }
};

// ------------------------------------------------------------------ options
enum OptType { OT_I32, OT_I64, OT_F64 };
struct OptEntry { const char* name; OptType type; size_t offset; };

#define PX_OPT(name, type) { #name, type, offsetof(proxsdp_options, name) }
inline const std::vector<OptEntry>& option_table() {
    static const std::vector<OptEntry> t = {
        PX_OPT(log_verbose, OT_I32), PX_OPT(log_freq, OT_I32), PX_OPT(timer_verbose, OT_I32),
        PX_OPT(timer_file, OT_I32), PX_OPT(disable_julia_logger, OT_I32), PX_OPT(warn_on_limit, OT_I32),
        PX_OPT(extended_log, OT_I32), PX_OPT(extended_log2, OT_I32), PX_OPT(log_repeat_header, OT_I32),
        PX_OPT(time_limit, OT_F64),
        PX_OPT(tol_gap, OT_F64), PX_OPT(tol_feasibility, OT_F64), PX_OPT(tol_feasibility_dual, OT_F64),
        PX_OPT(tol_primal, OT_F64), PX_OPT(tol_dual, OT_F64), PX_OPT(tol_psd, OT_F64), PX_OPT(tol_soc, OT_F64),
        PX_OPT(check_dual_feas, OT_I32), PX_OPT(check_dual_feas_freq, OT_I32),
        PX_OPT(max_obj, OT_F64), PX_OPT(min_iter_max_obj, OT_I32),
        PX_OPT(min_iter_time_infeas, OT_I32), PX_OPT(infeas_gap_tol, OT_F64),
        PX_OPT(infeas_limit_gap_tol, OT_F64), PX_OPT(infeas_stable_gap_tol, OT_F64),
        PX_OPT(infeas_feasibility_tol, OT_F64), PX_OPT(infeas_stable_feasibility_tol, OT_F64),
        PX_OPT(certificate_search, OT_I32), PX_OPT(certificate_obj_tol, OT_F64), PX_OPT(certificate_fail_tol, OT_F64),
        PX_OPT(min_beta, OT_F64), PX_OPT(max_beta, OT_F64), PX_OPT(initial_beta, OT_F64),
        PX_OPT(initial_adapt_level, OT_F64), PX_OPT(adapt_decay, OT_F64), PX_OPT(adapt_window, OT_I32),
        PX_OPT(convergence_window, OT_I32), PX_OPT(convergence_check, OT_I32),
        PX_OPT(max_iter, OT_I64), PX_OPT(min_iter, OT_I64), PX_OPT(divergence_min_update, OT_I64),
        PX_OPT(max_iter_lp, OT_I64), PX_OPT(max_iter_conic, OT_I64), PX_OPT(max_iter_local, OT_I64),
        PX_OPT(advanced_initialization, OT_I32), PX_OPT(line_search_flag, OT_I32),
        PX_OPT(max_linsearch_steps, OT_I32), PX_OPT(delta, OT_F64), PX_OPT(initial_theta, OT_F64),
        PX_OPT(linsearch_decay, OT_F64),
        PX_OPT(full_eig_decomp, OT_I32), PX_OPT(max_target_rank_krylov_eigs, OT_I32),
        PX_OPT(min_size_krylov_eigs, OT_I32), PX_OPT(warm_start_eig, OT_I32),
        PX_OPT(rank_increment, OT_I32), PX_OPT(rank_increment_factor, OT_I32),
        PX_OPT(eigsolver, OT_I32), PX_OPT(eigsolver_min_lanczos, OT_I32), PX_OPT(eigsolver_resid_seed, OT_I64),
        PX_OPT(arpack_tol, OT_F64), PX_OPT(arpack_resid_init, OT_I32), PX_OPT(arpack_reset_resid, OT_I32),
        PX_OPT(arpack_max_iter, OT_I64),
        PX_OPT(krylovkit_reset_resid, OT_I32), PX_OPT(krylovkit_resid_init, OT_I32),
        PX_OPT(krylovkit_tol, OT_F64), PX_OPT(krylovkit_max_iter, OT_I32), PX_OPT(krylovkit_eager, OT_I32),
        PX_OPT(krylovkit_verbose, OT_I32),
        PX_OPT(reduce_rank, OT_I32), PX_OPT(rank_slack, OT_I32),
        PX_OPT(full_eig_freq, OT_I64), PX_OPT(full_eig_len, OT_I64),
        PX_OPT(equilibration, OT_I32), PX_OPT(equilibration_iters, OT_I32),
        PX_OPT(equilibration_lb, OT_F64), PX_OPT(equilibration_ub, OT_F64), PX_OPT(equilibration_limit, OT_F64),
        PX_OPT(equilibration_force, OT_I32), PX_OPT(approx_norm, OT_I32),
        PX_OPT(device_id, OT_I32), PX_OPT(trace_capacity, OT_I32), PX_OPT(profile_symv_every, OT_I32),
        PX_OPT(support_path, OT_I32), PX_OPT(lanczos_operator, OT_I32), PX_OPT(initial_target_rank, OT_I32),
        PX_OPT(full_eig_lanczos, OT_I32), PX_OPT(lanczos_cycle_kernel, OT_I32), PX_OPT(lanczos_warm_start, OT_I32),
        PX_OPT(reconstruct_mfma, OT_I32), PX_OPT(small_block_batch, OT_I32), PX_OPT(full_eig_sign, OT_I32), PX_OPT(psd_sign_engine, OT_I32),
        PX_OPT(full_eig_lanczos_verify, OT_I32), PX_OPT(full_eig_lanczos_posres, OT_F64), PX_OPT(full_eig_lanczos_kdim10, OT_I32),
        PX_OPT(sign_small_tile_max, OT_I32), PX_OPT(host_eig_threads, OT_I32), PX_OPT(block_threads, OT_I32),
        PX_OPT(host_eig_merge, OT_I32), PX_OPT(block_batch, OT_I32),
        PX_OPT(rocsolver_warmup, OT_I32), PX_OPT(debug_fail_iteration, OT_I32), PX_OPT(host_wait_spin, OT_I32), PX_OPT(sign_start_row, OT_I32), PX_OPT(general_batch, OT_I32),
        PX_OPT(full_eig_lanczos_certify, OT_I32), PX_OPT(full_eig_lanczos_tol, OT_F64), PX_OPT(host_merge_threads, OT_I32),
        PX_OPT(equilibration_reference_aliasing, OT_I32), PX_OPT(block_batch_groups, OT_I32), PX_OPT(full_eig_lanczos_warm_pow, OT_F64),
    };
    return t;
}
#undef PX_OPT

inline void default_options(proxsdp_options* o) {      // options.jl:1-132
    std::memset(o, 0, sizeof(*o));
    o->struct_size = (int64_t)sizeof(*o);
    o->log_verbose = 0; o->log_freq = 1000; o->disable_julia_logger = 1;
    o->time_limit = 360000.0;
    o->tol_gap = 1e-4; o->tol_feasibility = 1e-4; o->tol_feasibility_dual = 1e-4;
    o->tol_primal = 1e-4; o->tol_dual = 1e-4; o->tol_psd = 1e-7; o->tol_soc = 1e-7;
    o->check_dual_feas = 0; o->check_dual_feas_freq = 1000;
    o->max_obj = 1e20; o->min_iter_max_obj = 10;
    o->min_iter_time_infeas = 1000; o->infeas_gap_tol = 1e-4; o->infeas_limit_gap_tol = 1e-1;
    o->infeas_stable_gap_tol = 1e-4; o->infeas_feasibility_tol = 1e-4;
    o->infeas_stable_feasibility_tol = 1e-8;
    o->certificate_search = 1; o->certificate_obj_tol = 1e-1; o->certificate_fail_tol = 1e-8;
    o->min_beta = 1e-5; o->max_beta = 1e5; o->initial_beta = 1.0;
    o->initial_adapt_level = 0.9; o->adapt_decay = 0.8; o->adapt_window = 50;
    o->convergence_window = 200; o->convergence_check = 50;
    o->max_iter = 0; o->min_iter = 40; o->divergence_min_update = 50;
    o->max_iter_lp = 10000000; o->max_iter_conic = 1000000; o->max_iter_local = 0;
    o->advanced_initialization = 1; o->line_search_flag = 1; o->max_linsearch_steps = 5000;
    o->delta = 0.9999; o->initial_theta = 1.0; o->linsearch_decay = 0.75;
    o->full_eig_decomp = 0; o->max_target_rank_krylov_eigs = 16; o->min_size_krylov_eigs = 100;
    o->warm_start_eig = 1; o->rank_increment = 1; o->rank_increment_factor = 1;
    o->eigsolver = 2; o->eigsolver_min_lanczos = 25; o->eigsolver_resid_seed = 1234;
    o->arpack_tol = 1e-10; o->arpack_resid_init = 3; o->arpack_reset_resid = 1; o->arpack_max_iter = 10000;
    o->krylovkit_reset_resid = 0; o->krylovkit_resid_init = 3; o->krylovkit_tol = 1e-12;
    o->krylovkit_max_iter = 100; o->krylovkit_eager = 0; o->krylovkit_verbose = 0;
    o->reduce_rank = 0; o->rank_slack = 3; o->full_eig_freq = 10000000; o->full_eig_len = 0;
    o->equilibration = 0; o->equilibration_iters = 1000; o->equilibration_lb = -10.0;
    o->equilibration_ub = 10.0; o->equilibration_limit = 0.9; o->equilibration_force = 0;
    o->approx_norm = 1;
    o->device_id = 0; o->trace_capacity = 0; o->profile_symv_every = 0; o->support_path = -1;
    o->lanczos_operator = -1; o->initial_target_rank = 2;
    o->full_eig_lanczos = -1; o->lanczos_cycle_kernel = -1; o->lanczos_warm_start = 0; o->reconstruct_mfma = -1;
    o->small_block_batch = -1; o->full_eig_sign = -1; o->psd_sign_engine = -1;
    o->full_eig_lanczos_verify = -1; o->full_eig_lanczos_posres = 1e-6; o->full_eig_lanczos_kdim10 = 30;
    o->sign_small_tile_max = 3072; o->host_eig_threads = 0; o->block_threads = -1;
    o->host_eig_merge = -1; o->block_batch = -1; o->rocsolver_warmup = 0; o->host_wait_spin = -1; o->sign_start_row = -1; o->general_batch = -1;
    o->full_eig_lanczos_certify = -1; o->full_eig_lanczos_tol = 0.0; o->host_merge_threads = -1;
    o->equilibration_reference_aliasing = 1; o->block_batch_groups = -1; o->full_eig_lanczos_warm_pow = 1.0;
}

inline int set_option(proxsdp_options* o, const char* name, double v) {
    for (const auto& e : option_table()) {
        if (std::strcmp(e.name, name) == 0) {
            char* base = reinterpret_cast<char*>(o) + e.offset;
            if (e.type == OT_I32) *reinterpret_cast<int32_t*>(base) = (int32_t)std::llround(v);
            else if (e.type == OT_I64) *reinterpret_cast<int64_t*>(base) = (int64_t)std::llround(v);
            else *reinterpret_cast<double*>(base) = v;
            return 0;
        }
    }
    return PROXSDP_E_INVALID;
}
inline int get_option(const proxsdp_options* o, const char* name, double* v) {
    for (const auto& e : option_table()) {
        if (std::strcmp(e.name, name) == 0) {
            const char* base = reinterpret_cast<const char*>(o) + e.offset;
            if (e.type == OT_I32) *v = (double)*reinterpret_cast<const int32_t*>(base);
            else if (e.type == OT_I64) *v = (double)*reinterpret_cast<const int64_t*>(base);
            else *v = *reinterpret_cast<const double*>(base);
            return 0;
        }
    }
    return PROXSDP_E_INVALID;
}

}  // namespace proxsdp
