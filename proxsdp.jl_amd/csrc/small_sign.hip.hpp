// PSD projection of a SMALL dense block (side 3 .. 64) by the matrix sign function in ONE launch, one workgroup per block,
// all iterates in LDS.
//
// Replaces full_eig! (prox_operators.jl:111-126: eigen!(Symmetric(X)) + the rank-1 loop) for blocks that never take the
// Krylov path (side <= min_size_krylov_eigs = 100, prox_operators.jl:46-49).  What these blocks cost before (round 5,
// tools/gpurun_small_rate.py, one projection per PDHG iteration): a lone block of side 3 .. 32 went to rocSOLVER's dsyevd
// (~500 us: a chain of ~30 tiny launches), side 33 .. 64 to the tiled sign projection of sign_project.hip.hpp (57 + launches of
// 10 .. 30 us: 1.2 ms at side 60), several blocks of side <= 32 to the batched Jacobi kernel (k_small_psd_project: 5 us at
// side 3, 950 us at side 22, 10 ms at side 60).  The reference's CPU does a 22 x 22 dsyevr in ~30 us, so small models ran
// at a tenth of its iteration rate.  Here the SAME iteration as sign_project.hip.hpp (same table, all 19 rows, 57 products:
// X+ = (X + X sign X) / 2, every |eigenvalue| >= 1e-10 s resolved) runs inside one workgroup: four n_pad x n_pad
// matrices in LDS (A, X, Y, Q; n_pad = side rounded up to 16, row stride n_pad + 2 doubles = 4 banks: the MFMA operand
// reads are conflict-free), every product on the upper block-triangle of 16 x 16 tiles with v_mfma_f64_16x16x4_f64 and
// mirrored (the iterates stay exactly symmetric, which also lets BOTH operands be read row-wise), one barrier per product,
// the scalars (|A|_F, |A A|_F, tr S, |S|_F^2) by workgroup reductions; 16 waves, one tile of the triangle each; the
// shortened, TESTED schedule of the tiled projection (start at row 8, 34 products) with the test and the fall-back to the
// remaining rows decided INSIDE the kernel.  One launch per projection instead of 30 .. 60:
// ~30 us at side 22, ~80 us at side 60 (profiles/r05_small_blocks.md).
// Accuracy is that of the tiled sign projection (absolute error <= 1e-10 x spectral scale in X+); as there, solves that
// ask for tolerances below 1e-8 keep the LAPACK-accurate engines (Jacobi / dsyevd).
#pragma once
#include "sign_project.hip.hpp"

namespace proxsdp {
namespace dev {

constexpr int SS_MAXN = 64;
constexpr int SS_TPB = 1024;              // 16 waves: ONE 16 x 16 tile of the upper block-triangle per wave (<= 10 tiles at side 64)
inline size_t small_sign_lds_bytes(int maxn) {
    const int np = 16 * ((maxn + 15) / 16), ld = np + 2;
    return ((size_t)4 * np * ld + 16) * sizeof(double);
}

// C = kc * P Q (+ kb * Ysrc + ka * I when POLY) on the upper block-triangle, mirrored.  P, Q, Ysrc symmetric (exactly), so the
// B operand Q[k][col] is read as Q[col][k]: both operands with lane (l15, l4) -> (row, k), conflict-free at stride np + 2.
// Wave w owns tile w of the triangle (column-major over I <= J); the other waves pass through.
template <bool POLY>
__device__ __forceinline__ void ss_product(const double* __restrict__ P, const double* __restrict__ Q, double* __restrict__ C,
                                           const double* __restrict__ Ysrc, int np, int ld, double ka, double kb, double kc,
                                           int lane, int tI, int tJ) {
    typedef double v4f64 __attribute__((ext_vector_type(4)));
    if (tJ < 0) return;
    const int l15 = lane & 15, l4 = lane >> 4;
    v4f64 acc = (v4f64){0.0, 0.0, 0.0, 0.0};
    const double* pa = P + (16 * tI + l15) * ld + l4;
    const double* pb = Q + (16 * tJ + l15) * ld + l4;
    for (int q0 = 0; q0 < np; q0 += 16) {                  // (np is a multiple of 16: four operand pairs in flight per turn)
        double av[4], bv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { av[u] = pa[q0 + 4 * u]; bv[u] = pb[q0 + 4 * u]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u], acc, 0, 0, 0);
    }
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const int r = 16 * tI + l4 + 4 * reg, c = 16 * tJ + l15;       // D: row = (lane >> 4) + 4 reg, column = lane & 15
        double v = kc * acc[reg];
        if (POLY) { v += kb * Ysrc[r * ld + c]; if (r == c) v += ka; }
        if (r <= c) { C[r * ld + c] = v; if (r != c) C[c * ld + r] = v; }
    }
}

// workgroup sum (16 waves); result in every thread
__device__ __forceinline__ double ss_sum(double v, double* __restrict__ s_red, double* __restrict__ s_out) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) s_red[w] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double r = 0.0;
#pragma unroll
        for (int i = 0; i < SS_TPB / 64; ++i) r += s_red[i];
        *s_out = r;
    }
    __syncthreads();
    return *s_out;
}
__device__ __forceinline__ double ss_fro2(const double* __restrict__ M, int np, int ld, double* __restrict__ s_red, double* __restrict__ s_out) {
    double s = 0.0;
    for (int t = threadIdx.x; t < np * np; t += SS_TPB) { const int i = t / np, j = t - i * np; const double v = M[i * ld + j]; s += v * v; }
    return ss_sum(s, s_red, s_out);
}

// blocks with nmin <= side <= nmax are projected; the others are left to the kernel that owns them.
// j0 > 0: SHORTENED SCHEDULE of sign_project.hip.hpp / Solver::full_eig_by_sign -- the iteration starts at row j0 of the table and
// the result is TESTED in the kernel (|S|_F^2 - |S S|_F^2 <= 4e-13 n: every eigenvalue of S is 0 or +-1); a failed test continues
// with the rows from r_fail on (same guarantee as the full table).  j0 = 0: the full table, untested.
__global__ void __launch_bounds__(SS_TPB)
k_small_sign_project(double* __restrict__ x, const long long* __restrict__ offs, const int* __restrict__ sides,
                     int nmin, int nmax, int* __restrict__ rank_out, int* __restrict__ npos_out, int j0, int r_fail,
                     int* __restrict__ short_stats /* per block: 1 = schedule test passed, 2 = failed; or null */) {
    extern __shared__ __attribute__((aligned(16))) double ss_mem[];
    __shared__ double s_red[SS_TPB / 64];
    __shared__ double s_sc[4];
    const int n = sides[blockIdx.x];
    if (n < nmin || n > nmax) return;
    double* __restrict__ xp = x + offs[blockIdx.x];
    const int np = 16 * ((n + 15) / 16), ld = np + 2;
    double* A = ss_mem;
    double* X = A + np * ld;
    double* Y = X + np * ld;
    double* Q = Y + np * ld;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // this wave's tile of the upper block-triangle (none: tJ = -1)
    int tI = 0, tJ = -1;
    {
        const int T = np >> 4;
        int t = 0;
        for (int J = 0; J < T; ++J) for (int I = 0; I <= J; ++I, ++t) if (t == w) { tI = I; tJ = J; }
    }
    // ---- unpack (off-diagonals carry sqrt 2 in the packed form), padding zero
    for (int t = tid; t < np * ld; t += SS_TPB) A[t] = 0.0;
    __syncthreads();
    for (int t = tid; t < n * n; t += SS_TPB) {
        const int i = t % n, j = t / n;
        const int lo = min(i, j), hi = max(i, j);
        const double v = xp[(long long)hi * (hi + 1) / 2 + lo];
        A[i * ld + j] = (i == j) ? v : v * INV_SQRT2;
    }
    __syncthreads();
    double f2 = ss_fro2(A, np, ld, s_red, &s_sc[0]);
    const int N = n * (n + 1) / 2;
    // a block of denormal-tiny entries (|A|_F^2 below ~1e-250: 1 / f2 would overflow): the projection is positively
    // homogeneous, so the block is scaled by an exact power of two for the iteration and the result scaled back
    double unscale = 1.0;
    if (f2 > 0.0 && f2 < 1e-250) {
        for (int t = tid; t < np * ld; t += SS_TPB) A[t] *= 0x1p+500;
        __syncthreads();
        f2 = ss_fro2(A, np, ld, s_red, &s_sc[0]);
        unscale = 0x1p-500;
    }
    if (!(f2 > 0.0)) {                                       // the zero matrix (or non-finite input: left alone)
        if (f2 == 0.0) for (int t = tid; t < N; t += SS_TPB) xp[t] = 0.0;
        if (tid == 0) {
            const int rk = (f2 == 0.0) ? 0 : -1;             // -1: non-finite input -- the host raises the error the tiled path raises
            rank_out[blockIdx.x] = rk; npos_out[blockIdx.x] = rk;
            if (short_stats != nullptr && j0 > 0) short_stats[blockIdx.x] = 1;     // (the first PDHG iterate is exactly zero: counted as resolved)
        }
        return;
    }
    const double f = sqrt(f2);
    // ---- Y0 = A A / f^2,  g = |Y0|_F,  s = f sqrt(g) >= |A|_2
    ss_product<false>(A, A, Y, nullptr, np, ld, 0.0, 0.0, 1.0 / f2, lane, tI, tJ);
    __syncthreads();
    const double g = sqrt(ss_fro2(Y, np, ld, s_red, &s_sc[1]));
    const double sinv = 1.0 / (f * sqrt(g));
    // rows [from, SIGN_STEPS); `first`: X does not exist yet (the row works on X0 = A / s: Y = X0 X0 = Y0 / g)
    auto run_rows = [&](int from, bool first) {
        for (int k = from; k < SIGN_STEPS; ++k) {
            const bool cubic = SIGN_LAST_CUBIC && k + 1 == SIGN_STEPS;
            const SignStep c = SIGN_TABLE[k];
            if (first && k == from) {
                ss_product<true>(Y, Y, Q, Y, np, ld, c.a, c.b / g, c.c / (g * g), lane, tI, tJ);
                __syncthreads();
                ss_product<false>(A, Q, X, nullptr, np, ld, 0.0, 0.0, sinv, lane, tI, tJ);
                __syncthreads();
                continue;
            }
            if (cubic) {
                ss_product<true>(X, X, Q, X, np, ld, 1.5, 0.0, -0.5, lane, tI, tJ);      // Q = (3 I - X X) / 2
                __syncthreads();
            } else {
                ss_product<false>(X, X, Y, nullptr, np, ld, 0.0, 0.0, 1.0, lane, tI, tJ);
                __syncthreads();
                ss_product<true>(Y, Y, Q, Y, np, ld, c.a, c.b, c.c, lane, tI, tJ);
                __syncthreads();
            }
            ss_product<false>(X, Q, Y, nullptr, np, ld, 0.0, 0.0, 1.0, lane, tI, tJ);   // X_{k+1} into the buffer Y no longer needs
            __syncthreads();
            double* tsw = X; X = Y; Y = tsw;
        }
    };
    run_rows(j0, true);
    double fro2 = ss_fro2(X, np, ld, s_red, &s_sc[2]);
    if (j0 > 0) {
        ss_product<false>(X, X, Y, nullptr, np, ld, 0.0, 0.0, 1.0, lane, tI, tJ);        // S S, only for its norm
        __syncthreads();
        const double m4 = ss_fro2(Y, np, ld, s_red, &s_sc[3]);
        const bool ok = (fro2 == fro2) && (m4 == m4) && fabs(fro2 - m4) <= 4e-13 * (double)n;
        if (tid == 0 && short_stats != nullptr) short_stats[blockIdx.x] = ok ? 1 : 2;
        if (!ok) {                                           // (uniform over the workgroup)
            run_rows(r_fail, false);
            fro2 = ss_fro2(X, np, ld, s_red, &s_sc[2]);
        }
    }
    // ---- S = X:  #{lambda > 0} = (tr S + |S|_F^2) / 2;  X+ = (A + A S) / 2 in packed form
    double tr = 0.0;
    for (int i = tid; i < n; i += SS_TPB) tr += X[i * ld + i];
    tr = ss_sum(tr, s_red, &s_sc[3]);
    if (tid == 0) {
        int npos = (int)llround(0.5 * (tr + fro2));
        npos = max(0, min(npos, n));
        if (!(fro2 == fro2) || !(tr == tr)) npos = -1;       // non-finite: the host raises an error
        rank_out[blockIdx.x] = npos; npos_out[blockIdx.x] = npos;
    }
    if (tJ >= 0) {
        typedef double v4f64 __attribute__((ext_vector_type(4)));
        const int l15 = lane & 15, l4 = lane >> 4;
        v4f64 acc = (v4f64){0.0, 0.0, 0.0, 0.0};
        const double* pa = A + (16 * tI + l15) * ld + l4;
        const double* pb = X + (16 * tJ + l15) * ld + l4;
        for (int q0 = 0; q0 < np; q0 += 16) {
            double av[4], bv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { av[u] = pa[q0 + 4 * u]; bv[u] = pb[q0 + 4 * u]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u], acc, 0, 0, 0);
        }
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int r = 16 * tI + l4 + 4 * reg, c = 16 * tJ + l15;
            if (r <= c && c < n) {
                const double v = 0.5 * (A[r * ld + c] + acc[reg]) * unscale;
                xp[(long long)c * (c + 1) / 2 + r] = (r == c) ? v : v * SQRT2;
            }
        }
    }
}

}  // namespace dev
}  // namespace proxsdp
