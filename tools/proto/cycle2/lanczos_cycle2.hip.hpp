// One whole Lanczos cycle (operator form) in ONE persistent launch, second design (round 5).
//
// lanczos_cycle.hip.hpp (round 2) keeps each workgroup's rows of the Krylov basis in LDS and measured 12.3 us per step
// at n = 4000 (tools/gpurun_cycle_ticks.py: 4.8 us in its two exchanges, 7.5 us in its own chain of LDS passes and
// serial record sums) against 12.2 us for the two step kernels.  This kernel is built on that breakdown:
//   * the workgroup's 128 rows of the basis live in REGISTERS, twice: row-per-lane (thread (row, j mod 4) holds columns
//     j = cw, cw + 4, ...: the row sums V q, V h2) and column-per-lane (lane = basis column, 32 rows per thread: the
//     measured pass V'w' is a per-lane serial sum with w' broadcast from LDS -- no cross-lane fold network at all);
//     512 threads, 128 + 128 of the 256 VGPRs of a wave at two waves per SIMD.  The previous projection's factors sit
//     in LDS once, with a column stride of 129 doubles: conflict-free by rows and by columns;
//   * the Vp'u partials of step k + 1 are taken from w' the moment it is formed and ride in the SAME exchange as the
//     measured dots (X2), so the exchange in the middle of a step (X1) carries one double per 64-row group (u'E u);
//   * everything of the recurrence that does not need alpha -- T h2 / beta, the row sums over columns < k, the Vp u
//     part, the insertion of v_k into the two register layouts -- runs UNDER the X1 hand-off; when alpha arrives one
//     fused multiply-add per row is left;
//   * the rows of w' other workgroups need for (E u)_i are GATHERED (<= 8 loads per thread) instead of staged whole.
// THE ARITHMETIC IS THAT OF THE STEP KERNELS, BIT FOR BIT: partial dots per 64-row group with the fold network's tree
// (lane partners 8, 4, 2, 1, 16, 32 -- restated as a serial tree per lane), records summed over the groups with
// tree_in_wave / tree_across, row sums in two accumulators over columns j = cw + 4c, the same expression forms (so the
// same fused multiply-adds).  tests/test_gpu_parity.py::test_cycle2_is_bit_identical_to_the_step_kernels compares whole
// solves bit for bit; every committed iteration count and golden trace holds for either engine.
//
// Replaces: the BLAS-1 work and mat-vecs of one KrylovKit Lanczos cycle (call site /root/reference/src/eigsolver.jl:802-812),
// as k_fop* / k_lz_* do.  A cycle may be cut into two launches (kb .. ke) so that the host can read the coefficients of the
// first part while the second runs (host_eig_merge): the state between launches is exactly the exchange buffers.
//
// Placement: the launch has 8 G workgroups of which those with blockIdx % 8 == 0 work: G <= 32 workgroups on ONE XCD, one
// per CU (LDS), hand-offs through that XCD's L2 (plain stores, drained, flag; L1-bypassing loads).  As in round 2 the
// placement is an assumption for SPEED only: every spin is bounded, a timeout sets *err and the host redoes the
// projection with the step kernels.
#pragma once
#include "lanczos_cycle.hip.hpp"
#include <type_traits>

namespace proxsdp {
namespace dev {

constexpr int C2_TPB = 512;
constexpr int C2_R = 128;                 // rows per workgroup = two 64-row groups of the step kernels
constexpr int C2_FS = 129;                // LDS stride (doubles) of a factor column
constexpr int C2_NE = 8;                  // ELL entries per thread (4 threads per row): ell_w <= 32
constexpr int C2_HLD = 128;               // stride of a group's record of measured dots (krylovdim <= 127)
constexpr int C2_TLD = 64;                // stride of a group's record of Vp'u partials (rp <= 64)
constexpr int C2_CO = 136;                // length of the replicated coefficient arrays
// exchange buffer (doubles): X2 measured dots [64 groups][C2_HLD] | |w'|^2 shares [64] | Vp'w' partials [64][C2_TLD] |
// Vp'v partials of the cycle's first vector [64][C2_TLD] (X1) | u'E u shares [64] (X1) | w' (all rows, X2)
constexpr int C2_XH = 0, C2_XN = C2_XH + 64 * C2_HLD, C2_XT = C2_XN + 64, C2_XT0 = C2_XT + 64 * C2_TLD,
              C2_XA = C2_XT0 + 64 * C2_TLD, C2_XW = C2_XA + 64, C2_XTOTAL = C2_XW + 32 * C2_R;

struct Cycle2Args {
    double* V; int ldv; int npad; int nt;  // nt = 64-row groups that hold rows (records of the others are zero)
    int kfirst, kd;                        // the cycle: first step (0 or the restart's keep), krylovdim
    int kb, ke;                            // this launch: steps kb .. ke (kb > kfirst: resumed from the exchange buffers)
    double tol;
    const double* Vp; int rp; const double* lam;
    const int* ell_col; const int* ell_sidx; int ell_w; const double* esv;
    const double* arrow;                   // f | D, MAXK each (valid below kfirst)
    double* rec;                           // alphas[MAXK] | betas[MAXK] | LanczosCtl (ONE record, EigWork::rec)
    double* hsum;
    double* xb;                            // ONE exchange buffer (one base pointer in scalar registers), offsets C2_X*
    unsigned* fb; unsigned epoch0;         // flags of X1 [32] | X2 [32]
    int G; int* err;
    long long* dbg;                        // optional: wall_clock64 ticks per phase (workgroup 0)
};

// The partial dots of the step kernels are products reduced over the 64 lanes of a wave by the fold network (fold16_all):
// lane partners 8, 4, 2, 1 inside a 16-lane row, then 16, then 32.  Here a lane owns a COLUMN and walks the rows, so the same
// tree is a serial sum per lane.  Contraction is off: the step kernels' products pass through DPP moves before they are
// added, so they are never fused; these must not be either.
#pragma clang fp contract(off)
__device__ __forceinline__ double c2_tree16(const double (&p)[16]) {
    double a[8], b4[4];
#pragma unroll
    for (int r = 0; r < 8; ++r) a[r] = p[r] + p[r + 8];
#pragma unroll
    for (int r = 0; r < 4; ++r) b4[r] = a[r] + a[r + 4];
    return (b4[0] + b4[2]) + (b4[1] + b4[3]);
}
// sum_{r < 16} v[OFF + r] * s[r] in that order (v in registers, s broadcast from LDS)
template <int OFF, int N>
__device__ __forceinline__ double c2_dot16_reg(const double (&v)[N], const double* __restrict__ s) {
    double p[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) p[r] = v[OFF + r] * s[r];
    return c2_tree16(p);
}
// the same with both operands in LDS
__device__ __forceinline__ double c2_dot16_lds(const double* __restrict__ f, const double* __restrict__ s) {
    double p[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) p[r] = f[r] * s[r];
    return c2_tree16(p);
}
__device__ __forceinline__ double c2_add(double x, double y) { return x + y; }
#pragma clang fp contract(fast)

// dynamic LDS bytes of k_lz_cycle2
inline size_t cycle2_lds_bytes() {
    const size_t d = (size_t)64 * C2_FS + 4 * 4 * 128 + 4 * 4 * 64 + 4 * 128 + 8 * 64 + 3 * 512 + 6 * C2_CO + 2 * 64 + 2 * 128 + 8 + C2_NE * C2_TPB;
    return d * sizeof(double);
}

template <bool TIMING>
__global__ void __launch_bounds__(C2_TPB, 1)
k_lz_cycle2(Cycle2Args a) {
    extern __shared__ __attribute__((aligned(16))) double c2_smem[];
    if ((blockIdx.x & 7) != 0) return;
    int b = blockIdx.x >> 3;
    int kfirst = a.kfirst, kd = a.kd, ke = a.ke, rp = a.rp, nt = a.nt;
    const int kb = a.kb, G = a.G;
    const bool resume = kb > kfirst;
    // (the index variables are re-derived from laundered copies of lane / wave at the top of every step: left to itself the
    // compiler hoists ~100 registers of loop-invariant address arithmetic out of the step loop and spills the basis)
    double* const xh_ = a.xb + C2_XH; double* const xn_ = a.xb + C2_XN; double* const xt_ = a.xb + C2_XT;
    double* const xt0_ = a.xb + C2_XT0; double* const xa_ = a.xb + C2_XA; double* const xw_ = a.xb + C2_XW;
    unsigned* const f1_ = a.fb; unsigned* const f2_ = a.fb + 32;
    double* const alphas_ = a.rec; double* const betas_ = a.rec + MAXK;
    LanczosCtl* const ctl_ = reinterpret_cast<LanczosCtl*>(a.rec + 2 * MAXK);
    int lane = threadIdx.x & 63;
    int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // row-per-lane layout: thread (row, cw) holds basis columns cw + 4c; column-per-lane layout: lane = column
    // 64 colblk + lane, rows 32 rowq .. + 31 of the workgroup
    int tid, half, cw, colblk, rowq, row, i, ic, g, colB;
    bool rowok;
    auto reindex = [&]() __attribute__((always_inline)) {
        asm volatile("" : "+v"(lane));
        asm volatile("" : "+s"(w));
        tid = w * 64 + lane;
        half = w & 1; cw = w >> 1; colblk = half; rowq = cw;
        row = half * 64 + lane;
        i = b * C2_R + row;
        rowok = i < a.npad;
        ic = rowok ? i : a.npad - 1;
        g = 2 * b + half;                              // 64-row group of `row`
        colB = colblk * 64 + lane;
    };
    reindex();

    double* sF = c2_smem;                              // [64][C2_FS]
    double* s_p = sF + 64 * C2_FS;                     // [4 waves][4 groups][128]: group sums of the measured-dot records
    double* s_t = s_p + 4 * 4 * 128;                   // [4][4][64]: the same for the Vp'u records
    double* s_m = s_t + 4 * 4 * 64;                    // [4 row quarters][128]
    double* s_tq = s_m + 4 * 128;                      // [8 row chunks][64]
    double* s_e = s_tq + 8 * 64;                       // [2][4][64]
    double* s_d = s_e + 512;
    double* s_acc = s_d + 512;
    double* s_h = s_acc + 512;
    double* s_q = s_h + C2_CO;
    double* s_al = s_q + C2_CO;
    double* s_be = s_al + C2_CO;
    double* s_f = s_be + C2_CO;
    double* s_D = s_f + C2_CO;
    double* s_u = s_D + C2_CO;                         // [64]
    double* s_lam = s_u + 64;                          // [64]
    double* s_wp = s_lam + 64;                         // [128] own rows of u (v_kfirst, then w')
    double* s_vk = s_wp + 128;                         // [128]
    double* s_sc = s_vk + 128;                         // [8]
    double* s_ev = s_sc + 8;                           // [C2_NE][512]: this thread's ELL values (off-diagonals carry 1/sqrt2)

    // ------------------------------------------------------------------ staging
    for (int idx = tid; idx < 64 * C2_R; idx += C2_TPB) {
        const int c = idx >> 7, r = idx & 127, gi = b * C2_R + r;
        sF[c * C2_FS + r] = (c < rp && gi < a.npad) ? a.Vp[(size_t)c * a.ldv + gi] : 0.0;
    }
    for (int j = tid; j < C2_CO; j += C2_TPB) {
        s_f[j] = (j < kfirst) ? a.arrow[j] : 0.0;
        s_D[j] = (j < kfirst) ? a.arrow[MAXK + j] : 0.0;
        s_h[j] = 0.0; s_q[j] = 0.0;
        s_al[j] = (resume && j >= kfirst && j < kb - 1) ? alphas_[j] : 0.0;
        s_be[j] = (resume && j >= kfirst && j < kb - 1) ? betas_[j] : 0.0;
    }
    if (tid < 64) { s_lam[tid] = (tid < rp) ? a.lam[tid] : 0.0; s_u[tid] = 0.0; }
    // this thread's ELL entries k = cw + 4q of its row: column and value (off-diagonals carry the svec 1/sqrt2)
    unsigned ecolp[C2_NE / 2];                         // two 16-bit column indices per register (side <= 4096)
#pragma unroll
    for (int q = 0; q < C2_NE; ++q) {
        const int kk = cw + 4 * q;
        int col = ic;
        double evq = 0.0;
        if (kk < a.ell_w && rowok) {
            col = a.ell_col[(size_t)kk * a.npad + i];
            const int sx = a.ell_sidx[(size_t)kk * a.npad + i];
            if (sx >= 0) { const double ev = a.esv[sx]; evq = (col == i) ? ev : ev * INV_SQRT2; }
        }
        s_ev[q * C2_TPB + tid] = evq;
        if (q & 1) ecolp[q >> 1] |= (unsigned)col << 16; else ecolp[q >> 1] = (unsigned)col;
    }
    auto ecol = [&](int q) __attribute__((always_inline)) { return (int)((q & 1) ? (ecolp[q >> 1] >> 16) : (ecolp[q >> 1] & 0xFFFFu)); };
    // basis columns that exist at the start of this launch
    const int ncol0 = resume ? kb : kfirst + 1;
    double regA[32], regB[32];
#pragma unroll
    for (int c0 = 0; c0 < 32; c0 += 8) {               // (eight loads in flight at a time: a register peak HERE makes the
#pragma unroll                                         //  allocator spill basis entries and reload them at every use in the loop)
        for (int c = c0; c < c0 + 8; ++c) regA[c] = (cw + 4 * c < ncol0 && rowok) ? a.V[(size_t)(cw + 4 * c) * a.ldv + i] : 0.0;
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int r0 = 0; r0 < 32; r0 += 8) {
#pragma unroll
        for (int r = r0; r < r0 + 8; ++r) {
            const int gi = b * C2_R + rowq * 32 + r;
            regB[r] = (colB < ncol0 && gi < a.npad) ? a.V[(size_t)colB * a.ldv + gi] : 0.0;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if (resume && ctl_->stop) return;                 // (uniform over the grid: written by the previous launch)

    unsigned epoch = a.epoch0;
    double carry = resume ? ctl_->carry : 0.0;
    double h1k = resume ? a.hsum[kb - 1] : 0.0;
    int kstop = -1;
    bool failed = false;
    const bool timing = TIMING && a.dbg != nullptr && b == 0;
    long long tk[TIMING ? 8 : 1] = {0};
    long long t0 = timing ? wall_clock64() : 0;
#define C2_TICK(q) if constexpr (TIMING) { if (timing) { const long long t1 = wall_clock64(); tk[q] += t1 - t0; t0 = t1; } }

    // the records of the exchange X2 of step k - 1 (written by this launch or the previous one) -> s_p, s_t, s_sc[0], xg
    auto fetch_x2 = [&](int kcols /* columns 0 .. kcols-1 carry dots */) __attribute__((always_inline)) {
        double hp[16], tp[8], hn = 0.0;
        const int wv4 = w >> 1;                        // records wv4 + 4u, columns 64 (w & 1) + lane
        // (records of groups >= nt are never written: zero since allocation; columns beyond the step's are masked by the caller)
        const double* xh_w = xh_ + (size_t)wv4 * C2_HLD + 64 * (w & 1) + lane;
        (void)kcols;
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            hp[u] = cy_get(xh_w + (size_t)(4 * u) * C2_HLD);
        }
        const int wt = w & 3, uh = w >> 2;             // records wt + 4u, u = 8 uh .. 8 uh + 7, columns = lane
        const double* xt_w = xt_ + (size_t)(wt + 32 * uh) * C2_TLD + lane;
#pragma unroll
        for (int u = 0; u < 8; ++u) tp[u] = cy_get(xt_w + (size_t)(4 * u) * C2_TLD);
        if (w == 0) hn = cy_get(xn_ + lane);
        double xg[C2_NE];                              // gathered entries of u = w' for this thread's ELL entries
#pragma unroll
        for (int q = 0; q < C2_NE; ++q) xg[q] = cy_get(xw_ + ecol(q));
        double q4[4];
        tree_in_wave(hp, q4);
#pragma unroll
        for (int aa = 0; aa < 4; ++aa) s_p[(wv4 * 4 + aa) * 128 + 64 * (w & 1) + lane] = q4[aa];
        s_t[(wt * 4 + 2 * uh) * 64 + lane] = (tp[0] + tp[2]) + (tp[1] + tp[3]);
        s_t[(wt * 4 + 2 * uh + 1) * 64 + lane] = (tp[4] + tp[6]) + (tp[5] + tp[7]);
        if (w == 0) { hn = wave_sum(hn); if (lane == 0) s_sc[0] = hn; }
        // this thread's share of (E u)_row, entries in ascending order (read after the next barrier)
        double e = 0.0;
#pragma unroll
        for (int q = 0; q < C2_NE; ++q) e += s_ev[q * C2_TPB + tid] * xg[q];
        s_e[(half * 4 + cw) * 64 + lane] = e;
    };
    // the Vp'u partials of this wave's 16 rows of s_wp, lane = factor column
    auto t_partial = [&]() __attribute__((always_inline)) {
        s_tq[w * 64 + lane] = c2_dot16_lds(sF + lane * C2_FS + 16 * w, s_wp + 16 * w);
    };
    // combine the row chunks of s_tq into the two groups' records (threads 256 .. 383)
    auto t_store = [&](double* xt) __attribute__((always_inline)) {
        if (tid >= 256 && tid < 384) {
            const int h = (tid - 256) >> 6, c = tid & 63, gg = 2 * b + h;
            if (c < rp && gg < nt)
                cy_put(xt + (size_t)gg * C2_TLD, c, (s_tq[(4 * h) * 64 + c] + s_tq[(4 * h + 1) * 64 + c]) +
                                                    (s_tq[(4 * h + 2) * 64 + c] + s_tq[(4 * h + 3) * 64 + c]));
        }
    };
    // sum_{c} regA[c] * coef[cw + 4c] in the step kernels' two accumulators (coef is zero beyond the live columns)
    auto rowdot = [&](const double* __restrict__ coef, int ncols, double& d0, double& d1) __attribute__((always_inline)) {
        d0 = 0.0; d1 = 0.0;
#pragma unroll
        for (int c0 = 0; c0 < 32; c0 += 8) {           // batches of eight coefficient reads: all 32 at once cost 64 registers
            if (cw + 4 * c0 < ncols) {                 // and push the basis into scratch memory
                double q[8];
#pragma unroll
                for (int c = 0; c < 8; ++c) q[c] = coef[cw + 4 * (c0 + c)];
#pragma unroll
                for (int c = 0; c < 8; c += 2) {
                    d0 += regA[c0 + c] * q[c];
                    d1 += regA[c0 + c + 1] * q[c + 1];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // ------------------------------------------------------------------ state at the top of the first step
    if (!resume) {
        // u = v_kfirst (exact, normalised): own rows into s_wp, the gathered entries straight from the basis
        const double* vk0 = a.V + (size_t)kfirst * a.ldv;
        if (cw == 0) s_wp[row] = rowok ? vk0[i] : 0.0;
        double e = 0.0;
#pragma unroll
        for (int q = 0; q < C2_NE; ++q) e += s_ev[q * C2_TPB + tid] * vk0[ecol(q)];
        s_e[(half * 4 + cw) * 64 + lane] = e;
        __syncthreads();
    } else {
        if (cw == 0) s_wp[row] = rowok ? xw_[i] : 0.0;
        fetch_x2(kb);
        __syncthreads();
        if (tid < 128) s_h[tid] = (tid < kb) ? tree_across(s_p, 128, tid) : 0.0;
        __syncthreads();
    }

    // one step; FIRST = the cycle's first step (u = v_kfirst exact: no closing, the Vp'u records travel with X1), peeled out of
    // the loop so that the loop body has ONE shape (the register allocator shuffles / spills the basis at every join otherwise).
    // Returns false when the step loop ends here.
    auto step = [&](auto first_tag, int k) __attribute__((always_inline)) -> bool {
        constexpr bool first = decltype(first_tag)::value;
        reindex();
        double beta = 1.0;
        // ================= close step k-1 =================
        if constexpr (!first) {
            double hh = 0.0;
            for (int j = lane; j < k; j += WAVE) hh += s_h[j] * s_h[j];
            hh = wave_sum(hh);                         // same value, same order in every wave of every workgroup
            beta = sqrt(fmax(s_sc[0] - hh, 0.0));
            const double hk1 = s_h[k - 1];
            const double cu = (k - 1 > kfirst) ? carry : 0.0;
            if (tid == 0) {
                const double al = h1k + hk1 - cu;
                s_al[k - 1] = al; s_be[k - 1] = beta;
                if (b == 0) { alphas_[k - 1] = al; betas_[k - 1] = beta; }
            }
            carry = hk1;
            if (beta <= a.tol) { kstop = k; return false; }   // invariant subspace (uniform over the grid)
            double d0, d1;
            rowdot(s_h, k, d0, d1);
            s_d[(half * 4 + cw) * 64 + lane] = d0 + d1;
        }
        if (k == kd) {                                 // the cycle's last closing: v_kd, no further mat-vec
            __syncthreads();
            if (cw == 0 && rowok) {
                const double tot = (s_d[(half * 4) * 64 + lane] + s_d[(half * 4 + 1) * 64 + lane]) +
                                   (s_d[(half * 4 + 2) * 64 + lane] + s_d[(half * 4 + 3) * 64 + lane]);
                a.V[(size_t)k * a.ldv + i] = (s_wp[row] - tot) / beta;
            }
            return false;
        }
        // ================= operator pieces on u (v_k at the start of a cycle, else w'_{k-1}) =================
        if constexpr (first) t_partial();                        // (s_e: filled where the gathered entries of u landed)
        __syncthreads();                                                                            // (B1)
        double ei = 0.0;
        if (cw == 0) {
            ei = (s_e[(half * 4) * 64 + lane] + s_e[(half * 4 + 1) * 64 + lane]) +
                 (s_e[(half * 4 + 2) * 64 + lane] + s_e[(half * 4 + 3) * 64 + lane]);
            const double ui = s_wp[row];
            if constexpr (!first) {
                const double tot = (s_d[(half * 4) * 64 + lane] + s_d[(half * 4 + 1) * 64 + lane]) +
                                   (s_d[(half * 4 + 2) * 64 + lane] + s_d[(half * 4 + 3) * 64 + lane]);
                const double vk = (ui - tot) / beta;                                        // v_k = (w' - V h2) / beta
                s_vk[row] = vk;
                if (rowok) a.V[(size_t)k * a.ldv + i] = vk;
            }
            const double av = wave_sum(ui * ei);                                             // u'E u of this group
            if (lane == 0 && g < nt) cy_put(xa_, g, av);
        }
        if constexpr (first) t_store(xt0_);
        C2_TICK(0)
        // ---- X1: publish
        ++epoch;
        cy_publish(f1_, b, epoch);                                                                 // (P1)
        // ---- everything of the recurrence that does not need alpha
        const double binv = first ? 1.0 : 1.0 / beta;
        double tl = 0.0, d0 = 0.0, d1 = 0.0, upart = 0.0;
        auto pre_alpha = [&]() __attribute__((always_inline)) {
            // t = Vp'u from the group sums in s_t;  tl = t' Lam t;  s_u = Lam t
            {
                const double t0v = tree_across(s_t, 64, lane);
                tl = (lane < rp) ? s_lam[lane] * t0v * t0v : 0.0;
                tl = wave_sum(tl);
                if (tid < 64) s_u[tid] = (tid < rp) ? s_lam[tid] * t0v : 0.0;
            }
            // coefficients of the predicted pass over V_{k-1} (column k, the one that needs alpha, is added below)
            const int j = tid;
            if constexpr (first) {
                if (j < C2_CO) s_q[j] = (j < kfirst) ? s_f[j] : 0.0;
            } else {
                double fh = 0.0;
                if (kfirst > 0 && k > kfirst) {           // f'h (row `kfirst` of the arrow)
                    if (lane < kfirst) fh = s_f[lane] * s_h[lane];
                    if (lane + WAVE < kfirst) fh += s_f[lane + WAVE] * s_h[lane + WAVE];
                    fh = wave_sum(fh);
                }
                if (j < k) {
                    const double hj = s_h[j];
                    double t;
                    if (j < kfirst) {
                        t = s_D[j] * hj + (k > kfirst ? s_f[j] * s_h[kfirst] : 0.0);
                    } else {
                        t = s_al[j] * hj;
                        if (j + 1 < k) t += s_be[j] * s_h[j + 1];
                        if (j == kfirst) t += fh;                             // fh = 0 when kfirst == 0
                        else if (j > 0) t += s_be[j - 1] * s_h[j - 1];
                    }
                    s_q[j] = t * binv + (j == k - 1 ? beta : 0.0);
                } else if (j < C2_CO) s_q[j] = 0.0;
            }
            __syncthreads();                                                                        // (B2)
            rowdot(s_q, k, d0, d1);
            __builtin_amdgcn_sched_barrier(0);         // (one batch of LDS coefficient reads at a time: registers)
            {
                double u0 = 0.0, u1 = 0.0;
#pragma unroll
                for (int c = 0; c < 16; c += 2) {
                    u0 += sF[(cw + 4 * c) * C2_FS + row] * s_u[cw + 4 * c];
                    u1 += sF[(cw + 4 * (c + 1)) * C2_FS + row] * s_u[cw + 4 * (c + 1)];
                }
                upart = u0 + u1;
            }
        };
        if constexpr (!first) pre_alpha();
        __builtin_amdgcn_sched_barrier(0);
        // v_k into the two register layouts (ONE site: two copies of this code leave the register arrays in different places
        // on the two paths and the join shuffles / spills them)
        // (branch-free: conditional updates of 64 values become 64 phi nodes, and the allocator answers with copies and spills)
        if constexpr (!first) {
            {
                double v = s_vk[row];
                asm volatile("" : "+v"(v));
                int cidx = (cw == (k & 3)) ? (k >> 2) : -1;
                asm volatile("" : "+v"(cidx));         // (a vector compare per element; 32 scalar masks would spill SGPRs)
#pragma unroll
                for (int c = 0; c < 32; ++c) regA[c] = (c == cidx) ? v : regA[c];
            }
            __builtin_amdgcn_sched_barrier(0);
            {
                const bool me = colblk == (k >> 6) && lane == (k & 63);
#pragma unroll
                for (int r0 = 0; r0 < 32; r0 += 8) {  // (eight LDS reads in flight at a time: registers)
                    double v[8];
#pragma unroll
                    for (int r = 0; r < 8; ++r) {
                        v[r] = s_vk[rowq * 32 + r0 + r];
                        asm volatile("" : "+v"(v[r]));   // (a select of VALUES: left alone the compiler selects between the LDS
                    }                                    //  and the register array's ADDRESS, which pins the array in scratch)
#pragma unroll
                    for (int r = 0; r < 8; ++r) regB[r0 + r] = me ? v[r] : regB[r0 + r];
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        C2_TICK(1)
        // ---- X1: wait, fetch
        {
            bool ok = true;
            if (w == 0) ok = cy_wait(f1_, G, epoch, lane);
            if (__syncthreads_or(!ok)) { failed = true; return false; }                                    // (P2)
        }
        double ap = cy_get(xa_ + lane);
        if constexpr (first) {
            // the Vp'u records of the cycle's first vector came with X1
            const int wt = w & 3, uh = w >> 2;
            double tp[8];
            const double* xt_w = xt0_ + (size_t)(wt + 32 * uh) * C2_TLD + lane;
#pragma unroll
            for (int u = 0; u < 8; ++u) tp[u] = cy_get(xt_w + (size_t)(4 * u) * C2_TLD);
            s_t[(wt * 4 + 2 * uh) * 64 + lane] = (tp[0] + tp[2]) + (tp[1] + tp[3]);
            s_t[(wt * 4 + 2 * uh + 1) * 64 + lane] = (tp[4] + tp[6]) + (tp[5] + tp[7]);
            __syncthreads();
            pre_alpha();
        }
        C2_TICK(2)
        ap = wave_sum(ap);
        const double alpha = (tl + ap) * binv * binv;
        double ck = alpha;
        if constexpr (!first) ck -= s_h[k - 1];
        h1k = ck;
        // column k of the predicted pass: the one term that needed alpha
        if (cw == (k & 3)) {
            const double vkr = first ? s_wp[row] : s_vk[row];
            if (((k >> 2) & 1) == 0) d0 += vkr * ck; else d1 += vkr * ck;
        }
        {
            double dsub = d0 + d1;
            dsub -= upart * binv;
            s_acc[(half * 4 + cw) * 64 + lane] = dsub;
        }
        __syncthreads();                                                                            // (B3)
        ++epoch;
        if (cw == 0) {
            const double wi = ei * binv;
            const double wp = wi - ((s_acc[(half * 4) * 64 + lane] + s_acc[(half * 4 + 1) * 64 + lane]) +
                                    (s_acc[(half * 4 + 2) * 64 + lane] + s_acc[(half * 4 + 3) * 64 + lane]));
            s_wp[row] = wp;
            if (rowok) cy_put(xw_, i, wp);
            const double r2 = wave_sum(wp * wp);
            if (lane == 0 && g < nt) cy_put(xn_, g, r2);
            if (b == 0 && tid == 0) a.hsum[k] = ck;
        }
        __syncthreads();                                                                            // (B4)
        // ---- measured pass V_k'w' (lane = basis column, this wave's 32 rows) and the Vp'w' partials of step k + 1
        if (64 * colblk <= k) {
            const double da = c2_dot16_reg<0>(regB, s_wp + rowq * 32);
            __builtin_amdgcn_sched_barrier(0);
            const double db = c2_dot16_reg<16>(regB, s_wp + rowq * 32 + 16);
            s_m[rowq * 128 + colB] = c2_add(da, db);
        }
        t_partial();
        __syncthreads();                                                                            // (B5)
        if (tid < 256) {
            const int h = tid >> 7, j = tid & 127, gg = 2 * b + h;
            if (j <= k && gg < nt) cy_put(xh_ + (size_t)gg * C2_HLD, j, s_m[(2 * h) * 128 + j] + s_m[(2 * h + 1) * 128 + j]);
        }
        t_store(xt_);
        C2_TICK(3)
        cy_publish(f2_, b, epoch);                                                                 // (P3)
        if (k == ke) return false;                            // the next launch goes on from the exchange buffers
        {
            bool ok = true;
            if (w == 0) ok = cy_wait(f2_, G, epoch, lane);
            if (__syncthreads_or(!ok)) { failed = true; return false; }                                    // (P4)
        }
        C2_TICK(4)
        fetch_x2(k + 1);
        __syncthreads();                                                                            // (B6)
        if (tid < 128) s_h[tid] = (tid <= k) ? tree_across(s_p, 128, tid) : 0.0;
        __syncthreads();                                                                            // (B7)
        C2_TICK(5)
        return true;
    };
    {
        int k = kb;
        bool go = true;
        if (!resume) { go = step(std::true_type{}, k); ++k; }
        for (; go && k <= ke; ++k) go = step(std::false_type{}, k);
    }
    if (failed) {
        if (tid == 0) atomicExch(a.err, 1);
        return;
    }
    if (b == 0 && tid == 0) {
        ctl_->carry = carry;
        if (kstop >= 0) { ctl_->kstop = kstop; ctl_->stop = 1; }
    }
    if constexpr (TIMING) {
        if (timing && tid == 0) {
            unsigned xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            atomicOr((unsigned long long*)(a.dbg + 15), 1ull << (xcc & 15));
            for (int q = 0; q < 8; ++q) atomicAdd((unsigned long long*)(a.dbg + q), (unsigned long long)tk[q]);
        }
    }
#undef C2_TICK
}

}  // namespace dev
}  // namespace proxsdp
