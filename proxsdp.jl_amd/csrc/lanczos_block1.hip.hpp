// One whole Lanczos cycle of a MEDIUM PSD block (side <= 512) in ONE launch of ONE workgroup (round 6).
//
// Why: every Lanczos step of the step kernels (kernels.hip.hpp) is two dependent launches of ~6 us each whatever the
// side; a block of side 101 .. 500 -- all but three instances of the reference's own benchmark script, test/runbench.jl --
// therefore spent ~370 us per PDHG iteration on 25-30 mat-vecs of a matrix that fits in one CU's caches.  Here one
// workgroup of 1024 threads owns the whole block for the whole cycle: the Krylov basis stays in REGISTERS (thread <-> row,
// wave <-> column residue, exactly the tiling of the step kernels), every record the step kernels exchange through
// global memory (mat-vec slots, partial dots, w', the reduced coefficients) lives in LDS together with the operator (the
// support matrix E in ELL form, or the packed triangle when it fits), and a kernel boundary becomes a workgroup barrier.
// Nothing is stored to global memory inside the step loop (a store would have to be acknowledged before the next barrier).
// The restart rotation V <- V U of the PREVIOUS cycle runs in the prologue, straight into the registers that hold the
// basis (the coefficients U and the arrow part are read from the host's pinned staging buffer): one launch per cycle.
//
// The ARITHMETIC is that of the step kernels, term by term and in the same order: the workgroup is organised as NV = 4
// "virtual workgroups" of 4 waves; virtual workgroup b does for row group / tile b, b + 4, ... exactly what workgroup b of
// a step kernel does (symv_load / symv_reduce, the bodies of k_fop, k_lz_orth<1, .>, k_lz_finish<1>, k_lz_rotate), and the
// sums that every real workgroup repeats redundantly (alpha, the reduced dots, beta) are formed once, by virtual workgroup
// 0, in the order the step kernels use.  Results, mat-vec counts and restart counts are therefore those of the step
// kernels BIT FOR BIT (tests/test_gpu_parity.py::test_block_cycle_kernel_reproduces_the_step_kernels_bit_for_bit); what
// changes is time.
//
// Replaces: the BLAS-1 work and the mat-vecs of one KrylovKit Lanczos cycle (call site /root/reference/src/eigsolver.jl:802-812,
// dsymv at :678) and its basistransform!, as k_symv_finish / k_fop_finish + k_lz_orth + k_lz_rotate do.  The K x K
// eigensolve and the restart logic stay on the host (Solver::lz_after_cycle), unchanged.
//
// Limits (Solver::block1_plan): side <= 512 (8 row groups; auto: <= 256, one row group per virtual workgroup -- with two, a step
// costs what two launches cost), Krylov dimension <= 31, operator form with <= 16 factor columns and no hub-row overflow list, or
// the packed triangle resident in LDS (side <= ~140); anything else takes the step kernels.
//
// Measured (profiles/r06_medium_blocks.md): 4.3-4.8 us per step on the operator form, 7.6 us on the packed operator, against ~12 us
// for the two launches; prologue (rotation + operator staging) 5-8 us, epilogue 0.6-1.1 us.  The bound is VALU issue of ONE CU:
// ~600 instructions per wave and step for 16 waves on 4 SIMDs.  Two things that cost a factor each while this was written: a
// `for c: if (c == cn) v[c] = x` loop over a register array is turned into one indexed store and demotes the array (the Krylov
// basis) to scratch memory (-> b1_set_slot); a global store inside the step loop has to be acknowledged before the next barrier
// (-> everything is written once, in the epilogue).
#pragma once
#include "kernels.hip.hpp"
#include <utility>

namespace proxsdp {
namespace dev {

constexpr int B1_NV = 4;                 // virtual workgroups (4 waves each)
constexpr int B1_TPB = B1_NV * TPB;      // 1024 threads
constexpr int B1_NC = 8;                 // basis columns held per wave (column j = wv + 4c)
constexpr int B1_NPV = 4;                // factor columns held per wave: rank of the previous projection <= 16
constexpr int B1_MAXNT = 8;              // side <= 512
constexpr int B1_KMAX = 31;              // Krylov dimension (k_lz_rotate is the scalar kernel below K = 32)
constexpr int B1_FOLD_LD = 72;           // column sums through LDS (b1_fold4): 64 rows + 2 x 4 padding per column ...
constexpr int B1_FOLD_STAGE = 4 * B1_FOLD_LD;   // ... four columns per wave

struct Block1Args {
    const double* xp;                    // packed block of the iterate (packed operator)
    int n, nt, npad;
    const double* Vin; double* Vout; int ldv;   // basis before (columns 0 .. kfirst, or 0 .. rotK when a rotation is pending) / after the cycle
    int kfirst, kd;                      // steps kfirst .. kd-1; the cycle ends with column kd
    double tol;
    const double* Vp; const double* lam; int rp;      // operator form: previous projection's factors
    const int* ell_col; const int* ell_sidx; int ell_w; const double* esv;
    int ell_in_lds;
    const double* arrow;                 // f | D: (MAXK each) on the device, or f at [0, kfirst), D at [MAXK, ..) behind U in the staging buffer
    double* alphas; double* betas; LanczosCtl* ctl;
    // pending restart rotation (rotK > 0): columns [0, kfirst) = Vin[:, 0..rotK) U, column kfirst = Vin[:, rotK]
    const double* U; int rotK;
    long long* dbg;                      // optional (PROXSDP_HIP_DEBUG_B1): accumulated 100 MHz ticks -- prologue | step loop | epilogue | steps | launches
};

// LDS plan (doubles).  Per virtual workgroup scratch is a union over the phases.
struct B1Lds {
    int vec, nw, X, Pp, Ap, eb, tp, apf, ellc, ellf, hp, hn, q, u, h2, hsum, al, be, f, D, red, common, vw, vw_stride, fold, total;
};
__host__ __device__ inline B1Lds b1_lds_plan(int nt, int npad, bool fop, long long xN, int ell_w_lds) {
    B1Lds L{};
    int o = 0;
    auto take = [&](long long cnt) { const int at = o; o += (int)((cnt + 1) & ~1LL); return at; };
    L.vec = take(npad);                          // operand of the mat-vec: v_k (first step) or w'
    L.nw = take(npad);                           // the new basis column, wave 0 -> the wave that holds it
    L.X = take(fop ? 0 : xN);                    // packed triangle of the block
    L.Pp = take(fop ? 0 : (long long)nt * npad); // mat-vec slots [slot][row]
    L.Ap = take(fop ? 0 : 64);                   // tiles' shares of w'P~w'
    L.eb = take(fop ? npad : 0);                 // (E v)_i
    L.tp = take(fop ? nt * 64 : 0);              // Vp'v partials [g][column]
    L.apf = take(fop ? 64 : 0);                  // v'Ev partials [g]
    L.ellc = take(fop ? ((long long)ell_w_lds * npad + 1) / 2 : 0);     // ELL columns (int) ...
    L.ellf = take(fop ? (long long)ell_w_lds * npad : 0);               // ... and coefficients (value, x 1/sqrt2 off the diagonal)
    L.hp = take(2 * nt * 64);                    // partial dots V'w' [parity][g][column]
    L.hn = take(2 * 64);                         // |w'|^2 partials [parity][g]
    L.q = take(64); L.u = take(64); L.h2 = take(66); L.hsum = take(64);
    L.al = take(64); L.be = take(64); L.f = take(64); L.D = take(64); L.red = take(8);
    L.common = take(NWAVE * 4 * 64);             // s_t | s_p of virtual workgroup 0; the rotation's U in the prologue
    L.vw_stride = 4 * NWAVE * LZ_ROWS;           // [0, 512): closing row sums / recurrence row sums | [512, 1024): (E v) partials / tile row+col sums
    L.vw = take(B1_NV * L.vw_stride);
    L.fold = take(B1_NV * NWAVE * B1_FOLD_STAGE);   // per wave: four columns x 64 rows (padded) of products for the column sums
    L.total = o;
    return L;
}

// ---- mat-vec tile, reduce phase: symv_reduce with the barriers hoisted out of the tile test (a virtual workgroup without a
// tile in this round still meets them).  Same arithmetic, same order.
__device__ __forceinline__ void b1_symv_reduce(bool active, int npad, const double* __restrict__ v, double* __restrict__ Ppart,
                                               int tile, int lane, int wv, double (&t)[CPW],
                                               double* __restrict__ s_row, double* __restrict__ s_col,
                                               double* __restrict__ Apart) {
    int I = 0, J = 0;
    if (active) tile_coords(tile, I, J);
    const int gi = I * TILE + lane;
    const int j0 = J * TILE + wv * CPW;
    const bool diag = (I == J);
    double vi = 0.0, cs = 0.0, aw = 0.0;
    if (active) {
        const double* __restrict__ vJ = v + j0;
        vi = v[gi];
        double racc = 0.0;
#pragma unroll
        for (int c = 0; c < CPW; ++c) {
            racc += t[c] * vJ[c];
            t[c] *= vi;
            if (diag && gi == j0 + c) t[c] = 0.0;
        }
        s_row[wv * TILE + lane] = racc;
        aw = diag ? 0.0 : wave_sum(vi * racc);
        fold_stage<8>(t, lane);
        fold_stage<4>(t, lane);
        fold_stage<2>(t, lane);
        fold_stage<1>(t, lane);
        cs = add_xor16(t[0]);
        cs = add_xor32(cs);
    }
    __syncthreads();
    if (active) {
        if (diag) { if (lane < CPW) s_col[wv * CPW + lane] = cs; }
        else if (lane == 0) s_col[wv] = aw;
    }
    __syncthreads();
    if (!active) return;
    if (diag) {
        if (wv == 0) {
            const double rs = (s_row[lane] + s_row[TILE + lane]) + (s_row[2 * TILE + lane] + s_row[3 * TILE + lane]);
            Ppart[I * npad + gi] = rs + s_col[lane];
            const double a = wave_sum(vi * (rs + s_col[lane]));
            if (lane == 0) Apart[tile] = a;
        }
    } else {
        if (wv == 0) {
            const double rs = (s_row[lane] + s_row[TILE + lane]) + (s_row[2 * TILE + lane] + s_row[3 * TILE + lane]);
            Ppart[J * npad + gi] = rs;
            if (lane == 0) Apart[tile] = 2.0 * ((s_col[0] + s_col[1]) + (s_col[2] + s_col[3]));
        }
        if (lane < CPW) Ppart[I * npad + j0 + lane] = cs;
    }
}

// v[cn] = nv without a loop over the slots: a loop `for c: if (c == cn) v[c] = nv` is rewritten by the optimiser into ONE store
// at a run-time index, which sends the whole register array to scratch memory (seen in the ISA: every row sum then re-read
// the basis through scratch loads)
template <int... Cs>
__device__ __forceinline__ void b1_set_slot(double (&v)[B1_NC], int cn, double nv, std::integer_sequence<int, Cs...>) {
    ((v[Cs] = (Cs == cn) ? nv : v[Cs]), ...);
}

// Column sums of FOUR columns of products over the wave's 64 rows, with the additions of the step kernels' fold network
// (fold16_all: per column a balanced tree that pairs the rows by bit 3, then 2, 1, 0, then 4, then 5 -- addition is commutative,
// so the tree alone fixes the bits), computed through an LDS transpose instead of 17 DPP exchanges of 64-bit values: the fold
// network is ~140 instructions per use, this ~30, and one CU issues every instruction of the step (profiles/r06_medium_blocks.md).
// Lane (c, a, m) = (lane >> 4, (lane >> 2) & 3, lane & 3) reads rows 16 a + m + {0, 4, 8, 12} of column c: levels "bit 3" and
// "bit 2" are in the lane, "bit 1" / "bit 0" between lanes m, "bit 4" / "bit 5" between lanes a.  Returns the sum of column
// lane >> 4 (in every lane of that 16-group).  Rows are stored with two doubles of padding per 16 (bank conflicts).
__device__ __forceinline__ double b1_fold4(double t0, double t1, double t2, double t3, double* __restrict__ stage, int lane) {
    const int wr = lane + 2 * (lane >> 4);
    __builtin_amdgcn_wave_barrier();
    stage[wr] = t0; stage[B1_FOLD_LD + wr] = t1; stage[2 * B1_FOLD_LD + wr] = t2; stage[3 * B1_FOLD_LD + wr] = t3;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    const double* rd = stage + (lane >> 4) * B1_FOLD_LD + 18 * ((lane >> 2) & 3) + (lane & 3);
    const double x0 = rd[0], x4 = rd[4], x8 = rd[8], x12 = rd[12];
    double y = (x0 + x8) + (x4 + x12);            // bit 3, then bit 2
    y += lane_xor<2>(y);                          // bit 1
    y += lane_xor<1>(y);                          // bit 0
    y += lane_xor<4>(y);                          // bit 4 (a ^ 1)
    y += lane_xor<8>(y);                          // bit 5 (a ^ 2)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    return y;
}

// RG = row groups per virtual workgroup (1: side <= 256, 2: side <= 512);  FOP = operator form
template <int RG, bool FOP>
__global__ void __launch_bounds__(B1_TPB)
k_lz_block1(Block1Args a) {
    extern __shared__ double b1_sm[];
    const int tid = threadIdx.x & (TPB - 1);
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)((threadIdx.x >> 6) & 3));
    const int vb = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8));
    const int nt = a.nt, npad = a.npad, ldv = a.ldv, kfirst = a.kfirst, kd = a.kd, keep = a.kfirst;
    const long long xN = (long long)a.n * (a.n + 1) / 2;
    const B1Lds L = b1_lds_plan(nt, npad, FOP, xN, a.ell_in_lds ? a.ell_w : 0);
    double* const s_vec = b1_sm + L.vec;
    double* const s_new = b1_sm + L.nw;
    double* const s_X = b1_sm + L.X;
    double* const s_Pp = b1_sm + L.Pp;
    double* const s_Ap = b1_sm + L.Ap;
    double* const s_eb = b1_sm + L.eb;
    double* const s_tp = b1_sm + L.tp;
    double* const s_apf = b1_sm + L.apf;
    int* const s_ellc = reinterpret_cast<int*>(b1_sm + L.ellc);
    double* const s_ellf = b1_sm + L.ellf;
    double* const s_hp = b1_sm + L.hp;
    double* const s_hn = b1_sm + L.hn;
    double* const s_q = b1_sm + L.q;
    double* const s_u = b1_sm + L.u;
    double* const s_h2 = b1_sm + L.h2;            // [0, 64): reduced dots of the last closed step (zero from its column count on), [64]: |w'|^2
    double* const s_hsum = b1_sm + L.hsum;
    double* const s_al = b1_sm + L.al;
    double* const s_be = b1_sm + L.be;
    double* const s_f = b1_sm + L.f;
    double* const s_D = b1_sm + L.D;
    double* const s_red = b1_sm + L.red;
    double* const s_com = b1_sm + L.common;
    double* const s_vw = b1_sm + L.vw + vb * L.vw_stride;
    double* const s_vwB = s_vw + 2 * NWAVE * LZ_ROWS;
    double* const s_fold = b1_sm + L.fold + (vb * NWAVE + wv) * B1_FOLD_STAGE;

    if (a.ctl->stop) return;
    const long long tk0 = (a.dbg != nullptr && threadIdx.x == 0) ? (long long)wall_clock64() : 0;
    // ---- prologue
    const bool rot = a.rotK > 0;
    if (threadIdx.x < 64) {
        const int j = threadIdx.x;
        s_f[j] = a.arrow[j]; s_D[j] = a.arrow[MAXK + j];
        s_hn[j] = 0.0; s_hn[64 + j] = 0.0;
        s_al[j] = 0.0; s_be[j] = 0.0; s_hsum[j] = 0.0; s_h2[j] = 0.0;
        if constexpr (FOP) s_apf[j] = 0.0; else s_Ap[j] = 0.0;
    }
    if (rot) for (int t = threadIdx.x; t < a.rotK * kfirst; t += B1_TPB) s_com[t] = a.U[t];      // [c][j], K x keep
    if constexpr (FOP) {
        if (a.ell_in_lds) {
            // (eight elements per thread in flight: the value load depends on the index load)
            const int tot = a.ell_w * npad;
            for (int t0 = threadIdx.x; t0 < tot; t0 += 8 * B1_TPB) {
                int col[8], sx[8];
                double ev[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int t = min(t0 + u * B1_TPB, tot - 1);
                    col[u] = a.ell_col[t]; sx[u] = a.ell_sidx[t];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) ev[u] = a.esv[max(sx[u], 0)];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int t = t0 + u * B1_TPB;
                    if (t < tot) {
                        s_ellc[t] = (sx[u] >= 0) ? col[u] : -1;
                        s_ellf[t] = (col[u] == t % npad) ? ev[u] : ev[u] * INV_SQRT2;
                    }
                }
            }
        }
    } else {
        for (long long t0 = threadIdx.x; t0 < xN; t0 += 8 * B1_TPB) {
            double xv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) xv[u] = a.xp[min(t0 + (long long)u * B1_TPB, xN - 1)];
#pragma unroll
            for (int u = 0; u < 8; ++u) if (t0 + (long long)u * B1_TPB < xN) s_X[t0 + (long long)u * B1_TPB] = xv[u];
        }
    }
    double vr[RG][B1_NC];
    double vrp[RG][FOP ? B1_NPV : 1];
    if (rot) __syncthreads();                    // U staged
#pragma unroll
    for (int r = 0; r < RG; ++r) {
        const int g = vb + B1_NV * r;
        const int i = min(g, nt - 1) * LZ_ROWS + lane;
        if (!rot) {
#pragma unroll
            for (int c = 0; c < B1_NC; ++c) vr[r][c] = (g < nt && wv + 4 * c <= kfirst) ? a.Vin[(long long)(wv + 4 * c) * ldv + i] : 0.0;
        } else {
            // k_lz_rotate: out[:, c] = (sum over even j) + (sum over odd j) of V[:, j] U[j, c], each a sequential chain; column c
            // belongs to wave c & 3 -- the wave that computes it -- so the result lands in the register that holds it
            const int K = a.rotK;
            const double vK = (g < nt) ? a.Vin[(long long)K * ldv + i] : 0.0;
            double a0[B1_NC], a1[B1_NC];
#pragma unroll
            for (int c = 0; c < B1_NC; ++c) { a0[c] = 0.0; a1[c] = 0.0; }
#pragma unroll
            for (int jh = 0; jh < 32; jh += 16) {          // two halves of the old basis in flight (registers): the chains run on in order
                double tv[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) tv[j] = (g < nt && jh + j < K) ? a.Vin[(long long)(jh + j) * ldv + i] : 0.0;
#pragma unroll
                for (int c = 0; c < B1_NC; ++c) {
                    const int col = wv + 4 * c;
                    if (col < kfirst) {
                        const double* u = s_com + col * K + jh;
#pragma unroll
                        for (int j = 0; j < 16; j += 2) {
                            if (jh + j + 1 < K) { a0[c] += tv[j] * u[j]; a1[c] += tv[j + 1] * u[j + 1]; }
                            else if (jh + j < K) a0[c] += tv[j] * u[j];
                        }
                    }
                }
            }
#pragma unroll
            for (int c = 0; c < B1_NC; ++c) {
                const int col = wv + 4 * c;
                vr[r][c] = (col < kfirst) ? (a0[c] + a1[c]) : (col == kfirst ? vK : 0.0);
            }
            if (g < nt && wv == (kfirst & 3)) s_vec[i] = vK;
        }
        if constexpr (FOP) {
#pragma unroll
            for (int c = 0; c < B1_NPV; ++c) vrp[r][c] = a.Vp[(long long)min(wv + 4 * c, max(a.rp - 1, 0)) * ldv + i];
        } else vrp[r][0] = 0.0;
    }
    if (!rot) for (int i = threadIdx.x; i < npad; i += B1_TPB) s_vec[i] = a.Vin[(long long)kfirst * ldv + i];
    double lam_j = 0.0, lam_l0 = 0.0;
    if constexpr (FOP) { lam_j = a.lam[min(tid, MAXK - 1)]; lam_l0 = a.lam[lane]; }
    double carry = 0.0;                      // ctl->carry of the step kernels (thread 0 keeps it)
    int stop_k = -1;
    int kdone = kfirst;                      // columns 0 .. kdone of the basis exist when the loop ends
    __syncthreads();

    const int ntile = nt * (nt + 1) / 2;
    const int rounds = (ntile + B1_NV - 1) / B1_NV;
    const long long tk1 = (a.dbg != nullptr && threadIdx.x == 0) ? (long long)wall_clock64() : 0;

    // ---------------------------------------------------------------------------------------------------------------------
    // pieces of a step (see the step kernels for the arithmetic; the comments name the body each piece restates)
    // F0: lz_finish_body -- this wave's share of the producers' records of step kc, |w'|^2
    auto F0 = [&](int kc) {
        const double* hp_in = s_hp + (kc & 1) * nt * 64;
        const double* hn_in = s_hn + (kc & 1) * 64;
        double pr[16], q4[4];
#pragma unroll
        for (int u = 0; u < 16; ++u) pr[u] = (wv + NWAVE * u < nt) ? hp_in[(wv + NWAVE * u) * 64 + lane] : 0.0;
        tree_in_wave(pr, q4);
#pragma unroll
        for (int q = 0; q < 4; ++q) s_com[(wv * 4 + q) * 64 + lane] = q4[q];
        if (wv == 0) {
            double hn = hn_in[lane];
            hn = wave_sum(hn);
            if (lane == 0) s_h2[64] = hn;
        }
    };
    auto F1 = [&](int kc) {                      // threads tid < 64 of virtual workgroup 0
        const double hj = tree_across(s_com, 64, tid);
        s_h2[tid] = (tid < kc + 1) ? hj : 0.0;
    };
    // F2: beta, V h2 row sums, alpha_kc / beta_kc; returns beta
    auto F2 = [&](int kc) -> double {
        const int kk = kc + 1;
        double hh = 0.0;
        for (int j = lane; j < kk; j += WAVE) hh += s_h2[j] * s_h2[j];
        hh = wave_sum(hh);
        const double beta = sqrt(fmax(s_h2[64] - hh, 0.0));
#pragma unroll
        for (int r = 0; r < RG; ++r) {
            double d0 = 0.0, d1 = 0.0;
#pragma unroll
            for (int c = 0; c < B1_NC; c += 2) {
                d0 += vr[r][c] * s_h2[wv + 4 * c];
                d1 += vr[r][c + 1] * s_h2[wv + 4 * (c + 1)];
            }
            s_vw[r * (NWAVE * LZ_ROWS) + wv * LZ_ROWS + lane] = d0 + d1;
        }
        if (threadIdx.x == 0) {
            const double al = s_hsum[kc] + s_h2[kc] - ((kc > kfirst) ? carry : 0.0);
            s_al[kc] = al; s_be[kc] = beta;
            carry = s_h2[kc];
        }
        return beta;
    };
    auto F3 = [&](int kc, double beta) {         // wave 0 of every virtual workgroup: v_{kc+1} rows
#pragma unroll
        for (int r = 0; r < RG; ++r) {
            const int g = vb + B1_NV * r;
            if (g < nt) {
                const int i = g * LZ_ROWS + lane;
                const double* sd = s_vw + r * (NWAVE * LZ_ROWS);
                const double wi = s_vec[i] - ((sd[lane] + sd[LZ_ROWS + lane]) + (sd[2 * LZ_ROWS + lane] + sd[3 * LZ_ROWS + lane]));
                s_new[i] = wi / beta;
            }
        }
    };
    auto F4 = [&](int kn) {                      // the wave that holds column kn takes it into its registers
        if (wv == (kn & 3)) {
            const int cn = kn >> 2;
#pragma unroll
            for (int r = 0; r < RG; ++r) {
                const int g = vb + B1_NV * r;
                const double nv = (g < nt) ? s_new[g * LZ_ROWS + lane] : 0.0;
                b1_set_slot(vr[r], cn, nv, std::make_integer_sequence<int, B1_NC>{});
            }
        }
    };
    // S0 / S1: fop_body -- (E v)_i partials per wave, Vp'v partials; then (E v)_i and the v'Ev partial
    auto S0 = [&]() {
#pragma unroll
        for (int r = 0; r < RG; ++r) {
            const int g = vb + B1_NV * r;
            if (g >= nt) continue;
            const int i = g * LZ_ROWS + lane;
            const double vi = s_vec[i];
            double e = 0.0;
            if (a.ell_in_lds) {
                for (int kq = wv; kq < a.ell_w; kq += NWAVE) {
                    const int col = s_ellc[kq * npad + i];
                    const double cf = s_ellf[kq * npad + i];
                    if (col >= 0) e += cf * s_vec[col];
                }
            } else {
                for (int k0 = wv; k0 < a.ell_w; k0 += 4 * NWAVE) {
                    int col[4], sx[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int kq = min(k0 + u * NWAVE, a.ell_w - 1);
                        col[u] = a.ell_col[(long long)kq * npad + i];
                        sx[u] = (k0 + u * NWAVE < a.ell_w) ? a.ell_sidx[(long long)kq * npad + i] : -1;
                    }
                    double ev[4], xv[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) { ev[u] = a.esv[max(sx[u], 0)]; xv[u] = s_vec[col[u]]; }
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (sx[u] >= 0) e += ((col[u] == i) ? ev[u] : ev[u] * INV_SQRT2) * xv[u];
                }
            }
            s_vwB[r * (NWAVE * LZ_ROWS) + wv * LZ_ROWS + lane] = e;
            static_assert(B1_NPV == 4, "one fold of four columns");
            const double ts = b1_fold4(vrp[r][0] * vi, vrp[r][1] * vi, vrp[r][2] * vi, vrp[r][3] * vi, s_fold, lane);
            const int cc = wv + 4 * (lane >> 4);
            if ((lane & 15) == 0 && cc < a.rp) s_tp[g * 64 + cc] = ts;
        }
    };
    auto S1 = [&]() {                            // wave 0 of every virtual workgroup
#pragma unroll
        for (int r = 0; r < RG; ++r) {
            const int g = vb + B1_NV * r;
            if (g >= nt) continue;
            const int i = g * LZ_ROWS + lane;
            const double* se = s_vwB + r * (NWAVE * LZ_ROWS);
            const double ei = (se[lane] + se[LZ_ROWS + lane]) + (se[2 * LZ_ROWS + lane] + se[3 * LZ_ROWS + lane]);
            s_eb[i] = ei;
            const double ap = wave_sum(s_vec[i] * ei);
            if (lane == 0) s_apf[g] = ap;
        }
    };
    // O-reduce (virtual workgroup 0): the sums every workgroup of k_lz_orth repeats
    auto Ored = [&]() {
        if constexpr (!FOP) {
            double av = (tid < 64) ? s_Ap[tid] : 0.0;            // (<= 40 tiles: the shares sit in wave 0)
            av = wave_sum(av);
            if (lane == 0) s_red[wv] = av;
        } else {
            double pr[16], q4[4];
#pragma unroll
            for (int u = 0; u < 16; ++u) pr[u] = (wv + NWAVE * u < nt) ? s_tp[(wv + NWAVE * u) * 64 + lane] : 0.0;
            tree_in_wave(pr, q4);
#pragma unroll
            for (int q = 0; q < 4; ++q) s_com[(wv * 4 + q) * 64 + lane] = q4[q];
            double ap = s_apf[lane];
            ap = wave_sum(ap);
            if (lane == 0) s_red[wv] = (wv == 0) ? ap : 0.0;
        }
    };
    // O-coefficients (virtual workgroup 0): alpha~, u = Lam t, the prediction's coefficients q
    auto Ocoef = [&](int k, bool first, double be_km, double binv) {
        double alpha;
        if constexpr (!FOP) {
            alpha = ((s_red[0] + s_red[1]) + (s_red[2] + s_red[3])) * INV_SQRT2 * binv * binv;
        } else {
            double tl = 0.0;
            if (lane < a.rp) { const double t0 = tree_across(s_com, 64, lane); tl = lam_l0 * t0 * t0; }
            tl = wave_sum(tl);
            alpha = (tl + s_red[0]) * binv * binv;
            if (tid < 64) s_u[tid] = (tid < a.rp) ? lam_j * tree_across(s_com, 64, tid) : 0.0;
        }
        const int j = tid;
        double ck = alpha;
        if (first) {
            if (j < keep) s_q[j] = s_f[j];
        } else {
            ck -= s_h2[k - 1];
            double fh = 0.0;
            if (keep > 0 && k > keep) {
                if (lane < keep) fh = s_f[lane] * s_h2[lane];
                fh = wave_sum(fh);
            }
            if (j < k) {
                const double hj = s_h2[j];
                double t;
                if (j < keep) {
                    t = s_D[j] * hj + (k > keep ? s_f[j] * s_h2[keep] : 0.0);
                } else {
                    t = s_al[j] * hj;
                    if (j + 1 < k) t += s_be[j] * s_h2[j + 1];
                    if (j == keep) t += fh;
                    else if (j > 0) t += s_be[j - 1] * s_h2[j - 1];
                }
                s_q[j] = t * binv + (j == k - 1 ? be_km : 0.0);
            }
        }
        if (j == k) { s_q[k] = ck; s_hsum[k] = ck; }
        else if (j > k && j < 64) s_q[j] = 0.0;
        if (first && j >= keep && j < k) s_q[j] = 0.0;
    };
    // O-rows: w - V q row sums of this wave's columns
    auto Orows = [&](double binv) {
#pragma unroll
        for (int r = 0; r < RG; ++r) {
            double d0 = 0.0, d1 = 0.0;
#pragma unroll
            for (int c = 0; c < B1_NC; c += 2) {
                d0 += vr[r][c] * s_q[wv + 4 * c];
                d1 += vr[r][c + 1] * s_q[wv + 4 * (c + 1)];
            }
            double dsub = d0 + d1;
            if constexpr (FOP) {
                double u0 = 0.0, u1 = 0.0;
#pragma unroll
                for (int c = 0; c < B1_NPV; c += 2) {
                    u0 += vrp[r][c] * s_u[wv + 4 * c];
                    u1 += vrp[r][c + 1] * s_u[wv + 4 * (c + 1)];
                }
                dsub -= (u0 + u1) * binv;
            }
            s_vw[r * (NWAVE * LZ_ROWS) + wv * LZ_ROWS + lane] = dsub;
        }
    };
    // O-measure: w', |w'|^2 partial, measured pass V_k' w'
    auto Omeas = [&](int k, const double (&wi)[RG]) {
        double* hp_out = s_hp + (k & 1) * nt * 64;
        double* hn_out = s_hn + (k & 1) * 64;
#pragma unroll
        for (int r = 0; r < RG; ++r) {
            const int g = vb + B1_NV * r;
            if (g >= nt) continue;
            const int i = g * LZ_ROWS + lane;
            const double* sa = s_vw + r * (NWAVE * LZ_ROWS);
            const double wp = wi[r] - ((sa[lane] + sa[LZ_ROWS + lane]) + (sa[2 * LZ_ROWS + lane] + sa[3 * LZ_ROWS + lane]));
            if (wv == 0) {
                s_vec[i] = wp;
                const double rr = wave_sum(wp * wp);
                if (lane == 0) hn_out[g] = rr;
            }
            static_assert(B1_NC == 8, "two folds of four columns");
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                if (wv + 16 * hf > k) break;                              // (wave-uniform: none of these four columns exists yet)
                double t[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) t[c] = (wv + 4 * (4 * hf + c) <= k) ? vr[r][4 * hf + c] * wp : 0.0;
                const double hs = b1_fold4(t[0], t[1], t[2], t[3], s_fold, lane);
                const int jc = wv + 4 * (4 * hf + (lane >> 4));
                if ((lane & 15) == 0 && jc <= k) hp_out[g * 64 + jc] = hs;
            }
        }
    };

    for (int k = kfirst; k <= kd; ++k) {
        const bool hasF = k > kfirst, hasSO = k < kd, first = (k == kfirst);
        double beta = 0.0;
        double wi[RG];
        if constexpr (FOP) {
            // ---- six segments per step: the closing work of step k-1 rides with the operator rows and the recurrence of step k
            if (hasF && vb == 0) F0(k - 1);
            if (hasSO) S0();
            __syncthreads();
            if (hasF && vb == 0 && tid < 64) F1(k - 1);
            if (hasSO && wv == 0) S1();
            __syncthreads();
            if (hasF) {
                beta = F2(k - 1);
                if (beta <= a.tol) { stop_k = k; break; }                 // (uniform: every wave holds the same beta)
            }
            if (hasSO && vb == 0) Ored();
            __syncthreads();
            const double be_km = first ? 1.0 : s_be[max(k - 1, 0)];
            const double binv = first ? 1.0 : 1.0 / be_km;
            if (hasF && wv == 0) F3(k - 1, beta);
            if (hasSO && vb == 0) Ocoef(k, first, be_km, binv);
            __syncthreads();
            if (hasF) kdone = k;
            if (!hasSO) break;
            if (hasF) F4(k);
#pragma unroll
            for (int r = 0; r < RG; ++r) wi[r] = s_eb[min(vb + B1_NV * r, nt - 1) * LZ_ROWS + lane] * binv;
            Orows(binv);
            __syncthreads();
            Omeas(k, wi);
            __syncthreads();
        } else {
          if constexpr (RG == 1) {
            // ---- packed operator (resident in LDS: side <= ~140, at most two rounds of tiles): the closing work of step k-1 rides with
            // the tile rounds of step k -- symv_reduce cut at its two barriers (A: products, row sums, folds | B: the waves' column sums /
            // shares meet | C: slots written) -- 8-9 barriers per step instead of 11-13
            struct TileSt { int tile, I, J, gi, j0; bool act, diag; double vi, cs, aw; };
            auto tileA = [&](int rd, TileSt& T) {
                T.tile = rd * B1_NV + vb; T.act = T.tile < ntile; T.I = T.J = 0; T.vi = T.cs = T.aw = 0.0;
                if (!T.act) { T.gi = T.j0 = 0; T.diag = false; return; }
                tile_coords(T.tile, T.I, T.J);
                T.gi = T.I * TILE + lane; T.j0 = T.J * TILE + wv * CPW; T.diag = (T.I == T.J);
                double t[CPW];
                symv_load(s_X, a.n, T.tile, lane, wv, t);
                double* s_row = s_vwB + (rd & 1) * (NWAVE * TILE);
                const double* vJ = s_vec + T.j0;
                T.vi = s_vec[T.gi];
                double racc = 0.0;
#pragma unroll
                for (int c = 0; c < CPW; ++c) {
                    racc += t[c] * vJ[c];
                    t[c] *= T.vi;
                    if (T.diag && T.gi == T.j0 + c) t[c] = 0.0;
                }
                s_row[wv * TILE + lane] = racc;
                T.aw = T.diag ? 0.0 : wave_sum(T.vi * racc);
                fold_stage<8>(t, lane);
                fold_stage<4>(t, lane);
                fold_stage<2>(t, lane);
                fold_stage<1>(t, lane);
                T.cs = add_xor32(add_xor16(t[0]));
            };
            auto tileB = [&](int rd, const TileSt& T) {
                if (!T.act) return;
                double* s_col = s_vw + 256 + (rd & 1) * TILE;
                if (T.diag) { if (lane < CPW) s_col[wv * CPW + lane] = T.cs; }
                else if (lane == 0) s_col[wv] = T.aw;
            };
            auto tileC = [&](int rd, const TileSt& T) {
                if (!T.act) return;
                const double* s_row = s_vwB + (rd & 1) * (NWAVE * TILE);
                const double* s_col = s_vw + 256 + (rd & 1) * TILE;
                if (T.diag) {
                    if (wv == 0) {
                        const double rs = (s_row[lane] + s_row[TILE + lane]) + (s_row[2 * TILE + lane] + s_row[3 * TILE + lane]);
                        s_Pp[T.I * npad + T.gi] = rs + s_col[lane];
                        const double aa = wave_sum(T.vi * (rs + s_col[lane]));
                        if (lane == 0) s_Ap[T.tile] = aa;
                    }
                } else {
                    if (wv == 0) {
                        const double rs = (s_row[lane] + s_row[TILE + lane]) + (s_row[2 * TILE + lane] + s_row[3 * TILE + lane]);
                        s_Pp[T.J * npad + T.gi] = rs;
                        if (lane == 0) s_Ap[T.tile] = 2.0 * ((s_col[0] + s_col[1]) + (s_col[2] + s_col[3]));
                    }
                    if (lane < CPW) s_Pp[T.I * npad + T.j0 + lane] = T.cs;
                }
            };
            TileSt T0{}, T1{};
            if (hasF && vb == 0) F0(k - 1);
            if (hasSO) tileA(0, T0);
            __syncthreads();
            if (hasF && vb == 0 && tid < 64) F1(k - 1);
            if (hasSO) tileB(0, T0);
            __syncthreads();
            if (hasF) {
                beta = F2(k - 1);
                if (beta <= a.tol) { stop_k = k; break; }
            }
            if (hasSO) { tileC(0, T0); if (rounds > 1) tileA(1, T1); }
            __syncthreads();
            if (hasF && wv == 0) F3(k - 1, beta);
            if (hasSO && rounds > 1) tileB(1, T1);
            __syncthreads();
            if (hasF) kdone = k;
            if (!hasSO) break;
            if (hasF) F4(k);
            if (rounds > 1) tileC(1, T1);
            for (int rd = 2; rd < rounds; ++rd) {
                const int tile = rd * B1_NV + vb;
                const bool act = tile < ntile;
                double t[CPW];
                if (act) symv_load(s_X, a.n, tile, lane, wv, t);
                else {
#pragma unroll
                    for (int c = 0; c < CPW; ++c) t[c] = 0.0;
                }
                b1_symv_reduce(act, npad, s_vec, s_Pp, tile, lane, wv, t, s_vwB + (rd & 1) * (NWAVE * TILE),
                               s_vw + 256 + (rd & 1) * TILE, s_Ap);
            }
            if (rounds > 1) __syncthreads();
          } else {
            if (hasF) {
                if (vb == 0) F0(k - 1);
                __syncthreads();
                if (vb == 0 && tid < 64) F1(k - 1);
                __syncthreads();
                beta = F2(k - 1);
                if (beta <= a.tol) { stop_k = k; break; }
                __syncthreads();
                if (wv == 0) F3(k - 1, beta);
                __syncthreads();
                kdone = k;
                if (!hasSO) break;
                F4(k);
            }
            for (int rd = 0; rd < rounds; ++rd) {
                const int tile = rd * B1_NV + vb;
                const bool act = tile < ntile;
                double t[CPW];
                if (act) symv_load(s_X, a.n, tile, lane, wv, t);
                else {
#pragma unroll
                    for (int c = 0; c < CPW; ++c) t[c] = 0.0;
                }
                b1_symv_reduce(act, npad, s_vec, s_Pp, tile, lane, wv, t, s_vwB + (rd & 1) * (NWAVE * TILE),
                               s_vw + (rd & 1) * TILE, s_Ap);
            }
            __syncthreads();
          }
            const double be_km = first ? 1.0 : s_be[max(k - 1, 0)];
            const double binv = first ? 1.0 : 1.0 / be_km;
            if (vb == 0) Ored();
#pragma unroll
            for (int r = 0; r < RG; ++r) {
                const int i = min(vb + B1_NV * r, nt - 1) * LZ_ROWS + lane;
                double pv[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) pv[u] = (wv + u * NWAVE < nt) ? s_Pp[(wv + u * NWAVE) * npad + i] : 0.0;
                const double acc = (((pv[0] + pv[1]) + (pv[2] + pv[3])) + ((pv[4] + pv[5]) + (pv[6] + pv[7]))) +
                                   (((pv[8] + pv[9]) + (pv[10] + pv[11])) + ((pv[12] + pv[13]) + (pv[14] + pv[15])));
                s_vw[r * (NWAVE * LZ_ROWS) + wv * LZ_ROWS + lane] = acc;
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < RG; ++r) {
                const double* sa = s_vw + r * (NWAVE * LZ_ROWS);
                wi[r] = ((sa[lane] + sa[LZ_ROWS + lane]) + (sa[2 * LZ_ROWS + lane] + sa[3 * LZ_ROWS + lane])) * (INV_SQRT2 * binv);
            }
            if (vb == 0) Ocoef(k, first, be_km, binv);
            __syncthreads();
            Orows(binv);
            __syncthreads();
            Omeas(k, wi);
            __syncthreads();
        }
    }
    // ---- epilogue: the basis (columns 0 .. kdone), the recurrence coefficients, the control block
    __syncthreads();
    const long long tk2 = (a.dbg != nullptr && threadIdx.x == 0) ? (long long)wall_clock64() : 0;
#pragma unroll
    for (int r = 0; r < RG; ++r) {
        const int g = vb + B1_NV * r;
        if (g >= nt) continue;
        const int i = g * LZ_ROWS + lane;
#pragma unroll
        for (int c = 0; c < B1_NC; ++c) {
            const int col = wv + 4 * c;
            if (col <= kdone && col < kd) a.Vout[(long long)col * ldv + i] = vr[r][c];
        }
        if (wv == 0 && kdone == kd) a.Vout[(long long)kd * ldv + i] = s_new[i];
    }
    const int kcl = (stop_k >= 0) ? stop_k : kdone;          // steps kfirst .. kcl-1 were closed
    if ((int)threadIdx.x >= kfirst && (int)threadIdx.x < kcl) {
        a.alphas[threadIdx.x] = s_al[threadIdx.x];
        a.betas[threadIdx.x] = s_be[threadIdx.x];
    }
    if (threadIdx.x == 0) {
        a.ctl->carry = carry;
        if (stop_k >= 0) { a.ctl->kstop = stop_k; a.ctl->stop = 1; }
    }
    if (a.dbg != nullptr) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (threadIdx.x == 0) {
            const long long tk3 = (long long)wall_clock64();
            a.dbg[0] += tk1 - tk0; a.dbg[1] += tk2 - tk1; a.dbg[2] += tk3 - tk2; a.dbg[3] += kcl - kfirst; a.dbg[4] += 1;
        }
    }
}

}  // namespace dev
}  // namespace proxsdp
