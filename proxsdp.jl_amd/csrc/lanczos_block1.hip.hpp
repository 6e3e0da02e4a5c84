// One whole Lanczos cycle of a MEDIUM PSD block (side <= 512) in ONE launch of ONE workgroup (round 6).
//
// Why: every Lanczos step of the step kernels (kernels.hip.hpp) is two dependent launches of ~6 us each whatever the
// side; a block of side 101 .. 500 -- all but three instances of the reference's own benchmark script, test/runbench.jl --
// therefore spent ~370 us per PDHG iteration on 25-30 mat-vecs of a matrix that fits in one CU's caches.  Here one
// workgroup of 1024 threads owns the whole block for the whole cycle: the Krylov basis stays in REGISTERS (thread <-> row,
// wave <-> column residue, exactly the tiling of the step kernels), every record the step kernels exchange through
// global memory (mat-vec slots, partial dots, w', the reduced coefficients) lives in LDS, and a kernel boundary becomes
// a workgroup barrier.
//
// The ARITHMETIC is that of the step kernels, term by term and in the same order: the workgroup is organised as NV = 4
// "virtual workgroups" of 4 waves; virtual workgroup b does for row group / tile b, b + 4, ... exactly what workgroup b of
// a step kernel does (symv_load / symv_reduce, the bodies of k_fop, k_lz_orth<1, .>, k_lz_finish<1>), and the sums that
// every real workgroup repeats redundantly (alpha, the reduced dots, beta) are formed once, by virtual workgroup 0, in the
// order the step kernels use.  Results, mat-vec counts and restart counts are therefore those of the step kernels BIT FOR
// BIT (tests/test_gpu_parity.py::test_block_cycle_kernel_reproduces_the_step_kernels_bit_for_bit); what changes is time.
//
// Replaces: the BLAS-1 work and the mat-vecs of one KrylovKit Lanczos cycle (call site /root/reference/src/eigsolver.jl:802-812,
// dsymv at :678), as k_symv_finish / k_fop_finish + k_lz_orth do.  The K x K eigensolve and the restart logic stay on the
// host (Solver::lz_after_cycle), unchanged.
//
// Limits (Solver::block1_plan): side <= 512 (8 row groups), Krylov dimension <= 32, operator form with <= 16 factor columns
// and no hub-row overflow list; anything else takes the step kernels.
#pragma once
#include "kernels.hip.hpp"

namespace proxsdp {
namespace dev {

constexpr int B1_NV = 4;                 // virtual workgroups (4 waves each)
constexpr int B1_TPB = B1_NV * TPB;      // 1024 threads
constexpr int B1_NC = 8;                 // basis columns held per wave (column j = wv + 4c): Krylov dimension <= 32
constexpr int B1_NPV = 4;                // factor columns held per wave: rank of the previous projection <= 16
constexpr int B1_MAXNT = 8;              // side <= 512
constexpr int B1_KMAX = 4 * B1_NC;

struct Block1Args {
    const double* xp;                    // packed block of the iterate (packed operator)
    int n, nt, npad;
    double* V; int ldv;                  // basis, column j at V + j*ldv; columns 0 .. kfirst valid on entry
    int kfirst, kd;                      // steps kfirst .. kd-1; the cycle ends with column kd written
    double tol;
    const double* Vp; const double* lam; int rp;      // operator form: previous projection's factors
    const int* ell_col; const int* ell_sidx; int ell_w; const double* esv;
    const double* arrow;                 // f | D (MAXK each), valid below kfirst
    double* alphas; double* betas; LanczosCtl* ctl;
};

// LDS plan (doubles).  Per virtual workgroup scratch is a union over the phases (mat-vec | recurrence | closing).
struct B1Lds {
    int vec, Pp, Ap, eb, tp, apf, hp, hn, nw, q, u, h, h2, hred, hsum, al, be, f, D, red, common, vw, vw_stride, total;
};
__host__ __device__ inline B1Lds b1_lds_plan(int nt, int npad, bool fop) {
    B1Lds L{};
    int o = 0;
    auto take = [&](int cnt) { const int at = o; o += (cnt + 1) & ~1; return at; };
    L.vec = take(npad);                          // operand of the mat-vec: v_k (first step) or w'
    L.nw = take(npad);                           // the new basis column, wave 0 -> the wave that holds it
    L.Pp = take(fop ? 0 : nt * npad);            // mat-vec slots [slot][row]
    L.Ap = take(fop ? 0 : 64);                   // tiles' shares of w'P~w'
    L.eb = take(fop ? npad : 0);                 // (E v)_i
    L.tp = take(fop ? nt * 64 : 0);              // Vp'v partials [g][column]
    L.apf = take(fop ? 64 : 0);                  // v'Ev partials [g]
    L.hp = take(2 * nt * 64);                    // partial dots V'w' [parity][g][column]
    L.hn = take(2 * 64);                         // |w'|^2 partials [parity][g]
    L.q = take(64); L.u = take(64); L.h = take(64); L.h2 = take(66); L.hred = take(64); L.hsum = take(64);
    L.al = take(64); L.be = take(64); L.f = take(64); L.D = take(64); L.red = take(8);
    L.common = take(NWAVE * 4 * 64);             // s_t | s_p of virtual workgroup 0
    L.vw_stride = 2 * NWAVE * TILE + 2 * TILE;   // s_row x2 + s_col x2 | s_acc[2][4][64] | s_d[2][256] | s_e x2
    L.vw = take(B1_NV * L.vw_stride);
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
    const B1Lds L = b1_lds_plan(nt, npad, FOP);
    double* const s_vec = b1_sm + L.vec;
    double* const s_new = b1_sm + L.nw;
    double* const s_Pp = b1_sm + L.Pp;
    double* const s_Ap = b1_sm + L.Ap;
    double* const s_eb = b1_sm + L.eb;
    double* const s_tp = b1_sm + L.tp;
    double* const s_apf = b1_sm + L.apf;
    double* const s_hp = b1_sm + L.hp;
    double* const s_hn = b1_sm + L.hn;
    double* const s_q = b1_sm + L.q;
    double* const s_u = b1_sm + L.u;
    double* const s_h = b1_sm + L.h;
    double* const s_h2 = b1_sm + L.h2;            // [0, 64): reduced dots, [64]: |w'|^2, [65]: beta
    double* const s_hred = b1_sm + L.hred;
    double* const s_hsum = b1_sm + L.hsum;
    double* const s_al = b1_sm + L.al;
    double* const s_be = b1_sm + L.be;
    double* const s_f = b1_sm + L.f;
    double* const s_D = b1_sm + L.D;
    double* const s_red = b1_sm + L.red;
    double* const s_com = b1_sm + L.common;
    double* const s_vw = b1_sm + L.vw + vb * L.vw_stride;

    if (a.ctl->stop) return;
    // ---- prologue: basis rows into registers, operand of the first mat-vec, arrow part, zeroed padding records
    double vr[RG][B1_NC];
    double vrp[RG][FOP ? B1_NPV : 1];
#pragma unroll
    for (int r = 0; r < RG; ++r) {
        const int g = vb + B1_NV * r;
        const int i = min(g, nt - 1) * LZ_ROWS + lane;
#pragma unroll
        for (int c = 0; c < B1_NC; ++c) vr[r][c] = (g < nt && wv + 4 * c <= kfirst) ? a.V[(long long)(wv + 4 * c) * ldv + i] : 0.0;
        if constexpr (FOP) {
#pragma unroll
            for (int c = 0; c < B1_NPV; ++c) vrp[r][c] = a.Vp[(long long)min(wv + 4 * c, max(a.rp - 1, 0)) * ldv + i];
        } else vrp[r][0] = 0.0;
    }
    for (int i = threadIdx.x; i < npad; i += B1_TPB) s_vec[i] = a.V[(long long)kfirst * ldv + i];
    if (threadIdx.x < 64) {
        const int j = threadIdx.x;
        s_f[j] = a.arrow[j]; s_D[j] = a.arrow[MAXK + j];
        s_hn[j] = 0.0; s_hn[64 + j] = 0.0;
        s_al[j] = 0.0; s_be[j] = 0.0; s_hred[j] = 0.0; s_hsum[j] = 0.0;
        if constexpr (FOP) s_apf[j] = 0.0; else s_Ap[j] = 0.0;
    }
    double lam_j = 0.0, lam_l0 = 0.0;
    if constexpr (FOP) { lam_j = a.lam[min(tid, MAXK - 1)]; lam_l0 = a.lam[lane]; }
    double carry = 0.0;                      // ctl->carry of the step kernels (thread 0 of virtual workgroup 0 keeps it)
    int stop_k = -1;
    __syncthreads();

    const int ntile = nt * (nt + 1) / 2;
    const int rounds = (ntile + B1_NV - 1) / B1_NV;

    for (int k = kfirst; k <= kd; ++k) {
        // =========================================================== closing work of step k-1 (k_lz_finish<1> / lz_finish_body)
        if (k > kfirst) {
            const int kc = k - 1, kk = kc + 1;
            const double* hp_in = s_hp + (kc & 1) * nt * 64;
            const double* hn_in = s_hn + (kc & 1) * 64;
            if (vb == 0) {
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
            }
            __syncthreads();
            if (vb == 0 && tid < 64) {
                const double hj = tree_across(s_com, 64, tid);
                s_h2[tid] = (tid < kk) ? hj : 0.0;
            }
            __syncthreads();
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
            if (vb == 0) {
                if (tid < kk) s_hred[tid] = s_h2[tid];
                if (tid == 0) {
                    const double al = s_hsum[kc] + s_h2[kc] - ((kc > kfirst) ? carry : 0.0);
                    s_al[kc] = al; s_be[kc] = beta;
                    a.alphas[kc] = al; a.betas[kc] = beta;
                    carry = s_h2[kc];
                }
            }
            if (beta <= a.tol) { stop_k = kc + 1; break; }            // (uniform: every wave holds the same beta)
            __syncthreads();
            if (wv == 0) {
#pragma unroll
                for (int r = 0; r < RG; ++r) {
                    const int g = vb + B1_NV * r;
                    if (g < nt) {
                        const int i = g * LZ_ROWS + lane;
                        const double* sd = s_vw + r * (NWAVE * LZ_ROWS);
                        const double wi = s_vec[i] - ((sd[lane] + sd[LZ_ROWS + lane]) + (sd[2 * LZ_ROWS + lane] + sd[3 * LZ_ROWS + lane]));
                        const double nv = wi / beta;
                        a.V[(long long)(kc + 1) * ldv + i] = nv;
                        s_new[i] = nv;
                    }
                }
            }
            __syncthreads();
            if (k < kd && wv == ((kc + 1) & 3)) {
                const int cn = (kc + 1) >> 2;
#pragma unroll
                for (int r = 0; r < RG; ++r) {
                    const int g = vb + B1_NV * r;
                    const double nv = (g < nt) ? s_new[g * LZ_ROWS + lane] : 0.0;
#pragma unroll
                    for (int c = 0; c < B1_NC; ++c) if (c == cn) vr[r][c] = nv;
                }
            }
        }
        if (k == kd) break;
        const bool first = (k == kfirst);
        // =========================================================== operator on s_vec (v_k at the first step, w' afterwards)
        if constexpr (!FOP) {
            for (int rd = 0; rd < rounds; ++rd) {
                const int tile = rd * B1_NV + vb;
                const bool act = tile < ntile;
                double t[CPW];
                if (act) symv_load(a.xp, a.n, tile, lane, wv, t);
                else {
#pragma unroll
                    for (int c = 0; c < CPW; ++c) t[c] = 0.0;
                }
                b1_symv_reduce(act, npad, s_vec, s_Pp, tile, lane, wv, t, s_vw + (rd & 1) * (NWAVE * TILE),
                               s_vw + 2 * NWAVE * TILE + (rd & 1) * TILE, s_Ap);
            }
        } else {
            // fop_body per row group: (E v)_i, Vp'v partials, v'Ev partials
#pragma unroll
            for (int r = 0; r < RG; ++r) {
                const int g = vb + B1_NV * r;
                const bool act = g < nt;
                const int i = min(g, nt - 1) * LZ_ROWS + lane;
                double* s_e = s_vw + (r & 1) * (NWAVE * LZ_ROWS);
                double vi = 0.0;
                if (act) {
                    vi = s_vec[i];
                    double e = 0.0;
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
                    s_e[wv * LZ_ROWS + lane] = e;
                    double t[16];
#pragma unroll
                    for (int c = 0; c < 16; ++c) t[c] = (c < B1_NPV) ? vrp[r][c < B1_NPV ? c : 0] * vi : 0.0;
                    const double ts = fold16_all(t, lane);
                    const int cc = wv + 4 * lane;
                    if (lane < 16 && cc < a.rp) s_tp[g * 64 + cc] = ts;
                }
                __syncthreads();
                if (act && wv == 0) {
                    const double ei = (s_e[lane] + s_e[LZ_ROWS + lane]) + (s_e[2 * LZ_ROWS + lane] + s_e[3 * LZ_ROWS + lane]);
                    s_eb[i] = ei;
                    const double ap = wave_sum(vi * ei);
                    if (lane == 0) s_apf[g] = ap;
                }
            }
        }
        __syncthreads();
        // =========================================================== recurrence + measured pass of step k (lz_orth_body<1, .>)
        const double be_km = first ? 1.0 : s_be[max(k - 1, 0)];
        const double binv = first ? 1.0 : 1.0 / be_km;
        double wi[RG];
        if (vb == 0) {
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
            if (!first && tid < 64) s_h[tid] = (tid < k) ? s_hred[tid] : 0.0;
        }
        if constexpr (!FOP) {
#pragma unroll
            for (int r = 0; r < RG; ++r) {
                const int g = vb + B1_NV * r;
                const int i = min(g, nt - 1) * LZ_ROWS + lane;
                double pv[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) pv[u] = (wv + u * NWAVE < nt) ? s_Pp[(wv + u * NWAVE) * npad + i] : 0.0;
                const double acc = (((pv[0] + pv[1]) + (pv[2] + pv[3])) + ((pv[4] + pv[5]) + (pv[6] + pv[7]))) +
                                   (((pv[8] + pv[9]) + (pv[10] + pv[11])) + ((pv[12] + pv[13]) + (pv[14] + pv[15])));
                s_vw[r * (NWAVE * LZ_ROWS) + wv * LZ_ROWS + lane] = acc;
            }
        }
        __syncthreads();
        if constexpr (!FOP) {
#pragma unroll
            for (int r = 0; r < RG; ++r) {
                const double* sa = s_vw + r * (NWAVE * LZ_ROWS);
                wi[r] = ((sa[lane] + sa[LZ_ROWS + lane]) + (sa[2 * LZ_ROWS + lane] + sa[3 * LZ_ROWS + lane])) * (INV_SQRT2 * binv);
            }
        } else {
#pragma unroll
            for (int r = 0; r < RG; ++r) {
                const int g = vb + B1_NV * r;
                wi[r] = s_eb[min(g, nt - 1) * LZ_ROWS + lane] * binv;
            }
        }
        if (vb == 0) {
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
                ck -= s_h[k - 1];
                double fh = 0.0;
                if (keep > 0 && k > keep) {
                    if (lane < keep) fh = s_f[lane] * s_h[lane];
                    fh = wave_sum(fh);
                }
                if (j < k) {
                    const double hj = s_h[j];
                    double t;
                    if (j < keep) {
                        t = s_D[j] * hj + (k > keep ? s_f[j] * s_h[keep] : 0.0);
                    } else {
                        t = s_al[j] * hj;
                        if (j + 1 < k) t += s_be[j] * s_h[j + 1];
                        if (j == keep) t += fh;
                        else if (j > 0) t += s_be[j - 1] * s_h[j - 1];
                    }
                    s_q[j] = t * binv + (j == k - 1 ? be_km : 0.0);
                }
            }
            if (j == k) { s_q[k] = ck; s_hsum[k] = ck; }
            else if (j > k && j < 64) s_q[j] = 0.0;
            if (first && j >= keep && j < k) s_q[j] = 0.0;
        }
        __syncthreads();
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
        __syncthreads();
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
            double t[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) t[c] = (c < B1_NC && wv + 4 * c <= k) ? vr[r][c < B1_NC ? c : 0] * wp : 0.0;
            const double hs = fold16_all(t, lane);
            const int jc = wv + 4 * lane;
            if (lane < 16 && jc <= k) hp_out[g * 64 + jc] = hs;
        }
        __syncthreads();
    }
    // ---- epilogue: control block (alphas / betas were stored as they were formed)
    if (threadIdx.x == 0) {
        a.ctl->carry = carry;
        if (stop_k >= 0) { a.ctl->kstop = stop_k; a.ctl->stop = 1; }
    }
}

}  // namespace dev
}  // namespace proxsdp
