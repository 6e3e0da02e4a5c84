// One whole Lanczos cycle (operator form) in ONE persistent launch.
//
// The step kernels (kernels.hip.hpp: k_fop_finish + k_lz_orth) cost ~14 us per Lanczos step at
// n = 4000 although they move ~2 MB of L2-resident data: each of the two launches pays a kernel
// start, loads of lines another XCD wrote, and an end-of-kernel write-back.
// Here G <= 64 workgroups stay resident for the whole cycle; workgroup g owns R rows (64 or 128) of
// every vector and keeps ITS ROWS OF THE KRYLOV BASIS (and of the previous projection's factors) IN
// LDS.  Per step only three things cross workgroups, all through global memory with the
// data-tagged-granule protocol of cdna_hip_programming.md Guideline 16 (R2: the data is the flag):
//   X1  partial dots  Vp' u and u'E u         (rp + 1 doubles per workgroup)
//   X2  partial dots  V_k' w' and |w'|^2      (k + 2 doubles per workgroup)
//   w'  itself (write-through stores), because (E u)_i needs entries of u from other rows
// Every workgroup sums all partials in the same fixed order, so the replicated scalars (alpha,
// beta, the re-orthogonalisation coefficients) are bit-identical everywhere: deterministic, no
// broadcast step.  The ARITHMETIC is that of the step kernels (predicted first Gram-Schmidt pass,
// measured second pass, beta^2 = |w'|^2 - |h2|^2, alpha correction by `carry`); only the grouping
// of the partial sums differs (R rows per partial instead of 64).
//
// Replaces: the BLAS-1 work and mat-vecs of one KrylovKit Lanczos cycle (call site
// /root/reference/src/eigsolver.jl:802-812), as k_fop* / k_lz_* do.
//
// Residency: G <= 64 workgroups of <= 160 KiB LDS, one per CU, on a GPU that runs nothing else on
// this stream: all are co-resident.  Every spin is bounded; a timeout sets *err and the host redoes
// the projection with the step kernels (Solver::lanczos).
#pragma once
#include "kernels.hip.hpp"

namespace proxsdp {
namespace dev {

typedef __attribute__((address_space(1))) unsigned long long gu64;

constexpr int CY_CMAX = 196;          // >= MAXK + 2: columns of any replicated coefficient array
constexpr int CY_NE = 16;             // ELL entries cached per thread (ell_w <= CY_NE * S)
constexpr unsigned CY_SPIN_LIMIT = 200000u;

struct CycleArgs {
    double* V; int ldv; int n; int npad;
    int kfirst; int kd;                  // steps kfirst .. kd-1 (kd = krylovdim); basis columns 0 .. kd
    double tol;
    const double* Vp; int rp; const double* lam;
    const int* ell_col; const int* ell_sidx; int ell_w; const double* esv;
    const double* arrow;                 // f | D, MAXK each (valid below kfirst)
    double* alphas; double* betas; LanczosCtl* ctl;
    double* x1; double* x2; int xs1; int xs2;   // records [G][xs] of doubles
    unsigned* f1; unsigned* f2;                 // their flags [G]
    unsigned epoch0;
    int R; int G; int f_in_lds;
    int* err;
    long long* dbg;                      // optional: accumulated wall_clock64 ticks per phase (workgroup 0)
};

// Hand-off protocol (all active workgroups share one XCD, see k_lz_cycle): the producer writes its
// record with plain stores (they stay in the XCD's L2), drains them (s_waitcnt vmcnt(0): acknowledged
// by L2), and one lane stores the record's flag = this phase's epoch.  Consumers: ONE wave polls the G
// flags (one lane each, L1-bypassing agent-scope loads), then every thread reads the payload with
// L1-bypassing loads (served by the shared L2).  If the placement assumption fails the flags never
// arrive: bounded spin -> *err -> host fallback.
__device__ __forceinline__ void cy_put(double* rec, int e, double v) {
    __hip_atomic_store((gu64*)(rec + e), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ double cy_get(const double* p) {
    return __longlong_as_double((long long)__hip_atomic_load((gu64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
// called by every thread after its cy_put()s: drain, barrier, raise this workgroup's flag
__device__ __forceinline__ void cy_publish(unsigned* flags, int g, unsigned epoch) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0)
        __hip_atomic_store((__attribute__((address_space(1))) unsigned*)(flags + g), epoch, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_WORKGROUP);
}
// wave 0 waits until all G flags carry `epoch`; returns false on timeout (valid in wave 0 only)
__device__ __forceinline__ bool cy_wait(const unsigned* flags, int G, unsigned epoch, int lane) {
    unsigned spins = 0;
    for (;;) {
        const unsigned f = (lane < G)
            ? __hip_atomic_load((__attribute__((address_space(1))) unsigned*)(flags + lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
            : epoch;
        if (__all(f == epoch)) return true;
        if (++spins > CY_SPIN_LIMIT) return false;
        __builtin_amdgcn_s_sleep(1);
    }
}

// dst[g * dstride + e] = rec[g * xs + off + e], e < len, g < G <= 32: wave w takes records w, w+4, ...
// with its lanes on consecutive entries (one coalesced load per record and 64-entry chunk), eight
// loads in flight per wave before the first LDS store
__device__ __forceinline__ void cy_fetch_records(const double* __restrict__ rec, int xs, int off, int G, int len,
                                                 double* __restrict__ dst, int dstride, int lane, int wv) {
    for (int e0 = 0; e0 < len; e0 += 64) {
        const int e = e0 + lane;
        const bool in = e < len;
        const int ec = in ? e : len - 1;
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int gg = min(wv + u * NWAVE, G - 1);
            v[u] = cy_get(rec + (size_t)gg * xs + off + ec);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int gg = wv + u * NWAVE;
            if (in && gg < G) dst[gg * dstride + e] = v[u];
        }
    }
}

// sum over j = s, s+S, ... < cnt of M[j*ld + row] * coef[j]  (M, coef in LDS; 8 loads in flight)
template <int S>
__device__ __forceinline__ double cy_dot_cols(const double* __restrict__ M, int ld, int row, int s,
                                              const double* __restrict__ coef, int cnt) {
    if (cnt <= 0) return 0.0;
    double a0 = 0.0, a1 = 0.0;
    for (int j0 = s; j0 < cnt; j0 += 8 * S) {
        double m[8], q[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int jj = j0 + u * S;
            const int jc = min(jj, cnt - 1);
            m[u] = M[jc * ld + row];
            q[u] = (jj < cnt) ? coef[jc] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; u += 2) { a0 += m[u] * q[u]; a1 += m[u + 1] * q[u + 1]; }
    }
    return a0 + a1;
}
// the same with M in global memory (previous factors that do not fit in LDS; L1/L2 resident)
template <int S>
__device__ __forceinline__ double cy_dot_cols_g(const double* __restrict__ M, size_t ld, int grow, int s,
                                                const double* __restrict__ coef, int cnt) {
    if (cnt <= 0) return 0.0;
    double a0 = 0.0, a1 = 0.0;
    for (int j0 = s; j0 < cnt; j0 += 8 * S) {
        double m[8], q[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int jj = j0 + u * S;
            const int jc = min(jj, cnt - 1);
            m[u] = M[(size_t)jc * ld + grow];
            q[u] = (jj < cnt) ? coef[jc] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; u += 2) { a0 += m[u] * q[u]; a1 += m[u + 1] * q[u + 1]; }
    }
    return a0 + a1;
}

// Workgroup b is dispatched to XCD b % 8 (observed, MI355X_MICROARCH.md "Workgroup dispatch"): the launch
// has 8 G workgroups and only those with b % 8 == 0 work, so the G <= 32 active ones share ONE XCD and
// its L2: a hand-off costs an L2 round trip instead of a fabric one (measured: the same kernel with
// agent-scope write-through granules over all 8 XCDs took 19 us per step, this one ~1 us per hand-off).
// Placement is an assumption made for SPEED ONLY: if it does not hold the flags never arrive, the
// bounded spins time out, *err is set and the host redoes the projection with the step kernels.
template <int R>
__global__ void __launch_bounds__(TPB)
k_lz_cycle(CycleArgs a) {
    extern __shared__ __attribute__((aligned(16))) double cy_smem[];
    constexpr int NWR = R / 64, S = NWAVE / NWR;
    if ((blockIdx.x & 7) != 0) return;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int rb = wv % NWR, s = wv / NWR;
    const int row = rb * 64 + lane;
    const int g = blockIdx.x >> 3;
    const int i = g * R + row;
    const bool rowok = i < a.npad;
    const int ic = rowok ? i : a.npad - 1;
    const int tid = threadIdx.x;
    const int kd = a.kd, kfirst = a.kfirst, rp = a.rp, G = a.G;
    const int lenmax = max(rp + 1, kd + 2);
    const bool fl = a.f_in_lds != 0;

    double* sV = cy_smem;
    double* sF = sV + (kd + 1) * R;
    double* p = sF + (fl ? rp * R : 0);
    double* s_full = p; p += G * R;                    // the whole vector u (every workgroup's rows)
    double* s_x = p; p += G * lenmax;                  // fetched partial records
    double* s_w = p; p += R;
    double* s_pc = p; p += S * R;                      // per-subset row partials (closing / w')
    double* s_pe = p; p += S * R;                      // per-subset row partials (E u)
    double* s_col = p; p += NWR * CY_CMAX;             // per-row-block column partials
    double* s_h = p; p += CY_CMAX;
    double* s_q = p; p += CY_CMAX;
    double* s_al = p; p += CY_CMAX;
    double* s_be = p; p += CY_CMAX;
    double* s_f = p; p += CY_CMAX;
    double* s_D = p; p += CY_CMAX;
    double* s_u = p; p += CY_CMAX;
    double* s_lam = p; p += CY_CMAX;
    double* s_sc = p; p += 8;

    // ---- stage: basis columns 0..kfirst, factors, coefficient data, the start vector as u
    for (int idx = tid; idx < (kfirst + 1) * R; idx += TPB) {
        const int j = idx / R, r = idx - j * R, gi = g * R + r;
        sV[idx] = (gi < a.npad) ? a.V[(size_t)j * a.ldv + gi] : 0.0;
    }
    if (fl)
        for (int idx = tid; idx < rp * R; idx += TPB) {
            const int cc = idx / R, r = idx - cc * R, gi = g * R + r;
            sF[idx] = (gi < a.npad) ? a.Vp[(size_t)cc * a.ldv + gi] : 0.0;
        }
    for (int idx = tid; idx < G * R; idx += TPB) s_full[idx] = (idx < a.npad) ? a.V[(size_t)kfirst * a.ldv + idx] : 0.0;
    for (int j = tid; j < CY_CMAX; j += TPB) {
        s_f[j] = (j < kfirst) ? a.arrow[j] : 0.0;
        s_D[j] = (j < kfirst) ? a.arrow[MAXK + j] : 0.0;
        s_lam[j] = (j < rp) ? a.lam[j] : 0.0;
        s_h[j] = 0.0; s_q[j] = 0.0; s_al[j] = 0.0; s_be[j] = 0.0; s_u[j] = 0.0;
    }
    // this thread's ELL entries (k = s, s+S, ...) of its row: column and value, fixed for the projection
    int ecol[CY_NE];
    double eval[CY_NE];
#pragma unroll
    for (int q = 0; q < CY_NE; ++q) {
        const int k = s + q * S;
        ecol[q] = ic; eval[q] = 0.0;
        if (k < a.ell_w && rowok) {
            const int col = a.ell_col[(size_t)k * a.npad + i];
            const int sx = a.ell_sidx[(size_t)k * a.npad + i];
            ecol[q] = col;
            if (sx >= 0) { const double ev = a.esv[sx]; eval[q] = (col == i) ? ev : ev * INV_SQRT2; }
        }
    }
    __syncthreads();

    unsigned epoch = a.epoch0;
    double carry = 0.0, beta_prev = 1.0, h1k = 0.0;
    int kstop = -1;
    bool failed = false;
    const bool timing = a.dbg != nullptr;
    long long tk[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    long long t0 = timing ? wall_clock64() : 0;
    const long long wc0 = t0, sc0 = timing ? clock64() : 0;
#define CY_TICK(q) if (timing) { const long long t1 = wall_clock64(); tk[q] += t1 - t0; t0 = t1; }

    for (int k = kfirst; k <= kd; ++k) {
        const bool first = (k == kfirst);
        // ================= close step k-1: beta, alpha (replicated), V h2 partials =================
        double beta = 1.0;
        if (!first) {
            double hh = 0.0;
            for (int j = lane; j < k; j += WAVE) hh += s_h[j] * s_h[j];
            hh = wave_sum(hh);                     // same value, same order in every wave of every workgroup
            beta = sqrt(fmax(s_sc[0] - hh, 0.0));
            const double hk1 = s_h[k - 1];
            if (tid == 0) { s_al[k - 1] = h1k + hk1 - carry; s_be[k - 1] = beta; }
            carry = hk1;
            beta_prev = beta;
            if (beta <= a.tol) { kstop = k; break; }             // invariant subspace (uniform over the grid)
            s_pc[s * R + row] = cy_dot_cols<S>(sV, R, row, s, s_h, k);
        }
        if (k == kd) {                             // the cycle's last closing: v_kd, no further mat-vec
            __syncthreads();
            if (s == 0) {
                double tot = s_pc[row];
#pragma unroll
                for (int q = 1; q < S; ++q) tot += s_pc[q * R + row];
                sV[k * R + row] = (s_w[row] - tot) / beta;
            }
            break;
        }
        // ================= operator pieces on u (= v_k at the start of a cycle, else w'_{k-1}; all rows in s_full) ========
        const double ui = s_full[min(i, G * R - 1)];
        {
            double e0 = 0.0, e1 = 0.0;
#pragma unroll
            for (int q = 0; q < CY_NE; q += 2) { e0 += eval[q] * s_full[ecol[q]]; e1 += eval[q + 1] * s_full[ecol[q + 1]]; }
            s_pe[s * R + row] = e0 + e1;
        }
        // Vp' u partials: wave (rb, s) reduces columns c = s, s+S, ... over its 64 rows
        for (int c0 = s; c0 < rp; c0 += 16 * S) {
            double t[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const int cc = c0 + c * S;
                const int ccc = min(cc, rp - 1);
                const double fv = fl ? sF[ccc * R + row] : a.Vp[(size_t)ccc * a.ldv + ic];
                t[c] = (cc < rp && rowok) ? fv * ui : 0.0;
            }
            const double cs = fold16_all(t, lane);
            const int cc = c0 + lane * S;
            if (lane < 16 && cc < rp) s_col[rb * CY_CMAX + cc] = cs;
        }
        CY_TICK(0)
        __syncthreads();                                                                    // (B)
        CY_TICK(1)
        double ei = s_pe[row];
#pragma unroll
        for (int q = 1; q < S; ++q) ei += s_pe[q * R + row];
        if (s == 0) {
            if (!first) {
                double tot = s_pc[row];
#pragma unroll
                for (int q = 1; q < S; ++q) tot += s_pc[q * R + row];
                sV[k * R + row] = (s_w[row] - tot) / beta;                                 // v_k = (w' - V h2) / beta
            }
            const double av = wave_sum(ui * ei);                                           // u'E u of this row block
            if (lane == 0) s_col[rb * CY_CMAX + rp] = av;
        }
        __syncthreads();                                                                    // (C)
        CY_TICK(2)
        // ---- X1: publish this workgroup's partials, fetch all records
        ++epoch;
        if (tid <= rp) {
            double v = s_col[tid];
#pragma unroll
            for (int q = 1; q < NWR; ++q) v += s_col[q * CY_CMAX + tid];
            cy_put(a.x1 + (size_t)g * a.xs1, tid, v);
        }
        cy_publish(a.f1, g, epoch);                                                         // (D)
        const int len1 = rp + 1;
        {
            const bool ok = cy_wait(a.f1, G, epoch, lane);
            if (ok) cy_fetch_records(a.x1, a.xs1, 0, G, len1, s_x, len1, lane, wv);
            if (__syncthreads_or(!ok)) { failed = true; break; }                            // (F)
        }
        CY_TICK(3)
        // ================= recurrence + predicted first pass (step k) =================
        // t = Vp' u (lane c and c + 64: sums over the records in ascending order), alpha~
        const double binv = first ? 1.0 : 1.0 / beta_prev;
        double tc0 = 0.0, tc1 = 0.0, vEv = 0.0;
        {
            const int c0 = min(lane, rp), c1 = min(lane + WAVE, rp);        // index rp = u'E u
            for (int q0 = 0; q0 < G; q0 += 8) {                            // 16 LDS reads in flight, fixed order
                double x0[8], x1[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int q = min(q0 + u, G - 1);
                    x0[u] = s_x[q * len1 + c0]; x1[u] = s_x[q * len1 + c1];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) if (q0 + u < G) { tc0 += x0[u]; tc1 += x1[u]; }
            }
            vEv = __shfl(lane < rp ? 0.0 : tc0, min(rp, WAVE - 1), WAVE);
            if (rp >= WAVE) vEv = __shfl(lane + WAVE < rp ? 0.0 : tc1, rp - WAVE, WAVE);
        }
        double tl = (lane < rp ? s_lam[lane] * tc0 * tc0 : 0.0) + (lane + WAVE < rp ? s_lam[lane + WAVE] * tc1 * tc1 : 0.0);
        tl = wave_sum(tl);
        const double alpha = (tl + vEv) * binv * binv;
        if (wv == 0) {
            if (lane < rp) s_u[lane] = s_lam[lane] * tc0;
            if (lane + WAVE < rp) s_u[lane + WAVE] = s_lam[lane + WAVE] * tc1;
        }
        double ck = alpha;
        if (first) {
            if (tid < kfirst) s_q[tid] = s_f[tid];
        } else {
            ck -= s_h[k - 1];
            double fh = 0.0;
            if (kfirst > 0 && k > kfirst) {
                for (int jj = lane; jj < kfirst; jj += WAVE) fh += s_f[jj] * s_h[jj];
                fh = wave_sum(fh);
            }
            if (tid < k) {
                const int j = tid;
                const double hj = s_h[j];
                double t;
                if (j < kfirst) {
                    t = s_D[j] * hj + (k > kfirst ? s_f[j] * s_h[kfirst] : 0.0);
                } else {
                    t = s_al[j] * hj;
                    if (j + 1 < k) t += s_be[j] * s_h[j + 1];
                    if (j == kfirst) t += fh;
                    else if (j > 0) t += s_be[j - 1] * s_h[j - 1];
                }
                s_q[j] = t * binv + (j == k - 1 ? beta_prev : 0.0);
            }
        }
        if (tid == k) s_q[k] = ck;
        h1k = ck;
        CY_TICK(4)
        __syncthreads();                                                                    // (H)
        CY_TICK(5)
        // w' = (e + Vp u) / beta - V q: partial over this thread's column subset
        {
            const double dv = cy_dot_cols<S>(sV, R, row, s, s_q, k + 1);
            const double du = fl ? cy_dot_cols<S>(sF, R, row, s, s_u, rp) : cy_dot_cols_g<S>(a.Vp, (size_t)a.ldv, ic, s, s_u, rp);
            s_pc[s * R + row] = dv - du * binv;
        }
        CY_TICK(6)
        __syncthreads();                                                                    // (I)
        ++epoch;                                    // epoch of X2 (partials and the rows of w' share the record)
        double wp;
        {
            double tot = s_pc[row];
#pragma unroll
            for (int q = 1; q < S; ++q) tot += s_pc[q * R + row];
            wp = rowok ? (ei * binv - tot) : 0.0;
        }
        if (s == 0) {
            s_w[row] = wp;
            cy_put(a.x2 + (size_t)g * a.xs2, (k + 2) + row, wp);            // w' rows ride behind the k+2 partials
            const double r2 = wave_sum(wp * wp);
            if (lane == 0) s_col[rb * CY_CMAX + k + 1] = r2;
        }
        // measured pass: V_k' w' partials
        for (int j0 = s; j0 <= k; j0 += 16 * S) {
            double t[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const int jj = j0 + c * S;
                t[c] = (jj <= k) ? sV[min(jj, k) * R + row] * wp : 0.0;
            }
            const double cs = fold16_all(t, lane);
            const int jj = j0 + lane * S;
            if (lane < 16 && jj <= k) s_col[rb * CY_CMAX + jj] = cs;
        }
        CY_TICK(7)
        __syncthreads();                                                                    // (K)
        CY_TICK(8)
        // ---- X2: publish the partials (the rows of w' are already in the record), fetch every record
        if (tid <= k + 1) {
            double v = s_col[tid];
#pragma unroll
            for (int q = 1; q < NWR; ++q) v += s_col[q * CY_CMAX + tid];
            cy_put(a.x2 + (size_t)g * a.xs2, tid, v);
        }
        cy_publish(a.f2, g, epoch);                                                         // (L)
        const int len2 = k + 2;
        {
            const bool ok = cy_wait(a.f2, G, epoch, lane);
            if (ok) {
                cy_fetch_records(a.x2, a.xs2, 0, G, len2, s_x, len2, lane, wv);
                cy_fetch_records(a.x2, a.xs2, len2, G, R, s_full, R, lane, wv);
            }
            if (__syncthreads_or(!ok)) { failed = true; break; }                            // (N)
        }
        if (tid <= k + 1) {
            double acc = 0.0;
            for (int q0 = 0; q0 < G; q0 += 8) {
                double x0[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) x0[u] = s_x[min(q0 + u, G - 1) * len2 + tid];
#pragma unroll
                for (int u = 0; u < 8; ++u) if (q0 + u < G) acc += x0[u];
            }
            if (tid <= k) s_h[tid] = acc; else s_sc[0] = acc;
        }
        __syncthreads();                                                                    // (A)
        CY_TICK(9)
    }
    if (failed) {
        if (tid == 0) atomicExch(a.err, 1);
        return;
    }
    if (timing && tid == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        atomicOr((unsigned long long*)(a.dbg + 15), 1ull << (xcc & 15));
        if (g == 0) {
            tk[10] = wall_clock64() - wc0; tk[11] = clock64() - sc0;    // shader cycles per 10 ns tick -> clock
            for (int q = 0; q < 12; ++q) atomicAdd((unsigned long long*)(a.dbg + q), (unsigned long long)tk[q]);
        }
    }
    // ---- write back the new basis columns and the scalars
    // columns kfirst+1 .. lastcol were created in this cycle (column kstop does not exist when stopped)
    const int lastcol = (kstop >= 0) ? kstop - 1 : kd;
    __syncthreads();
    for (int idx = tid; idx < (lastcol - kfirst) * R; idx += TPB) {
        const int j = kfirst + 1 + idx / R, r = idx % R, gi = g * R + r;
        if (gi < a.npad) a.V[(size_t)j * a.ldv + gi] = sV[j * R + r];
    }
    if (g == 0) {
        const int nsc = (kstop >= 0) ? kstop : kd;       // alphas/betas of steps kfirst .. nsc-1
        for (int j = kfirst + tid; j < nsc; j += TPB) { a.alphas[j] = s_al[j]; a.betas[j] = s_be[j]; }
        if (tid == 0) {
            a.ctl->carry = carry;
            if (kstop >= 0) { a.ctl->kstop = kstop; a.ctl->stop = 1; }
        }
    }
}

// dynamic LDS bytes of k_lz_cycle<R>
inline size_t cycle_lds_bytes(int kd, int rp, int R, int G, bool f_in_lds) {
    const int lenmax = std::max(rp + 1, kd + 2);
    const int S = NWAVE / (R / 64), NWR = R / 64;
    size_t d = (size_t)(kd + 1) * R + (f_in_lds ? (size_t)rp * R : 0) + (size_t)G * R + (size_t)G * lenmax + (size_t)R +
               2 * (size_t)S * R + (size_t)NWR * CY_CMAX + 8 * (size_t)CY_CMAX + 8;
    return d * sizeof(double);
}

}  // namespace dev
}  // namespace proxsdp
