// Hand-written gfx950 kernels of the PDHG hot path.  fp64 throughout; wave = 64.
//
// Data layout in HBM:
//   * the primal iterate x keeps the reference's order (scaling.jl:2-26): PSD
//     blocks first, each as the column-major upper triangle ("packed svec",
//     entry (i<=j) at j(j+1)/2+i, off-diagonals carrying the sqrt(2) factor),
//     then SOC variables, then free variables;
//   * there is NO dense n x n copy of a PSD block on the Lanczos path: the
//     symmetric mat-vec reads the packed triangle of x directly (8*N bytes per
//     mat-vec -- exactly what dsymv('U') reads in the reference, half of a GEMV)
//     and the projection is written back in packed form by the rank-r
//     reconstruction kernel.  psd_vec_to_square / psd_square_to_vec
//     (prox_operators.jl:1-31) therefore have no kernel of their own.
//
// All reductions are deterministic: fixed-shape shuffle trees inside a wave,
// LDS across waves, per-workgroup partials combined in a fixed order.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace proxsdp {
namespace dev {

constexpr int WAVE = 64;
constexpr int TILE = 64;            // symv / reconstruct tile side (rows = lanes)
constexpr int TPB = 256;            // threads per workgroup everywhere
constexpr int NWAVE = TPB / WAVE;
constexpr int CPW = TILE / NWAVE;   // tile columns per wave (16)
constexpr double INV_SQRT2 = 0.70710678118654752440;
constexpr double SQRT2 = 1.41421356237309504880;

// ---- intra-kernel timeline of the Lanczos step kernels (measurement builds only: -DPX_TIMELINE, tools/timeline/).
// Thread 0 of every workgroup stamps the 100 MHz device-wide clock (s_memrealtime) at fixed points of the step
// kernels: g_tl[kind][step k][workgroup][slot], kind 0 = k_fop_finish / k_symv_finish, 1 = k_lz_orth.
// The product library is built WITHOUT the macro: PX_TL / PX_TL_DRAIN expand to nothing.
#ifdef PX_TIMELINE
constexpr int TL_WG = 160, TL_SLOTS = 8, TL_K = 260;
__device__ unsigned long long g_tl[2][TL_K][TL_WG][TL_SLOTS];
#define PX_TL(kind, k, slot)                                                                                   \
    do {                                                                                                       \
        if (threadIdx.x == 0 && (int)blockIdx.x < ::proxsdp::dev::TL_WG && (k) >= 0 && (k) < ::proxsdp::dev::TL_K) { \
            asm volatile("" ::: "memory");                                                                     \
            ::proxsdp::dev::g_tl[kind][k][blockIdx.x][slot] = wall_clock64();                                    \
            asm volatile("" ::: "memory");                                                                     \
        }                                                                                                      \
    } while (0)
#define PX_TL_DRAIN() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#else
#define PX_TL(kind, k, slot) do { } while (0)
#define PX_TL_DRAIN() do { } while (0)
#endif

struct LanczosCtl {                 // device-resident control block of one PSD block
    int stop;                       // set when beta <= tol (invariant subspace)
    int kstop;                      // basis size at which it stopped
    int pad[2];
    double carry;                   // h2[k] of the last closed step (alpha correction, see k_symv_finish)
};

// ---- cross-lane primitives (gfx950).  __shfl_* lowers to ds_bpermute_b32 (an LDS
// round trip, ~100+ cycles) per 32-bit half; the reductions below are latency
// chains, so they use DPP row shifts / broadcasts (a few cycles each) instead.
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ double dpp_f64(double old, double v) {
    int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(v), CTRL, ROW_MASK, BANK_MASK, false);
    int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(v), CTRL, ROW_MASK, BANK_MASK, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double bcast_lane63(double v) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}
// sum over the 64 lanes, result in EVERY lane.  Fixed shape: row_shr 1,2,4,8 inside
// each row of 16, row_bcast15 into rows 1/3, row_bcast31 into rows 2/3, lane 63 holds
// the total.  Must be called by a full wave.
__device__ __forceinline__ double wave_sum(double v) {
    v += dpp_f64<0x111, 0xF, 0xF>(0.0, v);      // row_shr:1
    v += dpp_f64<0x112, 0xF, 0xF>(0.0, v);      // row_shr:2
    v += dpp_f64<0x114, 0xF, 0xF>(0.0, v);      // row_shr:4
    v += dpp_f64<0x118, 0xF, 0xF>(0.0, v);      // row_shr:8
    v += dpp_f64<0x142, 0xA, 0xF>(0.0, v);      // row_bcast:15 -> rows 1,3
    v += dpp_f64<0x143, 0xC, 0xF>(0.0, v);      // row_bcast:31 -> rows 2,3
    return bcast_lane63(v);
}
__device__ __forceinline__ double wave_max(double v) {
    v = fmax(v, dpp_f64<0x111, 0xF, 0xF>(v, v));
    v = fmax(v, dpp_f64<0x112, 0xF, 0xF>(v, v));
    v = fmax(v, dpp_f64<0x114, 0xF, 0xF>(v, v));
    v = fmax(v, dpp_f64<0x118, 0xF, 0xF>(v, v));
    v = fmax(v, dpp_f64<0x142, 0xA, 0xF>(v, v));
    v = fmax(v, dpp_f64<0x143, 0xC, 0xF>(v, v));
    return bcast_lane63(v);
}
// value of lane (lane ^ H); H = 1, 2, 4, 8 by DPP (quad_perm / row shifts / row_ror)
template <int H>
__device__ __forceinline__ double lane_xor(double v) {
    if constexpr (H == 1) return dpp_f64<0xB1, 0xF, 0xF>(v, v);          // quad_perm [1,0,3,2]
    else if constexpr (H == 2) return dpp_f64<0x4E, 0xF, 0xF>(v, v);     // quad_perm [2,3,0,1]
    else if constexpr (H == 4) {
        double t = dpp_f64<0x104, 0xF, 0x5>(v, v);                      // row_shl:4 into banks 0,2
        return dpp_f64<0x114, 0xF, 0xA>(t, v);                          // row_shr:4 into banks 1,3
    } else if constexpr (H == 8) return dpp_f64<0x128, 0xF, 0xF>(v, v);  // row_ror:8
    else return __shfl_xor(v, H, WAVE);
}
// v + (value of lane ^ 16) and v + (value of lane ^ 32) through the gfx950
// v_permlane16_swap / v_permlane32_swap instructions (no LDS)
__device__ __forceinline__ double add_xor16(double v) {
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
}
__device__ __forceinline__ double add_xor32(double v) {
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
}
// block-wide sum; result valid in thread 0.  `sm` needs NWAVE doubles.  Must be
// called by every thread of the workgroup.
__device__ __forceinline__ double block_sum(double v, double* sm) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sm[w] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < NWAVE; ++i) r += sm[i];
    }
    return r;
}
__device__ __forceinline__ double block_max(double v, double* sm) {
    v = wave_max(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sm[w] = v;
    __syncthreads();
    double r = sm[0];
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 1; i < NWAVE; ++i) r = fmax(r, sm[i]);
    }
    return r;
}

// linear tile id (column-major over the upper block-triangle) -> (I <= J)
__device__ __forceinline__ void tile_coords(int t, int& I, int& J) {
    int j = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
    while ((long long)(j + 1) * (j + 2) / 2 <= t) ++j;
    while ((long long)j * (j + 1) / 2 > t) --j;
    J = j;
    I = t - (int)((long long)j * (j + 1) / 2);
}

// ---------------------------------------------------------------------------
// Symmetric mat-vec on the packed triangle:  u = P~ v  with P~ the symmetric
// fill of the raw packed values, diagonal pre-multiplied by sqrt(2), so that
// smat(xp) v = u / sqrt(2)  (the off-diagonals of xp carry a sqrt(2) factor,
// prox_operators.jl:5-13).  Replaces dsymv('U') (eigsolver.jl:678 / KrylovKit).
// Bound: HBM -- 8*N bytes of the packed triangle per mat-vec, read once.
//
// One workgroup (4 waves) per 64x64 tile (I<=J) of the upper block-triangle.
// Lane = tile row, each wave owns 16 tile columns and issues its 16 column loads
// back to back (each load instruction reads 512 contiguous bytes of one packed
// column), then
//   row sums   : racc += T[r,c] * v[J*64+c]        (v[J*64+c] is wave-uniform:
//                scalar loads), the 4 waves' partial row sums meet once in LDS
//   column sums: in-register transpose-reduce of T[r,c]*v[I*64+r]: 4 fold stages
//                halve the live columns per lane (8+4+2+1 exchanges), two plain
//                exchanges finish -- 17 lane exchanges for 16 columns instead of
//                16 full wave reductions, and no LDS round trip of the tile.
// The tile contributes rows of block I (slot J, row sums) and rows of block J
// (slot I, column sums) to Ppart[slot][row]; every (slot,row) is written exactly
// once per mat-vec: no zero-fill, no atomics, and the consumer sums the nt slots
// in a fixed order (deterministic).
// ---------------------------------------------------------------------------
template <int H, int NARR>
__device__ __forceinline__ void fold_stage(double (&t)[NARR], int lane) {
    // lanes whose bit H is set keep columns [H,2H) of the current set, the others [0,H)
    const bool upper = (lane & H) != 0;
#pragma unroll
    for (int m = 0; m < H; ++m) {
        const double lo = t[m], hi = t[m + H];
        const double send = upper ? lo : hi;
        const double keep = upper ? hi : lo;
        t[m] = keep + lane_xor<H>(send);
    }
}

// One workgroup (4 waves) per 64x64 tile; wave w owns tile columns [16w, 16w+16).
// Load phase: the 16 column loads of this wave (512 contiguous bytes each).
__device__ __forceinline__ void symv_load(const double* __restrict__ xp, int n, int tile, int lane, int wv,
                                          double (&t)[CPW]) {
    int I, J;
    tile_coords(tile, I, J);
    const int gi = I * TILE + lane;
    const int j0 = J * TILE + wv * CPW;              // first global column of this wave
    const bool diag = (I == J);
    const bool interior = (I < J) && (J * TILE + TILE <= n);
    if (interior) {
        const double* __restrict__ xrow = xp + gi;
#pragma unroll
        for (int c = 0; c < CPW; ++c) {
            const long long gj = j0 + c;
            t[c] = xrow[gj * (gj + 1) / 2];
        }
    } else {
        // diagonal / ragged tiles: unconditional loads from clamped (always valid)
        // addresses, masked afterwards -- a branch per load would serialise the
        // memory round trips
        const int gic = min(gi, n - 1);
#pragma unroll
        for (int c = 0; c < CPW; ++c) {
            const int gj = j0 + c;
            const int gjc = min(gj, n - 1);
            const double a = xp[(long long)gjc * (gjc + 1) / 2 + min(gic, gjc)];
            const bool ok = (gj < n) && (gi < n) && (gi <= gj);
            // the diagonal entry is counted once (row sum), as X_ii = xp_ii
            t[c] = ok ? ((diag && gi == gj) ? a * SQRT2 : a) : 0.0;
        }
    }
}
// Reduce phase: row sums (LDS meeting of the 4 waves) and column sums (in-register fold).
// Also emits this tile's share of v' P~ v (2 * v_I' T v_J for an off-diagonal tile) into
// Apart[tile]: summed over tiles it gives the Rayleigh quotient alpha without a separate
// dot-product kernel.
__device__ __forceinline__ void symv_reduce(int npad, const double* __restrict__ v, double* __restrict__ Ppart,
                                            int tile, int lane, int wv, double (&t)[CPW],
                                            double* __restrict__ s_row, double* __restrict__ s_col,
                                            double* __restrict__ Apart) {
    int I, J;
    tile_coords(tile, I, J);
    const int gi = I * TILE + lane;
    const int j0 = J * TILE + wv * CPW;
    const double* __restrict__ vJ = v + j0;          // uniform address: scalar loads
    const double vi = v[gi];                         // v is zero-padded to npad
    const bool diag = (I == J);
    double racc = 0.0;
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
        racc += t[c] * vJ[c];
        t[c] *= vi;
        if (diag && gi == j0 + c) t[c] = 0.0;        // the diagonal entry is in the row sum only
    }
    s_row[wv * TILE + lane] = racc;
    const double aw = diag ? 0.0 : wave_sum(vi * racc);           // this wave's share of v_I' T v_J
    // column sums: fold 16 columns over lane bits 3..0, then all-reduce over bits 4,5
    fold_stage<8>(t, lane);
    fold_stage<4>(t, lane);
    fold_stage<2>(t, lane);
    fold_stage<1>(t, lane);
    double cs = add_xor16(t[0]);
    cs = add_xor32(cs);                              // column (lane & 15) of this wave's strip
    __syncthreads();
    if (diag) {
        // rows of block I get row sums and column sums (same slot)
        if (lane < CPW) s_col[wv * CPW + lane] = cs;
        __syncthreads();
        if (wv == 0) {
            const double rs = (s_row[lane] + s_row[TILE + lane]) + (s_row[2 * TILE + lane] + s_row[3 * TILE + lane]);
            Ppart[(long long)I * npad + gi] = rs + s_col[lane];
            const double a = wave_sum(vi * (rs + s_col[lane]));   // v_I' P~_II v_I
            if (lane == 0) Apart[tile] = a;
        }
    } else {
        if (lane == 0) s_col[wv] = aw;           // s_col is free on off-diagonal tiles
        __syncthreads();
        if (wv == 0) {
            const double rs = (s_row[lane] + s_row[TILE + lane]) + (s_row[2 * TILE + lane] + s_row[3 * TILE + lane]);
            Ppart[(long long)J * npad + gi] = rs;                    // rows of block I, slot J
            if (lane == 0) Apart[tile] = 2.0 * ((s_col[0] + s_col[1]) + (s_col[2] + s_col[3]));
        }
        if (lane < CPW) Ppart[(long long)I * npad + j0 + lane] = cs; // rows of block J, slot I
    }
}
// SYMV_TPW tiles per workgroup (tile ids first, first + stride, ...): the loads of the next
// tile are issued before the current tile is reduced, so the reduction (shuffles, LDS
// meeting, barrier) of one tile overlaps the memory latency of the next.
// XCD-aware tile order.  Workgroup b runs on XCD b % 8 (each XCD has its own L2).  A column
// segment of a tile is 512 bytes at an arbitrary 8-byte alignment, i.e. it straddles 5 cache
// lines of 128 B, and the vertically adjacent tile (I+1, J) needs the same boundary lines:
// with tiles dealt round-robin those lines are fetched twice (measured: 2*FETCH_SIZE =
// 79.8 MB for 64.1 MB algorithmic at n = 4000).  Here XCD x walks the contiguous tile range
// [x*q, (x+1)*q) in order, so the neighbour's lines are still in that XCD's L2.
__device__ __forceinline__ int xcd_tile(int b, int ntile) {
    const int q = (ntile + 7) >> 3;
    return (b & 7) * q + (b >> 3);            // >= ntile for the padding workgroups
}
constexpr int SYMV_TPW = 1;          // measured: 2 tiles/workgroup is not faster (16.0 vs 14.7 us at n=4000)
__device__ __forceinline__ void symv_tiles(const double* __restrict__ xp, int n, int npad, int ntile,
                                           const double* __restrict__ v, double* __restrict__ Ppart,
                                           int first, int stride, double* __restrict__ s_row,
                                           double* __restrict__ s_col, double* __restrict__ Apart) {
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    double ta[CPW], tb[CPW];
    int tile = xcd_tile(first, ntile);
    if (tile >= ntile) return;
    symv_load(xp, n, tile, lane, wv, ta);
#pragma unroll
    for (int it = 0; it < SYMV_TPW; ++it) {
        const int next = tile + stride;
        const bool has_next = (it + 1 < SYMV_TPW) && (next < ntile);
        if (has_next) symv_load(xp, n, next, lane, wv, tb);
        symv_reduce(npad, v, Ppart, tile, lane, wv, ta, s_row + (it & 1) * (NWAVE * TILE), s_col + (it & 1) * TILE, Apart);
        if (!has_next) break;
#pragma unroll
        for (int c = 0; c < CPW; ++c) ta[c] = tb[c];
        tile = next;
    }
}

__global__ void __launch_bounds__(TPB)
k_symv_packed(const double* __restrict__ xp, int n, int nt, int npad,
              const double* __restrict__ v, double* __restrict__ Ppart,
              const LanczosCtl* __restrict__ ctl, double* __restrict__ Apart) {
    if (ctl != nullptr && ctl->stop) return;
    __shared__ double s_row[2 * NWAVE * TILE];
    __shared__ double s_col[2 * TILE];
    symv_tiles(xp, n, npad, nt * (nt + 1) / 2, v, Ppart, blockIdx.x, gridDim.x, s_row, s_col, Apart);
}

// ---------------------------------------------------------------------------
// Lanczos recurrence with full re-orthogonalisation, scalars kept on the device:
// replaces the BLAS-1 work inside KrylovKit's LanczosIterator (call site
// eigsolver.jl:802; orth = two Gram-Schmidt passes against the whole basis).
// TWO dependent launches per Lanczos step:
//   k_symv_finish : closes step k-1 (beta^2 = |w'|^2 - |h2|^2, alpha, v_k = (w' - V h2)/beta,
//                   or stop) in nt workgroups while the other workgroups run the mat-vec
//                   tiles of step k on the un-corrected w'
//   k_lz_orth     : w = A w'/beta from the partial slots; first pass PREDICTED from the
//                   Lanczos relation (recurrence terms + image of the un-applied correction);
//                   second pass MEASURED: hpart = V' w', |w'|^2
// Every kernel boundary is a global reduction; a measured first pass would cost a third
// launch per step (14.1 + 7.2 + 8.5 us -> 14.1 + 7.4 us at n = 4000, K = 53).
// Grid = nt workgroups of 64 rows; the 4 waves of a workgroup split the slots and the
// basis columns j and meet in LDS.  These kernels are latency-bound (n*K*8 bytes of
// L2-resident basis), so the point of the layout is parallel width and loads in flight,
// not bytes.
// ---------------------------------------------------------------------------
constexpr int MAXK = 260;           // capacity of the Krylov basis (krylovdim + 1 <= 256: the step kernels hold
                                    // 16*NCH <= 64 basis columns per wave in registers, 4 waves, and map
                                    // thread j <-> basis column j) => target rank <= 127
constexpr int LZ_ROWS = TILE;       // rows per workgroup in the Lanczos vector kernels
constexpr int NRM_SLOT = MAXK - 1;  // slot of a partial-dots row that carries |w'|^2

// Partial dots live PRODUCER-major: workgroup g writes its record hpart[g * KLD + j] (column j of the basis),
// its |w'|^2 share at hpart[pld * KLD + g] (pld = number of workgroups rounded up to 64, padding records stay
// zero); likewise tpart[g * RLD + c] for the operator form's Vp'v partials.  The consumer maps LANE <-> COLUMN:
// wave wv holds the records g = wv, wv + 4, ... (one coalesced load per record, no cross-lane traffic), adds them in
// registers and the four waves meet once in LDS.  (Round 3 kept the partials column-major and reduced 16 columns at
// a time over the lanes with the mat-vec's fold network: by the step timeline, profiles/r04_step_timeline.md, that
// network was 1.7 us of the closing workgroups' 5.6 us.)  The ORDER of the additions is that network's balanced
// tree -- partner g^8, then g^4, g^2, g^1, g^16, g^32 -- restated on the new layout (tree_in_wave / tree_across), so
// every sum has the bits it had before: the committed iteration counts and rank schedules do not move.
//   g = wv + 4u:  g^8 <-> u^2, g^4 <-> u^1 (inside the wave);  g^2, g^1 <-> the other waves;  g^16 <-> u^4, g^32 <-> u^8
__device__ __forceinline__ void tree_in_wave(const double (&p)[16], double (&q)[4]) {   // levels g^8, g^4: 16 records -> 4 groups
#pragma unroll
    for (int a = 0; a < 4; ++a) q[a] = (p[4 * a] + p[4 * a + 2]) + (p[4 * a + 1] + p[4 * a + 3]);
}
// s[(wave * 4 + group) * stride + column]: levels g^2, g^1 (waves), then g^16, g^32 (groups)
__device__ __forceinline__ double tree_across(const double* __restrict__ s, int stride, int col) {
    double a[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
        a[q] = (s[(0 * 4 + q) * stride + col] + s[(2 * 4 + q) * stride + col]) + (s[(1 * 4 + q) * stride + col] + s[(3 * 4 + q) * stride + col]);
    return (a[0] + a[1]) + (a[2] + a[3]);
}
constexpr int KLD = 256;            // >= krylovdim + 1
constexpr int RLD = 128;            // >= rank of the previous projection's factors (<= 127)

// start of a projection: V[:,0] = start vector, control block cleared (one launch instead of a
// device-to-device copy plus a memset in the solve stream)
__global__ void __launch_bounds__(TPB)
k_lz_begin(double* __restrict__ V0, const double* __restrict__ resid, int npad, LanczosCtl* __restrict__ ctl) {
    const int i = blockIdx.x * TPB + threadIdx.x;
    if (i < npad) V0[i] = resid[i];
    if (i == 0) { ctl->stop = 0; ctl->kstop = 0; ctl->carry = 0.0; }
}

// library-only warm start (options.lanczos_warm_start): the start vector of a projection is the sum
// of the previous projection's Ritz vectors plus 1e-3 x the fixed start vector (so that the Krylov
// space still reaches directions the previous factors do not span), normalised.  Two launches:
// rows + per-workgroup sums of squares, then every workgroup adds the partials in the same order.
__global__ void __launch_bounds__(TPB)
k_lz_warm_sum(double* __restrict__ V0, const double* __restrict__ F, int ldf, int rp, const double* __restrict__ resid,
              int npad, double* __restrict__ part, const double* __restrict__ lam, double wpow) {
    __shared__ double sm[NWAVE];
    __shared__ double sw[TPB];
    // weights (wpow != 0, positive-part runs only): column c enters with (lam_0 / lam_c)^wpow -- the pairs with the SMALL
    // eigenvalues are the ones the run converges last, so they get the larger share of the start vector
    if (wpow != 0.0 && lam != nullptr) {
        const int c = threadIdx.x;
        // (the ratio is capped: a retained Ritz value at 1e-14 of the scale, or a denormal one, must not turn the start vector into
        // that one near-null direction -- or into NaN after the normalisation)
        sw[c] = (c < rp && lam[c] > 0.0 && lam[0] > 0.0) ? pow(fmin(lam[0] / lam[c], 1e6), wpow) : 1.0;
        __syncthreads();
    }
    const int i = blockIdx.x * TPB + threadIdx.x;
    double v = 0.0;
    if (i < npad) {
        if (wpow != 0.0 && lam != nullptr) { for (int c = 0; c < rp; ++c) v += sw[min(c, TPB - 1)] * F[(long long)c * ldf + i]; }
        else for (int c = 0; c < rp; ++c) v += F[(long long)c * ldf + i];
        v += 1e-3 * resid[i];
        V0[i] = v;
    }
    const double tot = block_sum(v * v, sm);
    if (threadIdx.x == 0) part[blockIdx.x] = tot;
}
__global__ void __launch_bounds__(TPB)
k_lz_warm_scale(double* __restrict__ V0, int npad, const double* __restrict__ part, int nparts, LanczosCtl* __restrict__ ctl) {
    double ss = 0.0;
    for (int q = 0; q < nparts; ++q) ss += part[q];               // same order in every thread
    const int i = blockIdx.x * TPB + threadIdx.x;
    if (i < npad) V0[i] = V0[i] / sqrt(ss);
    if (i == 0) { ctl->stop = 0; ctl->kstop = 0; ctl->carry = 0.0; }
}

// y = smat(xp) v from the mat-vec partial slots (test seam / residual checks):
// w = (sum of slots) / sqrt2, fixed order.
__global__ void __launch_bounds__(TPB)
k_symv_collect(const double* __restrict__ Ppart, int nt, int npad, double* __restrict__ wbuf) {
    __shared__ double s_acc[NWAVE][LZ_ROWS];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int i = blockIdx.x * LZ_ROWS + lane;
    double a0 = 0.0;
    for (int s = wv; s < nt; s += NWAVE) a0 += Ppart[(long long)s * npad + i];
    s_acc[wv][lane] = a0;
    __syncthreads();
    if (wv == 0) wbuf[i] = ((s_acc[0][lane] + s_acc[1][lane]) + (s_acc[2][lane] + s_acc[3][lane])) * INV_SQRT2;
}

// Lanczos recurrence and re-orthogonalisation measurements of step k in ONE kernel.
//
// The mat-vec of step k ran on the un-normalised, un-corrected w'_{k-1} (see k_symv_finish):
// with h = h2_{k-1} (the measured pass of step k-1), beta = beta_{k-1} and the Lanczos
// relation A V_{k-1} = V_{k-1} T_{k-1} + v_k r',
//     w = A w'_{k-1}/beta = A v_k + V_{k-1} (T_{k-1} h)/beta + v_k c,     c = r'h/beta = h[k-1].
// The bracketed terms are KNOWN (T is on the device: alphas/betas, and after a thick restart
// the arrow part D, f in `arrow`), so they are subtracted here together with the recurrence
// terms, instead of being measured by a separate dot-product launch.  Left unsubtracted they
// would be re-injected every step and grow like (|T|/beta)^k.
//     alpha~ = w' P~ w' / (sqrt2 beta^2)  from the mat-vec tiles' shares  (= alpha_k + 2c)
//     w'     = w - V_{k-1} (T h/beta + r) - v_k (alpha~ - c)
// then ONE measured full pass: hpart_out = V_k' w' and |w'|^2, applied by the closing kernel.
// hsum_out[k] = alpha~ - c, so the closing kernel's alpha_k = hsum[k] + h2[k] - carry holds.
// `first`: v_k is an exact, normalised basis vector (start of a cycle): no correction terms;
// after a restart (keep > 0, k == keep) r = f.
// arrow[0..MAXK) = f, arrow[MAXK..2 MAXK) = D (Ritz values), valid for indices < keep.
// Latency-bound kernel (63 workgroups at n = 4000): EVERY global load is issued before the
// first use, and each wave keeps its basis columns j = wv + 4c in registers for both the
// subtraction and the measured pass; 16 column sums are reduced together by the fold network
// of the mat-vec (15 + 2 DPP steps instead of 16 x 6).  NCH = 16-column chunks per wave
// (k + 1 <= 64 NCH).
__device__ __forceinline__ double fold16_all(double (&t)[16], int lane) {
    fold_stage<8>(t, lane);
    fold_stage<4>(t, lane);
    fold_stage<2>(t, lane);
    fold_stage<1>(t, lane);
    return add_xor32(add_xor16(t[0]));               // lane holds column (lane & 15)
}
// Operator-form inputs of k_lz_orth / k_fop* (see "Operator-form mat-vec" below)
struct FopArgs {
    const double* Vp;        // previous projection's Ritz vectors, column c at Vp + c*ldv
    const double* lam;       // their (positive) eigenvalues
    int rp;                  // how many (0: x_prev = 0 off the support)
    const double* tpart;     // [rp][pld] partial dots Vp' v written by k_fop*
    const double* ebuf;      // (E v)_i
    const double* apart;     // [pld] partials of v' E v
};
// NCHP = 0: the mat-vec came from the packed tiles (Ppart / Apart).  NCHP > 0: it is rebuilt from
// the operator-form pieces (16*NCHP*4 >= rp): A v = Vp (lam o (Vp' v)) + E v.
template <int NCH, int NCHP>
__device__ __forceinline__ void
lz_orth_body(const double* __restrict__ Ppart, int nt, int npad, const double* __restrict__ V, int ldv, int k,
             double* __restrict__ wbuf, const double* __restrict__ hred, double* __restrict__ hpart_out, int pld,
             double* __restrict__ hsum_out, const LanczosCtl* __restrict__ ctl,
             const double* __restrict__ alphas, const double* __restrict__ betas,
             const double* __restrict__ Apart, int napart, int first, const double* __restrict__ arrow, int keep,
             const FopArgs& fo) {
    PX_TL(1, k, 0);
    const int stop = ctl->stop;                      // tested after the loads below are in flight
    constexpr int NC = 16 * NCH;
    constexpr int NCP = 16 * (NCHP > 0 ? NCHP : 1);
    __shared__ double s_t[NWAVE * 4 * 4 * NCP];      // per-wave group sums of the producers' Vp'v records, [wave][group][column]
    __shared__ double s_u[4 * NCP];
    __shared__ double s_h[4 * NC];
    __shared__ double s_q[4 * NC];
    __shared__ double s_acc[NWAVE][LZ_ROWS];
    __shared__ double s_red[NWAVE];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int i = blockIdx.x * LZ_ROWS + lane;       // < npad always; rows >= n carry zeros
    // ---- all loads
    double pv[16];                                   // mat-vec partial slots wv, wv+4, ...
    double av[8];                                    // per-tile shares of w' P~ w'
    constexpr int NLP = (NCHP > 0 ? NCHP : 1);       // 64-column chunks of the Vp'v records (lane <-> column)
    double vrp[NCP], tp[16][NLP], eb = 0.0, ap = 0.0; // operator form: Vp columns, Vp'v partials, (E v)_i, v'Ev partials
    if constexpr (NCHP == 0) {
#pragma unroll
        for (int u = 0; u < 16; ++u) pv[u] = (wv + u * NWAVE < nt) ? Ppart[(long long)(wv + u * NWAVE) * npad + i] : 0.0;   // (uniform test)
#pragma unroll
        for (int u = 0; u < 8; ++u) av[u] = (u * TPB < napart) ? Apart[min((int)threadIdx.x + u * TPB, napart - 1)] : 0.0;
    } else {
        const int gl0 = min(lane, pld - 1);
#pragma unroll
        for (int c = 0; c < NCP; ++c) {
            const int cc = min(wv + 4 * c, max(fo.rp - 1, 0));
            vrp[c] = fo.Vp[(long long)cc * ldv + i];
        }
        // Vp'v partials: records of the producers wv, wv + 4, ... (lane <-> column)
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const long long rec = (long long)min(wv + NWAVE * u, pld - 1) * RLD;
#pragma unroll
            for (int ch = 0; ch < NLP; ++ch)
                tp[u][ch] = ((ch == 0 || 64 * ch < fo.rp) && (wv + NWAVE * u < nt || pld > WAVE)) ? fo.tpart[rec + 64 * ch + lane] : 0.0;
        }
        eb = fo.ebuf[i];
        ap = fo.apart[gl0];
    }
    double vr[NC];                                   // basis columns of this wave (column k included): groups of four,
#pragma unroll                                       // groups beyond column k are not loaded
    for (int c0 = 0; c0 < NC; c0 += 4) {
        if (wv + 4 * c0 <= k) {
#pragma unroll
            for (int c = c0; c < c0 + 4; ++c) vr[c] = V[(long long)min(wv + 4 * c, k) * ldv + i];
        } else {
#pragma unroll
            for (int c = c0; c < c0 + 4; ++c) vr[c] = 0.0;
        }
    }
    // h2 of step k-1, already reduced by the closing workgroup 0 of the previous launch
    const double hred_j = hred[min((int)threadIdx.x, MAXK - 1)];
    // coefficient data of the prediction (thread j <-> basis column j; lane-indexed copies for
    // the two short dot products): issued with everything else, used after the first barrier
    const int j = threadIdx.x;
    const int jc = min(j, MAXK - 1);
    const double al_j = alphas[jc], be_j = betas[jc], be_jm = betas[max(jc - 1, 0)];
    const double f_j = arrow[jc], d_j = arrow[MAXK + jc];
    const double be_km = betas[max(k - 1, 0)];
    const double f_l0 = arrow[lane], f_l1 = arrow[lane + WAVE];
    double lam_j = 0.0, lam_l0 = 0.0, lam_l1 = 0.0;
    if constexpr (NCHP > 0) {
        lam_j = fo.lam[jc]; lam_l0 = fo.lam[lane];
        if constexpr (NCHP > 1) lam_l1 = fo.lam[lane + WAVE];
    }
    if (stop) return;
    PX_TL(1, k, 1);
    PX_TL_DRAIN();
    PX_TL(1, k, 2);
    const double binv = first ? 1.0 : 1.0 / be_km;
    // ---- reductions
    if constexpr (NCHP == 0) {
#pragma unroll
        for (int u = 0; u < 16; ++u) if (wv + u * NWAVE >= nt) pv[u] = 0.0;
        double acc = (((pv[0] + pv[1]) + (pv[2] + pv[3])) + ((pv[4] + pv[5]) + (pv[6] + pv[7]))) +
                     (((pv[8] + pv[9]) + (pv[10] + pv[11])) + ((pv[12] + pv[13]) + (pv[14] + pv[15])));
        for (int s = wv + 16 * NWAVE; s < nt; s += NWAVE) acc += Ppart[(long long)s * npad + i];      // n > 4096
        s_acc[wv][lane] = acc;
#pragma unroll
        for (int u = 0; u < 8; ++u) if ((int)threadIdx.x + u * TPB >= napart) av[u] = 0.0;
        double a = ((av[0] + av[1]) + (av[2] + av[3])) + ((av[4] + av[5]) + (av[6] + av[7]));
        for (int t = threadIdx.x + 8 * TPB; t < napart; t += TPB) a += Apart[t];                      // n > 4096
        a = wave_sum(a);
        if (lane == 0) s_red[wv] = a;
    } else {
        // t = Vp' v: this wave's share of the producers' records, fixed order; the four waves meet in s_t
#pragma unroll
        for (int ch = 0; ch < NLP; ++ch) {
            double pr[16], q4[4];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                pr[u] = tp[u][ch];
                if (64 * ch < fo.rp)
                    for (int g = wv + NWAVE * u + WAVE; g < pld; g += WAVE) pr[u] += fo.tpart[(long long)g * RLD + 64 * ch + lane];   // n > 4096
            }
            tree_in_wave(pr, q4);
#pragma unroll
            for (int a = 0; a < 4; ++a) s_t[(wv * 4 + a) * (64 * NLP) + 64 * ch + lane] = q4[a];
        }
        if (lane >= pld) ap = 0.0;
        else for (int g = lane + WAVE; g < pld; g += WAVE) ap += fo.apart[g];
        ap = wave_sum(ap);
        if (lane == 0) s_red[wv] = (wv == 0) ? ap : 0.0;
    }
    if (!first && (int)threadIdx.x < 4 * NC) s_h[threadIdx.x] = ((int)threadIdx.x < k) ? hred_j : 0.0;
    __syncthreads();
    PX_TL(1, k, 3);
    // ---- coefficients: s_q[j], j < k, over V_{k-1}; s_q[k] = coefficient of v_k
    double alpha, wi;
    if constexpr (NCHP == 0) {
        alpha = ((s_red[0] + s_red[1]) + (s_red[2] + s_red[3])) * INV_SQRT2 * binv * binv;
        wi = ((s_acc[0][lane] + s_acc[1][lane]) + (s_acc[2][lane] + s_acc[3][lane])) * (INV_SQRT2 * binv);
    } else {
        // v'Av = t' Lam t + v'Ev;   u = Lam t for the row-wise rebuild below
        auto tsum = [&](int c) { return tree_across(s_t, 64 * NLP, c); };   // t_c from the waves' group sums (c < 64 NLP)
        double tl = 0.0;
        if (lane < fo.rp) { const double t0 = tsum(lane); tl = lam_l0 * t0 * t0; }
        if constexpr (NCHP > 1) if (lane + WAVE < fo.rp) { const double t1 = tsum(lane + WAVE); tl += lam_l1 * t1 * t1; }
        tl = wave_sum(tl);
        alpha = (tl + s_red[0]) * binv * binv;
        wi = eb * binv;                              // + Vp u / beta, folded into the exchange below
        if (j < 4 * NCP) s_u[j] = (j < fo.rp) ? lam_j * tsum(j) : 0.0;
    }
    double ck = alpha;
    if (first) {
        if (j < keep) s_q[j] = f_j;
    } else {
        ck -= s_h[k - 1];
        double fh = 0.0;
        if (keep > 0 && k > keep) {                   // f'h (row `keep` of the arrow)
            if (lane < keep) fh = f_l0 * s_h[lane];
            if (lane + WAVE < keep) fh += f_l1 * s_h[lane + WAVE];
            for (int jj = lane + 2 * WAVE; jj < keep; jj += WAVE) fh += arrow[jj] * s_h[jj];
            fh = wave_sum(fh);
        }
        if (j < k) {
            const double hj = s_h[j];
            double t;
            if (j < keep) {
                t = d_j * hj + (k > keep ? f_j * s_h[keep] : 0.0);
            } else {
                t = al_j * hj;
                if (j + 1 < k) t += be_j * s_h[j + 1];
                if (j == keep) t += fh;                               // fh = 0 when keep == 0
                else if (j > 0) t += be_jm * s_h[j - 1];
            }
            s_q[j] = t * binv + (j == k - 1 ? be_km : 0.0);
        }
    }
    if (j == k) s_q[k] = ck;
    else if (j > k && j < 4 * NC) s_q[j] = 0.0;       // (and `first` with j in [keep, k): set below)
    if (first && j >= keep && j < k) s_q[j] = 0.0;
    __syncthreads();
    PX_TL(1, k, 4);
    // ---- w' = w - V q   (each wave its own columns, from registers; s_q is zero beyond column k, so no test per term:
    // the LDS reads are issued together)
    double qv[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) qv[c] = s_q[wv + 4 * c];
    double d0 = 0.0, d1 = 0.0;
#pragma unroll
    for (int c = 0; c < NC; c += 2) {
        d0 += vr[c] * qv[c];
        d1 += vr[c + 1] * qv[c + 1];
    }
    double dsub = d0 + d1;
    if constexpr (NCHP > 0) {                         // (A v)_i = e_i + sum_c Vp[i,c] u_c
        double u0 = 0.0, u1 = 0.0;
#pragma unroll
        for (int c = 0; c < NCP; c += 2) {
            u0 += vrp[c] * s_u[wv + 4 * c];           // s_u = 0 beyond rp
            u1 += vrp[c + 1] * s_u[wv + 4 * (c + 1)];
        }
        dsub -= (u0 + u1) * binv;
    }
    s_acc[wv][lane] = dsub;                           // (all reads of s_acc for wi precede the barrier above)
    __syncthreads();
    PX_TL(1, k, 5);
    const double wp = wi - ((s_acc[0][lane] + s_acc[1][lane]) + (s_acc[2][lane] + s_acc[3][lane]));
    if (wv == 0) {
        wbuf[i] = wp;
        const double r = wave_sum(wp * wp);
        if (lane == 0) hpart_out[(long long)pld * KLD + blockIdx.x] = r;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) hsum_out[k] = ck;
    // ---- measured pass: V_k' w'
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        double t[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) t[c] = (wv + 4 * (16 * ch + c) <= k) ? vr[16 * ch + c] * wp : 0.0;
        const double hs = fold16_all(t, lane);
        const int jc = wv + 4 * (16 * ch + lane);
        if (lane < 16 && jc <= k) hpart_out[(long long)blockIdx.x * KLD + jc] = hs;
    }
    PX_TL(1, k, 6);
}
template <int NCH, int NCHP>
__global__ void __launch_bounds__(TPB)
k_lz_orth(const double* __restrict__ Ppart, int nt, int npad, const double* __restrict__ V, int ldv, int k,
          double* __restrict__ wbuf, const double* __restrict__ hred, double* __restrict__ hpart_out, int pld,
          double* __restrict__ hsum_out, const LanczosCtl* __restrict__ ctl,
          const double* __restrict__ alphas, const double* __restrict__ betas,
          const double* __restrict__ Apart, int napart, int first, const double* __restrict__ arrow, int keep,
          FopArgs fo) {
    lz_orth_body<NCH, NCHP>(Ppart, nt, npad, V, ldv, k, wbuf, hred, hpart_out, pld, hsum_out, ctl, alphas, betas, Apart,
                            napart, first, arrow, keep, fo);
}

// second pass applied and the step closed:
//   h2 = sum of partial dots;  beta^2 = |w'|^2 - |h2|^2  (= |w' - V h2|^2 exactly, V being
//   orthonormal; h2 is the rounding-level second-pass correction, so nothing cancels
//   unless w' itself is numerically inside span(V), where beta <= tol ends the run anyway);
//   alpha_k = h1[k] + h2[k] - carry;  V[:,k+1] = (w' - V h2)/beta, or stop when beta <= tol.
// `g` = index of the 64-row group handled by this workgroup.
// Same latency discipline as k_lz_orth: every global load (partial dots, the wave's basis
// columns, w', the scalars) is issued before the stop flag is tested; 16 column sums are reduced
// together by the fold network.  NCH = 16-column chunks per wave (k + 1 <= 64 NCH).
// s_h needs 4*16*NCH + 1 doubles, s_d NWAVE*64.
template <int NCH>
__device__ __forceinline__ void lz_finish_body(const double* __restrict__ wbuf, double* __restrict__ V, int ldv, int k,
                                               const double* __restrict__ hpart_in, int pld,
                                               const double* __restrict__ h1, double* __restrict__ alphas,
                                               double* __restrict__ betas, LanczosCtl* __restrict__ ctl,
                                               double tol, int use_carry, int g,
                                               double* __restrict__ s_h, double* __restrict__ s_d,
                                               double* __restrict__ s_beta, double* __restrict__ hred,
                                               int nprod /* workgroups that wrote a record (the rest are zero) */) {
    constexpr int NC = 16 * NCH;
    __shared__ double s_p[NWAVE * 4 * 64 * NCH];         // per-wave group sums of the producers' records, [wave][group][column]
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int kk = k + 1;
    const int i = g * LZ_ROWS + lane;
    PX_TL(0, k + 1, 0);
    // ---- all loads
    const int stop = ctl->stop;
    // partial dots: records of the producers wv, wv + 4, ... (16 of them cover pld = 64), lane <-> column
    double hp[16][NCH];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const long long rec = (long long)min(wv + NWAVE * u, pld - 1) * KLD;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch)
            hp[u][ch] = ((ch == 0 || 64 * ch < kk) && (wv + NWAVE * u < nprod || pld > WAVE))          // (uniform tests: chunks beyond k and
                            ? hpart_in[rec + 64 * ch + lane] : 0.0;                                        // records nobody wrote are not loaded)
    }
    // the wave's basis columns j = wv + 4c: groups of four, groups beyond column k are not loaded
    double vr[NC];
#pragma unroll
    for (int c0 = 0; c0 < NC; c0 += 4) {
        if (wv + 4 * c0 <= k) {
#pragma unroll
            for (int c = c0; c < c0 + 4; ++c) vr[c] = V[(long long)min(wv + 4 * c, k) * ldv + i];
        } else {
#pragma unroll
            for (int c = c0; c < c0 + 4; ++c) vr[c] = 0.0;
        }
    }
    double hn = hpart_in[(long long)pld * KLD + min(lane, pld - 1)];
    const double w0 = wbuf[i];
    const double h1k = h1[k];
    const double carry = use_carry ? ctl->carry : 0.0;
    if (stop) return;
    PX_TL(0, k + 1, 1);
    PX_TL_DRAIN();
    PX_TL(0, k + 1, 2);
    // ---- h2 = sums of the partial dots, |w'|^2
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        double pr[16], q4[4];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            pr[u] = hp[u][ch];                                        // records >= the producer count are zero
            if (64 * ch < kk)
                for (int q = wv + NWAVE * u + WAVE; q < pld; q += WAVE) pr[u] += hpart_in[(long long)q * KLD + 64 * ch + lane];   // n > 4096
        }
        tree_in_wave(pr, q4);
#pragma unroll
        for (int a = 0; a < 4; ++a) s_p[(wv * 4 + a) * (64 * NCH) + 64 * ch + lane] = q4[a];
    }
    if (wv == 0) {
        if (lane >= pld) hn = 0.0;
        else for (int q = lane + WAVE; q < pld; q += WAVE) hn += hpart_in[(long long)pld * KLD + q];
        hn = wave_sum(hn);
        if (lane == 0) s_h[4 * NC] = hn;
    }
    __syncthreads();
    if ((int)threadIdx.x < 64 * NCH) {
        const int j = threadIdx.x;
        const double hj = tree_across(s_p, 64 * NCH, j);
        s_h[j] = (j < kk) ? hj : 0.0;
    }
    __syncthreads();
    PX_TL(0, k + 1, 3);
    // ---- beta, v_{k+1} = (w' - V h2) / beta
    double hh = 0.0;
    for (int j = lane; j < kk; j += WAVE) hh += s_h[j] * s_h[j];
    hh = wave_sum(hh);                                   // every wave: same value, same order
    const double beta = sqrt(fmax(s_h[4 * NC] - hh, 0.0));
    // V h2 row sums: s_h is zero from column kk on (and skipped column groups hold zeros), so the products need no
    // test -- a uniform branch around every term made the LDS reads of s_h wait for one another (1.3 us of the closing
    // workgroups' time in profiles/r04_step_timeline.md)
    double hq[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) hq[c] = s_h[wv + 4 * c];
    double d0 = 0.0, d1 = 0.0;
#pragma unroll
    for (int c = 0; c < NC; c += 2) {
        d0 += vr[c] * hq[c];
        d1 += vr[c + 1] * hq[c + 1];
    }
    s_d[wv * LZ_ROWS + lane] = d0 + d1;
    // the reduced h2 for the prediction of the next step (k_lz_orth reads k values instead of
    // re-reducing the partials)
    if (g == 0 && (int)threadIdx.x < kk) hred[threadIdx.x] = s_h[threadIdx.x];
    if (g == 0 && threadIdx.x == 0) {
        alphas[k] = h1k + s_h[k] - carry;
        betas[k] = beta;
        ctl->carry = s_h[k];
        // every workgroup computes the same beta; the flag is only READ by later
        // launches (stream order), so a plain store by one thread is enough
        if (beta <= tol) { ctl->kstop = k + 1; ctl->stop = 1; }
    }
    if (beta <= tol) return;
    __syncthreads();
    PX_TL(0, k + 1, 4);
    if (wv == 0) {
        const double wi = w0 - ((s_d[lane] + s_d[LZ_ROWS + lane]) + (s_d[2 * LZ_ROWS + lane] + s_d[3 * LZ_ROWS + lane]));
        V[(long long)(k + 1) * ldv + i] = wi / beta;          // rows >= n stay zero
    }
    PX_TL(0, k + 1, 5);
    (void)s_beta;
}

// measured pass alone: partial dots V[:, 0..k]' w and |w|^2 into a partial-dot buffer (the records lz_finish_body reads),
// control block cleared.  Used to ORTHOGONALISE A FRESH VECTOR against a set of basis columns (k_lz_measure followed by
// k_lz_finish gives V[:, k+1] = (w - V V'w) / |.|): the start of the certificate run of a Lanczos-served full_eig!
// (Solver::lanczos_certificate).  k = -1: no columns, only |w|^2.
template <int NCH>
__global__ void __launch_bounds__(TPB)
k_lz_measure(const double* __restrict__ wbuf, const double* __restrict__ V, int ldv, int k,
             double* __restrict__ hpart_out, int pld, LanczosCtl* __restrict__ ctl) {
    constexpr int NC = 16 * NCH;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int i = blockIdx.x * LZ_ROWS + lane;
    const double wp = wbuf[i];
    double vr[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) vr[c] = (wv + 4 * c <= k) ? V[(long long)(wv + 4 * c) * ldv + i] : 0.0;
    if (blockIdx.x == 0 && threadIdx.x == 0) { ctl->stop = 0; ctl->kstop = 0; ctl->carry = 0.0; }
    if (wv == 0) {
        const double r = wave_sum(wp * wp);
        if (lane == 0) hpart_out[(long long)pld * KLD + blockIdx.x] = r;
    }
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        double t[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) t[c] = vr[16 * ch + c] * wp;
        const double hs = fold16_all(t, lane);
        const int jc = wv + 4 * (16 * ch + lane);
        if (lane < 16 && jc <= k) hpart_out[(long long)blockIdx.x * KLD + jc] = hs;
    }
}

template <int NCH>
__global__ void __launch_bounds__(TPB)
k_lz_finish(const double* __restrict__ wbuf, int n, double* __restrict__ V, int ldv, int k,
            const double* __restrict__ hpart_in, int pld, const double* __restrict__ h1,
            double* __restrict__ alphas, double* __restrict__ betas, LanczosCtl* __restrict__ ctl, double tol,
            int use_carry, double* __restrict__ hred) {
    __shared__ double s_h[64 * NCH + 1];
    __shared__ double s_d[NWAVE * LZ_ROWS];
    __shared__ double s_beta;
    lz_finish_body<NCH>(wbuf, V, ldv, k, hpart_in, pld, h1, alphas, betas, ctl, tol, use_carry, blockIdx.x, s_h, s_d, &s_beta, hred,
                        (int)gridDim.x);
}

// The step-closing work of step k and the mat-vec of step k+1 in ONE launch.
// v_{k+1} = (w' - V h2)/beta is only known after the closing reductions, but
//     A (w'/beta) = A v_{k+1} + [ V (T h2) + beta v_{k+1} h2[k] ] / beta
// and the bracket lies in span(V, v_{k+1}) and is known: k_lz_orth subtracts it with the
// recurrence terms; so the mat-vec runs on w' (ready before the closing work), the 1/beta
// is applied by k_lz_orth, and alpha_{k+1} gets the exact correction -h2[k] (`carry`).
// Workgroups [0, nt) close step k, workgroups [nt, nt + ntile) are mat-vec tiles.
template <int NCH>
__global__ void __launch_bounds__(TPB)
k_symv_finish(const double* __restrict__ xp, int n, int nt, int npad, double* __restrict__ Ppart,
              const double* __restrict__ wbuf, double* __restrict__ V, int ldv, int k,
              const double* __restrict__ hpart_in, int pld, const double* __restrict__ h1,
              double* __restrict__ alphas, double* __restrict__ betas, LanczosCtl* __restrict__ ctl, double tol,
              int use_carry, double* __restrict__ Apart, double* __restrict__ hred) {
    __shared__ double s_a[2 * NWAVE * TILE];                            // s_h   | s_row (double-buffered)
    __shared__ double s_b[NWAVE * LZ_ROWS];                             // s_d   | s_col (double-buffered)
    __shared__ double s_beta;
    if ((int)blockIdx.x < nt)
        lz_finish_body<NCH>(wbuf, V, ldv, k, hpart_in, pld, h1, alphas, betas, ctl, tol, use_carry, blockIdx.x,
                            s_a, s_b, &s_beta, hred, nt);
    else if (ctl->stop) return;
    else
        symv_tiles(xp, n, npad, nt * (nt + 1) / 2, wbuf, Ppart, (int)blockIdx.x - nt, (int)gridDim.x - nt, s_a, s_b, Apart);
}

// ---------------------------------------------------------------------------
// BATCHED Lanczos step over several PSD blocks of EQUAL side (the reference projects the blocks of a model one
// after the other, prox_operators.jl:40-61; north_star: "a batched symmetric mat-vec for the Lanczos
// recurrence").  grid.z = block: the workgroups of block z run exactly the single-block bodies above on that
// block's buffers, so results, mat-vec counts and restart counts per block are those of the one-block-at-a-time
// path; what changes is that ONE launch advances every block's (latency-bound) recurrence, instead of one
// stream + host thread per block contending for the hardware queues.  Blocks restart at different basis sizes
// (keep depends on the converged count), so the step index is PER BLOCK and a block that has finished its
// cycle idles (mode 0) until the others have.  Packed-triangle operator, krylovdim <= 63 (NCH = 1).
// ---------------------------------------------------------------------------
constexpr int LZB_MAX = 8;
struct LzBlk {
    const double* xp;            // packed block of the iterate
    double* Ppart; double* wbuf; double* V;
    double* hpart1; double* hpart2; double* hsum; double* alphas; double* betas;
    LanczosCtl* ctl;
    double* Apart; double* hred;
    const double* arrow; const double* resid;
    int k;                       // Lanczos step whose mat-vec this launch runs (basis column k)
    int keep;                    // first step of the block's current cycle (0, or the restart's keep)
    int mode;                    // k_lzb_mv: 0 idle | 1 first mat-vec of a cycle (on v_k) | 2 close step k-1 + mat-vec of
                                 // step k on w' | 3 close step k-1 only (end of the cycle).  k_lzb_orth: != 0 = active
    int pad;
};
struct LzBatch {
    LzBlk b[LZB_MAX];
    int n, nt, npad, pld, napart, nb;
    double tol;
};
__global__ void __launch_bounds__(TPB)
k_lzb_begin(LzBatch B) {
    const LzBlk& b = B.b[blockIdx.z];
    if (b.mode == 0) return;
    const int i = blockIdx.x * TPB + threadIdx.x;
    if (i < B.npad) b.V[i] = b.resid[i];
    if (i == 0) { b.ctl->stop = 0; b.ctl->kstop = 0; b.ctl->carry = 0.0; }
}
// workgroups [0, nt): closing work; [nt, nt + ntile): mat-vec tiles (as k_symv_finish / k_symv_packed)
// rec_out (pinned host memory) / rec_doubles: at the END of a block's cycle (mode 3) its closing workgroup 0 also copies the
// block's record [alphas | betas | ctl] out -- the separate gather launch of round 3 is gone (one launch less per cycle)
__global__ void __launch_bounds__(TPB)
k_lzb_mv(LzBatch B, double* __restrict__ rec_out, int rec_doubles) {
    __shared__ double s_a[2 * NWAVE * TILE];
    __shared__ double s_b[NWAVE * LZ_ROWS];
    __shared__ double s_beta;
    const LzBlk& b = B.b[blockIdx.z];
    const int mode = b.mode;
    if (mode == 0) return;
    const int nt = B.nt;
    if ((int)blockIdx.x < nt) {
        if (mode == 1) return;
        const int kc = b.k - 1;
        lz_finish_body<1>(b.wbuf, b.V, B.npad, kc, (kc & 1) ? b.hpart2 : b.hpart1, B.pld, b.hsum, b.alphas, b.betas, b.ctl,
                          B.tol, kc > b.keep ? 1 : 0, blockIdx.x, s_a, s_b, &s_beta, b.hred, nt);
        if (mode == 3 && blockIdx.x == 0 && rec_out != nullptr) {
            __syncthreads();                                 // thread 0's alphas[kc] / betas[kc] / ctl stores are visible to the workgroup
            const double* src = b.alphas;                    // the record starts at the alphas
            for (int t = threadIdx.x; t < rec_doubles; t += TPB) rec_out[(long long)blockIdx.z * rec_doubles + t] = src[t];
        }
    } else {
        if (mode == 3 || b.ctl->stop) return;
        const double* v = (mode == 1) ? b.V + (long long)b.k * B.npad : b.wbuf;
        symv_tiles(b.xp, B.n, B.npad, nt * (nt + 1) / 2, v, b.Ppart, (int)blockIdx.x - nt, (int)gridDim.x - nt, s_a, s_b,
                   b.Apart);
    }
}
__global__ void __launch_bounds__(TPB)
k_lzb_orth(LzBatch B) {
    const LzBlk& b = B.b[blockIdx.z];
    if (b.mode == 0) return;
    FopArgs fo{};
    lz_orth_body<1, 0>(b.Ppart, B.nt, B.npad, b.V, B.npad, b.k, b.wbuf, b.hred, (b.k & 1) ? b.hpart2 : b.hpart1, B.pld,
                       b.hsum, b.ctl, b.alphas, b.betas, b.Apart, B.napart, b.k == b.keep ? 1 : 0, b.arrow, b.keep, fo);
}

// ---------------------------------------------------------------------------
// Operator-form mat-vec.  On the support path the matrix handed to the projection is
//     X = x_prev - tau (M'y + c)   with  x_prev = Vp Lam Vp'  (the previous projection, rank rp)
// and the update supported on S (Max-Cut n = 4000: 32 000 of 8.0e6 entries).  So
//     A v = Vp (Lam (Vp' v)) + E v,      E = smat of the update, sparse symmetric,
// costs 16 n rp + O(|S|) bytes instead of the 8 N bytes of the packed triangle (64 MB -> 2 MB),
// and needs no tile kernel: the row-local pieces (Vp'v partials, E v, v'Ev partials) are produced
// by the workgroups below, the global sums are taken by k_lz_orth<., NCHP>.  The dense iterate
// is still written by the reconstruction (residuals, M x), only the Lanczos operator changes.
// E is kept in ELL form per PSD block: entry k of row i at [k*npad + i]: column, index into the
// support-value array (-1 = padding); off-diagonal values carry the svec sqrt(2).
// ---------------------------------------------------------------------------
// rows of E wider than the ELL part (hub vertices): their remaining entries, handled by the whole
// workgroup of the row's block with a fixed-order reduction
struct EllOverflow {
    const int* wr_ptr;       // [nt + 1] wide rows of each 64-row block
    const int* wr_row;       // local row (0..63)
    const int* wr_lo;        // range into ov_col / ov_sidx
    const int* wr_hi;
    const int* ov_col;
    const int* ov_sidx;
};
template <int NCHP>
__device__ __forceinline__ void fop_body(const double* __restrict__ v, const double* __restrict__ Vp, int ldv, int rp,
                                         const int* __restrict__ ell_col, const int* __restrict__ ell_sidx, int ell_w,
                                         int npad, const double* __restrict__ esv, double* __restrict__ tpart, int pld,
                                         double* __restrict__ ebuf, double* __restrict__ apart, int g,
                                         double* __restrict__ s_e /* NWAVE*64 */, const LanczosCtl* __restrict__ ctl,
                                         const EllOverflow& ov, int tlk = -1 /* timeline builds: step index */) {
    constexpr int NCP = 16 * NCHP;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int i = g * LZ_ROWS + lane;
    PX_TL(0, tlk, 0);
    const int stop = (ctl != nullptr) ? ctl->stop : 0;   // tested once the first loads are in flight
    const double vi = v[i];
    double vrp[NCP];
#pragma unroll
    for (int c = 0; c < NCP; ++c) vrp[c] = Vp[(long long)min(wv + 4 * c, max(rp - 1, 0)) * ldv + i];
    // E v: the waves split the ELL entries of the row (first 4 entries per wave prefetched)
    int col0[4], sx0[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int k = min(wv + u * NWAVE, ell_w - 1);
        col0[u] = ell_col[(long long)k * npad + i];
        sx0[u] = (wv + u * NWAVE < ell_w) ? ell_sidx[(long long)k * npad + i] : -1;
    }
    if (stop) return;
    PX_TL(0, tlk, 1);
    double e = 0.0;
    for (int k0 = wv; k0 < ell_w; k0 += 4 * NWAVE) {
        int col[4], sx[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (k0 == wv) { col[u] = col0[u]; sx[u] = sx0[u]; continue; }
            const int k = min(k0 + u * NWAVE, ell_w - 1);
            col[u] = ell_col[(long long)k * npad + i];
            sx[u] = (k0 + u * NWAVE < ell_w) ? ell_sidx[(long long)k * npad + i] : -1;
        }
        double ev[4], xv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { ev[u] = esv[max(sx[u], 0)]; xv[u] = v[col[u]]; }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (sx[u] >= 0) e += ((col[u] == i) ? ev[u] : ev[u] * INV_SQRT2) * xv[u];
    }
    s_e[wv * LZ_ROWS + lane] = e;
    PX_TL(0, tlk, 2);
    if (ov.wr_ptr != nullptr) {
        const int q0 = ov.wr_ptr[g], q1 = ov.wr_ptr[g + 1];
        if (q1 > q0) {                                   // uniform over the workgroup; rare
            __shared__ double s_w[NWAVE];
            __syncthreads();
            for (int q = q0; q < q1; ++q) {
                const int rloc = ov.wr_row[q], grow = g * LZ_ROWS + rloc;
                double a = 0.0;
                for (int k = ov.wr_lo[q] + (int)threadIdx.x; k < ov.wr_hi[q]; k += TPB) {
                    const int col = ov.ov_col[k];
                    const double evk = esv[ov.ov_sidx[k]];
                    a += ((col == grow) ? evk : evk * INV_SQRT2) * v[col];
                }
                a = wave_sum(a);
                if (lane == 0) s_w[wv] = a;
                __syncthreads();
                if (threadIdx.x == 0) s_e[rloc] += (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
                __syncthreads();
            }
        }
    }
    // Vp' v partials of this workgroup's rows
#pragma unroll
    for (int ch = 0; ch < NCHP; ++ch) {
        double t[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) t[c] = vrp[16 * ch + c] * vi;
        const double ts = fold16_all(t, lane);
        const int cc = wv + 4 * (16 * ch + lane);
        if (lane < 16 && cc < rp) tpart[(long long)g * RLD + cc] = ts;
    }
    __syncthreads();
    PX_TL(0, tlk, 3);
    if (wv == 0) {
        const double ei = (s_e[lane] + s_e[LZ_ROWS + lane]) + (s_e[2 * LZ_ROWS + lane] + s_e[3 * LZ_ROWS + lane]);
        ebuf[i] = ei;
        const double a = wave_sum(vi * ei);
        if (lane == 0) apart[g] = a;
    }
    PX_TL(0, tlk, 4);
}
// first mat-vec of a cycle (on the normalised v_k); grid = nt
template <int NCHP>
__global__ void __launch_bounds__(TPB)
k_fop(const double* __restrict__ v, const double* __restrict__ Vp, int ldv, int rp,
      const int* __restrict__ ell_col, const int* __restrict__ ell_sidx, int ell_w, int npad,
      const double* __restrict__ esv, double* __restrict__ tpart, int pld, double* __restrict__ ebuf,
      double* __restrict__ apart, const LanczosCtl* __restrict__ ctl, EllOverflow ov) {
    __shared__ double s_e[NWAVE * LZ_ROWS];
    fop_body<NCHP>(v, Vp, ldv, rp, ell_col, ell_sidx, ell_w, npad, esv, tpart, pld, ebuf, apart, blockIdx.x, s_e, ctl, ov);
}
// closing work of step k (workgroups [0, nt)) + operator rows of step k+1 on w' ([nt, 2 nt))
template <int NCHP, int NCH>
__global__ void __launch_bounds__(TPB)
k_fop_finish(const double* __restrict__ wbuf, double* __restrict__ V, int ldv, int k,
             const double* __restrict__ hpart_in, int pld, const double* __restrict__ h1,
             double* __restrict__ alphas, double* __restrict__ betas, LanczosCtl* __restrict__ ctl, double tol,
             int use_carry, int nt, const double* __restrict__ Vp, int rp,
             const int* __restrict__ ell_col, const int* __restrict__ ell_sidx, int ell_w, int npad,
             const double* __restrict__ esv, double* __restrict__ tpart, double* __restrict__ ebuf,
             double* __restrict__ apart, double* __restrict__ hred, EllOverflow ov) {
    __shared__ double s_a[2 * NWAVE * TILE];
    __shared__ double s_b[NWAVE * LZ_ROWS];
    __shared__ double s_beta;
    if ((int)blockIdx.x < nt) {
        lz_finish_body<NCH>(wbuf, V, ldv, k, hpart_in, pld, h1, alphas, betas, ctl, tol, use_carry, blockIdx.x,
                            s_a, s_b, &s_beta, hred, nt);
    } else {
        fop_body<NCHP>(wbuf, Vp, ldv, rp, ell_col, ell_sidx, ell_w, npad, esv, tpart, pld, ebuf, apart,
                       (int)blockIdx.x - nt, s_b, ctl, ov, k + 1);
    }
}

// out[:, c] = sum_j V[:, j] U[j, c]  (basis rotation at a thick restart and the final Ritz
// vectors B*v, KrylovKit eigsolve); optional extra column copy.  A small GEMM: each
// workgroup stages its 64-row tile of V (64 x K) and U (K x ncols) in LDS once, thread
// (row, c mod 4) then produces the outputs (row, c), c = c mod 4, +4, ...
__global__ void __launch_bounds__(TPB)
k_lz_rotate(const double* __restrict__ V, int ldv, int n, int K, const double* __restrict__ U, int ldu,
            int ncols, double* __restrict__ out, int ldo, int copy_src, int copy_dst) {
    extern __shared__ double s_mem[];
    double* s_U = s_mem;                          // [c][j], K x ncols
    double* s_V = s_mem + (size_t)K * ncols;      // [j][r], K x 64 (+1 pad per column)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int i = blockIdx.x * LZ_ROWS + lane;    // < npad (ldv = npad), padding rows are zero
    for (int t = threadIdx.x; t < K * ncols; t += TPB) {
        const int j = t % K, c = t / K;
        s_U[t] = U[(long long)c * ldu + j];
    }
    for (int j0 = wv; j0 < K; j0 += 16 * NWAVE) {       // batches of 16 loads in flight per wave (see k_lz_rotate_mfma)
        double tv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) tv[u] = V[(long long)min(j0 + NWAVE * u, K - 1) * ldv + i];
#pragma unroll
        for (int u = 0; u < 16; ++u) if (j0 + NWAVE * u < K) s_V[(j0 + NWAVE * u) * (LZ_ROWS + 1) + lane] = tv[u];
    }
    __syncthreads();
    for (int c = wv; c < ncols; c += NWAVE) {
        const double* u = s_U + (size_t)c * K;
        double a0 = 0.0, a1 = 0.0;
        int j = 0;
        for (; j + 1 < K; j += 2) {
            a0 += s_V[j * (LZ_ROWS + 1) + lane] * u[j];
            a1 += s_V[(j + 1) * (LZ_ROWS + 1) + lane] * u[j + 1];
        }
        if (j < K) a0 += s_V[j * (LZ_ROWS + 1) + lane] * u[j];
        out[(long long)c * ldo + i] = a0 + a1;
    }
    if (copy_src >= 0 && wv == 0) out[(long long)copy_dst * ldo + i] = V[(long long)copy_src * ldv + i];
}

// batched form (grid.z = block) for the restarts / final Ritz vectors of lanczos_batch: one launch instead of one per block
struct LzRot {
    const double* V; const double* U; double* out;
    int K, ncols, copy_src, copy_dst;
};
struct LzRotBatch {
    LzRot r[LZB_MAX];
    int npad, nb;
};
__global__ void __launch_bounds__(TPB)
k_lzb_rotate(LzRotBatch B) {
    extern __shared__ double s_mem[];
    const LzRot& q = B.r[blockIdx.z];
    const int K = q.K, ncols = q.ncols;
    if (ncols <= 0 && q.copy_src < 0) return;
    double* s_U = s_mem;
    double* s_V = s_mem + (size_t)K * ncols;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int i = blockIdx.x * LZ_ROWS + lane;
    const int ldv = B.npad;
    for (int t = threadIdx.x; t < K * ncols; t += TPB) s_U[t] = q.U[t];           // compact K x ncols
    for (int j0 = wv; j0 < K; j0 += 16 * NWAVE) {       // batches of 16 loads in flight per wave (see k_lz_rotate_mfma)
        double tv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) tv[u] = q.V[(long long)min(j0 + NWAVE * u, K - 1) * ldv + i];
#pragma unroll
        for (int u = 0; u < 16; ++u) if (j0 + NWAVE * u < K) s_V[(j0 + NWAVE * u) * (LZ_ROWS + 1) + lane] = tv[u];
    }
    __syncthreads();
    for (int c = wv; c < ncols; c += NWAVE) {
        const double* u = s_U + (size_t)c * K;
        double a0 = 0.0, a1 = 0.0;
        int j = 0;
        for (; j + 1 < K; j += 2) {
            a0 += s_V[j * (LZ_ROWS + 1) + lane] * u[j];
            a1 += s_V[(j + 1) * (LZ_ROWS + 1) + lane] * u[j + 1];
        }
        if (j < K) a0 += s_V[j * (LZ_ROWS + 1) + lane] * u[j];
        q.out[(long long)c * ldv + i] = a0 + a1;
    }
    if (q.copy_src >= 0 && wv == 0) q.out[(long long)q.copy_dst * ldv + i] = q.V[(long long)q.copy_src * ldv + i];
}
// every block's record [alphas | betas | ctl] into one buffer (pinned host memory: no copy engine round trip)
__global__ void __launch_bounds__(TPB)
k_lzb_gather_rec(LzBatch B, double* __restrict__ out, int rec_doubles) {
    const LzBlk& b = B.b[blockIdx.z];
    if (b.mode == 0) return;
    const double* src = b.alphas;                        // the record starts at the alphas
    for (int t = blockIdx.x * TPB + threadIdx.x; t < rec_doubles; t += gridDim.x * TPB)
        out[(long long)blockIdx.z * rec_doubles + t] = src[t];
}

// ---------------------------------------------------------------------------
// Rank-r reconstruction written directly in packed form:
//   xp[(i,j)] = s_ij * sum_k lambda_k Z[i,k] Z[j,k],  s = sqrt2 off-diagonal, 1 on it
// replaces fill! + r dense rank-1 dgemm updates of the full square +
// psd_square_to_vec (prox_operators.jl:92-106, 115-124, 17-31): one 8*N-byte
// write instead of (r+1) passes over 8*n^2 bytes.  Same tiling as the mat-vec.
// ---------------------------------------------------------------------------
constexpr int RCHUNK = 16;
// FUSE_RES: while x_new is still in registers, also accumulate over the entries NOT in
// the support set (bit mask) the two off-support terms of compute_residual!
// (residuals.jl:41-48): max |x_new - x_old| and max |x_old|  (Mty = 0 off the support,
// and x_old = the pre-projection buffer there).  Partials: respart[q*rstride + tile], q = 0,1.
template <bool FUSE_RES>
__global__ void __launch_bounds__(TPB)
k_reconstruct_packed(const double* __restrict__ Z, int ldz, const double* __restrict__ lam, int r,
                     int n, double* __restrict__ xp, const double* __restrict__ xold,
                     const unsigned* __restrict__ mask, long long mask_off, double* __restrict__ respart,
                     int rstride) {
    __shared__ double s_ZI[RCHUNK][TILE];      // [k][row]
    __shared__ double s_ZJ[RCHUNK][TILE];      // [k][col], pre-multiplied by lambda
    __shared__ double s_red[NWAVE];
    const int nt_ = (n + TILE - 1) / TILE;
    const int tile = xcd_tile(blockIdx.x, nt_ * (nt_ + 1) / 2);   // XCD-aware order (see k_symv_packed)
    if (tile >= nt_ * (nt_ + 1) / 2) return;
    int I, J;
    tile_coords(tile, I, J);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int gi = I * TILE + lane;
    // the old iterate and the support mask of this thread's 16 entries are requested first, so
    // that the 8 N bytes of x_old stream in while the factors are staged and multiplied
    double xo[CPW];
    unsigned mk[CPW];
    if (FUSE_RES) {
#pragma unroll
        for (int k = 0; k < CPW; ++k) {
            const int gjc = min(J * TILE + w * CPW + k, n - 1);
            const long long idxc = (long long)gjc * (gjc + 1) / 2 + min(gi, gjc);      // always a valid entry
            xo[k] = xold[idxc];
            mk[k] = mask[(mask_off + idxc) >> 5];
        }
    }
    double acc[CPW];
#pragma unroll
    for (int k = 0; k < CPW; ++k) acc[k] = 0.0;
    for (int k0 = 0; k0 < r; k0 += RCHUNK) {
        const int kc = min(RCHUNK, r - k0);
        __syncthreads();
        for (int t = threadIdx.x; t < RCHUNK * TILE; t += TPB) {
            const int kk = t / TILE, rr = t % TILE;
            double zi = 0.0, zj = 0.0;
            if (kk < kc) {
                const int gi2 = I * TILE + rr, gj2 = J * TILE + rr;
                if (gi2 < n) zi = Z[(long long)(k0 + kk) * ldz + gi2];
                if (gj2 < n) zj = Z[(long long)(k0 + kk) * ldz + gj2] * lam[k0 + kk];
            }
            s_ZI[kk][rr] = zi;
            s_ZJ[kk][rr] = zj;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < RCHUNK; ++kk) {
            const double zi = s_ZI[kk][lane];
#pragma unroll
            for (int k = 0; k < CPW; ++k) acc[k] += zi * s_ZJ[kk][w * CPW + k];
        }
    }
    double m0 = 0.0, m1 = 0.0;
#pragma unroll
    for (int k = 0; k < CPW; ++k) {
        const int gj = J * TILE + w * CPW + k;
        if (gj < n && gi <= gj) {
            const double sc = (gi == gj) ? 1.0 : SQRT2;
            const long long idx = (long long)gj * (gj + 1) / 2 + gi;
            const double xn = sc * acc[k];
            xp[idx] = xn;
            if (FUSE_RES) {
                const long long gidx = mask_off + idx;
                const bool on = (mk[k] >> (gidx & 31)) & 1u;
                if (!on) {
                    m0 = fmax(m0, fabs(xn - xo[k]));
                    m1 = fmax(m1, fabs(xo[k]));
                }
            }
        }
    }
    if (FUSE_RES) {
        const double r0 = block_max(m0, s_red);
        const double r1 = block_max(m1, s_red);
        if (threadIdx.x == 0) { respart[blockIdx.x] = r0; respart[rstride + blockIdx.x] = r1; }
    }
}

// ---------------------------------------------------------------------------
// The same rank-r reconstruction on the fp64 matrix cores: a SYRK on the packed triangle,
//   xp[(i,j)] = s_ij * sum_k Z[i,k] lambda_k Z[j,k],
// tiled for v_mfma_f64_16x16x4_f64 (BASELINE north_star: "MFMA-tiled V Lam+ V' rank-r SYRK").
// One workgroup per 64x64 tile (I <= J) of the block triangle, XCD-aware tile order as above; wave
// w owns the 16 tile COLUMNS [16w, 16w+16) of block J and all 64 tile rows of block I: four 16x16
// accumulators.  The product is formed transposed, D[jj][ii] = sum_k (lambda_k Z[J+jj,k]) Z[I+ii,k]:
// in the f64 MFMA's C/D layout (col = lane & 15, row = (lane >> 4) + 4 reg) a lane then holds 16
// consecutive ROWS ii of the output for its columns, so every store instruction writes 128-byte
// contiguous pieces of packed columns.  A and B fragments are ONE double per lane (A[i = l & 15]
// [k = l >> 4]); 16-k chunks of the two 64-row slabs of Z are staged in LDS (double-buffered, the
// next chunk's global loads fly during the current chunk's 16 MFMAs per wave), 5 conflict-free
// ds_read_b64 per 4 MFMAs instead of the scalar kernel's 17 LDS reads per 16 FMAs.  fp64 MFMA runs at the fp64 vector rate on
// gfx950 (78.6 TF), so this is about feeding the FMAs, not about a higher peak: the scalar kernel is
// LDS-issue-bound from r ~ 16 on, this one reaches the write roofline up to r ~ 64 and the MFMA
// roofline beyond (full_eig!: r+ ~ n/2).
// FUSE_RES as in k_reconstruct_packed.
// ---------------------------------------------------------------------------
typedef double v4f64 __attribute__((ext_vector_type(4)));
constexpr int MF_KC = 16;            // k per staged chunk (32: 2 workgroups per CU instead of 4, measured slower)
constexpr int MF_LD = TILE + 16;     // LDS row stride (doubles): k-rows 160 dwords apart -> the two k of a 32-lane group hit disjoint banks
template <bool FUSE_RES>
__global__ void __launch_bounds__(TPB)
k_reconstruct_mfma(const double* __restrict__ Z, int ldz, const double* __restrict__ lam, int r,
                   int n, double* __restrict__ xp, const double* __restrict__ xold,
                   const unsigned* __restrict__ mask, long long mask_off, double* __restrict__ respart,
                   int rstride) {
    __shared__ double sA[2][MF_KC][MF_LD];      // lambda_k Z[J*64 + row, k]   (columns of the output)
    __shared__ double sB[2][MF_KC][MF_LD];      // Z[I*64 + row, k]            (rows of the output)
    __shared__ double s_red[NWAVE];
    const int nt_ = (n + TILE - 1) / TILE;
    const int tile = xcd_tile(blockIdx.x, nt_ * (nt_ + 1) / 2);
    if (tile >= nt_ * (nt_ + 1) / 2) return;
    int I, J;
    tile_coords(tile, I, J);
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int l15 = lane & 15, l4 = lane >> 4;
    // staging role: row = lane, k = w + 4 u (u < 4) of the chunk
    const int gja = J * TILE + lane, gib = I * TILE + lane;
    const bool ja_ok = gja < n, ib_ok = gib < n;
    const double* __restrict__ zA = Z + (ja_ok ? gja : 0);
    const double* __restrict__ zB = Z + (ib_ok ? gib : 0);
    constexpr int MF_U = MF_KC / NWAVE;
    double pa[MF_U], pb[MF_U];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int u = 0; u < MF_U; ++u) {
            const int k = k0 + w + 4 * u;
            const bool kok = k < r;
            const long long ko = (long long)(kok ? k : 0) * ldz;
            const double lk = lam[kok ? k : 0];
            pa[u] = (kok && ja_ok) ? zA[ko] * lk : 0.0;
            pb[u] = (kok && ib_ok) ? zB[ko] : 0.0;
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int u = 0; u < MF_U; ++u) { sA[buf][w + 4 * u][lane] = pa[u]; sB[buf][w + 4 * u][lane] = pb[u]; }
    };
    v4f64 acc[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[b] = (v4f64){0.0, 0.0, 0.0, 0.0};
    // FUSE_RES: the old iterate and the support mask of this lane's 16 output entries are requested
    // first, so that the 8 N bytes of x_old stream in under the MFMA loop (a load per entry inside the
    // store loop made the kernel latency-bound: 73 us instead of 39 at r = 63)
    double xo[16];
    unsigned mk[16];
    if (FUSE_RES) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int gjc = min(J * TILE + w * 16 + l4 + 4 * reg, n - 1);
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int gic = min(I * TILE + b * 16 + l15, gjc);
                const long long idxc = (long long)gjc * (gjc + 1) / 2 + gic;      // always a valid entry
                xo[reg * 4 + b] = xold[idxc];
                mk[reg * 4 + b] = mask[(mask_off + idxc) >> 5];
            }
        }
    }
    fetch(0);
    stash(0);
    __syncthreads();
    int cur = 0;
    for (int k0 = 0; k0 < r; k0 += MF_KC) {
        const bool more = k0 + MF_KC < r;
        if (more) fetch(k0 + MF_KC);                       // global loads in flight during the MFMAs below
#pragma unroll
        for (int q = 0; q < MF_KC / 4; ++q) {
            const double a = sA[cur][4 * q + l4][w * 16 + l15];
            double bv[4];
#pragma unroll
            for (int b = 0; b < 4; ++b) bv[b] = sB[cur][4 * q + l4][b * 16 + l15];
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bv[b], acc[b], 0, 0, 0);
        }
        if (more) {
            stash(cur ^ 1);
            __syncthreads();
            cur ^= 1;
        }
    }
    // D[jj][ii]: jj = l4 + 4 reg (column of block J inside this wave's 16), ii = l15 (row inside 16-row block b)
    double m0 = 0.0, m1 = 0.0;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const int gj = J * TILE + w * 16 + l4 + 4 * reg;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int gi = I * TILE + b * 16 + l15;
            if (gj < n && gi <= gj) {
                const long long idx = (long long)gj * (gj + 1) / 2 + gi;
                const double xn = ((gi == gj) ? 1.0 : SQRT2) * acc[b][reg];
                if (FUSE_RES) {
                    const double xov = xo[reg * 4 + b];
                    const long long gidx = mask_off + idx;
                    const bool on = (mk[reg * 4 + b] >> (gidx & 31)) & 1u;
                    if (!on) { m0 = fmax(m0, fabs(xn - xov)); m1 = fmax(m1, fabs(xov)); }
                }
                xp[idx] = xn;
            }
        }
    }
    if (FUSE_RES) {
        const double r0 = block_max(m0, s_red);
        const double r1 = block_max(m1, s_red);
        if (threadIdx.x == 0) { respart[blockIdx.x] = r0; respart[rstride + blockIdx.x] = r1; }
    }
}

// ---------------------------------------------------------------------------
// Basis rotation out[:, c] = sum_j V[:, j] U[j, c] as fp64 MFMA tiles (round 3).  At K = 127 the scalar k_lz_rotate
// takes 55-60 us for 63-100 columns (4-5 % of the rank-63 iteration: 1.5 rotations per iteration); it is a skinny
// GEMM -- 64 rows x ncols x K per workgroup -- so: the workgroup's V tile (K x 64, zero padded to a multiple of 4) goes
// to LDS once, U is staged in groups of 48 columns, wave w owns rows [16 w, 16 w + 16) of the tile and forms
// D[c][i] = sum_j U[j][c] V[i][j] with v_mfma_f64_16x16x4_f64 (transposed like the reconstruction: a lane holds 16
// consecutive ROWS of one output column, so the stores are 128-byte pieces of columns).  U is the compact K x ncols
// column-major matrix the host uploads.  Same sums in a different order: results agree with the scalar kernel to rounding.
// ---------------------------------------------------------------------------
constexpr int RM_LDV = TILE + 16;        // LDS row stride of the V tile (as MF_LD)
constexpr int RM_CG = 16;                // U columns per workgroup (one MFMA column block per wave)
// Round 4: grid = (64-row tiles, groups of RM_CG output columns).  Round 3 gave one workgroup per row tile and let it walk
// the column groups of 48 one after the other: 63 workgroups, each a serial chain of staging rounds -- 39 us at K = 127
// (25 us once the staging loads went out in batches) for ~8 us of work.  Every output element is still the same sequence of
// MFMA steps over k = 0, 4, 8, ...: which workgroup forms it does not change its bits.
__global__ void __launch_bounds__(TPB)
k_lz_rotate_mfma(const double* __restrict__ V, int ldv, int K, const double* __restrict__ U, int ncols,
                 double* __restrict__ out, int ldo, int copy_src, int copy_dst) {
    extern __shared__ double s_mem[];
    const int Kp = (K + 3) & ~3;
    const int ldu = Kp + 2;
    double* s_V = s_mem;                           // [Kp][RM_LDV]: s_V[j][row]
    double* s_U = s_mem + (size_t)Kp * RM_LDV;     // [RM_CG][ldu]: s_U[c][j]
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int l15 = lane & 15, l4 = lane >> 4;
    const int i0 = blockIdx.x * LZ_ROWS;
    const int c0 = blockIdx.y * RM_CG;
    const int cn = min(RM_CG, ncols - c0);         // (>= 1 for every launched group; 0 only for the copy-only launch)
    // staging with the loads of a batch all in flight (one load per loop turn is a chain of memory round trips)
    if (cn > 0) {
        double tu[RM_CG / NWAVE][2];               // wave w stages columns w, w + 4, ... of the group, lanes along j (K <= 128 per pass)
        for (int jb = 0; jb < Kp; jb += 2 * WAVE) {
#pragma unroll
            for (int u = 0; u < RM_CG / NWAVE; ++u)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int c = w + NWAVE * u, j = jb + WAVE * h + lane;
                    tu[u][h] = (c < cn && j < K) ? U[(long long)(c0 + c) * K + j] : 0.0;
                }
#pragma unroll
            for (int u = 0; u < RM_CG / NWAVE; ++u)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int c = w + NWAVE * u, j = jb + WAVE * h + lane;
                    if (j < Kp) s_U[c * ldu + j] = tu[u][h];
                }
        }
        for (int j0 = w; j0 < Kp; j0 += 16 * NWAVE) {
            double tv[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int j = j0 + NWAVE * u;
                tv[u] = (j < K) ? V[(long long)j * ldv + i0 + lane] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int j = j0 + NWAVE * u;
                if (j < Kp) s_V[j * RM_LDV + lane] = tv[u];
            }
        }
        __syncthreads();
        v4f64 acc = (v4f64){0.0, 0.0, 0.0, 0.0};
        for (int q = 0; q < Kp / 4; ++q) {
            const double bv = s_V[(4 * q + l4) * RM_LDV + w * 16 + l15];
            const double av = s_U[l15 * ldu + 4 * q + l4];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
        }
        // D[c][i]: c = l4 + 4 reg (column of this group), i = w*16 + l15 (row of the tile)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int c = l4 + 4 * reg;
            if (c < cn) out[(long long)(c0 + c) * ldo + i0 + w * 16 + l15] = acc[reg];
        }
    }
    if (copy_src >= 0 && blockIdx.y == 0 && w == 0) out[(long long)copy_dst * ldo + i0 + lane] = V[(long long)copy_src * ldv + i0 + lane];
}

// ---------------------------------------------------------------------------
// Batched projection of SMALL PSD blocks (2 <= n <= 64): full_eig! (prox_operators.jl:111-126) for
// every block in ONE launch, one workgroup per block.  The reference (and the large-block path here)
// calls a dense eigensolver per block per iteration -- on multi-block SDPLIB models (truss, control,
// arch, qap: tens to hundreds of blocks of side 2..20) that is a serial chain of tiny LAPACK / rocSOLVER
// calls.  Here: the block is unpacked into LDS, diagonalised by a parallel-order cyclic Jacobi
// (n/2 disjoint rotations per round, round-robin schedule, eigenvectors accumulated; stops when
// off(A)^2 <= 1e-32 |diag|^2), and X+ = sum over lambda_k > 0 of lambda_k v_k v_k' is written back in
// packed form.  Jacobi's eigenpairs are accurate to a few ulp, as dsyevr's are.
// out: rank[b] = #{lambda > tol_psd} (current_rank), npos[b] = #{lambda > 0}.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(TPB)
k_small_psd_project(double* __restrict__ x, const long long* __restrict__ offs, const int* __restrict__ sides,
                    double tol_psd, int* __restrict__ rank_out, int* __restrict__ npos_out, int nmin, int nmax) {
    extern __shared__ __attribute__((aligned(16))) double sj_mem[];
    __shared__ double s_red[NWAVE];
    __shared__ int s_cnt[2];
    const int n = sides[blockIdx.x];
    if (n < nmin || n > nmax) return;                  // (sides outside the range belong to k_small_sign_project, small_sign.hip.hpp)
    double* __restrict__ xp = x + offs[blockIdx.x];
    const int ld = n | 1;                              // odd stride: conflict-free row and column walks
    double* A = sj_mem;                                // n x ld
    double* V = A + n * ld;                            // n x ld
    double* cs = V + n * ld;                           // c[i], s[i] of the round's pairs (2 x 32)
    int* pq = (int*)(cs + 64);                         // p[i], q[i]
    const int tid = threadIdx.x;
    for (int t = tid; t < n * n; t += TPB) {
        const int i = t % n, j = t / n;
        const int lo = min(i, j), hi = max(i, j);
        const double v = xp[(long long)hi * (hi + 1) / 2 + lo];
        A[i * ld + j] = (i == j) ? v : v * INV_SQRT2;
        V[i * ld + j] = (i == j) ? 1.0 : 0.0;
    }
    __syncthreads();
    const int m = n + (n & 1);                         // players of the round-robin (one dummy when n is odd)
    const int np = m / 2;
    for (int sweep = 0; sweep < 40; ++sweep) {
        // convergence: off-diagonal mass against the diagonal
        double off = 0.0, dg = 0.0;
        for (int t = tid; t < n * n; t += TPB) {
            const int i = t % n, j = t / n;
            const double v = A[i * ld + j];
            if (i == j) dg += v * v; else off += v * v;
        }
        const double offs_ = block_sum(off, s_red);
        if (tid == 0) cs[0] = offs_;
        __syncthreads();
        const double dgs = block_sum(dg, s_red);
        if (tid == 0) cs[1] = dgs;
        __syncthreads();
        if (cs[0] <= 1e-32 * cs[1] || cs[0] == 0.0) break;
        __syncthreads();
        for (int rd = 0; rd < m - 1; ++rd) {
            // the round's disjoint pairs and their rotations
            if (tid < np) {
                int p, q;
                if (tid == 0) { p = m - 1; q = rd; }
                else { p = (rd + tid) % (m - 1); q = (rd - tid + (m - 1)) % (m - 1); }
                if (p > q) { const int t = p; p = q; q = t; }
                double c = 1.0, s = 0.0;
                if (q < n) {
                    const double apq = A[p * ld + q];
                    if (apq != 0.0) {
                        const double tau = (A[q * ld + q] - A[p * ld + p]) / (2.0 * apq);
                        const double t = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
                        c = 1.0 / sqrt(1.0 + t * t);
                        s = t * c;
                    }
                } else { q = -1; }
                pq[tid] = p; pq[32 + tid] = q; cs[tid] = c; cs[32 + tid] = s;
            }
            __syncthreads();
            // rows: A <- J' A
            for (int t = tid; t < np * n; t += TPB) {
                const int i = t / n, k = t - i * n;
                const int p = pq[i], q = pq[32 + i];
                if (q >= 0) {
                    const double c = cs[i], s = cs[32 + i];
                    const double ap = A[p * ld + k], aq = A[q * ld + k];
                    A[p * ld + k] = c * ap - s * aq;
                    A[q * ld + k] = s * ap + c * aq;
                }
            }
            __syncthreads();
            // columns: A <- A J, V <- V J
            for (int t = tid; t < np * n; t += TPB) {
                const int i = t / n, k = t - i * n;
                const int p = pq[i], q = pq[32 + i];
                if (q >= 0) {
                    const double c = cs[i], s = cs[32 + i];
                    const double ap = A[k * ld + p], aq = A[k * ld + q];
                    A[k * ld + p] = c * ap - s * aq;
                    A[k * ld + q] = s * ap + c * aq;
                    const double vp = V[k * ld + p], vq = V[k * ld + q];
                    V[k * ld + p] = c * vp - s * vq;
                    V[k * ld + q] = s * vp + c * vq;
                }
            }
            __syncthreads();
        }
    }
    // eigenvalues on the diagonal; counts, then X+ in packed form
    if (tid == 0) { s_cnt[0] = 0; s_cnt[1] = 0; }
    __syncthreads();
    if (tid < n) {
        const double lamk = A[tid * ld + tid];
        cs[tid] = lamk > 0.0 ? lamk : 0.0;
        if (lamk > tol_psd) atomicAdd(&s_cnt[0], 1);
        if (lamk > 0.0) atomicAdd(&s_cnt[1], 1);
    }
    __syncthreads();
    const int N = n * (n + 1) / 2;
    for (int t = tid; t < N; t += TPB) {
        int j = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
        while ((j + 1) * (j + 2) / 2 <= t) ++j;
        while (j * (j + 1) / 2 > t) --j;
        const int i = t - j * (j + 1) / 2;
        double acc = 0.0;
        for (int k = 0; k < n; ++k) acc += cs[k] * V[i * ld + k] * V[j * ld + k];
        xp[t] = (i == j) ? acc : acc * SQRT2;
    }
    if (tid == 0) { rank_out[blockIdx.x] = s_cnt[0]; npos_out[blockIdx.x] = s_cnt[1]; }
}

// packed svec -> dense column-major upper triangle (only for the full-eig
// fallback, which hands the matrix to rocSOLVER): psd_vec_to_square :1-16
__global__ void __launch_bounds__(TPB)
k_unpack_upper(const double* __restrict__ xp, int n, double* __restrict__ A, int lda, double offscale) {
    int I, J;
    tile_coords(blockIdx.x, I, J);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int gi = I * TILE + lane;
#pragma unroll
    for (int k = 0; k < CPW; ++k) {
        const int gj = J * TILE + w * CPW + k;
        if (gj < n && gi <= gj) {
            double v = xp[(long long)gj * (gj + 1) / 2 + gi];
            A[(long long)gj * lda + gi] = (gi == gj) ? v : v * offscale;
        }
    }
}

// ---------------------------------------------------------------------------
// PDHG vector kernels (pdhg.jl:532-637, residuals.jl) -- fused passes.
// Floating-point contraction is OFF for the element-wise expressions below:
// Julia's broadcasts round every product and sum separately, and the solver's
// discrete decisions depend on it (at k = 1 the reference gets x = tau*c -
// tau*(0 + c) = exactly 0; an FMA leaves 1 ulp of tau*c behind, which flips the
// first 0 <= 0 linesearch acceptance and changes the whole trajectory).  These
// kernels are HBM-bound, so the extra rounding steps cost nothing.
// ---------------------------------------------------------------------------
#pragma clang fp contract(off)
// x_out = x_in - tau*(Mty + c)                                   pdhg.jl:622
__global__ void __launch_bounds__(TPB)
k_primal_update(double* __restrict__ xo, const double* __restrict__ xi, const double* __restrict__ Mty,
                const double* __restrict__ c, double tau, long long N) {
    long long i = (long long)blockIdx.x * TPB + threadIdx.x;
    const long long stride = (long long)gridDim.x * TPB;
    for (; i < N; i += stride) xo[i] = xi[i] - tau * (Mty[i] + c[i]);
}

// 1x1 PSD blocks: x = max(0,x); min_eig = x                      prox_operators.jl:43-45
__global__ void k_clamp_scalars(double* __restrict__ x, const long long* __restrict__ offs, int cnt,
                                double* __restrict__ mineig) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < cnt) { double v = fmax(0.0, x[offs[t]]); x[offs[t]] = v; mineig[t] = v; }
}

// second-order cone projection, one workgroup per cone           prox_operators.jl:138-158
__global__ void __launch_bounds__(TPB)
k_soc_project(double* __restrict__ x, const long long* __restrict__ soc_off, const int* __restrict__ soc_len) {
    __shared__ double sm[NWAVE];
    __shared__ double s_nv;
    const long long off = soc_off[blockIdx.x];
    const int len = soc_len[blockIdx.x];
    double ss = 0.0;
    for (int i = 1 + threadIdx.x; i < len; i += TPB) { double v = x[off + i]; ss += v * v; }
    double tot = block_sum(ss, sm);
    if (threadIdx.x == 0) s_nv = sqrt(tot);
    __syncthreads();
    const double nv = s_nv, s = x[off];
    __syncthreads();
    if (nv <= -s) {
        for (int i = threadIdx.x; i < len; i += TPB) x[off + i] = 0.0;
    } else if (nv <= s) {
        // inside the cone
    } else {
        const double val = 0.5 * (1.0 + s / nv);
        for (int i = 1 + threadIdx.x; i < len; i += TPB) x[off + i] *= val;
        if (threadIdx.x == 0) x[off] = val * nv;
    }
}
// soc_gap = |v| - s  per cone (residuals.jl:73-86); out[cone]
__global__ void __launch_bounds__(TPB)
k_soc_gap(const double* __restrict__ x, const long long* __restrict__ soc_off, const int* __restrict__ soc_len,
          double* __restrict__ out) {
    __shared__ double sm[NWAVE];
    const long long off = soc_off[blockIdx.x];
    const int len = soc_len[blockIdx.x];
    double ss = 0.0;
    for (int i = 1 + threadIdx.x; i < len; i += TPB) { double v = x[off + i]; ss += v * v; }
    double tot = block_sum(ss, sm);
    if (threadIdx.x == 0) out[blockIdx.x] = sqrt(tot) - x[off];
}

// Mx = M x, M in CSR; pdhg.jl:634 (the reference does a CSC scatter in pure Julia).  Three row
// classes: thread-per-row (short rows), wave-per-row (medium), and rows longer than
// `long_thresh` (gpp500-1 has one all-ones row of 125 250 entries) which the two kernels skip:
// those are cut into segments of SPMV_SEG entries, one workgroup per segment with the products
// reduced through LDS (k_spmv_csr_seg), and a fixed-order sum of the segment partials per row
// (k_spmv_seg_fin): parallel width for the long row, still deterministic.
constexpr int SPMV_SEG = 4096;
__global__ void __launch_bounds__(TPB)
k_spmv_csr_seg(const int* __restrict__ seg_lo, const int* __restrict__ seg_hi, const int* __restrict__ col,
               const double* __restrict__ val, const double* __restrict__ x, double* __restrict__ segpart) {
    __shared__ double sm[NWAVE];
    const int lo = seg_lo[blockIdx.x], hi = seg_hi[blockIdx.x];
    double a0 = 0.0, a1 = 0.0;
    int k = lo + threadIdx.x;
    for (; k + TPB < hi; k += 2 * TPB) {
        a0 += val[k] * x[col[k]];
        a1 += val[k + TPB] * x[col[k + TPB]];
    }
    if (k < hi) a0 += val[k] * x[col[k]];
    const double tot = block_sum(a0 + a1, sm);
    if (threadIdx.x == 0) segpart[blockIdx.x] = tot;
}
__global__ void __launch_bounds__(TPB)
k_spmv_seg_fin(const int* __restrict__ long_row, const int* __restrict__ long_ptr, int nlong,
               const double* __restrict__ segpart, double* __restrict__ y) {
    const int r = blockIdx.x * TPB + threadIdx.x;
    if (r >= nlong) return;
    double a = 0.0;
    for (int sg = long_ptr[r]; sg < long_ptr[r + 1]; ++sg) a += segpart[sg];
    y[long_row[r]] = a;
}
__global__ void __launch_bounds__(TPB)
k_spmv_csr_thread(const int* __restrict__ rowptr, const int* __restrict__ col, const double* __restrict__ val,
                  const double* __restrict__ x, double* __restrict__ y, int nrows, int long_thresh) {
    int r = blockIdx.x * TPB + threadIdx.x;
    if (r >= nrows) return;
    if (rowptr[r + 1] - rowptr[r] > long_thresh) return;      // segmented path
    // (four entries' loads in flight at a time -- a row of a few hundred entries walked one dependent round trip after the
    // other cost 64 us per launch on theta6; the additions keep the scalar loop's order, FP contraction is off here)
    double acc = 0.0;
    int k = rowptr[r];
    const int k1 = rowptr[r + 1];
    for (; k + 4 <= k1; k += 4) {
        const int c0 = col[k], c1 = col[k + 1], c2 = col[k + 2], c3 = col[k + 3];
        const double v0 = val[k], v1 = val[k + 1], v2 = val[k + 2], v3 = val[k + 3];
        const double x0 = x[c0], x1 = x[c1], x2 = x[c2], x3 = x[c3];
        acc += v0 * x0; acc += v1 * x1; acc += v2 * x2; acc += v3 * x3;
    }
    for (; k < k1; ++k) acc += val[k] * x[col[k]];
    y[r] = acc;
}
__global__ void __launch_bounds__(TPB)
k_spmv_csr_wave(const int* __restrict__ rowptr, const int* __restrict__ col, const double* __restrict__ val,
                const double* __restrict__ x, double* __restrict__ y, int nrows, int long_thresh) {
    const int lane = threadIdx.x & 63;
    int r = blockIdx.x * NWAVE + (threadIdx.x >> 6);
    if (r >= nrows) return;
    if (rowptr[r + 1] - rowptr[r] > long_thresh) return;      // segmented path
    double acc = 0.0;
    for (int k = rowptr[r] + lane; k < rowptr[r + 1]; k += WAVE) acc += val[k] * x[col[k]];
    acc = wave_sum(acc);
    if (lane == 0) y[r] = acc;
}

// y+ = ybar - bt*box(ybar/bt),  ybar = y + bt((1+theta)Mx - theta Mx_old)
// pdhg.jl:547-553 + box_projection! (prox_operators.jl:160-170), fused with the
// partial of |y+ - y_old|^2 (pdhg.jl:561-562; y_old == y at this point).
__global__ void __launch_bounds__(TPB)
k_dual_trial(const double* __restrict__ y, const double* __restrict__ Mx, const double* __restrict__ Mx_old,
             const double* __restrict__ bh, int p, int Q, double bt, double theta,
             double* __restrict__ yout, double* __restrict__ part, int addback) {
    __shared__ double sm[NWAVE];
    double ss = 0.0;
    for (int i = blockIdx.x * TPB + threadIdx.x; i < Q; i += gridDim.x * TPB) {
        const double yi = y[i];
        const double ybar = yi + bt * ((1.0 + theta) * Mx[i] - theta * Mx_old[i]);
        const double proj = (i < p) ? bh[i] : fmin(ybar / bt, bh[i]);
        const double yn = ybar - bt * proj;
        const double d = yn - yi;
        // linesearch! takes its norms "in place" (y_temp .-= y_old ... y_temp .+= y_old, pdhg.jl:560-575):
        // the y it keeps is fl(fl(y+ - y_old) + y_old); dual_step! (addback = 0) keeps y+ itself
        yout[i] = addback ? d + yi : yn;
        ss += d * d;
    }
    double tot = block_sum(ss, sm);
    if (threadIdx.x == 0) part[blockIdx.x] = tot;
}
// no-linesearch variant (dual_step!, pdhg.jl:584-609): ybar = y + sigma(2Mx - Mx_old)
// is the same kernel with theta = 1, bt = sigma.

// ---- long columns of M (round 6).  The transposed products below give every column to ONE thread, which walks its entries
// with two dependent memory round trips per entry (row index, then y[row]): ~250 ns per entry.  Sensor localisation has three
// columns -- the entries of the identity block, present in every anchor constraint -- with thousands of entries: ONE launch of
// k_spmvT_S_batch took 2.07 ms, 88 % of the GPU time of the reference's SENSORLOC benchmark family (profiles/r06a_kernel_stats_
// sensorloc400.md).  Columns longer than LONGCOL entries are now summed by a whole wave: the products val[k] * y[row[k]] of 64
// entries are formed in parallel (FP contraction is off here: the product is rounded on its own in the scalar loop too) and
// ADDED IN THE SCALAR LOOP'S ORDER by a serial chain over the lanes -- the same bits, ~7 ns per entry instead of ~250.
constexpr int LONGCOL = 192;         // entries from which a column goes to a wave
constexpr int LC_CAP = 64;           // long columns one workgroup can take per launch (the rest stay with their threads)
constexpr int LC_GROUP = 4 * WAVE;   // entries a wave stages per round (four per lane)
struct LongCols { int n; int key[LC_CAP]; double acc[LC_CAP]; double stage[NWAVE][LC_GROUP]; };
// One wave, one column: rounds of 256 entries.  Row indices / values are loaded two rounds ahead, the y gathers one round
// ahead (both stay in flight under the additions of the current round), the products go through the wave's LDS stage and
// are read back as BROADCAST loads, 16 at a time, for a chain of dependent additions in the scalar loop's order.
__device__ __forceinline__ double wave_col_dot_inorder(const int* __restrict__ row, const double* __restrict__ val,
                                                       const double* __restrict__ y, int k0, int k1, int lane,
                                                       double* __restrict__ stage) {
    double acc = 0.0;
    int rA[4]; double vA[4];                 // round r + 2: row, val
    double vB[4], yB[4];                     // round r + 1: val, y[row]
    double pC[4];                            // round r: products
    auto loadA = [&](int kb) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = min(kb + u * WAVE + lane, k1 - 1);
            rA[u] = row[k]; vA[u] = val[k];
        }
    };
    auto gatherB = [&]() {
#pragma unroll
        for (int u = 0; u < 4; ++u) { vB[u] = vA[u]; yB[u] = y[rA[u]]; }
    };
    auto mulC = [&]() {
#pragma unroll
        for (int u = 0; u < 4; ++u) pC[u] = vB[u] * yB[u];
    };
    loadA(k0); gatherB();
    if (k0 + LC_GROUP < k1) loadA(k0 + LC_GROUP);
    for (int kb = k0; kb < k1; kb += LC_GROUP) {
        mulC();                                                   // products of this round (its gathers were issued a round ago)
        if (kb + LC_GROUP < k1) gatherB();                        // next round's gathers
        if (kb + 2 * LC_GROUP < k1) loadA(kb + 2 * LC_GROUP);     // row / val of the round after
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int u = 0; u < 4; ++u) stage[u * WAVE + lane] = pC[u];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        const int cnt = min(LC_GROUP, k1 - kb);
        int e = 0;
        if (cnt >= 16) {
            double q[16], qn[16];
#pragma unroll
            for (int t = 0; t < 16; ++t) q[t] = stage[t];         // the same address in every lane: LDS broadcast
            for (; e + 16 <= cnt; e += 16) {
                const int en = min(e + 16, LC_GROUP - 16);        // (the next 16 are read while these are added; the last read is a repeat)
#pragma unroll
                for (int t = 0; t < 16; ++t) qn[t] = stage[en + t];
#pragma unroll
                for (int t = 0; t < 16; ++t) acc = acc + q[t];
#pragma unroll
                for (int t = 0; t < 16; ++t) q[t] = qn[t];
            }
        }
        for (; e < cnt; ++e) acc = acc + stage[e];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    return acc;
}
// the workgroup's long columns, summed by its waves; colof(s) = column of loop index s.  Must be called by every thread.
template <class ColOf>
__device__ __forceinline__ void long_cols_collect(LongCols& L, const int* __restrict__ colptr, const int* __restrict__ row,
                                                  const double* __restrict__ val, const double* __restrict__ y,
                                                  long long first, long long stride, long long count, ColOf colof) {
    if (threadIdx.x == 0) L.n = 0;
    __syncthreads();
    for (long long s = first; s < count; s += stride) {
        const int col = colof(s);
        if (colptr[col + 1] - colptr[col] > LONGCOL) {
            const int at = atomicAdd(&L.n, 1);
            if (at < LC_CAP) L.key[at] = col;
        }
    }
    __syncthreads();
    const int n = min(L.n, LC_CAP);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int q = wv; q < n; q += NWAVE) {
        const int col = L.key[q];
        const double a = wave_col_dot_inorder(row, val, y, colptr[col], colptr[col + 1], lane, L.stage[wv]);
        if (lane == 0) L.acc[q] = a;
    }
    __syncthreads();
}
__device__ __forceinline__ double col_dot(const LongCols& L, int col, const int* __restrict__ colptr, const int* __restrict__ row,
                                          const double* __restrict__ val, const double* __restrict__ y) {
    const int k0 = colptr[col], k1 = colptr[col + 1];
    if (k1 - k0 > LONGCOL) {
        const int n = min(L.n, LC_CAP);
        for (int q = 0; q < n; ++q) if (L.key[q] == col) return L.acc[q];
    }
    // (four entries' loads in flight at a time; the additions keep the scalar loop's order)
    double acc = 0.0;
    int k = k0;
    for (; k + 4 <= k1; k += 4) {
        const int r0 = row[k], r1 = row[k + 1], r2 = row[k + 2], r3 = row[k + 3];
        const double v0 = val[k], v1 = val[k + 1], v2 = val[k + 2], v3 = val[k + 3];
        const double y0 = y[r0], y1 = y[r1], y2 = y[r2], y3 = y[r3];
        acc += v0 * y0; acc += v1 * y1; acc += v2 * y2; acc += v3 * y3;
    }
    for (; k < k1; ++k) acc += val[k] * y[row[k]];
    return acc;
}

// Mty = M' y (M in CSC: one dot per column) fused with |Mty - Mty_old|^2
// pdhg.jl:556-563.  Thread per column; columns are mostly empty or short.
__global__ void __launch_bounds__(TPB)
k_spmv_csc_norm(const int* __restrict__ colptr, const int* __restrict__ row, const double* __restrict__ val,
                const double* __restrict__ y, double* __restrict__ Mty, const double* __restrict__ Mty_old,
                long long ncols, double* __restrict__ part, int addback) {
    __shared__ double sm[NWAVE];
    __shared__ LongCols lc;
    double ss = 0.0;
    long long j = (long long)blockIdx.x * TPB + threadIdx.x;
    const long long stride = (long long)gridDim.x * TPB;
    long_cols_collect(lc, colptr, row, val, y, j, stride, ncols, [](long long s) { return (int)s; });
    for (; j < ncols; j += stride) {
        const double acc = col_dot(lc, (int)j, colptr, row, val, y);
        const double o = Mty_old[j];
        const double d = acc - o;
        Mty[j] = addback ? d + o : acc;       // pdhg.jl:560,574 (a.Mty .-= a.Mty_old ... a.Mty .+= a.Mty_old)
        ss += d * d;
    }
    double tot = block_sum(ss, sm);
    if (threadIdx.x == 0) part[blockIdx.x] = tot;
}
// plain transposed product (test entry point)
__global__ void __launch_bounds__(TPB)
k_spmv_csc(const int* __restrict__ colptr, const int* __restrict__ row, const double* __restrict__ val,
           const double* __restrict__ y, double* __restrict__ out, long long ncols) {
    long long j = (long long)blockIdx.x * TPB + threadIdx.x;
    const long long stride = (long long)gridDim.x * TPB;
    for (; j < ncols; j += stride) {
        double acc = 0.0;
        for (int k = colptr[j]; k < colptr[j + 1]; ++k) acc += val[k] * y[row[k]];
        out[j] = acc;
    }
}

// compute_residual! (x part, residuals.jl:41-48) + prim_obj = c.x (residuals.jl:22)
// in ONE pass over (x, x_old, Mty, Mty_old, c).  part layout: [3][grid]:
//   0: max |(x - tau Mty) - (x_old - tau Mty_old)|   1: max |x_old - tau Mty_old|   2: sum c*x
__global__ void __launch_bounds__(TPB)
k_residual_x(const double* __restrict__ x, const double* __restrict__ xold, double xold_coef,
             const double* __restrict__ Mty, const double* __restrict__ Mty_old, const double* __restrict__ c,
             double tau, long long N, double* __restrict__ part) {
    __shared__ double sm[NWAVE];
    double m0 = 0.0, m1 = 0.0, s2 = 0.0;
    long long i = (long long)blockIdx.x * TPB + threadIdx.x;
    const long long stride = (long long)gridDim.x * TPB;
    for (; i < N; i += stride) {
        const double xi = x[i];
        const double pold = xold_coef * xold[i] - tau * Mty_old[i];
        const double pnew = xi - tau * Mty[i];
        m0 = fmax(m0, fabs(pnew - pold));
        m1 = fmax(m1, fabs(pold));
        s2 += c[i] * xi;
    }
    double r0 = block_max(m0, sm);
    double r1 = block_max(m1, sm);
    double r2 = block_sum(s2, sm);
    if (threadIdx.x == 0) {
        part[blockIdx.x] = r0;
        part[gridDim.x + blockIdx.x] = r1;
        part[2 * gridDim.x + blockIdx.x] = r2;
    }
}
// y part of compute_residual! (residuals.jl:51-58) + compute_gap! (residuals.jl:2-35)
// part layout [6][grid]: 0 max|dPy| 1 max|Py_old| 2 max|Mx-b| (eq) 3 max(Mx-h) (ineq, >=0)
//                        4 sum b*y_eq 5 sum h*y_in
__global__ void __launch_bounds__(TPB)
k_residual_y(const double* __restrict__ y, const double* __restrict__ yold,
             const double* __restrict__ Mx, const double* __restrict__ Mx_old,
             const double* __restrict__ bh, int p, int Q, double sigma, double* __restrict__ part) {
    __shared__ double sm[NWAVE];
    double m0 = 0.0, m1 = 0.0, m2 = 0.0, m3 = 0.0, s4 = 0.0, s5 = 0.0;
    for (int i = blockIdx.x * TPB + threadIdx.x; i < Q; i += gridDim.x * TPB) {
        const double yi = y[i], mx = Mx[i], rhs = bh[i];
        const double pold = yold[i] - sigma * Mx_old[i];
        const double pnew = yi - sigma * mx;
        m0 = fmax(m0, fabs(pnew - pold));
        m1 = fmax(m1, fabs(pold));
        if (i < p) { m2 = fmax(m2, fabs(mx - rhs)); s4 += rhs * yi; }
        else       { m3 = fmax(m3, mx - rhs);       s5 += rhs * yi; }
    }
    double r0 = block_max(m0, sm), r1 = block_max(m1, sm), r2 = block_max(m2, sm), r3 = block_max(m3, sm);
    double r4 = block_sum(s4, sm), r5 = block_sum(s5, sm);
    if (threadIdx.x == 0) {
        const int g = gridDim.x, b = blockIdx.x;
        part[b] = r0; part[g + b] = r1; part[2 * g + b] = r2; part[3 * g + b] = r3;
        part[4 * g + b] = r4; part[5 * g + b] = r5;
    }
}

// ---------------------------------------------------------------------------
// Support-aware variants.  S = {i : column i of M is non-empty or c_i != 0} is fixed for
// the solve; Mty and c vanish outside S, so  x .-= tau.*(Mty .+ c)  (pdhg.jl:622) only
// touches S, Mty = M'y (pdhg.jl:556) only needs the columns in S, and the residual
// (residuals.jl:41-48) splits into an off-support part (fused into the reconstruction)
// and an on-support part over |S| entries.  For Max-Cut n=4000: |S| ~ 32e3 of 8.0e6.
// ---------------------------------------------------------------------------
// xsave = x[S];  x[S] -= tau*(MtyS + cS)   (in place: x becomes the matrix to project)
__global__ void __launch_bounds__(TPB)
k_primal_update_S(double* __restrict__ x, const int* __restrict__ supp, const double* __restrict__ MtyS,
                  const double* __restrict__ cS, double tau, double* __restrict__ xsave, int ns,
                  double* __restrict__ esv) {
    const int s = blockIdx.x * TPB + threadIdx.x;
    if (s >= ns) return;
    const int i = supp[s];
    const double xi = x[i];
    xsave[s] = xi;
    const double upd = tau * (MtyS[s] + cS[s]);
    const double xn = xi - upd;
    x[i] = xn;
    // support values of E for the operator-form mat-vec: [0, ns) the update itself (x_prev held
    // in factored form), [ns, 2 ns) the whole entry (x_prev without factors, zero off the support)
    esv[s] = -upd;
    esv[ns + s] = xn;
}

struct TrialBatch {                 // up to 4 linesearch candidates per launch
    double bt[4], theta[4], tau[4], sigma[4];
    int nc;
    int plain;                      // 1: dual_step! without linesearch (pdhg.jl:584-609) -- y+ and M'y+ are kept as computed, not
                                    // through linesearch!'s in-place norm + revert
};

// candidates of y+ (pdhg.jl:547-553): grid.y = candidate; part[c][0][wg] = |y+ - y|^2 partials
__global__ void __launch_bounds__(TPB)
k_dual_trial_batch(const double* __restrict__ y, const double* __restrict__ Mx, const double* __restrict__ Mx_old,
                   const double* __restrict__ bh, int p, int Q, TrialBatch tb,
                   double* __restrict__ ycand, long long ystride, double* __restrict__ part, long long cstride,
                   const double* __restrict__ roww = nullptr) {
    __shared__ double sm[NWAVE];
    const int c = blockIdx.y;
    const double bt = tb.bt[c], theta = tb.theta[c];
    double* yout = ycand + (long long)c * ystride;
    double ss = 0.0;
    for (int i = blockIdx.x * TPB + threadIdx.x; i < Q; i += gridDim.x * TPB) {
        const double yi = y[i];
        const double ybar = yi + bt * ((1.0 + theta) * Mx[i] - theta * Mx_old[i]);
        const double proj = (i < p) ? bh[i] : fmin(ybar / bt, bh[i]);
        const double yn = ybar - bt * proj;
        const double d = yn - yi;
        yout[i] = tb.plain ? yn : d + yi;      // in-place norm + revert (pdhg.jl:561,575): fl(fl(y+ - y_old) + y_old)
        // roww: 0 for a coupling row another shard accounts for (block-sharded solve), else 1
        ss += (roww != nullptr ? roww[i] : 1.0) * (d * d);
    }
    const double tot = block_sum(ss, sm);
    if (threadIdx.x == 0) part[(long long)c * cstride + blockIdx.x] = tot;
}

// MtyS_c = (M' y_c)|S for every candidate + |MtyS_c - MtyS_old|^2 partials (pdhg.jl:556-563)
__global__ void __launch_bounds__(TPB)
k_spmvT_S_batch(const int* __restrict__ colptr, const int* __restrict__ row, const double* __restrict__ val,
                const int* __restrict__ supp, int ns, const double* __restrict__ ycand, long long ystride,
                double* __restrict__ MtyScand, long long mstride, const double* __restrict__ MtyS_old,
                double* __restrict__ part, long long cstride, int plain = 0) {
    __shared__ double sm[NWAVE];
    __shared__ LongCols lc;
    const int c = blockIdx.y;
    const double* y = ycand + (long long)c * ystride;
    double* out = MtyScand + (long long)c * mstride;
    double ss = 0.0;
    long_cols_collect(lc, colptr, row, val, y, (long long)blockIdx.x * TPB + threadIdx.x, (long long)gridDim.x * TPB, (long long)ns,
                      [supp](long long s) { return supp[s]; });
    for (int s = blockIdx.x * TPB + threadIdx.x; s < ns; s += gridDim.x * TPB) {
        const int col = supp[s];
        const double acc = col_dot(lc, col, colptr, row, val, y);
        const double o = MtyS_old[s];
        const double d = acc - o;
        out[s] = plain ? acc : d + o;          // pdhg.jl:560,574 (plain: dual_step!'s mul!(Mty, Mt, y), pdhg.jl:606)
        ss += d * d;
    }
    const double tot = block_sum(ss, sm);
    if (threadIdx.x == 0) part[(long long)c * cstride + blockIdx.x] = tot;
}

// on-support part of compute_residual! + c.x (residuals.jl:22,41-48), per candidate
// part[c][q][wg], q = 0: max |dPx|  1: max |Px_old|  2: sum c*x
__device__ __forceinline__ void
residual_xS_body(const double* __restrict__ xnew, const int* __restrict__ supp, int ns,
                 const double* __restrict__ xsave, double xold_coef,
                 const double* __restrict__ MtyScand, long long mstride, const double* __restrict__ MtyS_old,
                 const double* __restrict__ cS, const TrialBatch& tb, double* __restrict__ part, int pstride,
                 long long cstride, int gx, double* __restrict__ sm) {
    if ((int)blockIdx.x >= gx) return;
    const int c = blockIdx.y;
    const double tau = tb.tau[c];
    const double* MtyS = MtyScand + (long long)c * mstride;
    double m0 = 0.0, m1 = 0.0, s2 = 0.0;
    for (int s = blockIdx.x * TPB + threadIdx.x; s < ns; s += gx * TPB) {
        const double xi = xnew[supp[s]];
        const double pold = xold_coef * xsave[s] - tau * MtyS_old[s];
        const double pnew = xi - tau * MtyS[s];
        m0 = fmax(m0, fabs(pnew - pold));
        m1 = fmax(m1, fabs(pold));
        s2 += cS[s] * xi;
    }
    const double r0 = block_max(m0, sm), r1 = block_max(m1, sm), r2 = block_sum(s2, sm);
    if (threadIdx.x == 0) {
        double* pp = part + (long long)c * cstride;
        pp[blockIdx.x] = r0; pp[pstride + blockIdx.x] = r1; pp[2 * pstride + blockIdx.x] = r2;
    }
}

// y part of compute_residual! + compute_gap! per candidate; part[c][q][wg], q as in k_residual_y
__device__ __forceinline__ void
residual_y_body(const double* __restrict__ ycand, long long ystride, const double* __restrict__ yold,
                const double* __restrict__ Mx, const double* __restrict__ Mx_old,
                const double* __restrict__ bh, int p, int Q, const TrialBatch& tb,
                double* __restrict__ part, int pstride, long long cstride, int gx, double* __restrict__ sm,
                const double* __restrict__ roww) {
    if ((int)blockIdx.x >= gx) return;
    const int c = blockIdx.y;
    const double sigma = tb.sigma[c];
    const double* y = ycand + (long long)c * ystride;
    double m0 = 0.0, m1 = 0.0, m2 = 0.0, m3 = 0.0, s4 = 0.0, s5 = 0.0;
    for (int i = blockIdx.x * TPB + threadIdx.x; i < Q; i += gx * TPB) {
        const double yi = y[i], mx = Mx[i], rhs = bh[i];
        const double pold = yold[i] - sigma * Mx_old[i];
        const double pnew = yi - sigma * mx;
        m0 = fmax(m0, fabs(pnew - pold));
        m1 = fmax(m1, fabs(pold));
        const double wgt = roww != nullptr ? roww[i] : 1.0;
        if (i < p) { m2 = fmax(m2, fabs(mx - rhs)); s4 += wgt * (rhs * yi); }
        else       { m3 = fmax(m3, mx - rhs);       s5 += wgt * (rhs * yi); }
    }
    const double r0 = block_max(m0, sm), r1 = block_max(m1, sm), r2 = block_max(m2, sm), r3 = block_max(m3, sm);
    const double r4 = block_sum(s4, sm), r5 = block_sum(s5, sm);
    if (threadIdx.x == 0) {
        double* pp = part + (long long)c * cstride;
        const int b = blockIdx.x;
        pp[b] = r0; pp[pstride + b] = r1; pp[2 * pstride + b] = r2; pp[3 * pstride + b] = r3;
        pp[4 * pstride + b] = r4; pp[5 * pstride + b] = r5;
    }
}

// both residual parts of every candidate in ONE launch: grid (max(gs, gq), nc, 2), z = 0 the
// on-support x part, z = 1 the y part (they are independent; one link less in the per-iteration chain)
__global__ void __launch_bounds__(TPB)
k_residual_xy_batch(const double* __restrict__ xnew, const int* __restrict__ supp, int ns,
                    const double* __restrict__ xsave, double xold_coef,
                    const double* __restrict__ MtyScand, long long mstride, const double* __restrict__ MtyS_old,
                    const double* __restrict__ cS, int gs,
                    const double* __restrict__ ycand, long long ystride, const double* __restrict__ yold,
                    const double* __restrict__ Mx, const double* __restrict__ Mx_old,
                    const double* __restrict__ bh, int p, int Q, int gq,
                    TrialBatch tb, double* __restrict__ part, int pstride, long long cstride,
                    const double* __restrict__ roww = nullptr) {
    __shared__ double sm[NWAVE];
    if (blockIdx.z == 0)
        residual_xS_body(xnew, supp, ns, xsave, xold_coef, MtyScand, mstride, MtyS_old, cS, tb,
                         part + 2 * (long long)pstride, pstride, cstride, gs, sm);
    else
        residual_y_body(ycand, ystride, yold, Mx, Mx_old, bh, p, Q, tb,
                        part + 5 * (long long)pstride, pstride, cstride, gq, sm, roww);
}

// The same batch on the GENERAL path (no support set: full-vector passes).  Mty_c = M' y_c for every candidate
// + |Mty_c - Mty_old|^2 partials: per candidate the arithmetic of k_spmv_csc_norm (addback form).
__global__ void __launch_bounds__(TPB)
k_spmv_csc_norm_batch(const int* __restrict__ colptr, const int* __restrict__ row, const double* __restrict__ val,
                      const double* __restrict__ ycand, long long ystride, double* __restrict__ Mtycand, long long mstride,
                      const double* __restrict__ Mty_old, long long ncols, double* __restrict__ part, long long cstride) {
    __shared__ double sm[NWAVE];
    const int c = blockIdx.y;
    const double* y = ycand + (long long)c * ystride;
    double* out = Mtycand + (long long)c * mstride;
    __shared__ LongCols lc;
    double ss = 0.0;
    long long j = (long long)blockIdx.x * TPB + threadIdx.x;
    const long long stride = (long long)gridDim.x * TPB;
    long_cols_collect(lc, colptr, row, val, y, j, stride, ncols, [](long long s) { return (int)s; });
    for (; j < ncols; j += stride) {
        const double acc = col_dot(lc, (int)j, colptr, row, val, y);
        const double o = Mty_old[j];
        const double d = acc - o;
        out[j] = d + o;                        // pdhg.jl:560,574
        ss += d * d;
    }
    const double tot = block_sum(ss, sm);
    if (threadIdx.x == 0) part[(long long)c * cstride + blockIdx.x] = tot;
}
// x part of compute_residual! + c.x over the whole vector, per candidate (the arithmetic of k_residual_x);
// part[c][q][wg], q = 0: max |dPx|  1: max |Px_old|  2: sum c*x
__device__ __forceinline__ void
residual_x_full_body(const double* __restrict__ x, const double* __restrict__ xold, double xold_coef,
                     const double* __restrict__ Mtycand, long long mstride, const double* __restrict__ Mty_old,
                     const double* __restrict__ cv, long long N, const TrialBatch& tb, double* __restrict__ part,
                     int pstride, long long cstride, int gx, double* __restrict__ sm) {
    if ((int)blockIdx.x >= gx) return;
    const int c = blockIdx.y;
    const double tau = tb.tau[c];
    const double* Mty = Mtycand + (long long)c * mstride;
    double m0 = 0.0, m1 = 0.0, s2 = 0.0;
    for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < N; i += (long long)gx * TPB) {
        const double xi = x[i];
        const double pold = xold_coef * xold[i] - tau * Mty_old[i];
        const double pnew = xi - tau * Mty[i];
        m0 = fmax(m0, fabs(pnew - pold));
        m1 = fmax(m1, fabs(pold));
        s2 += cv[i] * xi;
    }
    const double r0 = block_max(m0, sm), r1 = block_max(m1, sm), r2 = block_sum(s2, sm);
    if (threadIdx.x == 0) {
        double* pp = part + (long long)c * cstride;
        pp[blockIdx.x] = r0; pp[pstride + blockIdx.x] = r1; pp[2 * pstride + blockIdx.x] = r2;
    }
}
// grid (max(gx, gq), nc, 2): z = 0 the x part over the whole vector, z = 1 the y part
__global__ void __launch_bounds__(TPB)
k_residual_xy_full_batch(const double* __restrict__ x, const double* __restrict__ xold, double xold_coef,
                         const double* __restrict__ Mtycand, long long mstride, const double* __restrict__ Mty_old,
                         const double* __restrict__ cv, long long N, int gx,
                         const double* __restrict__ ycand, long long ystride, const double* __restrict__ yold,
                         const double* __restrict__ Mx, const double* __restrict__ Mx_old,
                         const double* __restrict__ bh, int p, int Q, int gq,
                         TrialBatch tb, double* __restrict__ part, int pstride, long long cstride) {
    __shared__ double sm[NWAVE];
    if (blockIdx.z == 0)
        residual_x_full_body(x, xold, xold_coef, Mtycand, mstride, Mty_old, cv, N, tb,
                             part + 2 * (long long)pstride, pstride, cstride, gx, sm);
    else
        residual_y_body(ycand, ystride, yold, Mx, Mx_old, bh, p, Q, tb,
                        part + 5 * (long long)pstride, pstride, cstride, gq, sm, nullptr);
}

// non-PSD tail of x (SOC + free variables): x_new = x_trial copied to the other buffer,
// with the off-support residual terms (as in the fused reconstruction)
__global__ void __launch_bounds__(TPB)
k_tail_copy_res(const double* __restrict__ xin, double* __restrict__ xout, long long start, long long N,
                const unsigned* __restrict__ mask, double* __restrict__ respart, int rstride) {
    __shared__ double sm[NWAVE];
    double m0 = 0.0, m1 = 0.0;   // m0 stays 0: x_new == x_old here before the cone projections of the tail
    for (long long i = start + (long long)blockIdx.x * TPB + threadIdx.x; i < N; i += (long long)gridDim.x * TPB) {
        const double v = xin[i];
        xout[i] = v;
        const bool on = (mask[i >> 5] >> (i & 31)) & 1u;
        if (!on) m1 = fmax(m1, fabs(v));
    }
    const double r0 = block_max(m0, sm), r1 = block_max(m1, sm);
    if (threadIdx.x == 0) { respart[blockIdx.x] = r0; respart[rstride + blockIdx.x] = r1; }
}

#pragma clang fp contract(fast)
// final fixed-order combine of per-workgroup partials: out[q] = sum or max over
// part[q*stride .. q*stride+cnt).  One workgroup; ismax bit q selects max.
__global__ void __launch_bounds__(TPB)
k_combine(const double* __restrict__ part, int stride, int cnt, int nq, unsigned ismax, double* __restrict__ out) {
    __shared__ double sm[NWAVE];
    for (int q = 0; q < nq; ++q) {
        const bool mx = (ismax >> q) & 1u;
        double a = 0.0;
        for (int i = threadIdx.x; i < cnt; i += TPB) {
            const double v = part[(long long)q * stride + i];
            a = mx ? fmax(a, v) : a + v;
        }
        double r = mx ? block_max(a, sm) : block_sum(a, sm);
        if (threadIdx.x == 0) out[q] = r;
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------
// Dense constraint matrix (randSDP-scale models: every A_k dense, M = 64 GB at
// n = 2000, m = 4000).  M is the caller's ROW-MAJOR Q x n array, used in place:
// the sqrt(2)/2 column scaling of norm_scaling (scaling.jl:28-58) is applied to
// the vector on the way in (M x) or to the result on the way out (M' y), so the
// borrowed matrix is never modified or copied.  Both products stream M once:
// 8*Q*n bytes, HBM-bound.
// ---------------------------------------------------------------------------
// DMV_ROWS rows per workgroup in M x (x is re-used from registers), DMV_UNR column strips in
// flight per thread

// part[cs][r] = sum over the column slice cs of M[r, j] * s_j * x_j   (grid: row groups x slices)
template <int DMV_ROWS, int DMV_UNR>
__global__ void __launch_bounds__(TPB)
k_dense_mv(const double* __restrict__ M, long long ld, int Q, long long n, const double* __restrict__ x,
           const unsigned char* __restrict__ offdiag, double scale, double* __restrict__ part, int qpad) {
    __shared__ double sm[DMV_ROWS][NWAVE];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int r0 = blockIdx.x * DMV_ROWS;
    // column slice of this workgroup, in units of TPB * DMV_UNR columns
    const long long chunk = (long long)TPB * DMV_UNR;
    const long long nchunks = (n + chunk - 1) / chunk;
    const long long per = (nchunks + gridDim.y - 1) / gridDim.y;
    const long long c0 = (long long)blockIdx.y * per * chunk;
    const long long c1 = (c0 + per * chunk < n) ? c0 + per * chunk : n;
    const double* rowp[DMV_ROWS];
#pragma unroll
    for (int r = 0; r < DMV_ROWS; ++r) rowp[r] = M + (long long)min(r0 + r, Q - 1) * ld;
    double acc[DMV_ROWS];
#pragma unroll
    for (int r = 0; r < DMV_ROWS; ++r) acc[r] = 0.0;
    for (long long jb = c0; jb < c1; jb += chunk) {
        double xv[DMV_UNR], mv[DMV_ROWS][DMV_UNR];
#pragma unroll
        for (int u = 0; u < DMV_UNR; ++u) {
            const long long j = jb + (long long)u * TPB + threadIdx.x;
            const long long jc = (j < c1) ? j : c1 - 1;
            const double sc = (offdiag != nullptr && offdiag[jc]) ? scale : 1.0;
            xv[u] = (j < c1) ? x[jc] * sc : 0.0;
#pragma unroll
            for (int r = 0; r < DMV_ROWS; ++r) mv[r][u] = rowp[r][jc];
        }
#pragma unroll
        for (int r = 0; r < DMV_ROWS; ++r)
#pragma unroll
            for (int u = 0; u < DMV_UNR; ++u) acc[r] += mv[r][u] * xv[u];
    }
#pragma unroll
    for (int r = 0; r < DMV_ROWS; ++r) {
        const double w = wave_sum(acc[r]);
        if (lane == 0) sm[r][wv] = w;
    }
    __syncthreads();
    if (threadIdx.x < DMV_ROWS && r0 + (int)threadIdx.x < Q) {
        const int r = threadIdx.x;
        part[(long long)blockIdx.y * qpad + r0 + r] = (sm[r][0] + sm[r][1]) + (sm[r][2] + sm[r][3]);
    }
}
// y[r] = sum of the slices (fixed order)
__global__ void __launch_bounds__(TPB)
k_dense_mv_fin(const double* __restrict__ part, int qpad, int nslice, int Q, double* __restrict__ y) {
    const int r = blockIdx.x * TPB + threadIdx.x;
    if (r >= Q) return;
    double a = 0.0;
    for (int sidx = 0; sidx < nslice; ++sidx) a += part[(long long)sidx * qpad + r];
    y[r] = a;
}

// OUT_c[j] = s_j * sum_k M[k, j] * Y_c[k] (+ the sparse rows of [A;G]) for NC candidate vectors in ONE pass over M
// (the linesearch candidates tau, 3/4 tau, (3/4)^2 tau share the 8*Q*n bytes), plus the
// partials of |OUT_c - old|^2 (pdhg.jl:556-563) when `old` != NULL, plus `addc` (c_orig for
// the exit path's dual cone).  Thread = column; the loop over rows keeps DMT_UNR loads in flight.
// CPT columns per thread (TPB apart: a workgroup reads CPT * 2 KB contiguous bytes of every
// row), DMT_UNR rows in flight.
template <int NC, int CPT, int DMT_UNR>
__global__ void __launch_bounds__(TPB)
k_dense_mtv(const double* __restrict__ M, long long ld, int Q, long long n, const double* __restrict__ Y,
            long long ystride, const unsigned char* __restrict__ offdiag, double scale,
            double* __restrict__ OUT, long long ostride, const double* __restrict__ old,
            const double* __restrict__ addc, double* __restrict__ part, long long cstride,
            const int* __restrict__ sp_colptr, const int* __restrict__ sp_row, const double* __restrict__ sp_val,
            int addback) {
    __shared__ double sm[NWAVE];
    double ss[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) ss[c] = 0.0;
    const long long span = (long long)TPB * CPT;
    const long long nchunk = (n + span - 1) / span;
    // XCD-aware chunk order (workgroup b runs on XCD b % 8): within every sweep of gridDim.x
    // chunks XCD x walks a contiguous range, so the 128-byte lines that neighbouring 2 KB row
    // segments share (rows start at arbitrary 8-byte alignment) are found in the same L2
    // (PMC: 1.057x -> see profiles/r01_pmc_dense.md).  gridDim.x is a multiple of 8 or < 8.
    const int gx8 = (gridDim.x >= 8 && (gridDim.x & 7) == 0) ? gridDim.x >> 3 : 0;
    const long long bmine = gx8 ? (long long)(blockIdx.x & 7) * gx8 + (blockIdx.x >> 3) : blockIdx.x;
    for (long long b = bmine; b < nchunk; b += gridDim.x) {
        const double* col[CPT];
#pragma unroll
        for (int t = 0; t < CPT; ++t) {
            const long long j = b * span + (long long)t * TPB + threadIdx.x;
            col[t] = M + ((j < n) ? j : n - 1);
        }
        double acc[NC][CPT];
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int t = 0; t < CPT; ++t) acc[c][t] = 0.0;
        int k = 0;
        for (; k + DMT_UNR <= Q; k += DMT_UNR) {
            double mv[DMT_UNR][CPT];
#pragma unroll
            for (int u = 0; u < DMT_UNR; ++u)
#pragma unroll
                for (int t = 0; t < CPT; ++t) mv[u][t] = col[t][(long long)(k + u) * ld];
#pragma unroll
            for (int u = 0; u < DMT_UNR; ++u)
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    const double yv = Y[c * ystride + k + u];
#pragma unroll
                    for (int t = 0; t < CPT; ++t) acc[c][t] += mv[u][t] * yv;
                }
        }
        for (; k < Q; ++k) {
#pragma unroll
            for (int t = 0; t < CPT; ++t) {
                const double mvk = col[t][(long long)k * ld];
#pragma unroll
                for (int c = 0; c < NC; ++c) acc[c][t] += mvk * Y[c * ystride + k];
            }
        }
#pragma unroll
        for (int t = 0; t < CPT; ++t) {
            const long long j = b * span + (long long)t * TPB + threadIdx.x;
            if (j >= n) continue;
            const double sc = (offdiag != nullptr && offdiag[j]) ? scale : 1.0;
            const double o = (old != nullptr) ? old[j] : 0.0;
            const double ad = (addc != nullptr) ? addc[j] : 0.0;
            double spv[NC];                               // sparse rows of M (G), already scaled
#pragma unroll
            for (int c = 0; c < NC; ++c) spv[c] = 0.0;
            if (sp_colptr != nullptr)
                for (int q = sp_colptr[j]; q < sp_colptr[j + 1]; ++q)
#pragma unroll
                    for (int c = 0; c < NC; ++c) spv[c] += sp_val[q] * Y[c * ystride + sp_row[q]];
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const double v = acc[c][t] * sc + spv[c] + ad;
                const double d = v - o;
                OUT[c * ostride + j] = addback ? d + o : v;      // linesearch!: pdhg.jl:560,574
                ss[c] += d * d;
            }
        }
    }
    if (part != nullptr) {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const double tot = block_sum(ss[c], sm);
            if (threadIdx.x == 0) part[c * cstride + blockIdx.x] = tot;
        }
    }
}

// sum_kj (M[k,j] s_j)^2 partials: ||M||_F of the scaled matrix (pdhg.jl:121)
__global__ void __launch_bounds__(TPB)
k_dense_frob(const double* __restrict__ M, long long ld, int Q, long long n,
             const unsigned char* __restrict__ offdiag, double scale, double* __restrict__ part) {
    __shared__ double sm[NWAVE];
    double ss = 0.0;
    const long long nchunk = (n + TPB - 1) / TPB;
    for (long long b = blockIdx.x; b < nchunk; b += gridDim.x) {
        const long long j = b * TPB + threadIdx.x;
        if (j >= n) continue;
        const double sc = (offdiag != nullptr && offdiag[j]) ? scale : 1.0;
        double a0 = 0.0, a1 = 0.0;
        int k = 0;
        for (; k + 1 < Q; k += 2) {
            const double v0 = M[(long long)k * ld + j], v1 = M[(long long)(k + 1) * ld + j];
            a0 += v0 * v0; a1 += v1 * v1;
        }
        if (k < Q) { const double v0 = M[(long long)k * ld + j]; a0 += v0 * v0; }
        ss += (a0 + a1) * sc * sc;
    }
    const double tot = block_sum(ss, sm);
    if (threadIdx.x == 0) part[blockIdx.x] = tot;
}

// one workgroup per quantity: out[q] = sum or max of part[q*stride .. +cnt); workgroups
// q >= nq1 reduce a second family (maxima): out2[q - nq1] = max of part2[(q-nq1)*stride2 .. +cnt2)
__global__ void __launch_bounds__(TPB)
k_combine_multi(const double* __restrict__ part, int stride, int cnt, unsigned long long ismax,
                double* __restrict__ out, int nq1, const double* __restrict__ part2, int stride2, int cnt2,
                double* __restrict__ out2) {
    __shared__ double sm[NWAVE];
    int q = blockIdx.x;
    bool mx;
    if (q >= nq1) { q -= nq1; part = part2; stride = stride2; cnt = cnt2; out = out2; mx = true; }
    else mx = (ismax >> q) & 1ull;
    double a = 0.0;
    for (int i = threadIdx.x; i < cnt; i += TPB) {
        const double v = part[(long long)q * stride + i];
        a = mx ? fmax(a, v) : a + v;
    }
    const double r = mx ? block_max(a, sm) : block_sum(a, sm);
    if (threadIdx.x == 0) out[q] = r;
}

// the accepted linesearch candidate becomes the iterate: y <- y_c and (M'y)|S <- (M'y_c)|S in ONE launch
// (two device-to-device copies cost two runtime copy kernels, 4.2 us each, per iteration)
__global__ void __launch_bounds__(TPB)
k_copy2(double* __restrict__ d1, const double* __restrict__ s1, long long n1,
        double* __restrict__ d2, const double* __restrict__ s2, long long n2) {
    const long long stride = (long long)gridDim.x * TPB;
    for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < n1 + n2; i += stride) {
        if (i < n1) d1[i] = s1[i]; else d2[i - n1] = s2[i - n1];
    }
}

// coupling rows of a block-sharded solve: buf[k] = v[rows[k]] and back
__global__ void __launch_bounds__(TPB)
k_gather_rows(const double* __restrict__ v, const int* __restrict__ rows, int cnt, double* __restrict__ buf) {
    const int k = blockIdx.x * TPB + threadIdx.x;
    if (k < cnt) buf[k] = v[rows[k]];
}
__global__ void __launch_bounds__(TPB)
k_scatter_rows(double* __restrict__ v, const int* __restrict__ rows, int cnt, const double* __restrict__ buf) {
    const int k = blockIdx.x * TPB + threadIdx.x;
    if (k < cnt) v[rows[k]] = buf[k];
}

// v .*= d (removing the equilibration at the exit path, pdhg.jl:751-755)
__global__ void __launch_bounds__(TPB)
k_scale_by(double* __restrict__ v, const double* __restrict__ d, long long n) {
    for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < n; i += (long long)gridDim.x * TPB) v[i] *= d[i];
}

// v[offdiag] *= s over all PSD blocks (fix_diag_scaling, pdhg.jl:734-743) -- used
// on the exit path and when a certificate search snapshots the solution.
__global__ void __launch_bounds__(TPB)
k_scale_offdiag(double* __restrict__ xp, int n, double s) {
    int I, J;
    tile_coords(blockIdx.x, I, J);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int gi = I * TILE + lane;
#pragma unroll
    for (int k = 0; k < CPW; ++k) {
        const int gj = J * TILE + w * CPW + k;
        if (gj < n && gi < gj) xp[(long long)gj * (gj + 1) / 2 + gi] *= s;
    }
}

}  // namespace dev
}  // namespace proxsdp
