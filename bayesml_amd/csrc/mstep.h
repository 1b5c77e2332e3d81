// M-step kernel: per-component weighted sufficient statistics on f64 MFMA.
//
// Replaces the reference's _calc_n_x_bar_s (bayesml/gaussianmixture/_gaussianmixture.py:725-732:
// ns = r.sum(0); x_bar = r.T @ x; S_k = ((r_k * diff.T) @ diff) / ns_k, two passes over x and one
// [N, D] temporary pair per component) and the N-sized lower-bound term -sum xlogy(r, r) (:704)
// with one pass that accumulates, about a fixed pivot p,
//   ns_k = sum_n r_nk,  a_k = sum_n r_nk (x_n - p),  B_k = sum_n r_nk (x_n - p)(x_n - p)^T,
//   h_k = sum_n r_nk ln r_nk,
// which are linear in the rows (row shards and GPUs add).  r_nk = exp(ln rho_nk - lse_n) is
// recomputed from the E-step's output; the [N, K] responsibility matrix is never materialised.
//
// Mapping: WS waves share ONE component k and a contiguous row range (WS = 1 up to T = 7; WS = 2 at
// T = 8, where the 36 accumulator tiles = 288 registers exceed the 256 AGPRs and would otherwise
// bounce through v_accvgpr copies every step); the 4/WS component slots of a workgroup take
// consecutive k over the SAME rows (their x loads hit L1).  The sample index is the MFMA contraction
// index: D[i][j] += A[i][n] B[n][j] with A = r_n x'_n (feature tile t1), B = x'_n (feature tile t2),
// t1 <= t2 only (B_k is symmetric).  Lane l = (i = l & 15, g = l >> 4) handles sample n0 + g and
// features T*i .. T*i + T-1 (contiguous in memory: one vector load per sample), i.e. feature f
// lives in tile f % T at row f / T.  A wave's accumulator tiles (pairs p with p % WS == sub) stay in
// registers for the whole row range (f64 accumulation, no flush needed) and are written once as a
// slab in register order; reduce_stats sums slabs over row splits in a fixed order.
#pragma once
#include <utility>
#include "common.h"

namespace gmmvb {

// T > 8 (128 < D <= 256, round 4): ceil(T / 2) waves share a component, wave w owning the A-operand tiles w and T - 1 - w
// (T + 1 tile pairs each at even T: 136 accumulator registers at T = 16), one component per workgroup.
__host__ __device__ constexpr int mstep_ws(int t) { return t > 8 ? (t + 1) / 2 : (t >= 8 ? 2 : 1); }
// waves per workgroup.  8-wave workgroups (4 components per row stream at T = 8) were measured: L2-side
// fetch traffic drops 3x (96 -> 33 GB per launch at C3) but the kernel is 12 % slower (the 256-register cap
// of a 512-thread workgroup costs more than the traffic, which is nowhere near a bandwidth limit) -> 4.
__host__ __device__ constexpr int mstep_waves(int t, bool pre) { return t > 8 ? mstep_ws(t) : 4; }

// Which of the WS waves of a component owns A-operand tile t1 (and every pair (t2, t1) with it).
// WS = 2 (T = 8): t1 in {0,3,4,7} -> wave 0, {1,2,5,6} -> wave 1: 18 tile pairs each, and each wave
// only forms r * x' for its own four t1 (f64 VALU work competes with the f64 MFMA pipe on gfx950).
__host__ __device__ constexpr int mstep_owner(int ws, int t1) {
    // (ws > 2 only occurs with T = 2 ws or 2 ws - 1 feature tiles: T - 1 - t1 = the mirror column)
    return ws == 1 ? 0 : (ws == 2 ? (((t1 & 3) == 0 || (t1 & 3) == 3) ? 0 : 1) : (t1 < ws ? t1 : -1));
}
// ... with the number of tiles known (the mirror column of the wide form)
__host__ __device__ constexpr int mstep_owner_t(int t, int ws, int t1) {
    return ws <= 2 ? mstep_owner(ws, t1) : (t1 < ws ? t1 : t - 1 - t1);
}
// rank of pair (t2, t1) among the pairs owned by the same wave (its accumulator slot)
__host__ __device__ constexpr int mstep_slot_t(int t, int ws, int t2, int t1) {
    int n = 0;
    for (int b = 0; b <= t2; ++b)
        for (int a = 0; a <= b; ++a) {
            if (b == t2 && a == t1) return n;
            if (mstep_owner_t(t, ws, a) == mstep_owner_t(t, ws, t1)) ++n;
        }
    return n;
}
__host__ __device__ constexpr int mstep_owned(int ws, int sub, int t) {
    int n = 0;
    for (int b = 0; b < t; ++b)
        for (int a = 0; a <= b; ++a)
            if (mstep_owner_t(t, ws, a) == sub) ++n;
    return n;
}

// PRE = true: rows come from the workspace's centred f64 copy xc[n][16T] = (double)x - pivot (zero padded),
// made once per sample matrix by center_rows_kernel, so the loop has no convert/subtract/mask work.
// LIST = true: [lo, hi) are positions in `list` (ascending rows of x whose responsibility for this component is not
// negligible, see "sparse responsibilities" below); entry e stands for row list[e].
// FULL = true: D == 16 T is known (whole feature tiles): no per-element range test in the loop.
template <int T, int WS, int SUB, typename XT, bool VEC, bool PRE, bool LIST = false, bool FULL = false>
__device__ __forceinline__ void mstep_body(const XT* __restrict__ x, int64_t ldx, int64_t n_rows, int D,
                                           const double* __restrict__ pivot, const double* __restrict__ lr,
                                           const double* __restrict__ lse, const double* __restrict__ aux_k,
                                           int64_t lo, int64_t hi, int direct_r, double* __restrict__ out,
                                           const int* __restrict__ list = nullptr) {
    constexpr int P = tri_pairs(T);
    constexpr int NP = mstep_owned(WS, SUB, T);   // tile pairs owned by this wave
    const int lane = threadIdx.x & 63;
    const int i = lane & 15;
    const int g = lane >> 4;

    double pv[T];
#pragma unroll
    for (int t = 0; t < T; ++t) pv[t] = (!PRE && (FULL || T * i + t < D)) ? pivot[T * i + t] : 0.0;
    d4 acc[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) acc[p] = d4{0.0, 0.0, 0.0, 0.0};
    double asum[T];
#pragma unroll
    for (int t = 0; t < T; ++t) asum[t] = 0.0;
    double nsum = 0.0, hsum = 0.0;

    struct RawRow { XT v[T]; };
    auto load_row = [&](int64_t row) {
        // PRE: the centred copy has zero rows up to npad + 64, so no clamp (and no 64-bit compare/select/multiply
        // per step); otherwise clamp to the last valid row (r is 0 there)
        // (list entries are rows of the matrix by construction)
        if constexpr (!PRE && !LIST) {
            if (row >= n_rows) row = n_rows - 1;
        }
        const XT* xp = x + row * ldx + T * i;
        RawRow o;
        if constexpr (PRE) {
            typedef XT v2 __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int t = 0; t < T; t += 2) {
                if (t + 1 < T) {
                    const v2 v = *reinterpret_cast<const v2*>(xp + t);
                    o.v[t] = v[0];
                    o.v[t + 1] = v[1];
                } else {
                    o.v[t] = xp[t];
                }
            }
        } else if constexpr (VEC) {
            typedef XT vt __attribute__((ext_vector_type(T)));
            const vt v = *reinterpret_cast<const vt*>(xp);
#pragma unroll
            for (int t = 0; t < T; ++t) o.v[t] = v[t];
        } else {
#pragma unroll
            for (int t = 0; t < T; ++t) o.v[t] = (T * i + t < D) ? xp[t] : XT(0);
        }
        return o;
    };

    // Software pipeline, one step = 4 samples: the row loads and the responsibility broadcast of step
    // s+1 are issued before the MFMAs of step s and consumed after them.  The sched_barriers keep hipcc
    // from hoisting those loads above the VALU block, where a register it then reuses forces a
    // vmcnt(0) right behind the issue (measured: -7 % on the whole kernel).
    int idx_n = 0;                         // LIST: rows of the next batch of 64 entries, one per lane
    // (a list entry's sign bit marks a row that LEAVES the settled-row cache: weight -1, direct_r == 3)
    if constexpr (LIST) idx_n = (lo + lane < hi) ? list[lo + lane] : 0;
    RawRow nxt = load_row(LIST ? (int64_t)(__shfl(idx_n, g) & 0x7FFFFFFF) : lo + g);
    // Measured dead ends of the LIST form (T = 8 sits exactly at the 256 registers of two waves per SIMD - 144 of them
    // accumulators): rows requested TWO steps ahead, in registers (round 3: 2.30 instead of 1.81 ms per step; round 4 with the
    // pivot moved to LDS to pay for them: 1.76 instead of 1.57) or staged six steps ahead through an LDS ring by LDS-DMA
    // (round 4: 1.75 instead of 1.57): the loop is bound by what it issues beside the MFMAs, not by the gather's latency.
    // What did help (round 4): no per-element range test and no row clamp when D == 16 T (FULL): 1.87 -> 1.57 ms.
    for (int64_t c0 = lo; c0 < hi; c0 += 64) {
        // responsibilities of 64 samples, one per lane
        const int64_t nl = c0 + lane;
        const int idx_l = idx_n;
        if constexpr (LIST) idx_n = (nl + 64 < hi) ? list[nl + 64] : 0;
        double r_l = 0.0;
        if (nl < hi) {
            const int64_t src = LIST ? (int64_t)(idx_l & 0x7FFFFFFF) : nl;
            const double v = lr[src];
            if (LIST && direct_r == 3) {         // delta lists of the settled-row cache: whole rows in or out
                r_l = idx_l < 0 ? -1.0 : 1.0;
            } else if (direct_r == 2) {                 // HMM: r = gamma, h accumulates sum gamma * ln rho (aux)
                r_l = v;
                if (v > 0.0 && aux_k) hsum = fma(v, aux_k[src], hsum);      // (no ln rho array: h stays 0, hmmvb_emission_target)
            } else if (direct_r) {
                r_l = v;
                if (v > 0.0) hsum = fma(v, log(v), hsum);
            } else {
                const double t = v - lse[src];
                r_l = exp(t);
                hsum = fma(r_l, t, hsum);        // r ln r, with ln r = ln rho - lse exactly
            }
            nsum += r_l;
        }
        double rr_n = __shfl(r_l, g);
#pragma unroll 2
        for (int st = 0; st < 16; ++st) {
            if (LIST && c0 + 4 * st >= hi) break;      // the list ended inside this batch (wave uniform)
            const RawRow cur = nxt;
            const double rr = rr_n;
            double xq[T];
#pragma unroll
            for (int t = 0; t < T; ++t) {
                if constexpr (PRE) xq[t] = (double)cur.v[t];
                else if constexpr (FULL) xq[t] = (double)cur.v[t] - pv[t];
                else xq[t] = (T * i + t < D) ? (double)cur.v[t] - pv[t] : 0.0;
            }
            double ra[T];
#pragma unroll
            for (int t = 0; t < T; ++t) {
                if (mstep_owner_t(T, WS, t) == SUB) {
                    ra[t] = rr * xq[t];
                    asum[t] += ra[t];
                } else {
                    ra[t] = 0.0;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (LIST)
                nxt = load_row((int64_t)(__shfl(st == 15 ? idx_n : idx_l, (4 * (st + 1) + g) & 63) & 0x7FFFFFFF));
            else
                nxt = load_row(c0 + 4 * (st + 1) + g);
            rr_n = __shfl(r_l, (4 * (st + 1) + g) & 63);      // st = 15: unused (next batch recomputes)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t2 = 0; t2 < T; ++t2) {
#pragma unroll
                for (int t1 = 0; t1 <= t2; ++t1) {
                    if (mstep_owner_t(T, WS, t1) == SUB)
                        acc[mstep_slot_t(T, WS, t2, t1)] = mfma_f64(ra[t1], xq[t2], acc[mstep_slot_t(T, WS, t2, t1)]);
                }
            }
            // keep every loaded element live to here: a dead half of a 16-byte load (tile 0 is never a B
            // operand of wave 1) otherwise gets its register reused while the load is in flight -> vmcnt(0)
#pragma unroll
            for (int t = 0; t < T; ++t) asm volatile("" ::"v"(xq[t]));
        }
    }

#pragma unroll
    for (int t2 = 0; t2 < T; ++t2) {
#pragma unroll
        for (int t1 = 0; t1 <= t2; ++t1) {
            if (mstep_owner_t(T, WS, t1) == SUB) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    out[(pair_index(t2, t1) * 4 + r) * 64 + lane] = acc[mstep_slot_t(T, WS, t2, t1)][r];
            }
        }
    }
#pragma unroll
    for (int t = 0; t < T; ++t) {
        if (mstep_owner_t(T, WS, t) == SUB) {
            const double v = sum_groups(asum[t]);
            if (g == 0) out[P * 256 + T * i + t] = v;
        }
    }
    if (SUB == 0) {
        nsum = sum_wave(nsum);
        hsum = sum_wave(hsum);
        if (lane == 0) {
            out[P * 256 + 16 * T + 0] = nsum;
            out[P * 256 + 16 * T + 1] = hsum;
        }
    }
}

template <int T, typename XT, bool VEC, bool PRE>
__global__ __launch_bounds__(64 * mstep_waves(T, PRE)) void mstep_mfma_f64(
    const XT* __restrict__ x, int64_t ldx, int64_t n_rows, int D,
    const double* __restrict__ pivot,      // [D]
    const double* __restrict__ lnrho,      // [K][npad]  (ln rho, or r itself when direct_r)
    const double* __restrict__ lse,        // [npad]
    const double* __restrict__ aux,        // [K][npad] (direct_r == 2 only)
    int64_t npad, int K, int KG, int S, int64_t rows_per_split, int direct_r,
    double* __restrict__ slabs /*[S][K][slab_len(T)]*/) {
    constexpr int WS = mstep_ws(T);
    constexpr int KPW = mstep_waves(T, PRE) / WS;   // components per workgroup
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    // XCD-aware decode: blocks b and b+8 share an XCD (round-robin dispatch), so give every XCD
    // whole row splits: all KG component groups of a split stream the same rows through one L2.
    const int bid = blockIdx.x;
    const int xcd = bid & 7;
    const int j = bid >> 3;
    const int kg = j % KG;
    const int split = (j / KG) * 8 + xcd;
    if (split >= S) return;
    const int k = kg * KPW + wave / WS;
    if (k >= K) return;
    const int sub = wave % WS;

    const int64_t lo = (int64_t)split * rows_per_split;
    int64_t hi = lo + rows_per_split;
    if (hi > n_rows) hi = n_rows;
    const double* lr = lnrho + (int64_t)k * npad;
    const double* aux_k = aux ? aux + (int64_t)k * npad : nullptr;
    double* out = slabs + ((int64_t)split * K + k) * slab_len(T);
    if constexpr (WS == 1) {
        mstep_body<T, 1, 0, XT, VEC, PRE>(x, ldx, n_rows, D, pivot, lr, lse, aux_k, lo, hi, direct_r, out);
    } else if constexpr (WS == 2) {
        if (sub == 0)
            mstep_body<T, 2, 0, XT, VEC, PRE>(x, ldx, n_rows, D, pivot, lr, lse, aux_k, lo, hi, direct_r, out);
        else
            mstep_body<T, 2, 1, XT, VEC, PRE>(x, ldx, n_rows, D, pivot, lr, lse, aux_k, lo, hi, direct_r, out);
    } else {
        static_assert(WS <= 2, "more than eight feature tiles: mstep_wide_f64");
    }
}

// ---- 128 < D <= 256: T = 10, 12, 14, 16 feature tiles ----------------------------------------------------------------
// A component's T (T + 1) / 2 accumulator tiles (136 at T = 16: 1088 registers) are spread over T / 2 waves: wave w owns
// the A-operand tiles w and T - 1 - w, i.e. the tile pairs (t2, w) for t2 >= w and (t2, T - 1 - w) for t2 >= T - 1 - w:
// T + 1 pairs each.  One component per workgroup; rows from the centred f64 copy (16 T doubles per row, T per lane); the
// per-step vector work of a wave is two multiplies and two additions beside its T + 1 MFMAs.  Same operations per tile pair
// in the same order as mstep_body, same slab layout, same reduce.
// Round 5: the rows go through LDS.  Every one of the workgroup's T / 2 waves used to load the whole row from global memory -
// 16 T doubles per step and wave, the same bytes T / 2 times through one L1 - and at T = 16 that L1 traffic took as long as the
// wave's T + 1 MFMAs (0.51 of the f64 peak).  Now the workgroup copies a batch of 16 rows (contiguous in the centred copy)
// into LDS once, a batch ahead through registers, in the order the MFMA operands are read - element (row r, tile t, lane
// feature i) at ((r >> 2) T + t) 64 + (r & 3) 16 + i, so that the operand of (step, tile) is the lane-linear read
// sx[(st T + t) 64 + lane]: conflict-free - and a wave reads only the T - SUB tiles it multiplies.  Two buffers, one
// barrier per batch of four steps (4 (T + 1) MFMAs per wave).
constexpr int kWideBatch = 16;      // rows per LDS batch
template <int T>
__device__ __forceinline__ int wide_lds_index(int r, int f) {       // row r of the batch, feature f = T i + t of the row
    const int i = f / T, t = f - T * i;
    return ((r >> 2) * T + t) * 64 + (r & 3) * 16 + i;
}
template <int T, int SUB>
__device__ __forceinline__ void mstep_wide_body(const double* __restrict__ xc, int64_t n_rows, const double* __restrict__ lr,
                                                const double* __restrict__ lse, const double* __restrict__ aux_k, int64_t lo,
                                                int64_t hi, int direct_r, double* __restrict__ out, double* __restrict__ sx) {
    static_assert(T > 8 && T <= 16 && T % 2 == 0, "even tile counts (the workspace rounds an odd one up)");
    constexpr int P = tri_pairs(T);
    constexpr int C1 = SUB, C2 = T - 1 - SUB;      // the wave's A-operand tiles (C1 < C2)
    constexpr int N1 = T - C1, N2 = T - C2;        // tile pairs (t2, C1), t2 = C1 .. T - 1, and (t2, C2), t2 = C2 .. T - 1
    constexpr int BUF = kWideBatch * 16 * T;       // doubles per LDS buffer
    const int tid = threadIdx.x, lane = tid & 63, i = lane & 15, g = lane >> 4;
    d4 acc1[N1], acc2[N2];
#pragma unroll
    for (int p = 0; p < N1; ++p) acc1[p] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int p = 0; p < N2; ++p) acc2[p] = d4{0.0, 0.0, 0.0, 0.0};
    double asum1 = 0.0, asum2 = 0.0, nsum = 0.0, hsum = 0.0;
    // staging: the batch is 256 T contiguous doubles of the centred copy, the workgroup has 32 T threads: eight each
    // (zero rows up to npad + 64: no clamp)
    d4 s0 = {0.0, 0.0, 0.0, 0.0}, s1 = s0;
    auto request = [&](int64_t row0) {
        const double* src = xc + row0 * (16 * T) + 8 * tid;
        s0 = *reinterpret_cast<const d4*>(src);
        s1 = *reinterpret_cast<const d4*>(src + 4);
    };
    auto deposit = [&](double* dst) {
        const int r = (8 * tid) / (16 * T), f0 = 8 * tid - r * (16 * T);      // (16 T is a multiple of 8: one row per thread)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            dst[wide_lds_index<T>(r, f0 + e)] = s0[e];
            dst[wide_lds_index<T>(r, f0 + 4 + e)] = s1[e];
        }
    };
    request(lo);
    deposit(sx);
    __syncthreads();
    int b = 0;
    for (int64_t c0 = lo; c0 < hi; c0 += 64) {
        const int64_t nl = c0 + lane;
        double r_l = 0.0;
        if (nl < hi) {
            const double v = lr[nl];
            if (direct_r == 2) {                   // HMM: r = gamma, h accumulates sum gamma * ln rho (aux)
                r_l = v;
                if (v > 0.0 && aux_k) hsum = fma(v, aux_k[nl], hsum);
            } else if (direct_r) {
                r_l = v;
                if (v > 0.0) hsum = fma(v, log(v), hsum);
            } else {
                const double t = v - lse[nl];
                r_l = exp(t);
                hsum = fma(r_l, t, hsum);
            }
            nsum += r_l;
        }
#pragma unroll 1
        for (int q = 0; q < 64 / kWideBatch; ++q, b ^= 1) {
            const int64_t r0 = c0 + kWideBatch * q;
            if (r0 >= hi) break;                   // (workgroup uniform: lo, hi are)
            const bool more = r0 + kWideBatch < hi;
            if (more) request(r0 + kWideBatch);    // in flight while this batch is worked through
            const double* cur = sx + b * BUF;
#pragma unroll
            for (int st = 0; st < kWideBatch / 4; ++st) {
                const double rr = __shfl(r_l, kWideBatch * q + 4 * st + g);
                double xv[N1];                     // tiles C1 .. T - 1 of the step's four rows (C2 .. T - 1 are among them)
#pragma unroll
                for (int p = 0; p < N1; ++p) xv[p] = cur[(st * T + C1 + p) * 64 + lane];
                const double ra1 = rr * xv[0], ra2 = rr * xv[C2 - C1];
                asum1 += ra1;
                asum2 += ra2;
#pragma unroll
                for (int p = 0; p < N1; ++p) acc1[p] = mfma_f64(ra1, xv[p], acc1[p]);
#pragma unroll
                for (int p = 0; p < N2; ++p) acc2[p] = mfma_f64(ra2, xv[C2 - C1 + p], acc2[p]);
            }
            if (more) deposit(sx + (b ^ 1) * BUF);
            __syncthreads();                       // the next batch is in LDS; everybody is done with this one
        }
    }
#pragma unroll
    for (int p = 0; p < N1; ++p)
#pragma unroll
        for (int r = 0; r < 4; ++r) out[(pair_index(C1 + p, C1) * 4 + r) * 64 + lane] = acc1[p][r];
#pragma unroll
    for (int p = 0; p < N2; ++p)
#pragma unroll
        for (int r = 0; r < 4; ++r) out[(pair_index(C2 + p, C2) * 4 + r) * 64 + lane] = acc2[p][r];
    {
        const double a1 = sum_groups(asum1), a2 = sum_groups(asum2);
        if (g == 0) {
            out[P * 256 + T * i + C1] = a1;
            out[P * 256 + T * i + C2] = a2;
        }
    }
    if (SUB == 0) {
        nsum = sum_wave(nsum);
        hsum = sum_wave(hsum);
        if (lane == 0) {
            out[P * 256 + 16 * T + 0] = nsum;
            out[P * 256 + 16 * T + 1] = hsum;
        }
    }
}

template <int T, int... I>
__device__ __forceinline__ void mstep_wide_dispatch(int sub, const double* __restrict__ xc, int64_t n_rows,
                                                    const double* __restrict__ lr, const double* __restrict__ lse,
                                                    const double* __restrict__ aux_k, int64_t lo, int64_t hi, int direct_r,
                                                    double* __restrict__ out, double* __restrict__ sx,
                                                    std::integer_sequence<int, I...>) {
    ((sub == I ? mstep_wide_body<T, I>(xc, n_rows, lr, lse, aux_k, lo, hi, direct_r, out, sx) : (void)0), ...);
}

template <int T>
__global__ __launch_bounds__(32 * T) void mstep_wide_f64(const double* __restrict__ xc, int64_t n_rows,
                                                         const double* __restrict__ lnrho, const double* __restrict__ lse,
                                                         const double* __restrict__ aux, int64_t npad, int K, int KG, int S,
                                                         int64_t rows_per_split, int direct_r,
                                                         double* __restrict__ slabs /*[S][K][slab_len(T)]*/) {
    __shared__ __attribute__((aligned(16))) double sx[2 * kWideBatch * 16 * T];      // two batches of 16 rows (64 KB at T = 16)
    const int sub = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // T / 2 waves, one component
    const int bid = blockIdx.x;
    const int xcd = bid & 7;
    const int j = bid >> 3;
    const int k = j % KG;                          // (KG = K: one component per workgroup)
    const int split = (j / KG) * 8 + xcd;
    if (split >= S || k >= K) return;
    const int64_t lo = (int64_t)split * rows_per_split;
    int64_t hi = lo + rows_per_split;
    if (hi > n_rows) hi = n_rows;
    const double* lr = lnrho + (int64_t)k * npad;
    const double* aux_k = aux ? aux + (int64_t)k * npad : nullptr;
    double* out = slabs + ((int64_t)split * K + k) * slab_len(T);
    mstep_wide_dispatch<T>(sub, xc, n_rows, lr, lse, aux_k, lo, hi, direct_r, out, sx, std::make_integer_sequence<int, T / 2>{});
}

// ---- one feature tile (D <= 16): many components per wave ----------------------------------------------------------
// With T = 1 a component's whole second moment is ONE 16 x 16 accumulator tile, and a step of mstep_body is a row load,
// a shuffle, two multiplies and one MFMA: the kernel above spends its time issuing the loop, re-reading the rows once per
// component group (HMM config 5, K = 32, D = 16: 8.3 ms for 1 ms of MFMA work).  Here a wave keeps CW components'
// tiles and walks the rows ONCE for all of them; the four waves of a workgroup take 4 CW consecutive components over the
// same rows.  Same operations per component in the same order as mstep_body<1, ...> (bit-identical slabs), same slab
// layout, same reduce.  Rows come from the centred f64 copy ([npad + 64][16], zero padded).
template <int CW>
__global__ __launch_bounds__(256) void mstep_small_f64(const double* __restrict__ xc, int64_t n_rows,
                                                       const double* __restrict__ lnrho, const double* __restrict__ lse,
                                                       const double* __restrict__ aux, int64_t npad, int K, int KGW, int S,
                                                       int64_t rows_per_split, int direct_r,
                                                       double* __restrict__ slabs /*[S][K][slab_len(1)]*/) {
    const int lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int bid = blockIdx.x;
    const int xcd = bid & 7, j = bid >> 3;
    const int kgw = j % KGW;
    const int split = (j / KGW) * 8 + xcd;
    if (split >= S) return;
    const int k0 = (kgw * 4 + wave) * CW;
    if (k0 >= K) return;
    const int64_t lo = (int64_t)split * rows_per_split;
    int64_t hi = lo + rows_per_split;
    if (hi > n_rows) hi = n_rows;
    d4 acc[CW];
    double asum[CW], nsum[CW], hsum[CW];
#pragma unroll
    for (int c = 0; c < CW; ++c) {
        acc[c] = d4{0.0, 0.0, 0.0, 0.0};
        asum[c] = nsum[c] = hsum[c] = 0.0;
    }
    double xn = xc[(lo + g) * 16 + i];
    for (int64_t c0 = lo; c0 < hi; c0 += 64) {
        const int64_t nl = c0 + lane;
        double r_l[CW];
#pragma unroll
        for (int c = 0; c < CW; ++c) {
            const int k = k0 + c;
            double r = 0.0;
            if (k < K && nl < hi) {
                const double v = lnrho[(int64_t)k * npad + nl];
                if (direct_r == 2) {                 // HMM: r = gamma, h accumulates sum gamma * ln rho (aux)
                    r = v;
                    if (v > 0.0 && aux) hsum[c] = fma(v, aux[(int64_t)k * npad + nl], hsum[c]);
                } else if (direct_r) {
                    r = v;
                    if (v > 0.0) hsum[c] = fma(v, log(v), hsum[c]);
                } else {
                    const double t = v - lse[nl];
                    r = exp(t);
                    hsum[c] = fma(r, t, hsum[c]);
                }
                nsum[c] += r;
            }
            r_l[c] = r;
        }
#pragma unroll 4
        for (int st = 0; st < 16; ++st) {
            const double xq = xn;
            xn = xc[(c0 + 4 * (st + 1) + g) * 16 + i];          // (zero rows up to npad + 64: no clamp)
#pragma unroll
            for (int c = 0; c < CW; ++c) {
                const double ra = __shfl(r_l[c], 4 * st + g) * xq;
                asum[c] += ra;
                acc[c] = mfma_f64(ra, xq, acc[c]);
            }
        }
    }
#pragma unroll
    for (int c = 0; c < CW; ++c) {
        const int k = k0 + c;
        if (k >= K) break;
        double* out = slabs + ((int64_t)split * K + k) * slab_len(1);
#pragma unroll
        for (int r = 0; r < 4; ++r) out[r * 64 + lane] = acc[c][r];
        const double a = sum_groups(asum[c]);
        if (g == 0) out[256 + i] = a;
        const double n = sum_wave(nsum[c]), h = sum_wave(hsum[c]);
        if (lane == 0) {
            out[256 + 16 + 0] = n;
            out[256 + 16 + 1] = h;
        }
    }
}

// ---- the same for the HMM, with gamma read as the forward-backward pass leaves it --------------------------------------
// hmmvb_forward_backward writes gamma time-major ([T][Kp], "lane order" within every 16-state block, hmm.h); round 3
// transposed it to component-major for mstep_small_f64 (hmm_gamma_to_cm_kernel: 1.8 ms and 5.1 GB per iteration at config 5)
// and then moved every responsibility across lanes twice per use (a 64-bit shuffle per step and component).  Here the
// workgroup - four waves x CW = 8 components = the 32 states of two 16-blocks - copies the 64 rows x 256 bytes of its states
// into LDS once per batch (contiguous in memory, one batch ahead), and every use is an LDS read: the row's own value for
// ns / h, a 16-lane broadcast of sample 4 st + g for the MFMA operand.  Same operations per component in the same order as
// mstep_small_f64 with direct_r = 2: bit-identical slabs.  aux = ln rho, component-major (the emission E-step's output).
// a gamma below the relevance line of common.h (2^-80) is left out of the HMM M-step's first and second moments
constexpr double kHmmGammaFloor = 8.271806125530277e-25;        // 2^-80
static_assert(kRelevanceBits == 80, "kHmmGammaFloor is 2^-kRelevanceBits");
// a component with this many of a batch's 16 MFMA steps above the line takes the dense form's unrolled loop instead of the
// bit walk (an iteration of the walk waits for its LDS operands).  Measured at config 5, ms per VB iteration over passes 3-7
// of a fit (profiles/r5_experiments.md): 6 -> 11.0-11.2, 12 -> 10.65-10.78, never -> 10.58-10.68, every step dense 12.3-12.5;
// 12 keeps the walk's worst case (all 16 steps of all 8 components) at the dense kernel's time.
#ifndef GMMVB_HMM_DENSE_FROM
#define GMMVB_HMM_DENSE_FROM 12
#endif
constexpr int kHmmDenseFrom = GMMVB_HMM_DENSE_FROM;
__host__ __device__ constexpr int lane_order_pos(int state) {          // hmm.h: hmm_pos
    const int w = state & 15;
    return (state & ~15) + 4 * (w & 3) + (w >> 2);
}

// AUX = false: the emission went straight to the forward-backward buffers and no ln rho array exists (hmm.h H0 + H1): h
// stays 0 - the host takes sum gamma ln rho from the moments, in closed form - and a third of the kernel's reads is gone.
// SPARSE (round 5): gamma is as sparse as a mixture's responsibilities - with informative emissions a time step belongs to
// one or two states - and the dense form spends K MFMAs per four steps on products whose weight is below the relevance line
// (common.h: gamma < 2^-80 is invisible in every f64 sum over fewer than 2^27 terms).  A ballot of the lanes' own values
// (lane = row of the batch) gives every component of the wave a 64-bit row mask, folded to one bit per MFMA step (four
// rows); the wave walks the SET bits only - a scalar loop: the masks are wave-uniform - so a (step, component) whose four
// rows are all below the line costs nothing.  ns stays the sum of ALL gamma values (one add per row and component, as
// before); a and B drop the terms below the line and nothing else - the steps that run form the same products in the same
// order.  Config 5 (sticky chain, 3-sigma separated states): 1.3 of 32 components per step are walked, the kernel goes from
// the matrix pipe's time (3.6 ms) to its memory traffic's.  SPARSE = false (GMMVB_HMM_MSTEP_DENSE) is the round-4 kernel.
template <int CW, bool AUX = true, bool SPARSE = false>
__global__ __launch_bounds__(256) void hmm_mstep_small_kernel(const double* __restrict__ xc, int64_t n_rows,
                                                              const double* __restrict__ gamma_tm, int Kp,
                                                              const double* __restrict__ aux, int64_t npad, int K, int KGW, int S,
                                                              int64_t rows_per_split, double* __restrict__ slabs) {
    static_assert(CW == 8, "four waves x CW components = one 32-state slice of a gamma row");
    constexpr int LD = 33;                                  // padded row: lane stride 33 doubles, conflict-free
    __shared__ double sg[2][64 * LD];
    // (round 4) the batch's 64 rows of x go through LDS too, a batch ahead like gamma: the four waves share them, and the
    // MFMA operand of step st is the lane-linear read sx[64 st + lane].  Before, every wave loaded its operand from global
    // memory one step (eight MFMAs, 0.2 us) ahead - far less than a memory round trip, and the kernel ran at 48 % of what
    // its MFMAs need.  The batch's ln rho values (h = sum gamma ln rho) are requested a batch ahead as well.
    __shared__ double sx[2][64 * 16];
    const int tid = threadIdx.x, lane = tid & 63, i = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bid = blockIdx.x;
    const int xcd = bid & 7, j = bid >> 3;
    const int kgw = j % KGW;
    const int split = (j / KGW) * 8 + xcd;
    if (split >= S) return;                                 // (whole workgroup)
    const int k0 = (kgw * 4 + wave) * CW;
    const int64_t lo = (int64_t)split * rows_per_split;
    int64_t hi = lo + rows_per_split;
    if (hi > n_rows) hi = n_rows;
    // staging: thread -> (row, quarter of the 32-state slice)
    const int srow = tid >> 2, sq = tid & 3;
    const int scol = 32 * kgw + 8 * sq;                     // first position of the eight this thread copies
    double stage[8];
    auto request = [&](int64_t c0) {
        const int64_t t = c0 + srow;
        const bool ok = t < hi && scol < Kp;
        const double* src = gamma_tm + (ok ? t : lo) * Kp + (scol < Kp ? scol : 0);
#pragma unroll
        for (int q = 0; q < 8; q += 2) {
            typedef double d2 __attribute__((ext_vector_type(2)));
            const d2 v = *reinterpret_cast<const d2*>(src + q);
            stage[q] = ok ? v[0] : 0.0;
            stage[q + 1] = ok ? v[1] : 0.0;
        }
    };
    d4 xstage = {0.0, 0.0, 0.0, 0.0};
    auto request_x = [&](int64_t c0) { xstage = *reinterpret_cast<const d4*>(xc + c0 * 16 + tid * 4); };      // (zero rows past n_rows: no clamp)
    auto deposit = [&](int b) {
#pragma unroll
        for (int q = 0; q < 8; ++q) sg[b][srow * LD + 8 * sq + q] = stage[q];
        *reinterpret_cast<d4*>(&sx[b][tid * 4]) = xstage;
    };
    int pc[CW];                                             // position of component k0 + c inside the slice
#pragma unroll
    for (int c = 0; c < CW; ++c) pc[c] = lane_order_pos(k0 + c) - 32 * kgw;
    d4 acc[CW];
    double asum[CW], nsum[CW], hsum[CW];
#pragma unroll
    for (int c = 0; c < CW; ++c) {
        acc[c] = d4{0.0, 0.0, 0.0, 0.0};
        asum[c] = nsum[c] = hsum[c] = 0.0;
    }
    double aux_n[CW];                                       // ln rho of the NEXT batch's row of this lane (clamped address)
    auto request_aux = [&](int64_t c0) {
        const int64_t nl = c0 + lane < hi ? c0 + lane : hi - 1;
#pragma unroll
        for (int c = 0; c < CW; ++c) aux_n[c] = AUX ? aux[(int64_t)(k0 + c < K ? k0 + c : 0) * npad + nl] : 0.0;
    };
    request(lo);
    request_x(lo);
    request_aux(lo);
    deposit(0);
    __syncthreads();
    int b = 0;
    for (int64_t c0 = lo; c0 < hi; c0 += 64, b ^= 1) {
        double aux_c[CW];
#pragma unroll
        for (int c = 0; c < CW; ++c) aux_c[c] = aux_n[c];
        if (c0 + 64 < hi) {                                 // in flight while this batch is worked through
            request(c0 + 64);
            request_x(c0 + 64);
            request_aux(c0 + 64);
        }
        const double* sb = sg[b];
        const double* xb = sx[b];
        const int64_t nl = c0 + lane;
        if (k0 < K) {
            unsigned long long act[CW];
#pragma unroll
            for (int c = 0; c < CW; ++c) {
                const int k = k0 + c;
                double v = 0.0;
                if (k < K && nl < hi) {
                    v = sb[lane * LD + pc[c]];
                    if (AUX && v > 0.0) hsum[c] = fma(v, aux_c[c], hsum[c]);
                    nsum[c] += v;
                }
                if constexpr (SPARSE) {
                    // rows of the batch whose gamma for this component is above the relevance line, one bit per MFMA step
                    unsigned long long m = __builtin_amdgcn_ballot_w64(v >= kHmmGammaFloor);
                    m |= m >> 1;
                    m |= m >> 2;
                    act[c] = m & 0x1111111111111111ull;
                }
            }
            if constexpr (SPARSE) {
#pragma unroll
                for (int c = 0; c < CW; ++c) {
                    unsigned long long m = act[c];
                    if (__builtin_popcountll(m) >= kHmmDenseFrom) {
                        // most of the component's steps matter (early iterations, a state that holds the whole batch): the
                        // dense form's static schedule - its LDS reads run ahead of the MFMAs - beats the bit walk
#pragma unroll
                        for (int st = 0; st < 16; ++st) {
                            const double xq = xb[64 * st + lane];
                            const double ra = sb[(4 * st + g) * LD + pc[c]] * xq;
                            asum[c] += ra;
                            acc[c] = mfma_f64(ra, xq, acc[c]);
                        }
                        continue;
                    }
                    while (m != 0ull) {                      // (wave-uniform: a scalar loop over the steps that matter, two at a time)
                        const int st0 = __builtin_ctzll(m) >> 2;
                        m &= m - 1ull;
                        const bool two = m != 0ull;
                        const int st1 = two ? __builtin_ctzll(m) >> 2 : st0;
                        m &= m - 1ull;                       // (0 & anything = 0)
                        const double xq0 = xb[64 * st0 + lane], g0 = sb[(4 * st0 + g) * LD + pc[c]];
                        const double xq1 = xb[64 * st1 + lane], g1 = sb[(4 * st1 + g) * LD + pc[c]];
                        const double ra0 = g0 * xq0;
                        asum[c] += ra0;
                        acc[c] = mfma_f64(ra0, xq0, acc[c]);
                        if (two) {
                            const double ra1 = g1 * xq1;
                            asum[c] += ra1;
                            acc[c] = mfma_f64(ra1, xq1, acc[c]);
                        }
                    }
                }
            } else {
#ifndef GMMVB_HMM_MSTEP_UNROLL
#define GMMVB_HMM_MSTEP_UNROLL 4
#endif
#pragma unroll GMMVB_HMM_MSTEP_UNROLL
                for (int st = 0; st < 16; ++st) {
                    const double xq = xb[64 * st + lane];
#pragma unroll
                    for (int c = 0; c < CW; ++c) {
                        const double ra = (k0 + c < K ? sb[(4 * st + g) * LD + pc[c]] : 0.0) * xq;
                        asum[c] += ra;
                        acc[c] = mfma_f64(ra, xq, acc[c]);
                    }
                }
            }
        }
        if (c0 + 64 < hi) deposit(b ^ 1);
        __syncthreads();                                    // the next batch is in LDS; everybody is done with this one
    }
    if (k0 >= K) return;
#pragma unroll
    for (int c = 0; c < CW; ++c) {
        const int k = k0 + c;
        if (k >= K) break;
        double* out = slabs + ((int64_t)split * K + k) * slab_len(1);
#pragma unroll
        for (int r = 0; r < 4; ++r) out[r * 64 + lane] = acc[c][r];
        const double a = sum_groups(asum[c]);
        if (g == 0) out[256 + i] = a;
        const double n = sum_wave(nsum[c]), h = sum_wave(hsum[c]);
        if (lane == 0) {
            out[256 + 16 + 0] = n;
            out[256 + 16 + 1] = h;
        }
    }
}

// ---- sparse responsibilities ---------------------------------------------------------------------------------
// After the first VB iterations most responsibilities are negligible: a sample belongs to a handful of the K
// components.  A term with r_nk < 2^-80 max_n r_nk cannot change any of component k's f64 sums (there are fewer
// than 2^27 terms and the sums are at least as large as their largest term), so the statistics are unchanged to the
// last bit of rounding when such samples are skipped - and skipping them removes their row loads and MFMAs.
// thr[k] = max_n (ln rho_nk - lse_n) - 80 ln 2 over a sample of the rows (row_lse_kernel / thr_kernel); lse_mask_kernel,
// scan_counts and fill_lists (aux_kernels.h) write, per component, the ascending list of its active rows; mstep_list_f64 is
// mstep_body over list positions instead of rows (LIST form: one index load per 64 entries, the row addresses go
// through it).
// Work distribution: the list of component k is cut into chunks of R entries; plan[k] = index of its first chunk,
// plan[K] = number of chunks, plan[K + 1] = R (mstep_plan_kernel: R >= r_min, raised so that the chunks fit the slab
// capacity).  Chunk c is one (component, list range) whatever rows it spans, so the work is balanced whether the rows
// of a component are scattered over the matrix or (after the rows were grouped by component) contiguous; every chunk
// writes one slab, reduce_chunks_kernel adds a component's slabs in chunk order (fixed order: run-to-run identical).
static __global__ void mstep_plan_kernel(const int* __restrict__ counts, int K, int cap_chunks, int r_min, int* __restrict__ plan) {
    // One workgroup (one wave).  The counts arrive with parallel loads, their sum and the chunk counts (a 64-bit division
    // each) are formed by all lanes, and only the prefix over the chunk counts - additions from LDS - is left to one thread:
    // a single thread's 2 K dependent loads and K divisions were 42 us at K = 256.
    __shared__ int sc[1025];
    __shared__ long long s_total;
    if (blockIdx.x != 0) return;
    if (K > 1024) {                                   // (no caller: the lists stop at 256 components)
        if (threadIdx.x != 0) return;
        long long total = 0;
        for (int k = 0; k < K; ++k) total += counts[k];
        long long R = r_min;
        const long long room = cap_chunks - K > 0 ? cap_chunks - K : 1;
        if ((total + R - 1) / R > room) R = ((total + room - 1) / room + 63) / 64 * 64;
        int c = 0;
        for (int k = 0; k < K; ++k) {
            plan[k] = c;
            c += (int)((counts[k] + R - 1) / R);
        }
        plan[K] = c;
        plan[K + 1] = (int)R;
        return;
    }
    long long part = 0;
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        sc[k] = counts[k];
        part += sc[k];
    }
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);        // (integers: any order)
    if (threadIdx.x == 0) s_total = part;
    __syncthreads();
    const long long total = s_total;
    long long R = r_min;
    const long long room = cap_chunks - K > 0 ? cap_chunks - K : 1;      // every component may end on a partial chunk
    if ((total + R - 1) / R > room) R = ((total + room - 1) / room + 63) / 64 * 64;
    for (int k = threadIdx.x; k < K; k += blockDim.x) sc[k] = (int)((sc[k] + R - 1) / R);
    __syncthreads();
    if (threadIdx.x == 0) {
        int c = 0;
        for (int k = 0; k < K; ++k) {
            const int n = sc[k];
            sc[k] = c;
            c += n;
        }
        sc[K] = c;
        plan[K + 1] = (int)R;
    }
    __syncthreads();
    for (int k = threadIdx.x; k <= K; k += blockDim.x) plan[k] = sc[k];
}

template <int T>
__global__ __launch_bounds__(64 * mstep_waves(T, true)) void mstep_list_f64(
    const double* __restrict__ xc,         // [npad + 64][16 T] centred rows
    const double* __restrict__ lnrho,      // [K][npad]
    const double* __restrict__ lse,        // [npad]
    const int* __restrict__ lists,         // [K][cap] active rows, ascending
    int64_t cap,
    const int* __restrict__ counts,        // [K] list lengths
    const int* __restrict__ plan,          // [K + 2] chunk plan
    int64_t npad, int K,
    double* __restrict__ slabs /*[chunks][slab_len(T)]*/, int direct_r /*0, or 3: delta lists*/) {
    constexpr int WS = mstep_ws(T);
    constexpr int KPW = mstep_waves(T, true) / WS;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = (int)blockIdx.x * KPW + wave / WS;
    if (c >= plan[K]) return;
    int k = 0;
    {
        int lo = 0, hi = K;                      // largest k with plan[k] <= c (empty components share their successor's start)
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (plan[mid] <= c) lo = mid;
            else hi = mid;
        }
        k = lo;
    }
    const int R = plan[K + 1];
    const int sub = wave % WS;
    const int64_t lo = (int64_t)(c - plan[k]) * R;
    int64_t hi = lo + R;
    if (hi > counts[k]) hi = counts[k];
    const double* lr = lnrho + (int64_t)k * npad;
    const int* list = lists + (int64_t)k * cap;
    double* out = slabs + (int64_t)c * slab_len(T);
    if constexpr (WS == 1) {
        mstep_body<T, 1, 0, double, true, true, true>(xc, 16 * T, npad, 16 * T, nullptr, lr, lse, nullptr, lo, hi, direct_r, out, list);
    } else {
        if (sub == 0)
            mstep_body<T, 2, 0, double, true, true, true>(xc, 16 * T, npad, 16 * T, nullptr, lr, lse, nullptr, lo, hi, direct_r, out, list);
        else
            mstep_body<T, 2, 1, double, true, true, true>(xc, 16 * T, npad, 16 * T, nullptr, lr, lse, nullptr, lo, hi, direct_r, out, list);
    }
}

// The same over the caller's f32 rows (or the workspace's regrouped f32 copy) instead of the centred f64 copy: with
// one or two active components per row every listed row is read about once, so the kernel's HBM traffic is the rows
// themselves - 4 D bytes instead of 8 D - and the conversion and pivot subtraction (the very operations that made the
// centred copy: identical values) ride in the shadow of the MFMAs.
template <int T>
__global__ __launch_bounds__(64 * mstep_waves(T, true)) void mstep_list_x32_f64(
    const float* __restrict__ x, int64_t ldx, int64_t n_rows, int D, const double* __restrict__ pivot,
    const double* __restrict__ lnrho, const double* __restrict__ lse, const int* __restrict__ lists, int64_t cap,
    const int* __restrict__ counts, const int* __restrict__ plan, int64_t npad, int K, double* __restrict__ slabs,
    int direct_r /*0, or 3: delta lists*/) {
    constexpr int WS = mstep_ws(T);
    constexpr int KPW = mstep_waves(T, true) / WS;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = (int)blockIdx.x * KPW + wave / WS;
    if (c >= plan[K]) return;
    int k = 0;
    {
        int lo = 0, hi = K;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (plan[mid] <= c) lo = mid;
            else hi = mid;
        }
        k = lo;
    }
    const int R = plan[K + 1];
    const int sub = wave % WS;
    const int64_t lo = (int64_t)(c - plan[k]) * R;
    int64_t hi = lo + R;
    if (hi > counts[k]) hi = counts[k];
    const double* lr = lnrho + (int64_t)k * npad;
    const int* list = lists + (int64_t)k * cap;
    double* out = slabs + (int64_t)c * slab_len(T);
    if constexpr (WS == 1) {
        mstep_body<T, 1, 0, float, true, false, true, true>(x, ldx, n_rows, D, pivot, lr, lse, nullptr, lo, hi, direct_r, out, list);
    } else {
        if (sub == 0)
            mstep_body<T, 2, 0, float, true, false, true, true>(x, ldx, n_rows, D, pivot, lr, lse, nullptr, lo, hi, direct_r, out, list);
        else
            mstep_body<T, 2, 1, float, true, false, true, true>(x, ldx, n_rows, D, pivot, lr, lse, nullptr, lo, hi, direct_r, out, list);
    }
}

}  // namespace gmmvb
