// Per-row candidate records and per-pair bounds of the pruned E-step.
//
// Once the responsibilities are sparse (a handful of the K components matter per sample), the E-step only has to
// PROVE the other pairs irrelevant (r_nk < 2^-80, invisible in every f64 sum of the reference's _update_q_z /
// _calc_n_x_bar_s, bayesml/gaussianmixture/_gaussianmixture.py:772-784, 725-732).  What is carried from one E-step to
// the next is one f32 upper bound of ln rho per PAIR (ub, [K][npad], every entry rounded up).  A parameter update
// (m, U) -> (m', U') with
//   gamma_k <= sigma_min(U'_k U_k^-1),  Gamma_k >= sigma_max(U'_k U_k^-1),  delta_k >= || U'_k (m'_k - m_k) ||
// (gmmvb_set_drift) turns a bound u under the old parameters into one under the new without touching x:
//   d = sqrt(2 (c_k - u)_+) <= || U_k (x - m_k) ||,   u' = c'_k - (gamma_k d - delta_k)_+^2 / 2      (rec_sweep_kernel),
// and a distance d of an exactly known pair into a LOWER bound of its new value, c'_k - (Gamma_k d + delta_k)^2 / 2.
// With thr = (the row's reference value) - 80 ln 2:
//   u' < thr    -> the pair is irrelevant this pass, nothing to compute;
//   u' >= thr   -> the pair is a candidate: listed, evaluated exactly (f64 MFMA, estep_gather_dev_f64) - or, for a settled
//                  row, bounded from both sides on the int8 pipe first (estep_i8_proof, rec_proof_decide_kernel).
// The row's reference value is its best exact value of this pass (pairs that were active in the previous pass are
// evaluated before the sweep), or for a settled row a lower bound of its one component's ln rho.
//
// The per-row RECORD is the by-product every pass leaves for its own tail and for the read-outs (55 bytes per row):
//   up to C = 8 slots (k_j, d_j):   d_j <= || U_k (x_n - m_k) ||  (for slots evaluated exactly in this pass the distance
//                                   itself, flagged "exact"), the components of largest bound;
//   one rest bound B:               ln rho_nk <= B  for EVERY component k that has no slot;
//   flags:  1 overflow row (no usable reference, or more than 24 candidates without a slot: all K pairs are evaluated and
//           the record rebuilt from them), 2 refreshed row (candidates without a slot were listed too), 4 settled row that
//           stays settled (nothing evaluated; the lazy sweep writes no record for it), 8 settled row in the proof round,
//           16 candidates of a row with an exact reference in the proof round (GMMVB_PROOF=all).
// rec_build_kernel / rec_select_kernel make records from a whole row of bounds (bound pass), rec_sweep_kernel while it
// carries the bounds, rec_finish_kernel refreshes them from the exact values of the pass.
// Nothing here needs the host: lists, gather grid and statistics are sized on the device, so an E-step is a fixed
// sequence of launches without a synchronisation.  The dense [K][npad] ln rho array stays the exchange buffer for
// exact values (gather kernel -> rec_finish_kernel / M-step / read-outs) and for the proof kernel's lower bounds; only
// listed entries of it are touched.
//
// rec_finish_kernel keeps three more things per row (workspace.h):
//   rthr    the relevance threshold of the pass (best exact value - 80 ln 2): the candidate gather may stop a pair whose
//           partial sum already lies below it, and a stored value below it is treated as a bound, never flagged exact;
//   lock / lcomp   whether the row's addend sits in the M-step's cache of single-component rows (r = 1.0 exactly), and for
//           which component; the pass's changes leave through the delta masks (dmask), the M-step's own lists are mmask;
//   dlock   for a settled row (cached AND left out of the E-step's lists) the upper bound of its distance to that
//           component, from which the next sweep takes the row's reference value instead of an exact evaluation.
#pragma once
#include <type_traits>
#include "aux_kernels.h"

namespace gmmvb {

// Sorted insertion of (distance cd, component ck, value cv) into the ascending list of the C nearest components;
// whatever falls off its end belongs to the rest, whose largest value is kept in `rest` (NaN sticks) - unless it is
// marked kRecListed (it is being evaluated exactly in this pass and will compete for a slot again in rec_finish_kernel).
// Component codes: bits 0..13 the component, kRecListed, kRecExactBit (value is exact).
__device__ __forceinline__ void rec_insert(float (&ds)[kRecSlots], unsigned short (&ks)[kRecSlots], float (&vs)[kRecSlots],
                                           float& rest, float cd, unsigned short ck, float cv) {
    if (!(cd < ds[kRecSlots - 1])) {          // not among the C nearest (the usual case: most components are far away)
        if (ck != kRecEmpty && !(ck & kRecListed)) rest = (cv > rest || cv != cv) ? cv : rest;
        return;
    }
#pragma unroll
    for (int j = 0; j < kRecSlots; ++j) {
        const bool lt = cd < ds[j];
        const float td = lt ? ds[j] : cd, tv = lt ? vs[j] : cv;
        const unsigned short tk = lt ? ks[j] : ck;
        ds[j] = lt ? cd : ds[j];
        vs[j] = lt ? cv : vs[j];
        ks[j] = lt ? ck : ks[j];
        cd = td;
        cv = tv;
        ck = tk;
    }
    if (ck != kRecEmpty && !(ck & kRecListed)) rest = (cv > rest || cv != cv) ? cv : rest;
}

// From a dense ln rho row (exact values and / or upper bounds under the parameters in force) to a record:
// the C components of smallest distance get slots, B = the largest value among the others.  GIVEN: khat[n] is the one
// pair known to be exact (the bound pass evaluated it), it gets a slot and the exact flag; otherwise every value is
// exact (dense pass), all slots are flagged and khat is not read.
template <bool GIVEN>
__global__ __launch_bounds__(256) void rec_build_kernel(const double* __restrict__ lnrho, int64_t npad, int64_t n_rows,
                                                        int K, const double* __restrict__ cvec,
                                                        const int* __restrict__ khat, RecArrays rec,
                                                        float* __restrict__ ub32 /*[K][npad]: every value rounded up*/) {
    const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (n >= n_rows) return;
    float ds[kRecSlots], vs[kRecSlots];
    unsigned short ks[kRecSlots];
#pragma unroll
    for (int j = 0; j < kRecSlots; ++j) {
        ds[j] = __builtin_huge_valf();
        vs[j] = -__builtin_huge_valf();
        ks[j] = kRecEmpty;
    }
    float rest = -__builtin_huge_valf();
    const int kb = GIVEN ? khat[n] : -1;
    for (int k0 = 0; k0 < K; k0 += 8) {            // eight loads in flight per thread
        // GIVEN: the bound pass has left its bounds in ub32 (f32, rounded up) - only khat's exact value is in lnrho;
        // otherwise every entry of lnrho is an exact value (dense pass) and ub32 is made from it here
        double pre[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if (k0 + q >= K) pre[q] = 0.0;
            else if (GIVEN) pre[q] = (double)ub32[(int64_t)(k0 + q) * npad + n];
            else pre[q] = lnrho[(int64_t)(k0 + q) * npad + n];
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int k = k0 + q;
            if (k >= K) break;
            const double v = pre[q];
            if (!GIVEN) ub32[(int64_t)k * npad + n] = f32_up(v);
            if (GIVEN && k == kb) continue;
            rec_insert(ds, ks, vs, rest, f32_down(dist_of(cvec[k], v)), (unsigned short)k, f32_up(v));
        }
    }
    unsigned ex = 0;
    if (GIVEN) {               // the exact pair takes the last slot; what was there joins the rest
        const int at = kRecSlots - 1;
        if (ks[at] != kRecEmpty) rest = (vs[at] > rest || vs[at] != vs[at]) ? vs[at] : rest;
        ks[at] = (unsigned short)kb;
        const double vb = lnrho[(int64_t)kb * npad + n];
        ds[at] = f32_down(dist_of(cvec[kb], vb));
        ub32[(int64_t)kb * npad + n] = f32_up(vb);
        ex = 1u << at;
    } else {
#pragma unroll
        for (int j = 0; j < kRecSlots; ++j) ex |= (ks[j] != kRecEmpty) ? (1u << j) : 0u;
    }
#pragma unroll
    for (int j = 0; j < kRecSlots; ++j) {
        rec.k[(int64_t)j * rec.npad + n] = ks[j];
        rec.d[(int64_t)j * rec.npad + n] = ds[j];
    }
    rec.B[n] = rest;
    rec.exact[n] = (unsigned char)ex;
    rec.sel[n] = 0;
    rec.flags[n] = 0;
}

// After a bound pass: select the candidates from the records rec_build_kernel<true> has just made (no parameter update in
// between: every slot's bound is what the bound pass computed, the one exact slot - the row's best component - gives the
// threshold best - 80 ln 2).  Slots whose bound clears the threshold are done with; the others are listed, and so is
// every component WITHOUT a slot if the rest bound B does not clear it ("refreshed row": B becomes the largest bound among
// the components not listed; more than 24 such components, or no exact slot: all K pairs are evaluated, "overflow row").
// Outputs: masks (candidate components per row), per-block counts for scan_counts / fill_lists, epart[block] = listed
// pairs, opart[block] = overflow rows.
static __global__ __launch_bounds__(kSelRows) void rec_select_kernel(RecArrays rec, int64_t n_rows, int K,
                                                              const double* __restrict__ cvec,
                                                              unsigned long long* __restrict__ masks, int64_t npad,
                                                              int* __restrict__ blk_cnt, double* __restrict__ epart,
                                                              double* __restrict__ opart,
                                                              float* __restrict__ rthr /*[npad] the row's relevance threshold*/) {
    __shared__ int wcnt[4][256];
    __shared__ double sc[256];
    __shared__ int wsum[2][4];
    const int tid = threadIdx.x, wave = tid >> 6;
    const int W = (K + 63) / 64;
    for (int k = tid & 63; k < K; k += 64) wcnt[wave][k] = 0;
    for (int k = tid; k < K; k += kSelRows) sc[k] = cvec[k];
    __syncthreads();
    const int64_t n = (int64_t)blockIdx.x * kSelRows + tid;
    const bool valid = n < n_rows;
    unsigned long long mk[4] = {0ull, 0ull, 0ull, 0ull};
    int listed = 0, over_i = 0;
    if (valid) {
        const unsigned ex = rec.exact[n];
        const double ninf = -__builtin_huge_val();
        double lb = ninf;
        double ub[kRecSlots];
        unsigned short kk[kRecSlots];
#pragma unroll
        for (int j = 0; j < kRecSlots; ++j) {
            const unsigned short k = rec.k[(int64_t)j * rec.npad + n];
            kk[j] = k;
            ub[j] = ninf;
            if (k == kRecEmpty) continue;
            const double d = (double)rec.d[(int64_t)j * rec.npad + n];
            const float yf = f32_down(d * (1.0 - 1e-12));                // NaN -> the trivial bound c
            const double c = sc[k];
            ub[j] = c - 0.5 * (double)yf * (double)yf * (1.0 - 1e-12) + 1e-12 * fabs(c);
            if ((ex >> j) & 1u) {
                const double du = d * (1.0 + 2.4e-7) * (1.0 + 1e-12);
                const double l = c - 0.5 * du * du * (1.0 + 1e-12) - 1e-12 * fabs(c);
                lb = l > lb ? l : lb;                                // NaN never raises the threshold
            }
        }
        const double thr = lb - kRelNats;
        unsigned sel = 0;
#pragma unroll
        for (int j = 0; j < kRecSlots; ++j) {
            if (kk[j] == kRecEmpty || ((ex >> j) & 1u)) continue;        // (exact slots are evaluated already)
            if (!(ub[j] < thr)) {
                sel |= 1u << j;
                mk[kk[j] >> 6] |= 1ull << (kk[j] & 63);
                ++listed;
            }
        }
        // every component without a slot is bounded by B: if B does not clear the threshold they are all listed
        // (B = -inf afterwards: nothing is left without a slot or a listing)
        const double B = (double)rec.B[n];
        bool over = !(lb > ninf);                                    // no exact slot: nothing to compare with
        int extra = 0;
        if (!over && (B > ninf || B != B) && !(B < thr)) {
            unsigned long long inslot[4] = {0ull, 0ull, 0ull, 0ull};
#pragma unroll
            for (int j = 0; j < kRecSlots; ++j)
                if (kk[j] != kRecEmpty) inslot[kk[j] >> 6] |= 1ull << (kk[j] & 63);
            for (int k = 0; k < K; ++k) {
                if ((inslot[k >> 6] >> (k & 63)) & 1ull) continue;
                mk[k >> 6] |= 1ull << (k & 63);
                ++extra;
            }
            rec.B[n] = -__builtin_huge_valf();
            over = extra > 24;
        }
        listed += extra;
        if (over) {
            for (int w = 0; w < W; ++w) mk[w] = (K - 64 * w >= 64) ? ~0ull : ((1ull << (K - 64 * w)) - 1ull);
            listed = K;
            over_i = 1;
            sel = 0;
        }
        rec.sel[n] = (unsigned char)sel;
        rec.flags[n] = (unsigned char)(over ? 1 : (extra > 0 ? 2 : 0));
        rthr[n] = (over || !(thr > ninf)) ? -__builtin_huge_valf() : f32_down(thr);
        for (int w = 0; w < W; ++w) masks[(int64_t)w * npad + n] = mk[w];
    }
    for (int w = 0; w < W; ++w) count_word(mk[w], w, wave, wcnt);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        listed += __shfl_xor(listed, o);
        over_i += __shfl_xor(over_i, o);
    }
    if ((tid & 63) == 0) {
        wsum[0][wave] = listed;
        wsum[1][wave] = over_i;
    }
    __syncthreads();
    for (int k = tid; k < K; k += kSelRows)
        blk_cnt[blk_at(k, blockIdx.x, K)] = wcnt[0][k] + wcnt[1][k] + wcnt[2][k] + wcnt[3][k];
    if (tid == 0) {
        epart[blockIdx.x] = (double)(wsum[0][0] + wsum[0][1] + wsum[0][2] + wsum[0][3]);
        opart[blockIdx.x] = (double)(wsum[1][0] + wsum[1][1] + wsum[1][2] + wsum[1][3]);
    }
}

// The carried E-step: the f32 array ub ([K][npad], every entry rounded UP) holds, for EVERY pair, an upper bound of
// ln rho under the previous parameters.  One sweep over it
//   carries every entry over the update with its own component's (gamma, delta):  u' = c'_k - (gamma_k d - delta_k)_+^2 / 2,
//     d = sqrt(2 (c_k - u)_+), and writes it back (the array stays valid for the next sweep);
//   takes as the row's reference value v the largest of the pairs that have just been evaluated exactly under the NEW
//     parameters (this early the components still shrink by factors - Gamma = 2 .. 4 -, a lower bound of the best value
//     carried through Gamma would be hundreds of nats too low):
//       PREV   all pairs that were active in the previous pass (the M-step's lists, still in the workspace: no list
//              building for this round); `masks` holds them on entry;
//       !PREV  the row's previous best component khat[n];
//   lists every other pair with u' >= v - 80 ln 2, and builds the row's record (C slots + rest bound) on the way,
// so that the pass continues like a bound pass (gather -> rec_finish_kernel).
//
// The kernel is bound by its instruction count (K pairs per row, 8 bytes of traffic each), so the per-pair work is
// ~45 f32 / integer instructions and free of branches:
//   the carry is done in f32 with the rounding slack folded into the per-component constants (gamma (1 - 1e-6) rounded
//     down, delta and c' rounded up, the old c rounded down; 0.5 (1 - 1e-6); the result pushed up by 2.4e-7 relative) -
//     v_sqrt_f32 is good to one ulp, every other operation to half an ulp, 1e-6 covers them several times over;
//   the slots go to the C components of LARGEST carried bound (they are the ones that may matter), selected with a
//     chain of nine min / max pairs on 32-bit keys: the bound's bits in descending order (upper 24 bits) | component
//     (K <= 256).  The ninth key is the largest bound without a slot: the rest bound.  Slot distances are recovered
//     from the keys (bounds rounded up to 15 mantissa bits: 3e-5 relative); exact slots read their value again.
// A settled row's reference is a lower bound of ln rho of its one component under the new parameters.  Carried from the
// previous pass (d' = Gamma d + delta) it loses about d^2 (Gamma - 1) + d delta nats - hundreds while the components still
// move by per cents, and every nat it is too low lets more components' bounds through as candidates.  For components that
// moved by more than this the settled rows' own pair goes through the proof round BEFORE the sweep (the lower bound in
// the ln rho array, estep_i8_proof without its upper-bound store); both kernels decide with this one predicate.

// One step of the carry for a stored bound `old`, p = (gamma, delta, c' up, c the bound was stored under, down): the
// kernel's per-pair arithmetic.  Every operation is monotone in `old` (round-to-nearest is), so applied to the LARGEST
// bound of a set of pairs of one component it gives a value >= the carried bound of every pair of the set - which is what
// the lazy form of the sweep lives on.
__device__ __forceinline__ float sweep_carry(float old, float4 p) {
    const float qd = p.w - old;
    const float sq = __builtin_amdgcn_sqrtf(qd + qd);              // NaN for qd < 0 or NaN: no information
    const float t = fmaxf(fmaf(p.x, sq, -p.y), 0.0f);              // NaN -> 0: the trivial bound c'
    const float w = fminf(t * t * 0.4999995f, 3.0e38f);
    const float r = p.z - w;
    return fmaf(fabsf(r), 2.4e-7f, r);
}

// v_max_f32 as it is (IEEE maxNum like fmaxf, without the two canonicalising copies the compiler puts in front of it)
__device__ __forceinline__ float max_f32_raw(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// largest value over the wave's lanes, in every lane (NaN operands lose against numbers: callers' values are never NaN).
// Four v_max_f32 with a DPP operand inside the rows of 16 lanes, then the four row results through scalar registers.  (Written
// out: from fmaxf on a DPP copy the compiler makes a move, two canonicalising copies and the maximum per step - 40
// instructions per call, and the lazy sweep calls this once per re-opened column and wave.)
__device__ __forceinline__ float wave_max_f32(float v) {
    asm volatile(
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(v));
    const float a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
    const float b = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
    const float c = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
    const float d = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
    return max_f32_raw(max_f32_raw(a, b), max_f32_raw(c, d));
}

// LAZY (PREV only): the sweep does not touch what it can prove irrelevant for a whole tile of 256 rows (= workgroup).
// Per tile and component it keeps (tmeta, [tiles][K] float4)
//   tub   the largest stored bound of the tile's rows (+inf: unknown - entries of the column have been or will be replaced
//         by fresh values behind the sweep's back: rec_finish_kernel, estep_i8_proof),
//   gc, dc   the carry composed over the passes since the column's entries were last rewritten
//            (d'' >= g2 (g1 d - d1) - d2 = g1 g2 d - (g2 d1 + d2)),
//   ce    the constant c_k those entries were stored under.
// sweep_carry(tub, composed) bounds the carried value of every pair of the column; if it lies below the LOWEST threshold of
// the tile's rows, no row has a candidate there and neither the column's entries nor the composition change hands: the
// entries stay as they are (bounds under the parameters of the pass that wrote them), gc / dc take the step in.  A column
// is opened - its entries carried with the composed step and rewritten under the new parameters - if that test fails, if
// a row of the tile holds an exact value there (fresh pairs, a settled row's own component: later kernels write fresh
// values into such entries) or if tub is unknown.  The carry being monotone, the largest entry of an opened column in
// which nothing is a candidate is sweep_carry(tub) itself; a column with candidates is marked unknown (the candidates'
// entries are about to be replaced by tighter values) and its true maximum is taken the next time it is opened.  With the
// rows grouped by dominant component a tile's rows share their near components, and the far ones - most of the K - are
// never read: the sweep was 5 GB of traffic per pass at the benchmark shape, the largest kernel of a converged step.
// WC > 0: the number of 64-component mask words is a compile-time constant (the loops over them unroll and the
// four-element mask arrays stay in registers without select chains: a third fewer vector instructions at K <= 64).
// (Three and four mask words: five waves per SIMD asked for - 96 registers and 24 bytes of scratch outside the column loop
// instead of 108: the kernel lives on the other waves hiding its per-pair LDS and memory waits; six - 80 registers, 144
// bytes of scratch - loses again.  3.79 -> 3.35 ms at K = 256.)
template <bool PREV, bool LAZY = false, int WC = 0>
__global__ __launch_bounds__(kSelRows, (LAZY && WC >= 3 ? 5 : 1)) void rec_sweep_kernel(float* __restrict__ ub, const double* __restrict__ u, int64_t npad,
                                                             int64_t n_rows, int K,
                                                             const double* __restrict__ drift,
                                                             const double* __restrict__ c_new,
                                                             const int* __restrict__ khat, RecArrays rec,
                                                             unsigned long long* __restrict__ masks,
                                                             int* __restrict__ blk_cnt, double* __restrict__ epart,
                                                             double* __restrict__ opart,
                                                             unsigned char* __restrict__ lock /*null: no settled rows*/,
                                                             float* __restrict__ dlock,
                                                             float* __restrict__ rthr /*[npad] the row's relevance threshold*/,
                                                             const unsigned char* __restrict__ lcomp /*[npad] component a cached row is in*/,
                                                             unsigned long long* __restrict__ pmask /*null: no proof round*/,
                                                             int* __restrict__ pblk,
                                                             int proof_all /*candidates of rows with an exact reference too*/,
                                                             int own_fresh /*the ln rho array holds fresh lower bounds of the
                                                                             settled rows' own pairs (own_first components)*/,
                                                             float4* __restrict__ tmeta = nullptr /*LAZY: [tiles][K]*/,
                                                             int tmeta_reset = 0 /*LAZY: the stored tile state is void*/,
                                                             unsigned long long* __restrict__ col_ctr = nullptr /*LAZY: += columns opened*/,
                                                             double* __restrict__ ppre = nullptr /*[blocks] pairs listed for the proof
                                                                 round (fill_lists_kernel skips the blocks without any)*/) {
    static_assert(!LAZY || PREV, "the lazy sweep is a form of the PREV sweep");
    __shared__ int wcnt[4][256];
    __shared__ int pcnt[4][256];
    __shared__ float4 sp[256];          // gamma (1 - 1e-6) down, delta up, c' up, c old down  (LAZY: composed since the column was written)
    __shared__ float2 sq[256];          // Gamma (1 + 1e-6) up, c' down (settled rows)
    __shared__ double sc[256];
    __shared__ unsigned char sfirst[256];
    __shared__ float s_delta[256];      // this pass's delta, rounded up
    __shared__ int wsum[3][4];
    __shared__ unsigned char s_force[LAZY ? 256 : 1];      // a row of the tile holds an exact value in this column
    __shared__ float s_thr[LAZY ? 4 : 1];                  // per wave: lowest threshold of its rows
    __shared__ unsigned long long s_open[LAZY ? 4 : 1];    // columns the tile's rows go through
    __shared__ unsigned long long s_redo[LAZY ? 4 : 1];    // ... of which the true maximum is wanted (unknown, not forced)
    __shared__ unsigned long long s_keep[LAZY ? 4 : 1];    // ... which only the tile test keeps open: read, not rewritten
    __shared__ unsigned long long s_any[LAZY ? 4 : 1][4];  // per wave: columns in which one of its rows has a candidate
    __shared__ float s_red[LAZY ? 4 : 1][LAZY ? 256 : 1];  // per wave: largest carried bound of the redo columns
    __shared__ float s_skip[LAZY ? 4 : 1];                 // per wave (of columns): largest bound among the closed columns
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int W = WC > 0 ? WC : (K + 63) / 64;
    const int64_t n = (int64_t)blockIdx.x * kSelRows + tid;
    const bool valid = n < n_rows;
    // what the row needs from global memory and does not depend on the component constants is requested before the first
    // barrier: the kernel is a chain of dependent loads per workgroup (0.67 ms of its 1.0 ms at the benchmark shape do not
    // depend on how many columns it opens)
    unsigned long long fresh[4] = {0ull, 0ull, 0ull, 0ull};      // pairs already exact under the new parameters
    unsigned lk_pre = 0u;
    int kset_pre = 0;
    float dl_pre = 0.0f;
    double lb_pre = 0.0;                                         // (own_fresh: the settled row's fresh lower bound, if it has one)
    float4 meta_pre = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (valid && PREV) {
        for (int w = 0; w < W; ++w) fresh[w] = masks[(int64_t)w * npad + n];
        if (lock != nullptr) {
            lk_pre = lock[n];
            kset_pre = lcomp[n];
            dl_pre = dlock[n];
            if (own_fresh && lk_pre == 1u) lb_pre = u[(int64_t)kset_pre * npad + n];
        }
    }
    if constexpr (LAZY)
        if (tid < K && !tmeta_reset) meta_pre = tmeta[(int64_t)blockIdx.x * K + tid];
    for (int k = lane; k < K; k += 64) wcnt[wave][k] = pcnt[wave][k] = 0;
    float4 step = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    for (int k = tid; k < K; k += kSelRows) {
        sfirst[k] = (own_fresh && own_first(drift[3 * K + k], drift[K + k])) ? 1 : 0;
        const double g = drift[k] * (1.0 - 1e-6);
        step = make_float4(g > 0.0 ? f32_down(g) : 0.0f, f32_up(drift[K + k]), f32_up(c_new[k]), f32_down(drift[2 * K + k]));
        if constexpr (!LAZY) sp[k] = step;
        s_delta[k] = step.y;
        sq[k] = make_float2(f32_up(drift[3 * K + k] * (1.0 + 1e-6)), f32_down(c_new[k]));
        sc[k] = c_new[k];
        if constexpr (LAZY) s_force[k] = 0;
    }
    // ---- the row's reference value ------------------------------------------------------------------------------------
    const double ninf = -__builtin_huge_val();
    double vb = ninf;
    if (valid && PREV) {
        for (int w = 0; w < W; ++w) {
            unsigned long long m = fresh[w];
            while (m) {
                const int b = __builtin_ctzll(m);
                m &= m - 1;
                const double v = u[(int64_t)(64 * w + b) * npad + n];
                vb = (v > vb || v != v) ? v : vb;
            }
        }
    }
    __syncthreads();
    unsigned long long mk[4] = {0ull, 0ull, 0ull, 0ull};
    unsigned long long pm[4] = {0ull, 0ull, 0ull, 0ull};        // pairs of the proof round (a settled row with candidates)
    int listed = 0, over_i = 0;
    // ---- ... and threshold -------------------------------------------------------------------------------------------
    unsigned long long nocand[4] = {0ull, 0ull, 0ull, 0ull};
    bool by_bound = false, over = false;
    int kset = -1;
    float d_set = 0.0f, thr_f = __builtin_huge_valf();
    if (valid) {
        const int kb = PREV ? -1 : khat[n];
        if constexpr (!PREV) {
            fresh[kb >> 6] = 1ull << (kb & 63);
            vb = u[(int64_t)kb * npad + n];
        }
        const double thr = vb - kRelNats;
        // A settled row (cached - its single active component kset has r = 1.0 exactly and its addend sits in the
        // statistics cache - and left out of the lists by rec_finish_kernel) has nothing evaluated for it: the reference
        // is a LOWER bound of ln rho under the new parameters, from the carried upper bound of its distance,
        // d' = Gamma d + delta.  No candidate against it: the row stays settled.
        by_bound = PREV && lock != nullptr && lk_pre == 1u && (fresh[0] | fresh[1] | fresh[2] | fresh[3]) == 0ull;
        float thr_set = 0.0f;
        if (by_bound) {
            kset = kset_pre;
            // the distance bound: from the lower bound the proof round has just made for the new parameters (components
            // that moved), or the previous pass's carried through Gamma and delta
            float dn;
            if (sfirst[kset]) {
                const double lbn = lb_pre;
                dn = f32_up(dist_of(sc[kset], lbn) * (1.0 + 1e-9));            // (-inf: +inf, every component a candidate)
            } else {
                dn = fmaf(sq[kset].x, dl_pre, s_delta[kset]) * (1.0f + 2.4e-7f);
            }
            const float lb = sq[kset].y - dn * dn * 0.5000005f;
            d_set = dn;
            thr_set = (lb - fabsf(lb) * 2.4e-7f) - ((float)kRelevanceNats + 0.2f);
        }
        over = !by_bound && !(thr > ninf);          // NaN / -inf: nothing to compare with
        thr_f = by_bound ? thr_set : (over ? -__builtin_huge_valf() : f32_down(thr));   // over: every pair is a candidate
        if (over) fresh[0] = fresh[1] = fresh[2] = fresh[3] = 0ull;
        rthr[n] = thr_f;                                       // (-inf: everything is evaluated in full)
#pragma unroll
        for (int w = 0; w < 4; ++w) nocand[w] = fresh[w];
        if (by_bound) nocand[kset >> 6] |= 1ull << (kset & 63);
    }
    // ---- LAZY: which columns the tile has to open --------------------------------------------------------------------
    float4 meta = make_float4(0.0f, 0.0f, 0.0f, 0.0f), comp = meta;
    bool col_open = false, col_forced = false, col_redo = false, col_keep = false;
    float col_bound = 0.0f;
    if constexpr (LAZY) {
        if (valid) {
            for (int w = 0; w < W; ++w) {
                unsigned long long m = nocand[w];
                while (m) {
                    const int b = __builtin_ctzll(m);
                    m &= m - 1;
                    s_force[64 * w + b] = 1;
                }
            }
        }
        float tmin = thr_f == thr_f ? thr_f : -__builtin_huge_valf();       // (invalid rows: +inf)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) tmin = fminf(tmin, __shfl_xor(tmin, o));
        if (lane == 0) s_thr[wave] = tmin;
        __syncthreads();
        const float tile_thr = fminf(fminf(s_thr[0], s_thr[1]), fminf(s_thr[2], s_thr[3]));
        float skip = -__builtin_huge_valf();
        if (tid < K) {
            const int k = tid;
            meta = tmeta_reset ? make_float4(__builtin_huge_valf(), 1.0f, 0.0f, step.w) : meta_pre;
            // the composition takes this pass's step in
            // (products of two floats are exact in f64)
            comp = make_float4(f32_down((double)meta.y * (double)step.x),
                               f32_up(((double)step.x * (double)meta.z + (double)step.y) * (1.0 + 1e-15)), step.z, meta.w);
            col_bound = sweep_carry(meta.x, comp);
            col_forced = s_force[k] != 0;
            const bool unknown = !(meta.x < __builtin_huge_valf());
            col_open = col_forced || unknown || !(col_bound < tile_thr);
            col_redo = col_open && unknown && !col_forced;
            // opened by the tile test alone: its entries are carried on the fly and stay as they are (bounds under the
            // parameters of the pass that wrote them, like a closed column's) unless one of them turns out a candidate -
            // half of the sweep's traffic at K = 256 were such rewrites
            col_keep = col_open && !unknown && !col_forced;
            sp[k] = comp;
            if (!col_open) {
                tmeta[(int64_t)blockIdx.x * K + k] = make_float4(meta.x, comp.x, comp.y, meta.w);
                skip = col_bound;
            }
        }
        const unsigned long long ob = __builtin_amdgcn_ballot_w64(col_open), rb = __builtin_amdgcn_ballot_w64(col_redo);
        const unsigned long long kb2 = __builtin_amdgcn_ballot_w64(col_keep);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) skip = fmaxf(skip, __shfl_xor(skip, o));
        if (lane == 0) {
            s_open[wave] = ob;
            s_redo[wave] = rb;
            s_keep[wave] = kb2;
            s_skip[wave] = skip;
            if (col_ctr && ob) atomicAdd(col_ctr, (unsigned long long)__builtin_popcountll(ob));
        }
        __syncthreads();
    }
    // ---- the sweep over the row's bounds ---------------------------------------------------------------------------------
    unsigned long long anyw[4] = {0ull, 0ull, 0ull, 0ull};       // LAZY: columns with a candidate among the wave's rows
    float mine[4] = {-__builtin_huge_valf(), -__builtin_huge_valf(), -__builtin_huge_valf(), -__builtin_huge_valf()};
    bool stays = false;
    {
        unsigned s[kRecSlots + 1];
#pragma unroll
        for (int j = 0; j <= kRecSlots; ++j) s[j] = 0xFFFFFFFFu;
        float restmax = -__builtin_huge_valf();                // largest bound among the pairs that are not listed
        // uniform base + 32-bit row offset: one address register for the whole loop
        const bool near = npad < (int64_t(1) << 29);
        const unsigned off = (unsigned)(near ? (valid ? n : 0) : 0) * 4u;
        const int64_t nn = valid ? n : 0;
        unsigned long long red_w = 0ull, keep_w = 0ull;
        float mine_w = -__builtin_huge_valf();
        // LAZY: a settled row's record has no slots (if the row comes loose, its candidates are listed and the rest bound
        // covers everything else: "refreshed row"), so a wave whose rows are all settled skips the selection chain - a third
        // of the per-pair instructions
        const bool need_chain = !LAZY || __builtin_amdgcn_ballot_w64(valid && !by_bound) != 0ull;
        // (CH: the slot-selection chain runs - a compile-time flag of the column loop, which exists twice: with the test inside
        // the loop the chain's nine registers met the skipping path at a join behind every pair and were copied there, eight
        // moves per pair)
        auto pair = [&](auto ch, int k, int bit, float old, unsigned long long fw, unsigned long long nw, unsigned long long& mw,
                        unsigned long long& aw) {
            constexpr bool CH = decltype(ch)::value;
            const float ubn = sweep_carry(old, sp[k]);
            char* base = (char*)(ub + (int64_t)k * npad + (near ? 0 : nn));
            const unsigned long long b1 = 1ull << bit;
            if (valid && !(LAZY && (keep_w & b1) != 0ull)) *(float*)(base + off) = ubn;
            const bool isf = (fw & b1) != 0ull;                            // (its exact value is written below)
            const bool cand = valid && !(ubn < thr_f) && (nw & b1) == 0ull;
            mw |= cand ? b1 : 0ull;
            restmax = max_f32_raw(restmax, (cand || isf) ? -__builtin_huge_valf() : ubn);
            if constexpr (CH) sweep_chain(s, (isf || !valid) ? 0xFFFFFFFFu : sweep_key(ubn, (unsigned)k));
            if constexpr (LAZY) {
                aw |= __builtin_amdgcn_ballot_w64(cand) != 0ull ? b1 : 0ull;
                if (red_w & b1) {                                              // (uniform)
                    const float m = wave_max_f32(valid ? ubn : -__builtin_huge_valf());
                    mine_w = lane == bit ? m : mine_w;
                }
            }
        };
        auto old_of = [&](int k) {
            const char* base = (const char*)(ub + (int64_t)k * npad + (near ? 0 : nn));
            return *(const float*)(base + off);
        };
        for (int w = 0; w < W; ++w) {
            const unsigned long long fw = w == 0 ? fresh[0] : (w == 1 ? fresh[1] : (w == 2 ? fresh[2] : fresh[3]));
            const unsigned long long nw = w == 0 ? nocand[0] : (w == 1 ? nocand[1] : (w == 2 ? nocand[2] : nocand[3]));
            unsigned long long mw = 0ull, aw = 0ull;
            if constexpr (LAZY) {
                const unsigned long long o0 = s_open[w];
                unsigned long long ow = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(o0 >> 32)) << 32) |
                                        (unsigned)__builtin_amdgcn_readfirstlane((int)o0);
                const unsigned long long r0 = s_redo[w];
                red_w = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(r0 >> 32)) << 32) |
                        (unsigned)__builtin_amdgcn_readfirstlane((int)r0);
                const unsigned long long q0 = s_keep[w];
                keep_w = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(q0 >> 32)) << 32) |
                         (unsigned)__builtin_amdgcn_readfirstlane((int)q0);
                mine_w = -__builtin_huge_valf();
                auto columns = [&](auto ch) {
                    while (ow) {
                        int kq[8];
#pragma unroll
                        for (int q = 0; q < 8; ++q) {
                            kq[q] = ow ? __builtin_ctzll(ow) : -1;
                            ow &= ow - 1;                                   // (0 stays 0)
                        }
                        float pre[8];                                       // up to eight loads in flight per thread
#pragma unroll
                        for (int q = 0; q < 8; ++q)
                            if (kq[q] >= 0) pre[q] = old_of(64 * w + kq[q]);
#pragma unroll
                        for (int q = 0; q < 8; ++q)
                            if (kq[q] >= 0) pair(ch, 64 * w + kq[q], kq[q], pre[q], fw, nw, mw, aw);
                    }
                };
                if (need_chain) columns(std::true_type{});
                else columns(std::false_type{});
                if (w == 0) mine[0] = mine_w;
                else if (w == 1) mine[1] = mine_w;
                else if (w == 2) mine[2] = mine_w;
                else mine[3] = mine_w;
            } else {
                if (valid) {
                    const int kend = K < 64 * w + 64 ? K : 64 * w + 64;
                    int k0 = 64 * w;
                    for (; k0 + 8 <= kend; k0 += 8) {
                        float pre[8];                                  // eight loads in flight per thread
#pragma unroll
                        for (int q = 0; q < 8; ++q) pre[q] = old_of(k0 + q);
#pragma unroll
                        for (int q = 0; q < 8; ++q) pair(std::true_type{}, k0 + q, k0 + q - 64 * w, pre[q], fw, nw, mw, aw);
                    }
                    for (; k0 < kend; ++k0) pair(std::true_type{}, k0, k0 - 64 * w, old_of(k0), fw, nw, mw, aw);
                }
            }
            if (w == 0) { mk[0] = mw; anyw[0] = aw; }
            else if (w == 1) { mk[1] = mw; anyw[1] = aw; }
            else if (w == 2) { mk[2] = mw; anyw[2] = aw; }
            else { mk[3] = mw; anyw[3] = aw; }
        }
        if (valid) {
        bool proof_row = false, proof_cand = false;
        if (pmask != nullptr && proof_all && !by_bound && !over && (mk[0] | mk[1] | mk[2] | mk[3]) != 0ull) {
            // a row with an exact reference: its candidates (carried bounds that no longer clear the threshold) get fresh
            // int8 bounds first; only those that still do not clear it are evaluated exactly (rec_proof_decide_kernel)
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                pm[w] = mk[w];
                mk[w] = 0ull;
            }
            proof_cand = true;
        }
        if (by_bound) {
            stays = (mk[0] | mk[1] | mk[2] | mk[3]) == 0ull;
            if (stays) {
                dlock[n] = d_set;
            } else if (pmask != nullptr) {
                // Proof round: the row's component and the candidates get fresh two-sided bounds from three int8 digits
                // (estep_i8_proof) before anything is evaluated in f64; rec_proof_decide_kernel then either keeps the row
                // settled or hands it - with the candidates that survived - to the exact gather.  Until then the row has
                // no pair in the pass's lists, and its record holds carried bounds only (no slot selected).
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    pm[w] = mk[w];
                    mk[w] = 0ull;
                }
                if (!sfirst[kset]) pm[kset >> 6] |= 1ull << (kset & 63);      // (else its lower bound is fresh already)
                proof_row = true;
                // (the new distance bound, should the table of project.h clear every candidate before the proof round:
                // proj_filter_kernel then keeps the row settled; rec_proof_decide_kernel overwrites it otherwise)
                dlock[n] = d_set;
            } else {
                mk[kset >> 6] |= 1ull << (kset & 63);          // loose: its component and the candidates are evaluated -
                // in full: a candidate that left the gather early would keep a loose bound, erode back to the threshold
                // within a pass or two and bring the row loose again; its exact value keeps the row settled for longer
                rthr[n] = -__builtin_huge_valf();
            }
        }
        unsigned sel = 0, ex = 0;
        int in_slots = 0;
        // (LAZY: a row that stays settled has no exact pair, no candidate and needs no record - nothing reads it before the
        // next pass that rebuilds it, and the read-outs answer for such rows from the log-normaliser alone,
        // rec_readout_kernel: the whole section is skipped, by whole waves where the rows are grouped)
        if (!(LAZY && stays)) {
        if (LAZY && by_bound) {              // (whatever the wave's other rows made the chain collect)
#pragma unroll
            for (int j = 0; j <= kRecSlots; ++j) s[j] = 0xFFFFFFFFu;
        }
        // the exact pairs: their values replace the carried bounds, and they compete for slots by value (the single
        // reference pair of the !PREV form always gets one)
        for (int w = 0; w < W; ++w) {
            unsigned long long m = fresh[w];
            while (m) {
                const int b = __builtin_ctzll(m);
                m &= m - 1;
                const int k = 64 * w + b;
                const float fu = f32_up(u[(int64_t)k * npad + n]);
                ub[(int64_t)k * npad + n] = fu;
                sweep_chain(s, PREV ? sweep_key(fu, (unsigned)k) : (unsigned)k);
            }
        }
        float ds[kRecSlots];
        unsigned short ks[kRecSlots];
        unsigned long long in_slot[4] = {0ull, 0ull, 0ull, 0ull};
#pragma unroll
        for (int j = 0; j < kRecSlots; ++j) {
            ks[j] = kRecEmpty;
            ds[j] = __builtin_huge_valf();
            if (s[j] == 0xFFFFFFFFu) continue;
            const int k = (int)(s[j] & 0xFFu);
            ks[j] = (unsigned short)k;
            const unsigned long long bit = 1ull << (k & 63);
            const int kw = WC == 1 ? 0 : (k >> 6);            // (one mask word: no select chain over the four)
            if (fresh[kw] & bit) {                         // exact, in a slot: not a candidate
                in_slot[kw] |= bit;
                ex |= 1u << j;
                ds[j] = f32_down(dist_of(sc[k], PREV ? u[(int64_t)k * npad + n] : vb));
            } else {
                ds[j] = dist_lower_f32(sc[k], (double)sweep_key_bound(s[j]));
                if (mk[kw] & bit) {
                    sel |= 1u << j;
                    ++in_slots;
                }
            }
        }
        // exact pairs that did not get a slot are listed again (rec_finish_kernel finds the evaluated pairs of a row
        // through its slots and its candidate mask)
        for (int w = 0; w < W; ++w) {
            mk[w] |= fresh[w] & ~in_slot[w];
            listed += __builtin_popcountll(mk[w]);
        }
        // rest bound: the ninth key bounds every pair without a slot; if that pair is itself listed (nine or more
        // candidates: a refreshed row), the largest bound among the pairs that are not listed is the tighter one
        float rest = -__builtin_huge_valf();
        if (s[kRecSlots] != 0xFFFFFFFFu) {
            const int k9 = (int)(s[kRecSlots] & 0xFFu);
            rest = ((mk[k9 >> 6] >> (k9 & 63)) & 1ull) ? restmax : sweep_key_bound(s[kRecSlots]);
        }
        if constexpr (LAZY) {    // (the columns the tile left closed: every pair in them lies below this)
            if (by_bound) rest = restmax;
            rest = fmaxf(rest, fmaxf(fmaxf(s_skip[0], s_skip[1]), fmaxf(s_skip[2], s_skip[3])));
        }
        {
#pragma unroll
            for (int j = 0; j < kRecSlots; ++j) {
                rec.k[(int64_t)j * rec.npad + n] = ks[j];
                rec.d[(int64_t)j * rec.npad + n] = ds[j];
            }
            rec.B[n] = rest;
            rec.exact[n] = (unsigned char)(over ? 0 : ex);
            rec.sel[n] = (unsigned char)(over ? 0 : sel);
        }
        }
        rec.flags[n] = (unsigned char)(stays ? 4 : (proof_row ? 8 : (over ? 1 : ((listed > in_slots ? 2 : 0) | (proof_cand ? 16 : 0)))));
        over_i = over ? 1 : 0;
        for (int w = 0; w < W; ++w) {
            masks[(int64_t)w * npad + n] = mk[w];
            if (pmask) pmask[(int64_t)w * npad + n] = pm[w];
        }
        }
    }
    int plisted = 0;
    for (int w = 0; w < W; ++w) {
        count_word(mk[w], w, wave, wcnt);
        if (pmask) count_word(pm[w], w, wave, pcnt);
        plisted += __builtin_popcountll(pm[w]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        listed += __shfl_xor(listed, o);
        over_i += __shfl_xor(over_i, o);
        plisted += __shfl_xor(plisted, o);
    }
    if (lane == 0) {
        wsum[0][wave] = listed;
        wsum[1][wave] = over_i;
        wsum[2][wave] = plisted;
    }
    if constexpr (LAZY) {
        if (lane == 0)
            for (int w = 0; w < 4; ++w) s_any[wave][w] = anyw[w];
        for (int w = 0; w < W; ++w) s_red[wave][64 * w + lane] = w == 0 ? mine[0] : (w == 1 ? mine[1] : (w == 2 ? mine[2] : mine[3]));
    }
    __syncthreads();
    if constexpr (LAZY) {
        // a kept column in which a row did turn out a candidate: its entries are about to be replaced by fresh values under
        // the new parameters, so the whole column goes over to them now (the same carry once more, this time stored)
        for (int w = 0; w < W; ++w) {
            unsigned long long fix = s_keep[w] & (s_any[0][w] | s_any[1][w] | s_any[2][w] | s_any[3][w]);
            while (fix) {
                const int b = __builtin_ctzll(fix);
                fix &= fix - 1;
                const int k = 64 * w + b;
                if (valid) {
                    float* e = ub + (int64_t)k * npad + n;
                    *e = sweep_carry(*e, sp[k]);
                }
            }
        }
        if (tid < K && col_open) {
            const int k = tid, w = k >> 6;
            const bool any = (((s_any[0][w] | s_any[1][w] | s_any[2][w] | s_any[3][w]) >> (k & 63)) & 1ull) != 0ull;
            if (col_keep && !any) {                        // nothing was written: the column stays in its frame
                tmeta[(int64_t)blockIdx.x * K + k] = make_float4(meta.x, comp.x, comp.y, meta.w);
            } else {
                float tub = col_bound;                     // nothing replaced: the carry of the largest entry is the largest
                if (col_redo) tub = fmaxf(fmaxf(s_red[0][k], s_red[1][k]), fmaxf(s_red[2][k], s_red[3][k]));
                if (col_forced || any) tub = __builtin_huge_valf();
                tmeta[(int64_t)blockIdx.x * K + k] = make_float4(tub, 1.0f, 0.0f, sq[k].y);
            }
        }
    }
    for (int k = tid; k < K; k += kSelRows) {
        blk_cnt[blk_at(k, blockIdx.x, K)] = wcnt[0][k] + wcnt[1][k] + wcnt[2][k] + wcnt[3][k];
        if (pmask) pblk[blk_at(k, blockIdx.x, K)] = pcnt[0][k] + pcnt[1][k] + pcnt[2][k] + pcnt[3][k];
    }
    if (tid == 0) {
        epart[blockIdx.x] = (double)(wsum[0][0] + wsum[0][1] + wsum[0][2] + wsum[0][3]);
        opart[blockIdx.x] = (double)(wsum[1][0] + wsum[1][1] + wsum[1][2] + wsum[1][3]);
        if (ppre) ppre[blockIdx.x] = (double)(wsum[2][0] + wsum[2][1] + wsum[2][2] + wsum[2][3]);
    }
}

// After the proof round (estep_i8_proof over the pairs of pmask): every row the sweep flagged 8 has a lower bound of its
// component's ln rho in lb[kset][n] and fresh upper bounds of its candidates in ub32.  A candidate whose bound lies below
// lb - 80 ln 2 is irrelevant (r < 2^-80 against the row's log-normaliser >= lb): if all are, the row stays settled - its
// responsibility is 1.0 to the last bit whatever the exact values (flag 4, new distance bound in dlock); otherwise the row
// comes loose: its component and the surviving candidates join the pass's lists for the exact gather, slots that hold
// one of them are marked selected (rec_finish_kernel finds the evaluated pairs through the slots and the mask), and the
// row is evaluated in full (threshold -inf).  The kernel visits every row: it also recounts the pass's candidate masks
// per block (blk_cnt, epart) now that rows have joined, and counts the proof round's pairs (ppart).
static __global__ __launch_bounds__(kSelRows) void rec_proof_decide_kernel(RecArrays rec, const unsigned long long* __restrict__ pmask,
                                                                    unsigned long long* __restrict__ masks, int64_t npad,
                                                                    int64_t n_rows, int K, const double* __restrict__ cvec,
                                                                    const float* __restrict__ ub32, const double* __restrict__ lb,
                                                                    const unsigned char* __restrict__ lcomp,
                                                                    float* __restrict__ dlock, float* __restrict__ rthr,
                                                                    int* __restrict__ blk_cnt, double* __restrict__ epart,
                                                                    double* __restrict__ ppart,
                                                                    const double* __restrict__ own_part /*[blocks] pairs of the
                                                                        round before the sweep (own_first), or null*/) {
    __shared__ int wcnt[4][256];
    __shared__ int wsum[2][4];
    const int tid = threadIdx.x, wave = tid >> 6;
    const int W = (K + 63) / 64;
    for (int k = tid & 63; k < K; k += 64) wcnt[wave][k] = 0;
    const int64_t n = (int64_t)blockIdx.x * kSelRows + tid;
    const bool valid = n < n_rows;
    // (round 6) only the rows of the proof round - flags 8 and 16 - are read and rewritten; the per-block counts and the
    // listed-pair count the sweep left are UPDATED by the pairs that join, instead of recounted over all rows' masks
    // (64 bytes per row at K = 256 for rows that have nothing to decide: 0.5-0.9 ms per pass at config 4)
    unsigned long long mk[4] = {0ull, 0ull, 0ull, 0ull}, old[4] = {0ull, 0ull, 0ull, 0ull};
    int listed = 0, proved = 0;
    const unsigned fl0 = valid ? rec.flags[n] : 0u;
    if (valid && ((fl0 & 16u) || fl0 == 8u)) {
        for (int w = 0; w < W; ++w) old[w] = mk[w] = masks[(int64_t)w * npad + n];
        if (fl0 & 16u) {
            // candidates of a row whose reference is exact (rthr = its best exact value - 80 ln 2): those whose fresh bound
            // clears the threshold are done with; the others join the pass's lists (and may still leave the exact gather early)
            const double thr = (double)rthr[n];
            unsigned sel = rec.sel[n];
            int in_slots = __builtin_popcount(sel), total = 0;
            unsigned long long kept[4] = {0ull, 0ull, 0ull, 0ull};
            for (int w = 0; w < W; ++w) {
                unsigned long long m = pmask[(int64_t)w * npad + n];
                proved += __builtin_popcountll(m);
                // (four scattered loads in flight per round: one at a time, every proof pair of the row was a memory round
                // trip of its own - six of them in a row at K = 256)
                while (m) {
                    int b[4];
                    float uf[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        b[q] = m ? __builtin_ctzll(m) : -1;
                        m &= m - 1;                                          // (0 stays 0)
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) uf[q] = b[q] >= 0 ? ub32[(int64_t)(64 * w + b[q]) * npad + n] : 0.0f;
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (b[q] >= 0 && !((double)uf[q] < thr)) kept[w] |= 1ull << b[q];      // also NaN
                }
            }
            unsigned long long done[4] = {0ull, 0ull, 0ull, 0ull};          // done with, and without a slot
            for (int w = 0; w < W; ++w) done[w] = pmask[(int64_t)w * npad + n] & ~kept[w];
#pragma unroll
            for (int j = 0; j < kRecSlots; ++j) {
                const unsigned short k = rec.k[(int64_t)j * rec.npad + n];
                if (k == kRecEmpty) continue;
                if (((kept[k >> 6] >> (k & 63)) & 1ull) && !((sel >> j) & 1u)) {
                    sel |= 1u << j;
                    ++in_slots;
                } else if ((done[k >> 6] >> (k & 63)) & 1ull) {
                    // done with by its fresh bound: the slot takes it over (tighter than the carried one)
                    rec.d[(int64_t)j * rec.npad + n] = dist_lower_f32(cvec[k], (double)ub32[(int64_t)k * npad + n]);
                    done[k >> 6] &= ~(1ull << (k & 63));
                }
            }
            // ... without one it belongs to the rest, whose bound need not cover it yet: when the ninth largest key of the
            // sweep was an exact pair, B is the largest bound among the pairs that were NOT candidates (restmax)
            float done_max = -__builtin_huge_valf();
            for (int w = 0; w < W; ++w) {
                unsigned long long m = done[w];
                while (m) {
                    const int b = __builtin_ctzll(m);
                    m &= m - 1;
                    done_max = fmaxf(done_max, ub32[(int64_t)(64 * w + b) * npad + n]);
                }
            }
            if (done_max > rec.B[n]) rec.B[n] = done_max;
            for (int w = 0; w < W; ++w) {
                mk[w] |= kept[w];
                total += __builtin_popcountll(mk[w]);
                masks[(int64_t)w * npad + n] = mk[w];
            }
            rec.sel[n] = (unsigned char)sel;
            rec.flags[n] = (unsigned char)(total > in_slots ? 2 : 0);
        }
        if (fl0 == 8u) {
            const int kset = lcomp[n];
            const double l = lb[(int64_t)kset * npad + n];
            const double thr = l - kRelNats;                              // -inf when the proof kernel had no bound
            unsigned long long kept[4] = {0ull, 0ull, 0ull, 0ull};
            bool any = false;
            float done_max = -__builtin_huge_valf();       // largest fresh bound among the candidates that are done with
            for (int w = 0; w < W; ++w) {
                unsigned long long m = pmask[(int64_t)w * npad + n];
                proved += __builtin_popcountll(m);
                if ((kset >> 6) == w) m &= ~(1ull << (kset & 63));           // (the row's own pair is not a candidate)
                while (m) {
                    int b[4];
                    float uf[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        b[q] = m ? __builtin_ctzll(m) : -1;
                        m &= m - 1;
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) uf[q] = b[q] >= 0 ? ub32[(int64_t)(64 * w + b[q]) * npad + n] : 0.0f;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        if (b[q] < 0) continue;
                        if (!((double)uf[q] < thr)) {                        // also NaN
                            kept[w] |= 1ull << b[q];
                            any = true;
                        } else {
                            done_max = fmaxf(done_max, uf[q]);
                        }
                    }
                }
            }
            if (!any && l > -__builtin_huge_val()) {
                rec.flags[n] = 4;
                dlock[n] = f32_up(dist_of(cvec[kset], l) * (1.0 + 1e-9));
            } else {
                kept[kset >> 6] |= 1ull << (kset & 63);
                unsigned sel = 0;
                int in_slots = 0, total = 0;
#pragma unroll
                for (int j = 0; j < kRecSlots; ++j) {
                    const unsigned short k = rec.k[(int64_t)j * rec.npad + n];
                    if (k != kRecEmpty && ((kept[k >> 6] >> (k & 63)) & 1ull)) {
                        sel |= 1u << j;
                        ++in_slots;
                    }
                }
                for (int w = 0; w < W; ++w) {
                    mk[w] = kept[w];
                    total += __builtin_popcountll(kept[w]);
                    masks[(int64_t)w * npad + n] = kept[w];
                }
                rec.sel[n] = (unsigned char)sel;
                rec.exact[n] = 0;
                rec.flags[n] = (unsigned char)(total > in_slots ? 2 : 0);
                // (the lazy sweep gives a settled row no slots: a candidate that is done with is covered by the rest bound)
                const float Bn = rec.B[n];
                if (done_max > Bn) rec.B[n] = done_max;
                rthr[n] = -__builtin_huge_valf();
            }
        }
        for (int w = 0; w < W; ++w) {
            mk[w] &= ~old[w];                                  // what joined the lists (nothing ever leaves them here)
            listed += __builtin_popcountll(mk[w]);
        }
    }
    for (int w = 0; w < W; ++w) count_word(mk[w], w, wave, wcnt);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        listed += __shfl_xor(listed, o);
        proved += __shfl_xor(proved, o);
    }
    if ((tid & 63) == 0) {
        wsum[0][wave] = listed;
        wsum[1][wave] = proved;
    }
    __syncthreads();
    const int joined = wsum[0][0] + wsum[0][1] + wsum[0][2] + wsum[0][3];
    if (joined != 0)
        for (int k = tid; k < K; k += kSelRows) {
            const int c = wcnt[0][k] + wcnt[1][k] + wcnt[2][k] + wcnt[3][k];
            if (c != 0) blk_cnt[blk_at(k, blockIdx.x, K)] += c;
        }
    if (tid == 0) {
        if (joined != 0) epart[blockIdx.x] += (double)joined;
        ppart[blockIdx.x] = (double)(wsum[1][0] + wsum[1][1] + wsum[1][2] + wsum[1][3]) + (own_part ? own_part[blockIdx.x] : 0.0);
    }
}

// Bound pass with a proof round: the candidates rec_select_kernel has listed carry bounds from the bound pass's leading
// output blocks only; estep_i8_proof has given each of them a bound from ALL blocks (ub32).  Those that now clear the row's
// threshold (its best component's exact value - 80 ln 2) are done with: taken off the pass's lists; a slot keeps the fresh
// bound, a component without a slot rejoins the rest bound B.  Overflow rows are left alone.  Recounts the lists per block.
static __global__ __launch_bounds__(kSelRows) void rec_prune_kernel(RecArrays rec, unsigned long long* __restrict__ masks, int64_t npad,
                                                             int64_t n_rows, int K, const double* __restrict__ cvec,
                                                             const float* __restrict__ ub32, const float* __restrict__ rthr,
                                                             int* __restrict__ blk_cnt, double* __restrict__ epart,
                                                             double* __restrict__ ppart) {
    __shared__ int wcnt[4][256];
    __shared__ int wsum[2][4];
    const int tid = threadIdx.x, wave = tid >> 6;
    const int W = (K + 63) / 64;
    for (int k = tid & 63; k < K; k += 64) wcnt[wave][k] = 0;
    const int64_t n = (int64_t)blockIdx.x * kSelRows + tid;
    const bool valid = n < n_rows;
    unsigned long long mk[4] = {0ull, 0ull, 0ull, 0ull};
    int listed = 0, proved = 0;
    if (valid) {
        for (int w = 0; w < W; ++w) mk[w] = masks[(int64_t)w * npad + n];
        const unsigned fl = rec.flags[n];
        if (fl != 1u && (mk[0] | mk[1] | mk[2] | mk[3]) != 0ull) {
            const double thr = (double)rthr[n];
            unsigned long long inslot[4] = {0ull, 0ull, 0ull, 0ull};
            unsigned sel = rec.sel[n];
            float rest = rec.B[n];
            bool changed = false;
#pragma unroll
            for (int j = 0; j < kRecSlots; ++j) {
                const unsigned short k = rec.k[(int64_t)j * rec.npad + n];
                if (k == kRecEmpty) continue;
                inslot[k >> 6] |= 1ull << (k & 63);
                if (!((sel >> j) & 1u)) continue;
                const float u = ub32[(int64_t)k * npad + n];
                if ((double)u < thr) {                                    // done with: the slot keeps the fresh bound
                    sel &= ~(1u << j);
                    mk[k >> 6] &= ~(1ull << (k & 63));
                    rec.d[(int64_t)j * rec.npad + n] = dist_lower_f32(cvec[k], (double)u);
                    changed = true;
                }
                ++proved;
            }
            int extra = 0;
            for (int w = 0; w < W; ++w) {
                unsigned long long m = mk[w] & ~inslot[w];
                while (m) {
                    const int b = __builtin_ctzll(m);
                    m &= m - 1;
                    const float u = ub32[(int64_t)(64 * w + b) * npad + n];
                    ++proved;
                    if ((double)u < thr) {                                // rejoins the components the rest bound speaks for
                        mk[w] &= ~(1ull << b);
                        rest = (u > rest) ? u : rest;
                        changed = true;
                    } else {
                        ++extra;
                    }
                }
            }
            if (changed) {
                for (int w = 0; w < W; ++w) masks[(int64_t)w * npad + n] = mk[w];
                rec.sel[n] = (unsigned char)sel;
                rec.B[n] = rest;
                rec.flags[n] = (unsigned char)(extra > 0 ? 2 : 0);
            }
        }
        for (int w = 0; w < W; ++w) listed += __builtin_popcountll(mk[w]);
    }
    for (int w = 0; w < W; ++w) count_word(mk[w], w, wave, wcnt);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        listed += __shfl_xor(listed, o);
        proved += __shfl_xor(proved, o);
    }
    if ((tid & 63) == 0) {
        wsum[0][wave] = listed;
        wsum[1][wave] = proved;
    }
    __syncthreads();
    for (int k = tid; k < K; k += kSelRows)
        blk_cnt[blk_at(k, blockIdx.x, K)] = wcnt[0][k] + wcnt[1][k] + wcnt[2][k] + wcnt[3][k];
    if (tid == 0) {
        epart[blockIdx.x] = (double)(wsum[0][0] + wsum[0][1] + wsum[0][2] + wsum[0][3]);
        ppart[blockIdx.x] = (double)(wsum[1][0] + wsum[1][1] + wsum[1][2] + wsum[1][3]);
    }
}

// first[k] = index of component k's first gather chunk (chunk = per_wg list entries), first[K] = number of chunks
// (one workgroup: the counts come in with parallel loads - a single thread's K dependent loads were 24 us at K = 256 - and
// the prefix over them is taken from LDS)
static __global__ void gather_plan_kernel(const int* __restrict__ counts, int K, int per_wg, int* __restrict__ first) {
    __shared__ int sc[1025];
    if (blockIdx.x != 0) return;
    if (K > 1024) {
        if (threadIdx.x != 0) return;
        int total = 0;
        for (int k = 0; k < K; ++k) {
            first[k] = total;
            total += (counts[k] + per_wg - 1) / per_wg;
        }
        first[K] = total;
        return;
    }
    for (int k = threadIdx.x; k < K; k += blockDim.x) sc[k] = (counts[k] + per_wg - 1) / per_wg;
    __syncthreads();
    if (threadIdx.x == 0) {
        int total = 0;
        for (int k = 0; k < K; ++k) {
            const int c = sc[k];
            sc[k] = total;
            total += c;
        }
        sc[K] = total;
    }
    __syncthreads();
    for (int k = threadIdx.x; k <= K; k += blockDim.x) first[k] = sc[k];
}

// After the exact evaluation of the listed pairs: refresh the records from the exact values (distances, exact
// flags; overflow rows are rebuilt from all K values), and produce what the rest of the pass needs per row:
// lse_n, the best component, the M-step's active mask (r_nk >= 2^-80) with its block counts, apart[block] = active pairs.
static __global__ __launch_bounds__(kSelRows) void rec_finish_kernel(RecArrays rec, const double* __restrict__ lnrho, int64_t npad,
                                                              int64_t n_rows, int K, const double* __restrict__ cvec,
                                                              double* __restrict__ lse, int* __restrict__ khat,
                                                              unsigned long long* __restrict__ masks,
                                                              int* __restrict__ blk_cnt, double* __restrict__ apart,
                                                              double* __restrict__ mpart /*rows whose best component changed*/,
                                                              float* __restrict__ ub32 /*[K][npad] per-pair bounds (sweeps)*/,
                                                              unsigned char* __restrict__ lock /*null: no cache of single-component rows*/,
                                                              float* __restrict__ dlock, double settle_margin /*< 0: rows never settle; >= 1e300: every single-component row does*/,
                                                              unsigned long long* __restrict__ dmask, int* __restrict__ dblk,
                                                              unsigned long long* __restrict__ mmask, int* __restrict__ mblk,
                                                              double* __restrict__ spart, double* __restrict__ gpart,
                                                              double* __restrict__ qpart /*pairs the M-step accumulates*/,
                                                              const float* __restrict__ rthr /*[npad] relevance thresholds*/,
                                                              unsigned char* __restrict__ lcomp /*[npad] component a cached row is in*/) {
    __shared__ int wcnt[4][256];
    __shared__ int dcnt[4][256];
    __shared__ int mcnt[4][256];
    __shared__ int wact[4], wmov[4], wset[4], wlist[4], wacc[4];
    const int tid = threadIdx.x, wave = tid >> 6;
    const int W = (K + 63) / 64;
    for (int k = tid & 63; k < K; k += 64) wcnt[wave][k] = dcnt[wave][k] = mcnt[wave][k] = 0;
    const int64_t n = (int64_t)blockIdx.x * kSelRows + tid;
    const bool valid = n < n_rows;
    unsigned long long mk[4] = {0ull, 0ull, 0ull, 0ull};
    unsigned long long dm[4] = {0ull, 0ull, 0ull, 0ull};     // the row enters (its best component) / leaves (khat_before) the cache
    int active = 0, settled_i = 0;
    const int khat_before = valid ? khat[n] : 0;
    const unsigned fl = valid ? rec.flags[n] : 0u;
    // what the decision to settle a row needs: its log-normaliser, best component and value, the largest value or bound
    // among the other components (+inf: unknown - the row is left alone)
    double row_l = 0.0, row_best = 0.0, row_second = __builtin_huge_val();
    int row_arg = -1;
    // a stored value below the row's relevance threshold may be a bound (the gather's early way out, estep.h): it is
    // irrelevant either way (2^-80 below the best), counts as evaluated, but is never flagged exact
    const double thr_row = valid ? (double)rthr[n] : 0.0;
    if (valid && fl == 4u) {
        // settled row: nothing was evaluated, nothing changes (lse[n] and the ln rho entries are stale until a read-out
        // asks for them); it still counts as one active pair
        active = 1;
        settled_i = 1;
    } else if (valid && fl == 2u) {
        // refreshed row: slots (exact where listed, carried bounds otherwise) and the listed components without a slot
        // compete for the C slots again; what does not get one joins the rest bound (B holds the unlisted ones already)
        const unsigned live = (unsigned)rec.sel[n] | (unsigned)rec.exact[n];
        float ds[kRecSlots], vs[kRecSlots];
        unsigned short ks[kRecSlots];
#pragma unroll
        for (int j = 0; j < kRecSlots; ++j) {
            ds[j] = __builtin_huge_valf();
            vs[j] = -__builtin_huge_valf();
            ks[j] = kRecEmpty;
        }
        float rest = rec.B[n];
        unsigned long long cand[4] = {0ull, 0ull, 0ull, 0ull}, ev[4] = {0ull, 0ull, 0ull, 0ull};
        for (int w = 0; w < W; ++w) cand[w] = masks[(int64_t)w * npad + n];
        double mx = -__builtin_huge_val(), ssum = 0.0;
        int arg = 0x7fffffff;
        bool nan = false;
        auto take_exact = [&](int k) {
            const double x = lnrho[(int64_t)k * npad + n];
            ub32[(int64_t)k * npad + n] = f32_up(x);
            ev[k >> 6] |= 1ull << (k & 63);
            nan = nan || x != x;
            if (x > mx) {
                ssum = fma(ssum, exp(mx - x), 1.0);
                arg = k;
                mx = x;
            } else {
                if (x == mx && k < arg) arg = k;
                ssum += exp(x - mx);
            }
            rec_insert(ds, ks, vs, rest, f32_down(dist_of(cvec[k], x)),
                       (unsigned short)(k | (x < thr_row ? 0 : kRecExactBit)), f32_up(x));
        };
#pragma unroll
        for (int j = 0; j < kRecSlots; ++j) {
            const unsigned short k = rec.k[(int64_t)j * rec.npad + n];
            if (k == kRecEmpty) continue;
            cand[k >> 6] &= ~(1ull << (k & 63));
            if ((live >> j) & 1u) {
                take_exact(k);
            } else {
                const float d = rec.d[(int64_t)j * rec.npad + n];
                rec_insert(ds, ks, vs, rest, d, k, f32_up(cvec[k] - 0.5 * (double)d * (double)d));
            }
        }
        for (int w = 0; w < W; ++w) {
            unsigned long long m = cand[w];
            while (m) {
                const int b = __builtin_ctzll(m);
                m &= m - 1;
                take_exact(64 * w + b);
            }
        }
        double l = mx + log(ssum);
        if (nan) l = __builtin_nan("");
        lse[n] = l;
        khat[n] = arg == 0x7fffffff ? 0 : arg;
        unsigned ex = 0;
#pragma unroll
        for (int j = 0; j < kRecSlots; ++j) {
            const bool e1 = ks[j] != kRecEmpty && (ks[j] & kRecExactBit);
            rec.k[(int64_t)j * rec.npad + n] = ks[j] == kRecEmpty ? kRecEmpty : (unsigned short)(ks[j] & kRecCompMask);
            rec.d[(int64_t)j * rec.npad + n] = ds[j];
            ex |= e1 ? (1u << j) : 0u;
        }
        rec.B[n] = rest;
        rec.exact[n] = (unsigned char)ex;
        for (int w = 0; w < W; ++w) {
            unsigned long long m = ev[w];
            while (m) {
                const int b = __builtin_ctzll(m);
                m &= m - 1;
                const double t = lnrho[(int64_t)(64 * w + b) * npad + n] - l;
                if (!(t < -kRelNats)) {
                    mk[w] |= 1ull << b;
                    ++active;
                }
            }
        }
        if (!nan) {
            float sec = rest;
#pragma unroll
            for (int j = 0; j < kRecSlots; ++j)
                if (ks[j] != kRecEmpty && (int)(ks[j] & kRecCompMask) != arg) sec = (vs[j] > sec || vs[j] != vs[j]) ? vs[j] : sec;
            row_l = l;
            row_best = mx;
            row_arg = arg == 0x7fffffff ? -1 : arg;
            row_second = (double)sec;
        }
    } else if (valid && fl == 0u) {
        unsigned live = (unsigned)rec.sel[n] | (unsigned)rec.exact[n];
        unsigned bounds_only = 0;
        double v[kRecSlots];
        unsigned short kk[kRecSlots];
        double mx = -__builtin_huge_val();
        int arg = 0x7fffffff;
        bool nan = false;
        // (the slots' components, then all their exact values, then the arithmetic: with the loads next to their use every
        // evaluated pair of the row was a memory round trip of its own)
#pragma unroll
        for (int j = 0; j < kRecSlots; ++j) kk[j] = rec.k[(int64_t)j * rec.npad + n];
#pragma unroll
        for (int j = 0; j < kRecSlots; ++j)
            v[j] = ((live >> j) & 1u) ? lnrho[(int64_t)kk[j] * npad + n] : 0.0;
#pragma unroll
        for (int j = 0; j < kRecSlots; ++j) {
            if (!((live >> j) & 1u)) continue;
            const double x = v[j];
            ub32[(int64_t)kk[j] * npad + n] = f32_up(x);
            rec.d[(int64_t)j * rec.npad + n] = f32_down(dist_of(cvec[kk[j]], x));
            if (x < thr_row) bounds_only |= 1u << j;
            nan = nan || x != x;
            if (x > mx || (x == mx && (int)kk[j] < arg)) {          // first maximiser, like numpy.argmax
                mx = x;
                arg = kk[j];
            }
        }
        double s = 0.0;
#pragma unroll
        for (int j = 0; j < kRecSlots; ++j)
            if ((live >> j) & 1u) s += exp(v[j] - mx);
        double l = mx + log(s);
        if (nan) {                                                  // poisoned row: NaN everywhere, like the dense path
            l = __builtin_nan("");
            arg = arg == 0x7fffffff ? 0 : arg;
        }
        lse[n] = l;
        khat[n] = arg == 0x7fffffff ? 0 : arg;
        rec.exact[n] = (unsigned char)(live & ~bounds_only);
#pragma unroll
        for (int j = 0; j < kRecSlots; ++j) {
            if (!((live >> j) & 1u)) continue;
            if (!(v[j] - l < -kRelNats)) {                           // NaN stays active
                mk[kk[j] >> 6] |= 1ull << (kk[j] & 63);
                ++active;
            }
        }
        if (!nan) {
            row_l = l;
            row_best = mx;
            row_arg = arg == 0x7fffffff ? -1 : arg;
        }
        if (!nan && active == 1 && settle_margin >= 0.0 && lock != nullptr) {
            double sec = (double)rec.B[n];
#pragma unroll
            for (int j = 0; j < kRecSlots; ++j) {
                if (kk[j] == kRecEmpty || (int)kk[j] == arg) continue;
                double o = v[j];
                if (!((live >> j) & 1u)) {
                    const double d = (double)rec.d[(int64_t)j * rec.npad + n];
                    o = cvec[kk[j]] - 0.5 * d * d * (1.0 - 1e-6);
                }
                sec = (o > sec || o != o) ? o : sec;
            }
            row_second = sec;
        }
    } else if (valid) {
        // overflow row: every value of the dense row is exact.  One sweep: running max / sum and the C nearest
        float ds[kRecSlots], vs[kRecSlots];
        unsigned short ks[kRecSlots];
#pragma unroll
        for (int j = 0; j < kRecSlots; ++j) {
            ds[j] = __builtin_huge_valf();
            vs[j] = -__builtin_huge_valf();
            ks[j] = kRecEmpty;
        }
        float rest = -__builtin_huge_valf();
        double mx = lnrho[n], s = 1.0;
        int arg = 0;
        ub32[n] = f32_up(mx);
        rec_insert(ds, ks, vs, rest, f32_down(dist_of(cvec[0], mx)), 0, f32_up(mx));
        for (int k = 1; k < K; ++k) {
            const double x = lnrho[(int64_t)k * npad + n];
            ub32[(int64_t)k * npad + n] = f32_up(x);
            rec_insert(ds, ks, vs, rest, f32_down(dist_of(cvec[k], x)), (unsigned short)k, f32_up(x));
            if (x > mx) {
                s = fma(s, exp(mx - x), 1.0);
                mx = x;
                arg = k;
            } else {
                s += exp(x - mx);
            }
        }
        const double l = mx + log(s);
        lse[n] = l;
        khat[n] = arg;
        row_l = l;
        row_best = mx;
        row_arg = (l == l) ? arg : -1;
        unsigned ex = 0;
#pragma unroll
        for (int j = 0; j < kRecSlots; ++j) {
            rec.k[(int64_t)j * rec.npad + n] = ks[j];
            rec.d[(int64_t)j * rec.npad + n] = ds[j];
            ex |= (ks[j] != kRecEmpty) ? (1u << j) : 0u;
        }
        rec.B[n] = rest;
        rec.exact[n] = (unsigned char)ex;
        rec.sel[n] = (unsigned char)ex;
        for (int k = 0; k < K; ++k) {
            const double t = lnrho[(int64_t)k * npad + n] - l;
            if (!(t < -kRelNats)) {
                mk[k >> 6] |= 1ull << (k & 63);
                ++active;
            }
        }
    }
    // The cache of single-component rows (workspace.h): a row whose ONLY active component is k has r_nk = 1.0 to the last
    // bit (the other terms of its log-normaliser are below 2^-80), so its addend to component k's statistics is the same
    // in every pass in which that holds - it is kept in the cache and the row left out of the M-step's lists (mmask).
    //   lock 0 -> 3  the row enters the cache of its component;  1 -> 1  it stays;  1 -> 2  it leaves;  1 -> 4  it moves
    //   to another component (fill_lists_kernel gives the delta lists' entries their signs and settles the state).
    // Settling goes one step further: the row is also left out of the E-step's lists (masks); the next sweep only checks
    // its carried bounds, and what they no longer prove goes through the int8 proof round instead of an exact evaluation.
    // By default every single-component row settles (whatever its carried bounds say, the proof round is cheaper than the
    // exact evaluation of the row's component plus its candidates); with a margin only rows whose other components all
    // lie at least that many nats below the 2^-80 line.
    unsigned long long mm[4] = {mk[0], mk[1], mk[2], mk[3]};         // the M-step's lists
    int in_lists = 0, m_pairs = 0;
    if (valid && lock != nullptr && fl != 4u) {
        const unsigned lk = lock[n];
        const bool single = row_arg >= 0 && active == 1;
        if (lk == 1u) {
            const int was = lcomp[n];                   // (khat may have been rewritten by a bound pass on the way here)
            if (!(single && row_arg == was)) {
                dm[was >> 6] |= 1ull << (was & 63);
                if (single) dm[row_arg >> 6] |= 1ull << (row_arg & 63);
                lock[n] = single ? 4 : 2;
            }
        } else if (single) {
            dm[row_arg >> 6] |= 1ull << (row_arg & 63);
            lock[n] = 3;
        }
        if (single) {
            lcomp[n] = (unsigned char)row_arg;
            mm[row_arg >> 6] &= ~(1ull << (row_arg & 63));
            if (settle_margin >= 1e300 || (settle_margin >= 0.0 && row_second < row_l - kRelNats - settle_margin)) {
                dlock[n] = f32_up(dist_of(cvec[row_arg], row_best) * (1.0 + 1e-9));
                mk[row_arg >> 6] &= ~(1ull << (row_arg & 63));
            }
        }
    }
    if (valid) {
        for (int w = 0; w < W; ++w) {
            masks[(int64_t)w * npad + n] = mk[w];
            in_lists += __builtin_popcountll(mk[w]);
            m_pairs += __builtin_popcountll(mm[w]) + __builtin_popcountll(dm[w]);
            if (dmask) {
                dmask[(int64_t)w * npad + n] = dm[w];
                mmask[(int64_t)w * npad + n] = mm[w];
            }
        }
    }
    for (int w = 0; w < W; ++w) {
        count_word<false>(mk[w], w, wave, wcnt);
        if (dmask) {
            count_word<false>(dm[w], w, wave, dcnt);
            count_word<false>(mm[w], w, wave, mcnt);
        }
    }
    int moved = (valid && khat[n] != khat_before) ? 1 : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        active += __shfl_xor(active, o);
        moved += __shfl_xor(moved, o);
        settled_i += __shfl_xor(settled_i, o);
        in_lists += __shfl_xor(in_lists, o);
        m_pairs += __shfl_xor(m_pairs, o);
    }
    if ((tid & 63) == 0) {
        wact[wave] = active;
        wmov[wave] = moved;
        wset[wave] = settled_i;
        wlist[wave] = in_lists;
        wacc[wave] = m_pairs;
    }
    __syncthreads();
    for (int k = tid; k < K; k += kSelRows) {
        blk_cnt[blk_at(k, blockIdx.x, K)] = wcnt[0][k] + wcnt[1][k] + wcnt[2][k] + wcnt[3][k];
        if (dblk) {
            dblk[blk_at(k, blockIdx.x, K)] = dcnt[0][k] + dcnt[1][k] + dcnt[2][k] + dcnt[3][k];
            mblk[blk_at(k, blockIdx.x, K)] = mcnt[0][k] + mcnt[1][k] + mcnt[2][k] + mcnt[3][k];
        }
    }
    if (tid == 0) {
        apart[blockIdx.x] = (double)(wact[0] + wact[1] + wact[2] + wact[3]);
        mpart[blockIdx.x] = (double)(wmov[0] + wmov[1] + wmov[2] + wmov[3]);
        spart[blockIdx.x] = (double)(wset[0] + wset[1] + wset[2] + wset[3]);
        gpart[blockIdx.x] = (double)(wlist[0] + wlist[1] + wlist[2] + wlist[3]);
        qpart[blockIdx.x] = (double)(wacc[0] + wacc[1] + wacc[2] + wacc[3]);
    }
}

// ctr[0] = sum apart (active pairs), ctr[1] = sum epart (exactly evaluated pairs), ctr[2] = sum opart (overflow rows),
// ctr[3] = sum mpart (rows whose best component changed), ctr[4] = sum spart (settled rows), ctr[5] = sum gpart (pairs
// in the E-step's lists), ctr[6] = sum qpart (pairs the M-step accumulates), ctr[7] = sum ppart (pairs of the proof round);
// a null part leaves its counter as it is.
// One workgroup per counter.
static __global__ __launch_bounds__(1024) void sum_parts_kernel(const double* __restrict__ apart, const double* __restrict__ epart,
                                                         const double* __restrict__ opart, const double* __restrict__ mpart,
                                                         const double* __restrict__ spart, const double* __restrict__ gpart,
                                                         const double* __restrict__ qpart, const double* __restrict__ ppart,
                                                         int blocks, double* __restrict__ ctr) {
    __shared__ double part[16];
    const double* src = blockIdx.x == 0 ? apart : (blockIdx.x == 1 ? epart : (blockIdx.x == 2 ? opart :
                        (blockIdx.x == 3 ? mpart : (blockIdx.x == 4 ? spart : (blockIdx.x == 5 ? gpart :
                        (blockIdx.x == 6 ? qpart : ppart))))));
    if (!src) return;
    double a = 0.0;                                   // (integer-valued addends below 2^53: any order is exact)
    for (int b = threadIdx.x; b < blocks; b += 1024) a += src[b];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int i = 0; i < 16; ++i) t += part[i];
        ctr[blockIdx.x] = t;
    }
}

// Read-outs while rows are settled: their component's ln rho is evaluated for the parameters in force on request only.
// settled_mask_kernel lists (row, khat) of every settled row; after the gather settled_lse_kernel sets lse[n] to that
// value (the row's only active pair: its log-normaliser to the last bit).
__device__ __forceinline__ bool row_settled(const unsigned char* __restrict__ lock,
                                            const unsigned long long* __restrict__ emask, int64_t npad, int W, int64_t n) {
    if (lock[n] != 1) return false;
    unsigned long long any = 0ull;
    for (int w = 0; w < W; ++w) any |= emask[(int64_t)w * npad + n];
    return any == 0ull;                   // in the cache and in none of the E-step's lists
}

static __global__ __launch_bounds__(kSelRows) void settled_mask_kernel(const unsigned char* __restrict__ lock,
                                                                const unsigned long long* __restrict__ emask,
                                                                const unsigned char* __restrict__ lcomp, int64_t npad, int64_t n_rows,
                                                                int K, unsigned long long* __restrict__ masks,
                                                                int* __restrict__ blk_cnt,
                                                                const double* __restrict__ drift = nullptr /*only the rows of
                                                                    own_first components (the sweep's first proof round)*/,
                                                                double* __restrict__ listed_part = nullptr /*[blocks] rows listed*/) {
    __shared__ int wcnt[4][256];
    const int64_t n = (int64_t)blockIdx.x * kSelRows + threadIdx.x;
    const bool valid = n < n_rows;
    const int W = (K + 63) / 64;
    const int wave = threadIdx.x >> 6;
    for (int k = threadIdx.x & 63; k < K; k += 64) wcnt[wave][k] = 0;
    // (the three per-row loads at once: lock -> mask -> component in turn was three memory round trips per workgroup)
    const int64_t nn = valid ? n : 0;
    const unsigned lk = lock[nn];
    const int lc = (int)lcomp[nn];
    unsigned long long any = 0ull;
    for (int w = 0; w < W; ++w) any |= emask[(int64_t)w * npad + nn];
    int kh = (valid && lk == 1u && any == 0ull) ? lc : -1;          // in the cache and in none of the E-step's lists
    if (kh >= 0 && drift != nullptr && !own_first(drift[3 * K + kh], drift[K + kh])) kh = -1;
    for (int w = 0; w < W; ++w) {
        const unsigned long long mk = (kh >= 0 && (kh >> 6) == w) ? 1ull << (kh & 63) : 0ull;
        if (valid) masks[(int64_t)w * npad + n] = mk;
        count_word(mk, w, wave, wcnt);
    }
    __syncthreads();
    int tot = 0;
    for (int k = threadIdx.x; k < K; k += kSelRows) {
        const int c = wcnt[0][k] + wcnt[1][k] + wcnt[2][k] + wcnt[3][k];
        blk_cnt[blk_at(k, blockIdx.x, K)] = c;
        tot += c;
    }
    if (listed_part) {
        __shared__ int wtot[4];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o);
        if ((threadIdx.x & 63) == 0) wtot[wave] = tot;
        __syncthreads();
        if (threadIdx.x == 0) listed_part[blockIdx.x] = (double)(wtot[0] + wtot[1] + wtot[2] + wtot[3]);
    }
}

static __global__ void settled_lse_kernel(const unsigned char* __restrict__ lock, const unsigned long long* __restrict__ emask,
                                   const unsigned char* __restrict__ lcomp, const double* __restrict__ lnrho, int64_t npad,
                                   int64_t n_rows, int K, double* __restrict__ lse) {
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n < n_rows && row_settled(lock, emask, npad, (K + 63) / 64, n)) lse[n] = lnrho[(int64_t)lcomp[n] * npad + n];
}

// Read-outs of a pass that lived on records.  The active mask rec_finish_kernel left (r_nk >= 2^-80) marks the pairs
// whose exact value is in the dense array.  mode 0: ln rho - exact for active pairs and for exact slots, otherwise the
// record's upper bound (at least 80 ln 2 below the row's log-normaliser); mode 1: responsibilities, exactly 0 for inactive pairs.
static __global__ void rec_readout_kernel(RecArrays rec, const unsigned long long* __restrict__ masks,
                                   const double* __restrict__ lnrho, const double* __restrict__ lse,
                                   const double* __restrict__ cvec, int64_t npad, int64_t row0, int64_t n_rows, int K,
                                   int mode, double* __restrict__ out, const int* __restrict__ iperm,
                                   const unsigned char* __restrict__ lock, const unsigned char* __restrict__ lcomp) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_rows * K) return;
    const int64_t n = iperm ? iperm[row0 + e / K] : row0 + e / K;
    const int k = (int)(e % K);
    bool exact = ((masks[(int64_t)(k >> 6) * npad + n] >> (k & 63)) & 1ull) != 0 || (rec.flags[n] & 1) != 0;
    // a settled row's pair (in the cache instead of the M-step's mask; refreshed by the caller before this read-out)
    if (lock && (lock[n] == 1 || lock[n] >= 3) && (int)lcomp[n] == k) exact = true;
    if (mode == 1) {
        out[e] = exact ? exp(lnrho[(int64_t)k * npad + n] - lse[n]) : 0.0;
        return;
    }
    // (a row the lazy sweep left settled has no record of this pass: the cap below answers for it)
    const bool no_record = rec.flags[n] == 4;
    double ub = no_record ? __builtin_huge_val() : (double)rec.B[n];
    if (!exact && !no_record) {
        const unsigned live = rec.exact[n];
#pragma unroll
        for (int j = 0; j < kRecSlots; ++j) {
            if (rec.k[(int64_t)j * rec.npad + n] == k) {
                exact = (live >> j) & 1u;
                const double d = (double)rec.d[(int64_t)j * rec.npad + n];
                ub = cvec[k] - 0.5 * d * d;
            }
        }
    }
    if (!exact) {
        // every pair that is not active lies at least 80 ln 2 below the row's log-normaliser (evaluated and found so,
        // or proven so by its bound): B alone may be the value of a ninth near component
        const double cap = lse[n] - kRelNats;
        ub = cap < ub ? cap : ub;
    }
    out[e] = exact ? lnrho[(int64_t)k * npad + n] : ub;
}

}  // namespace gmmvb
