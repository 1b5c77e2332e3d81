// Scaled forward-backward for the Gaussian-emission HMM on f64 MFMA, chunk-parallel over time.
//
// Replaces the reference's Python loops over T (bayesml/hiddenmarkovnormal/_hiddenmarkovnormal.py:
// _forward :999-1006, _backward :1008-1011), _update_gamma :1013-1014, _update_xi :1016-1018 (a
// materialised [T, K, K] array, 82 GB at config 5) and the ms/ns part of _calc_n_m_x_bar_s :837-845.
//
// With rho'_t = exp(ln rho_t - max_k ln rho_tk) (the reference exponentiates unshifted, :997, and
// underflows) the recursions are
//   alpha_t  = rho'_t o (alpha_{t-1} A~) / c'_t,   c'_t = sum(...)          ln c_t = ln c'_t + max_t
//   beta~_i  ~ A~ (rho'_{i+1} o beta~_{i+1})        (any per-step scale: gamma and xi renormalise)
//   gamma_t  = alpha_t o beta~_t / (alpha_t . beta~_t)
//   ms       = A~ o sum_{t>=1} alpha_{t-1}^T w_t,   w_t = rho'_t o beta~_t / (c'_t (alpha_t . beta~_t))
// Both directions are linear maps per step, M_t = A~ diag(rho'_t), so time is cut into chunks of L
// steps (chunk c = steps 1 + cL .. (c+1)L; t = 0 is the start vector) and
//   H2  chunk_products : P_c = prod_{t in c} M_t, one wave per chunk, K^3 per step on MFMA (both
//                        directions use the same P_c: forward as row-vector x P_c, backward as P_c x column)
//   H3  boundary_scan  : the short sequential pass over chunks (start vector of every chunk)
//   (round 4: for long sequences H2 / H3 run only behind a gate - the start vectors come from sweeps of H4 / H5 started at
//    the uniform vector, which the recursions forget, checked against the replays' own end vectors: hmm_capi.hip, run<KT>)
//   H4  forward_replay : 16 chunks per wave as the 16 MFMA columns, K^2 per step; writes alpha, c'
//   H5  backward_replay: same shape, descending; writes gamma (time-major) and w
//   H6  xi_sum         : the [K x T] x [T x K] product over time on MFMA, slabs per wave (up to 32 states: inside H5)
//   H0+H1 (D <= 16, K <= 32): the emission itself writes rho' and the row maxima (hmm_emission_mfma16_kernel); otherwise
//                        hmm_prep_kernel makes them from the E-step's ln rho array
// State vectors live in the MFMA C/D layout (state on register/lane-group, chunk on lane & 15) and the
// contraction index of step s is taken as {16 kt + g + 4 s}, which makes the accumulator of one time
// step directly the B operand of the next: no cross-lane movement, no LDS, in the whole recursion.
// Time-major arrays store state 16 it + (g + 4 r) at position 16 it + 4 g + r ("lane order"), so each
// lane reads/writes 4 contiguous doubles per 16-state block.
#pragma once
#include "common.h"
#include "estep.h"

namespace gmmvb {

__host__ __device__ constexpr int hmm_pos(int state) {          // natural state -> lane-order position
    const int w = state & 15;
    return (state & ~15) + 4 * (w & 3) + (w >> 2);
}
__host__ __device__ constexpr int hmm_state(int pos) {          // inverse
    const int w = pos & 15;
    return (pos & ~15) + (w >> 2) + 4 * (w & 3);
}

// H1: ln rho [K][npad] (component-major) -> rho' [T][Kp] lane order, mx[T].  A workgroup transposes 64 time steps
// through LDS: component-major reads and time-major writes are both contiguous (a thread per time step wrote 256-byte
// strided rows: 4.9 ms at config 5; this form moves the same 5 GB in 1.x ms).
constexpr int kPrepSteps = 64;
__global__ __launch_bounds__(256) void hmm_prep_kernel(const double* __restrict__ lnrho, int64_t npad, int64_t T, int K, int Kp,
                                                       double* __restrict__ rho_tm, double* __restrict__ mx) {
    extern __shared__ double tile[];            // [Kp][kPrepSteps + 1] ln rho, then [kPrepSteps] row maxima
    constexpr int LD = kPrepSteps + 1;
    double* smx = tile + (size_t)Kp * LD;
    const int tid = threadIdx.x, tq = tid & 63, kk = tid >> 6;
    const int64_t t0 = (int64_t)blockIdx.x * kPrepSteps;
    const int64_t tc = t0 + tq < T ? t0 + tq : T - 1;
    for (int k = kk; k < K; k += 4) tile[k * LD + tq] = lnrho[(int64_t)k * npad + tc];
    __syncthreads();
    if (tid < kPrepSteps) {
        double m = tile[tid];
        for (int k = 1; k < K; ++k) m = fmax(m, tile[k * LD + tid]);
        smx[tid] = m;
        if (t0 + tid < T) mx[t0 + tid] = m;
    }
    __syncthreads();
    for (int e = tid; e < kPrepSteps * Kp; e += 256) {
        const int t = e / Kp, p = e - t * Kp;
        if (t0 + t >= T) break;
        const int k = hmm_state(p);
        rho_tm[(t0 + t) * Kp + p] = k < K ? exp(tile[k * LD + t] - smx[t]) : 0.0;
    }
}

// exp(x) for x <= 0 in 13 f64 instructions (round 5): x = n ln2/64 + r, |r| <= ln2/128, exp(x) = 2^(n div 64) T[n mod 64] e^r
// with T[j] = 2^(j/64) from a 64-entry table in LDS and e^r - 1 = r + r^2 (1/2 + r (1/6 + r (1/24 + r/120))) (remainder
// r^6/720 < 4e-17); n ln2/64 is subtracted in two parts (the leading one has 20 trailing zero bits: exact for |n| < 2^20).
// About one ulp, like the library's exp - which is 30 f64 instructions, and in hmm_emission_mfma16_kernel the f64 vector
// instructions run on the units of the f64 MFMAs beside them: its 16 exponentials per lane and tile were two thirds of that
// kernel's vector work (0.7 of its 3.2 ms at config 5).  x is clamped at -800 (the result underflows to 0 through ldexp).
__device__ const double kExp2Table64[64] = {
    0x1.0000000000000p+0, 0x1.02c9a3e778061p+0, 0x1.059b0d3158574p+0, 0x1.0874518759bc8p+0,
    0x1.0b5586cf9890fp+0, 0x1.0e3ec32d3d1a2p+0, 0x1.11301d0125b51p+0, 0x1.1429aaea92de0p+0,
    0x1.172b83c7d517bp+0, 0x1.1a35beb6fcb75p+0, 0x1.1d4873168b9aap+0, 0x1.2063b88628cd6p+0,
    0x1.2387a6e756238p+0, 0x1.26b4565e27cddp+0, 0x1.29e9df51fdee1p+0, 0x1.2d285a6e4030bp+0,
    0x1.306fe0a31b715p+0, 0x1.33c08b26416ffp+0, 0x1.371a7373aa9cbp+0, 0x1.3a7db34e59ff7p+0,
    0x1.3dea64c123422p+0, 0x1.4160a21f72e2ap+0, 0x1.44e086061892dp+0, 0x1.486a2b5c13cd0p+0,
    0x1.4bfdad5362a27p+0, 0x1.4f9b2769d2ca7p+0, 0x1.5342b569d4f82p+0, 0x1.56f4736b527dap+0,
    0x1.5ab07dd485429p+0, 0x1.5e76f15ad2148p+0, 0x1.6247eb03a5585p+0, 0x1.6623882552225p+0,
    0x1.6a09e667f3bcdp+0, 0x1.6dfb23c651a2fp+0, 0x1.71f75e8ec5f74p+0, 0x1.75feb564267c9p+0,
    0x1.7a11473eb0187p+0, 0x1.7e2f336cf4e62p+0, 0x1.82589994cce13p+0, 0x1.868d99b4492edp+0,
    0x1.8ace5422aa0dbp+0, 0x1.8f1ae99157736p+0, 0x1.93737b0cdc5e5p+0, 0x1.97d829fde4e50p+0,
    0x1.9c49182a3f090p+0, 0x1.a0c667b5de565p+0, 0x1.a5503b23e255dp+0, 0x1.a9e6b5579fdbfp+0,
    0x1.ae89f995ad3adp+0, 0x1.b33a2b84f15fbp+0, 0x1.b7f76f2fb5e47p+0, 0x1.bcc1e904bc1d2p+0,
    0x1.c199bdd85529cp+0, 0x1.c67f12e57d14bp+0, 0x1.cb720dcef9069p+0, 0x1.d072d4a07897cp+0,
    0x1.d5818dcfba487p+0, 0x1.da9e603db3285p+0, 0x1.dfc97337b9b5fp+0, 0x1.e502ee78b3ff6p+0,
    0x1.ea4afa2a490dap+0, 0x1.efa1bee615a27p+0, 0x1.f50765b6e4540p+0, 0x1.fa7c1819e90d8p+0};
__device__ __forceinline__ double exp_nonpos_fast(double x, const double* __restrict__ tab /*[64], LDS*/) {
    x = x < -800.0 ? -800.0 : x;            // (not fmax: a NaN ln rho must stay NaN, as with exp())
    const double n = __builtin_rint(x * 0x1.71547652b82fep+6);
    double r = fma(-n, 0x1.62e42fef00000p-7, x);
    r = fma(-n, 0x1.473de6af278edp-40, r);
    double q = fma(r, 1.0 / 120.0, 1.0 / 24.0);
    q = fma(r, q, 1.0 / 6.0);
    q = fma(r, q, 0.5);
    const double p = fma(r * r, q, r);
    const int ni = (int)n;
    const double t = tab[ni & 63];
    return ldexp(fma(t, p, t), ni >> 6);
}

// H0 + H1 in one kernel for one feature tile (D <= 16) and up to 64 states: the emission on the matrix pipe (estep.h:
// estep_component, four MFMAs per component and 16 rows) leaves ||U_k (x_n - m_k)||^2 of sample n = lane & 15 in all four
// lane groups, so group g keeps the components k = g (mod 4) - which is exactly what a lane owns of a time-major row in
// lane order: states 16 it + g + 4 r at positions 16 it + 4 g + r, four contiguous doubles per 16-state block.  A wave
// therefore holds the K values of ln rho of its 64 rows in registers (8 per lane and row tile at K = 32), takes the row
// maxima with two shuffles and writes rho' = exp(ln rho - max) as whole 256-byte rows, and the maxima - what
// hmm_prep_kernel makes of the [K][npad] array, without the array: its round trip (2 x 2.56 GB at config 5) and a launch
// are gone.  The component loop is unrolled (the register index of a kept value must be static; ~40 instructions per
// component).  (A first form kept the row-per-lane vector-ALU emission of estep_rows16_f64 and its K values per row in
// LDS: 17 KB per wave = nine waves per CU, 4.7 ms against 2.4 + 1.0 for the two kernels it replaced.)
// No ln rho array is formed: what reads it (hmmvb_viterbi, the ln rho read-out) needs a pass with the other target
// (hmmvb_emission_target).
template <typename XT, bool VEC, int KT>
__global__ __launch_bounds__(256, 2) void hmm_emission_mfma16_kernel(const XT* __restrict__ x, int64_t ldx, int64_t T, int D,
                                                                  const double* __restrict__ img /*[K][img_doubles(1)]*/,
                                                                  const double* __restrict__ cvec, int K,
                                                                  double* __restrict__ rho_tm, double* __restrict__ mx) {
    constexpr int NB = KT == 1 ? 4 : 2, Kp = 16 * KT, IMG = img_doubles(1);      // (row tiles per wave: what the registers hold)
    __shared__ double s_exp2[64];
    if (threadIdx.x < 64) s_exp2[threadIdx.x] = kExp2Table64[threadIdx.x];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = lane & 15, g = lane >> 4;
    const int64_t n_tiles = (T + 16 * NB - 1) / (16 * NB);
    double cl[4 * KT];                                   // c of the components this lane keeps: 4 kq + g
#pragma unroll
    for (int kq = 0; kq < 4 * KT; ++kq) cl[kq] = cvec[4 * kq + g < K ? 4 * kq + g : 0];
    for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < n_tiles; tile += (int64_t)gridDim.x * 4) {
        int64_t ld[NB], stv[NB];
        tile_rows<NB>(tile * 16 * NB, n, T, ld, stv);
        XT xr[NB][1][4];
        load_x_tile<1, NB, XT, VEC>(x, ldx, D, ld, g, xr);
        double mine[NB][4 * KT];
        // The component loop is unrolled (static register indices) and the operands of component k + 1 are requested before
        // component k is computed.  Left alone the compiler requests ALL operands at the loop's top (8 doubles per lane and
        // component; loads from a const __restrict__ array are "invariant" and move across memory barriers, but not above the
        // empty asm their offset goes through) and issues the MFMAs of many components ahead, keeping their accumulators:
        // the asm statements on offsets and partial sums pin the order.  The f64 vector ALU does not run beside f64 MFMAs
        // (same units), so what is issued per component counts: the squares are summed in the lane, and every four
        // components one butterfly over the lane groups leaves component 4 kq + g's total in group g - where it is kept.
        // States past K (padding) run on component 0's operands and are discarded.
        auto fetch = [&](int kk) {
            int off = (kk < K ? kk : 0) * IMG;
            asm volatile("" : "+s"(off));
            return load_component16(img + off, lane, g);
        };
        Comp16 nxt = fetch(0);
        double part[4][NB];
#pragma unroll
        for (int k = 0; k < Kp; ++k) {
            const Comp16 cur = nxt;
            if (k + 1 < Kp) nxt = fetch(k + 1);
            d4 acc[NB];
            component16_mfma<NB, XT>(cur, xr, acc);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                double t = 0.0;
#pragma unroll
                for (int r = 0; r < 4; ++r) t = fma(acc[nb][r], acc[nb][r], t);
                asm volatile("" : "+v"(t));
                part[k & 3][nb] = t;
            }
            if ((k & 3) == 3) {
                const int kq = k >> 2;
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const double q = sum_groups_scatter4(part[0][nb], part[1][nb], part[2][nb], part[3][nb]);
                    mine[nb][kq] = 4 * kq + g < K ? cl[kq] - 0.5 * q : -__builtin_huge_val();
                }
            }
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            double m = mine[nb][0];
#pragma unroll
            for (int e = 1; e < 4 * KT; ++e) m = fmax(m, mine[nb][e]);
            m = max_groups(m);
            if (stv[nb] >= 0) {
#pragma unroll
                for (int it = 0; it < KT; ++it) {
                    d4 o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = 16 * it + g + 4 * r < K ? exp_nonpos_fast(mine[nb][4 * it + r] - m, s_exp2) : 0.0;
                    *reinterpret_cast<d4*>(rho_tm + stv[nb] * Kp + 16 * it + 4 * g) = o;
                }
                if (g == 0) mx[stv[nb]] = m;
            }
        }
    }
}

// A operand registers of a constant Kp x Kp matrix M for D' = M . (state-major operand):
//   aop[it][kt][s] = M[16 it + i][16 kt + g + 4 s]      lane = (i = l & 15, g = l >> 4)
template <int KT>
__device__ __forceinline__ void load_aop(const double* __restrict__ m, int ld, bool transpose, int K,
                                         double (&aop)[KT][KT][4]) {
    const int lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4;
#pragma unroll
    for (int it = 0; it < KT; ++it)
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int r = 16 * it + i, c = 16 * kt + g + 4 * s;
                double v = 0.0;
                if (r < K && c < K) v = transpose ? m[c * ld + r] : m[r * ld + c];
                aop[it][kt][s] = v;
            }
}

// out[it] = M . in  for one 16-column block (in/out in C/D layout: reg r of tile it = state 16 it + g + 4 r)
template <int KT>
__device__ __forceinline__ void apply(const double (&aop)[KT][KT][4], const d4 (&in)[KT], d4 (&out)[KT]) {
#pragma unroll
    for (int it = 0; it < KT; ++it) {
        d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int s = 0; s < 4; ++s) acc = mfma_f64(aop[it][kt][s], in[kt][s], acc);
        out[it] = acc;
    }
}

__device__ __forceinline__ double max_wave(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
    return v;
}

// H2: P_c = prod_{t = 1 + cL}^{min((c+1)L, T-1)} A~ diag(rho'_t).  One wave per chunk; P^T is kept as KT
// column blocks in C/D layout: P^T <- diag(rho'_t) A~^T P^T.  Rescaled (any positive factor) every 4 steps.
template <int KT>
__global__ __launch_bounds__(256) void hmm_chunk_products_kernel(const double* __restrict__ rho_tm,
                                                                 const double* __restrict__ a_tilde, int K, int64_t T,
                                                                 int64_t L, int64_t n_chunks,
                                                                 double* __restrict__ prod /*[n_chunks][Kp][Kp] (P, natural order)*/, const int* __restrict__ gate = nullptr) {
    if (gate != nullptr && *gate == 0) return;      // (the boundary vectors of the forgetting pass stand: hmm_capi.hip)
    constexpr int Kp = 16 * KT;
    const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
    const int64_t c = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= n_chunks) return;
    double aop[KT][KT][4];
    load_aop<KT>(a_tilde, K, /*transpose=*/true, K, aop);
    d4 pt[KT][KT];   // [column block jt][row tile it] of P^T
#pragma unroll
    for (int jt = 0; jt < KT; ++jt)
#pragma unroll
        for (int it = 0; it < KT; ++it)
#pragma unroll
            for (int r = 0; r < 4; ++r) pt[jt][it][r] = (it == jt && (g + 4 * r) == j) ? 1.0 : 0.0;
    const int64_t t0 = 1 + c * L;
    int64_t t1 = t0 + L;
    if (t1 > T) t1 = T;
    for (int64_t t = t0; t < t1; ++t) {
        d4 rho[KT];
#pragma unroll
        for (int it = 0; it < KT; ++it) rho[it] = *reinterpret_cast<const d4*>(rho_tm + t * Kp + 16 * it + 4 * g);
#pragma unroll
        for (int jt = 0; jt < KT; ++jt) {
            d4 nw[KT];
            apply<KT>(aop, pt[jt], nw);
#pragma unroll
            for (int it = 0; it < KT; ++it)
#pragma unroll
                for (int r = 0; r < 4; ++r) pt[jt][it][r] = nw[it][r] * rho[it][r];
        }
        if (((t - t0) & 3) == 3) {
            double m = 0.0;
#pragma unroll
            for (int jt = 0; jt < KT; ++jt)
#pragma unroll
                for (int it = 0; it < KT; ++it)
#pragma unroll
                    for (int r = 0; r < 4; ++r) m = fmax(m, pt[jt][it][r]);
            m = max_wave(m);
            const double sc = m > 0.0 ? 1.0 / m : 1.0;
#pragma unroll
            for (int jt = 0; jt < KT; ++jt)
#pragma unroll
                for (int it = 0; it < KT; ++it)
#pragma unroll
                    for (int r = 0; r < 4; ++r) pt[jt][it][r] *= sc;
        }
    }
    // P[row = column index of P^T][col = row index of P^T]
    double* out = prod + c * Kp * Kp;
#pragma unroll
    for (int jt = 0; jt < KT; ++jt)
#pragma unroll
        for (int it = 0; it < KT; ++it)
#pragma unroll
            for (int r = 0; r < 4; ++r) out[(16 * jt + j) * Kp + 16 * it + g + 4 * r] = pt[jt][it][r];
}

// H3: sequential pass over chunk boundaries; wave 0 forward, wave 1 backward (natural state order, lane =
// state, the running vector lives in one register per lane and is broadcast with shuffles: no LDS, no barrier).
// The lane's column (forward) / row (backward) of the NEXT chunk product is loaded while the current one is
// multiplied, so the dependent chain per chunk is Kp shuffles + FMAs, not Kp memory round trips.
//   fstart[c] = alpha at t = c L (normalised), fstart[0] = alpha_0;  cprime[0] = c'_0
//   bend[c]   = beta~ at the last step of chunk c (normalised to sum 1), bend[last] = uniform
//   T == 1 (no chunks): gamma_0 = alpha_0, w_0 = 0 are written here.
template <int KT>
__global__ __launch_bounds__(128) void hmm_boundary_scan_kernel(const double* __restrict__ rho_tm,
                                                                const double* __restrict__ pi_tilde,
                                                                const double* __restrict__ prod, int K,
                                                                int64_t n_chunks, double* __restrict__ fstart,
                                                                double* __restrict__ bend, double* __restrict__ cprime,
                                                                double* __restrict__ alpha_tm, double* __restrict__ gamma_tm,
                                                                double* __restrict__ w_tm, const int* __restrict__ gate = nullptr) {
    if (gate != nullptr && *gate == 0) return;      // (the boundary vectors of the forgetting pass stand: hmm_capi.hip)
    constexpr int Kp = 16 * KT;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int ln = lane < Kp ? lane : 0;
    double cur[Kp], nxt[Kp];
    if (wave == 0) {
        double v = lane < K ? rho_tm[hmm_pos(lane)] * pi_tilde[lane] : 0.0;
        const double s = sum_wave(v);
        if (lane == 0) cprime[0] = s;
        v = s > 0.0 ? v / s : 0.0;
        if (lane < Kp) {
            fstart[lane] = v;
            if (n_chunks == 0) {
                alpha_tm[hmm_pos(lane)] = v;
                gamma_tm[hmm_pos(lane)] = v;
                w_tm[hmm_pos(lane)] = 0.0;
            }
        }
        if (n_chunks > 1) {
#pragma unroll
            for (int i = 0; i < Kp; ++i) nxt[i] = prod[i * Kp + ln];
        }
        for (int64_t c = 0; c + 1 < n_chunks; ++c) {
#pragma unroll
            for (int i = 0; i < Kp; ++i) cur[i] = nxt[i];
            if (c + 2 < n_chunks) {
                const double* P = prod + (c + 1) * Kp * Kp;
#pragma unroll
                for (int i = 0; i < Kp; ++i) nxt[i] = P[i * Kp + ln];
            }
            double acc = 0.0;
#pragma unroll
            for (int i = 0; i < Kp; ++i) acc = fma(__shfl(v, i), cur[i], acc);
            if (lane >= Kp) acc = 0.0;
            const double tot = sum_wave(acc);
            v = tot > 0.0 ? acc / tot : 0.0;
            if (lane < Kp) fstart[(c + 1) * Kp + lane] = v;
        }
    } else {
        double v = lane < K ? 1.0 / K : 0.0;
        if (lane < Kp && n_chunks > 0) bend[(n_chunks - 1) * Kp + lane] = v;
        if (n_chunks > 1) {
            const double* P = prod + (n_chunks - 1) * Kp * Kp;
#pragma unroll
            for (int jj = 0; jj < Kp; ++jj) nxt[jj] = P[ln * Kp + jj];
        }
        for (int64_t c = n_chunks - 1; c >= 1; --c) {
#pragma unroll
            for (int jj = 0; jj < Kp; ++jj) cur[jj] = nxt[jj];
            if (c >= 2) {
                const double* P = prod + (c - 1) * Kp * Kp;
#pragma unroll
                for (int jj = 0; jj < Kp; ++jj) nxt[jj] = P[ln * Kp + jj];
            }
            double acc = 0.0;
#pragma unroll
            for (int jj = 0; jj < Kp; ++jj) acc = fma(cur[jj], __shfl(v, jj), acc);
            if (lane >= Kp) acc = 0.0;
            const double tot = sum_wave(acc);
            v = tot > 0.0 ? acc / tot : 0.0;
            if (lane < Kp) bend[(c - 1) * Kp + lane] = v;
        }
    }
}

// ---- two-level boundary pass (long sequences) -------------------------------------------------------------------------
// With one level the chunk length has to balance the sequential pass over T / L chunk products against the L
// dependent steps of a replay wave (L ~ sqrt(T): 2048 at T = 1e7, 4883 sequential products = 5.3 ms, and only 77
// replay workgroups).  With two levels the chunks are short (256 steps: 39063 chunks, 610 replay workgroups), the
// products of kHmmSuper consecutive chunks are multiplied up in parallel (H3a), the sequential pass runs over the 611
// super-chunk products (H3 itself), and every super-chunk then fills in its own chunks' start / end vectors (H3b).
#ifndef GMMVB_HMM_SUPER
#define GMMVB_HMM_SUPER 64
#endif
#ifndef GMMVB_HMM_LONG_CHUNK
#define GMMVB_HMM_LONG_CHUNK 256
#endif
constexpr int kHmmSuper = GMMVB_HMM_SUPER;
constexpr int kHmmLongChunk = GMMVB_HMM_LONG_CHUNK;      // chunk length of sequences past 2^15 steps (kHmmLongFrom, hmm_capi.hip) (two-level boundary pass)

// H3a: Q_s = P_{sG} P_{sG+1} ... (G = kHmmSuper chunks), rescaled to max 1 after every product.  One workgroup per s.
template <int KT>
__global__ __launch_bounds__(256) void hmm_super_products_kernel(const double* __restrict__ prod, int64_t n_chunks,
                                                                 double* __restrict__ qprod /*[n_super][Kp][Kp]*/, const int* __restrict__ gate = nullptr) {
    if (gate != nullptr && *gate == 0) return;      // (the boundary vectors of the forgetting pass stand: hmm_capi.hip)
    constexpr int Kp = 16 * KT, E = (Kp * Kp + 255) / 256;
    __shared__ double A[Kp * Kp], B[Kp * Kp];
    __shared__ double red[4];
    const int tid = threadIdx.x;
    const int64_t c0 = (int64_t)blockIdx.x * kHmmSuper;
    const int64_t c1 = c0 + kHmmSuper < n_chunks ? c0 + kHmmSuper : n_chunks;
    for (int e = tid; e < Kp * Kp; e += 256) A[e] = prod[c0 * Kp * Kp + e];
    for (int64_t c = c0 + 1; c < c1; ++c) {
        for (int e = tid; e < Kp * Kp; e += 256) B[e] = prod[c * Kp * Kp + e];
        __syncthreads();
        double out[E];
        double m = 0.0;
#pragma unroll
        for (int q = 0; q < E; ++q) {
            const int e = tid + 256 * q;
            double acc = 0.0;
            if (e < Kp * Kp) {
                const int i = e / Kp, j = e % Kp;
                for (int p2 = 0; p2 < Kp; ++p2) acc = fma(A[i * Kp + p2], B[p2 * Kp + j], acc);
            }
            out[q] = acc;
            m = fmax(m, acc);
        }
        m = max_wave(m);
        if ((tid & 63) == 0) red[tid >> 6] = m;
        __syncthreads();                                   // (also: every read of A and B is done)
        m = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
        const double sc = m > 0.0 ? 1.0 / m : 1.0;
#pragma unroll
        for (int q = 0; q < E; ++q) {
            const int e = tid + 256 * q;
            if (e < Kp * Kp) A[e] = out[q] * sc;
        }
        __syncthreads();
    }
    __syncthreads();
    for (int e = tid; e < Kp * Kp; e += 256) qprod[(int64_t)blockIdx.x * Kp * Kp + e] = A[e];
}

// H3b: from the super-chunk boundary vectors (H3 run on the Q_s) to every chunk's: wave 0 forward, wave 1 backward, the
// loops of H3 over the chunks of one super-chunk.  fstart_s[s] = alpha at the start of super-chunk s, bend_s[s] = beta~ at
// the end of its last chunk.
template <int KT>
__global__ __launch_bounds__(128) void hmm_boundary_fill_kernel(const double* __restrict__ prod, int64_t n_chunks,
                                                                const double* __restrict__ fstart_s,
                                                                const double* __restrict__ bend_s,
                                                                double* __restrict__ fstart, double* __restrict__ bend, const int* __restrict__ gate = nullptr) {
    if (gate != nullptr && *gate == 0) return;      // (the boundary vectors of the forgetting pass stand: hmm_capi.hip)
    constexpr int Kp = 16 * KT;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int ln = lane < Kp ? lane : 0;
    const int64_t s = blockIdx.x;
    const int64_t c0 = s * kHmmSuper;
    const int64_t c1 = c0 + kHmmSuper < n_chunks ? c0 + kHmmSuper : n_chunks;
    double cur[Kp], nxt[Kp];
    if (wave == 0) {
        double v = lane < Kp ? fstart_s[s * Kp + lane] : 0.0;
        if (lane < Kp) fstart[c0 * Kp + lane] = v;
        if (c0 + 1 < c1) {
#pragma unroll
            for (int i = 0; i < Kp; ++i) nxt[i] = prod[c0 * Kp * Kp + i * Kp + ln];
        }
        for (int64_t c = c0; c + 1 < c1; ++c) {
#pragma unroll
            for (int i = 0; i < Kp; ++i) cur[i] = nxt[i];
            if (c + 2 < c1) {
                const double* P = prod + (c + 1) * Kp * Kp;
#pragma unroll
                for (int i = 0; i < Kp; ++i) nxt[i] = P[i * Kp + ln];
            }
            double acc = 0.0;
#pragma unroll
            for (int i = 0; i < Kp; ++i) acc = fma(__shfl(v, i), cur[i], acc);
            if (lane >= Kp) acc = 0.0;
            const double tot = sum_wave(acc);
            v = tot > 0.0 ? acc / tot : 0.0;
            if (lane < Kp) fstart[(c + 1) * Kp + lane] = v;
        }
    } else {
        double v = lane < Kp ? bend_s[s * Kp + lane] : 0.0;
        if (lane < Kp) bend[(c1 - 1) * Kp + lane] = v;
        if (c1 - 1 > c0) {
            const double* P = prod + (c1 - 1) * Kp * Kp;
#pragma unroll
            for (int jj = 0; jj < Kp; ++jj) nxt[jj] = P[ln * Kp + jj];
        }
        for (int64_t c = c1 - 1; c > c0; --c) {
#pragma unroll
            for (int jj = 0; jj < Kp; ++jj) cur[jj] = nxt[jj];
            if (c - 1 > c0) {
                const double* P = prod + (c - 1) * Kp * Kp;
#pragma unroll
                for (int jj = 0; jj < Kp; ++jj) nxt[jj] = P[ln * Kp + jj];
            }
            double acc = 0.0;
#pragma unroll
            for (int jj = 0; jj < Kp; ++jj) acc = fma(cur[jj], __shfl(v, jj), acc);
            if (lane >= Kp) acc = 0.0;
            const double tot = sum_wave(acc);
            v = tot > 0.0 ? acc / tot : 0.0;
            if (lane < Kp) bend[(c - 1) * Kp + lane] = v;
        }
    }
}

#ifndef GMMVB_REPLAY_CHUNKS
#define GMMVB_REPLAY_CHUNKS 16
#endif
constexpr int kReplayChunks = GMMVB_REPLAY_CHUNKS;      // chunks per replay wave (8 or 16)
#ifndef GMMVB_HMM_REPLAY_AHEAD
#define GMMVB_HMM_REPLAY_AHEAD 4
#endif
constexpr int kHmmReplayAhead = GMMVB_HMM_REPLAY_AHEAD;      // steps whose rho' rows a replay / sweep wave keeps in flight
// H4: forward replay.  One wave = kReplayChunks chunks (MFMA columns).  alpha_tm / rho_tm in lane order.
template <int KT>
__device__ __forceinline__ void hmm_forward_replay_body(const double* __restrict__ rho_tm, const double* __restrict__ a_tilde, int K,
                                                        int64_t T, int64_t L, int64_t n_chunks, const double* __restrict__ fstart,
                                                        double* __restrict__ alpha_tm, double* __restrict__ cprime, int sweep,
                                                        double* __restrict__ end_out, int64_t s_from = 0 /*sweep: first step walked*/) {
    constexpr int Kp = 16 * KT;
    const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
    // kReplayChunks chunks per wave: with fewer than 16 the MFMA columns j and j + kReplayChunks carry the same chunk (only
    // the first copy stores).  Measured with 8 (twice as many waves for these chains of dependent steps): 24.7 / 25.0 ms per
    // iteration against 24.4 / 24.9 with 16 - no gain, 16 it stays (round 4)
    const int64_t c = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * kReplayChunks + (j % kReplayChunks);
    const bool live = c < n_chunks;
    const bool first_copy = j < kReplayChunks;
    double aop[KT][KT][4];
    load_aop<KT>(a_tilde, K, /*transpose=*/true, K, aop);      // alpha^T_new = A~^T alpha^T_old
    d4 al[KT];
#pragma unroll
    for (int it = 0; it < KT; ++it)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            al[it][r] = !live ? 0.0 : ((sweep && (c > 0 || s_from > 0)) ? (16 * it + g + 4 * r < K ? 1.0 / K : 0.0) : fstart[c * Kp + 16 * it + g + 4 * r]);
    if (live && c == 0 && !sweep) {                              // alpha_0 itself is part of the output
#pragma unroll
        for (int it = 0; it < KT; ++it) *reinterpret_cast<d4*>(alpha_tm + 16 * it + 4 * g) = al[it];
    }
    const int64_t t0 = 1 + c * L;
    // The wave's 16 columns read 16 rows of rho' that lie a chunk apart - 16 DRAM pages per step - and a step is ~20 dependent
    // MFMAs: without help every step waited out an HBM round trip (2.8 us per step at config 5, five times its arithmetic).
    // The rows of the next kHmmReplayAhead steps are kept in flight in registers (round 5; a rotating buffer, the loop
    // unrolled by its depth so that every slot is a fixed register).
    constexpr int PF = kHmmReplayAhead;
    d4 rbuf[PF][KT];
    auto fetch = [&](int64_t s, d4 (&dst)[KT]) {      // (unconditional loads from a clamped row: a predicated load would make
        const int64_t t = t0 + s;                       // the compiler wait for everything in flight; the consumer masks)
        const int64_t tc = (live && s < L && t < T) ? t : 0;
#pragma unroll
        for (int it = 0; it < KT; ++it) dst[it] = *reinterpret_cast<const d4*>(rho_tm + tc * Kp + 16 * it + 4 * g);
    };
#pragma unroll
    for (int p = 0; p < PF; ++p) fetch(s_from + p, rbuf[p]);
    for (int64_t s0 = s_from; s0 < L; s0 += PF) {
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            const int64_t s = s0 + p;
            if (s >= L) break;                       // (L - s_from need not be a multiple of the depth)
            const int64_t t = t0 + s;
            const bool on = live && t < T;
            const bool st = on && first_copy && !sweep;
            d4 nw[KT];
            apply<KT>(aop, al, nw);
            double part = 0.0;
            d4 rho[KT];
#pragma unroll
            for (int it = 0; it < KT; ++it) rho[it] = on ? rbuf[p][it] : d4{0.0, 0.0, 0.0, 0.0};
            fetch(s + PF, rbuf[p]);                  // the slot's next tenant, kHmmReplayAhead steps ahead
#pragma unroll
            for (int it = 0; it < KT; ++it) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    nw[it][r] *= rho[it][r];
                    part += nw[it][r];
                }
            }
            const double cp = sum_groups(part);
            const double inv = cp > 0.0 ? 1.0 / cp : 0.0;
#pragma unroll
            for (int it = 0; it < KT; ++it) {
#pragma unroll
                for (int r = 0; r < 4; ++r) al[it][r] = nw[it][r] * inv;
                if (st) *reinterpret_cast<d4*>(alpha_tm + t * Kp + 16 * it + 4 * g) = al[it];
            }
            if (st && g == 0) cprime[t] = cp;
        }
    }
    if (end_out != nullptr && live && first_copy && c + 1 < n_chunks) {      // (every chunk but the last is whole)
#pragma unroll
        for (int it = 0; it < KT; ++it)
#pragma unroll
            for (int r = 0; r < 4; ++r) end_out[(c + 1) * Kp + 16 * it + g + 4 * r] = al[it][r];
    }
}

template <int KT>
__global__ __launch_bounds__(256) void hmm_forward_replay_kernel(const double* __restrict__ rho_tm,
                                                                 const double* __restrict__ a_tilde, int K, int64_t T,
                                                                 int64_t L, int64_t n_chunks,
                                                                 const double* __restrict__ fstart,
                                                                 double* __restrict__ alpha_tm, double* __restrict__ cprime,
                                                                 int sweep = 0 /*1: no stores, chunks past the first start from the uniform vector*/,
                                                                 double* __restrict__ end_out = nullptr /*[n_chunks][Kp]: alpha behind chunk c -> row c + 1*/,
                                                                 const int* __restrict__ gate = nullptr) {
    if (gate != nullptr && *gate == 0) return;
    hmm_forward_replay_body<KT>(rho_tm, a_tilde, K, T, L, n_chunks, fstart, alpha_tm, cprime, sweep, end_out);
}

// The backward recursion alone (no gamma, no xi): beta~ in front of chunk c from the uniform vector behind it -> row c - 1.
template <int KT>
__device__ __forceinline__ void hmm_backward_sweep_body(const double* __restrict__ rho_tm, const double* __restrict__ a_tilde, int K,
                                                        int64_t T, int64_t L, int64_t n_chunks, double* __restrict__ bend_out,
                                                        int64_t W /*steps walked: the chunk's first W, from the uniform vector behind them*/) {
    constexpr int Kp = 16 * KT;
    const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
    const int64_t c = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * kReplayChunks + (j % kReplayChunks);
    const bool live = c < n_chunks;
    double aop[KT][KT][4];
    load_aop<KT>(a_tilde, K, /*transpose=*/false, K, aop);
    d4 be[KT];
#pragma unroll
    for (int it = 0; it < KT; ++it)
#pragma unroll
        for (int r = 0; r < 4; ++r) be[it][r] = (live && 16 * it + g + 4 * r < K) ? 1.0 / K : 0.0;
    const int64_t t0 = 1 + c * L;
    for (int64_t s = W - 1; s >= 0; --s) {
        const int64_t t = t0 + s;
        const bool on = live && t < T;
        d4 y[KT];
#pragma unroll
        for (int it = 0; it < KT; ++it) {
            d4 rho = {0.0, 0.0, 0.0, 0.0};
            if (on) rho = *reinterpret_cast<const d4*>(rho_tm + t * Kp + 16 * it + 4 * g);
#pragma unroll
            for (int r = 0; r < 4; ++r) y[it][r] = rho[r] * be[it][r];
        }
        d4 nb[KT];
        apply<KT>(aop, y, nb);
        double part = 0.0;
#pragma unroll
        for (int it = 0; it < KT; ++it)
#pragma unroll
            for (int r = 0; r < 4; ++r) part += nb[it][r];
        const double tot = sum_groups(part);
        const double inv = tot > 0.0 ? 1.0 / tot : 0.0;
        if (on) {
#pragma unroll
            for (int it = 0; it < KT; ++it)
#pragma unroll
                for (int r = 0; r < 4; ++r) be[it][r] = nb[it][r] * inv;
        }
    }
    if (live && j < kReplayChunks && c >= 1) {
#pragma unroll
        for (int it = 0; it < KT; ++it)
#pragma unroll
            for (int r = 0; r < 4; ++r) bend_out[(c - 1) * Kp + 16 * it + g + 4 * r] = be[it][r];
    }
}

// The two sweeps of the forgetting pass in one launch (blockIdx.y = direction): chains of dependent steps both, twice the waves
// in flight.
// W (round 5): a recursion that forgets its start does so within a few informative steps, not within a chunk: the sweeps walk
// only the W steps next to the boundary they are after - the LAST W of chunk c - 1 for alpha in front of chunk c, the FIRST W
// of chunk c for beta~ in front of it - from the uniform vector.  The replays' own boundary vectors are compared with these
// exactly as before (hmm_boundary_check_kernel), so a W that is too short for the sequence opens a gate like any other
// start vector that does not stand - the FIRST of two: behind it hmm_capi.hip has enqueued the whole-chunk sweeps, the replays
// and their check, and only if those do not stand either does the second gate open the chunk-product path.  W = 64 at
// config 5: a quarter of the sweeps' steps and of their 5 GB (1.0 -> 0.3 ms; 9.8 against 10.6 ms per iteration).  W = 32 was
// not always enough there: with duplicate components (two states with nearly the same emission) the split between them is
// forgotten at the chain's own mixing rate, 0.9 per step - one pass in seven of the benchmark's needed the whole chunks.
template <int KT>
__global__ __launch_bounds__(256) void hmm_sweeps_kernel(const double* __restrict__ rho_tm, const double* __restrict__ a_tilde, int K,
                                                         int64_t T, int64_t L, int64_t n_chunks, double* __restrict__ fstart,
                                                         double* __restrict__ bend, int64_t W, const int* __restrict__ gate = nullptr) {
    if (gate != nullptr && *gate == 0) return;      // (the second stage of the forgetting pass: only if the first did not stand)
    if (blockIdx.y == 0)
        hmm_forward_replay_body<KT>(rho_tm, a_tilde, K, T, L, n_chunks, fstart, nullptr, nullptr, 1, fstart, L - W);
    else
        hmm_backward_sweep_body<KT>(rho_tm, a_tilde, K, T, L, n_chunks, bend, W);
}

// alpha_0 and c'_0 (what hmm_boundary_scan_kernel does first), and the uniform vector behind the last chunk (K <= 256)
__global__ __launch_bounds__(256) void hmm_alpha0_kernel(const double* __restrict__ rho_tm, const double* __restrict__ pi_tilde, int K,
                                                        int Kp, int64_t n_chunks, double* __restrict__ fstart,
                                                        double* __restrict__ bend, double* __restrict__ cprime) {
    __shared__ double red[4];
    const int tid = threadIdx.x;
    double v = tid < K ? rho_tm[hmm_pos(tid)] * pi_tilde[tid] : 0.0;
    const double w = sum_wave(v);
    if ((tid & 63) == 0) red[tid >> 6] = w;
    __syncthreads();
    const double s = (red[0] + red[1]) + (red[2] + red[3]);
    if (tid == 0) cprime[0] = s;
    v = s > 0.0 ? v / s : 0.0;
    if (tid < Kp) {
        fstart[tid] = v;
        if (n_chunks > 0) bend[(n_chunks - 1) * Kp + tid] = tid < K ? 1.0 / K : 0.0;
    }
}

// The forgetting pass's test: the boundary vectors the replays arrived at (started from the sweeps' vectors) against the
// sweeps' own (started from the uniform vector).  *gate = 1 if they differ: the products path runs.
// REL (the forward-backward pass; round 5): the comparison is RELATIVE, entry by entry - |a - b| <= tol max(a, b) - i.e. a test
// in the projective (Hilbert) metric d(a, b) = max_ij ln(a_i b_j / (a_j b_i)) <= 2 ln((1 + tol) / (1 - tol)).  The normalised
// recursions alpha' = rho' o (A~^T alpha) / sum are non-expansive in THAT metric (Birkhoff), not in the sup norm: an absolute
// test (rounds 4's 2e-14) cannot see an entry that is truly 1e-40 where the uniform start re-seeded 1e-15, and a following
// chunk whose emissions favour that state can amplify exactly that entry to O(1) (a~_ij ~ exp(psi(0.01)) ~ 1e-44 is reachable
// with a sparse h0_zeta prior).  With the relative test the induction closes: d(s_c, a_c) <= d(s_c, r_c) + d(r_c, a_c)
// <= tol' + d(s_(c-1), a_(c-1)) (s: sweep end = the start the next replay used, r: replay end, a: the exact vector), so the
// start vectors are within chunks x tol' of the exact ones in the Hilbert metric (8e-9 at 4e4 chunks), which bounds the
// relative error of every gamma / xi entry.  Entries below 1e-290 in both vectors are exempt (sub-normal range: their
// relative precision is gone, and no chunk can amplify by 1e290 - rho' itself underflows to 0 long before).
// !REL (the Viterbi pass's coalescence test on omega - max, log domain): absolute, max-plus maps are 1-Lipschitz in the sup norm.
template <bool REL>
__global__ __launch_bounds__(256) void hmm_boundary_check_kernel(const double* __restrict__ fa, const double* __restrict__ fb,
                                                                 const double* __restrict__ ba, const double* __restrict__ bb,
                                                                 int64_t n /*entries of rows 1 .. n_chunks - 1 (forward), 0 .. n_chunks - 2 (backward)*/,
                                                                 int Kp, double tol, int* __restrict__ gate,
                                                                 const int* __restrict__ gate_in = nullptr) {
    if (gate_in != nullptr && *gate_in == 0) return;
    bool bad = false;
    auto differs = [tol](double a, double b) {
        const double d = fabs(a - b);
        if constexpr (REL) {
            const double m = fmax(fabs(a), fabs(b));
            return !(d <= tol * m || m < 1e-290);            // (NaN: differs)
        } else {
            return !(d <= tol);
        }
    };
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256)
        bad = bad || differs(fa[Kp + e], fb[Kp + e]) || differs(ba[e], bb[e]);
    if (__builtin_amdgcn_ballot_w64(bad) != 0ull && (threadIdx.x & 63) == 0) *gate = 1;
}

// H5: backward replay, descending in time.  Writes gamma_tm (lane order) and w_tm; gamma_0 too.
// XI (round 4): the xi sum of H6 inside this kernel, and no w_tm.  The term of time t, alpha_{t-1} (x) w_t, pairs the w_t this
// iteration makes with the alpha row the NEXT iteration loads anyway; summed over the wave's 16 chunks it is a
// [Kp x 16] x [16 x Kp] product with the chunk as contraction index - four MFMAs per output tile - whose operands are the
// transposes of what the lanes hold (chunk on lane & 15, positions on lane >> 4 and registers): both go through a
// per-wave LDS tile (2 x 16 x (Kp + 2) doubles), a d4 write per 16-state block and lane, 8-byte reads.  The replay runs
// at a tenth of the matrix pipe and ~4 TB/s: the extra MFMAs and LDS traffic ride in its shadow, and the w array's
// round trip (written here, read by hmm_xi_sum_kernel with alpha: 7.7 GB per iteration at config 5) and a launch are
// gone.  One slab per wave (same layout as H6's: hmm_finish_kernel adds them in wave order).  The xi read-out derives
// w_t = rho'_t gamma_t / (alpha_t c'_t) instead (hmm_readout_kernel).
constexpr int hmm_xi_ldw(int Kp) { return Kp + 2; }
template <int KT, bool XI = false>
__global__ __launch_bounds__(256, XI ? 3 : 1) void hmm_backward_replay_kernel(const double* __restrict__ rho_tm,
                                                                  const double* __restrict__ a_tilde, int K, int64_t T,
                                                                  int64_t L, int64_t n_chunks,
                                                                  const double* __restrict__ bend,
                                                                  const double* __restrict__ alpha_tm,
                                                                  const double* __restrict__ cprime,
                                                                  double* __restrict__ gamma_tm, double* __restrict__ w_tm,
                                                                  double* __restrict__ xi_slabs = nullptr,
                                                                  double* __restrict__ bend_out = nullptr /*beta~ in front of chunk c -> row c - 1*/,
                                                                  const int* __restrict__ gate = nullptr) {
    if (gate != nullptr && *gate == 0) return;
    constexpr int Kp = 16 * KT;
    constexpr int LDW = hmm_xi_ldw(Kp);
    static_assert(!XI || kReplayChunks == 16, "the fused xi sum takes the 16 MFMA columns as 16 distinct chunks");
    __shared__ __attribute__((aligned(16))) double xs[XI ? 4 * 2 * 16 * LDW : 2];
    const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
    double* const sa = xs + (XI ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) * 2 * 16 * LDW : 0);
    double* const sb = sa + (XI ? 16 * LDW : 0);
    d4 xacc[XI ? KT : 1][XI ? KT : 1];
    d4 wprev[XI ? KT : 1];
    if constexpr (XI) {
#pragma unroll
        for (int it = 0; it < KT; ++it) {
            wprev[it] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int jt = 0; jt < KT; ++jt) xacc[it][jt] = d4{0.0, 0.0, 0.0, 0.0};
        }
    }
    // xacc += sum over the wave's chunks of al (x) wv   (al, wv: one row per chunk, positions as the lanes hold them)
    auto xi_add = [&](const d4 (&al)[KT], const d4 (&wv)[XI ? KT : 1]) {
        if constexpr (XI) {
#pragma unroll
            for (int it = 0; it < KT; ++it) {
                *reinterpret_cast<d4*>(sa + j * LDW + 16 * it + 4 * g) = al[it];
                *reinterpret_cast<d4*>(sb + j * LDW + 16 * it + 4 * g) = wv[it];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                double a[KT], b[KT];
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) {
                    a[kt] = sa[(4 * q + g) * LDW + 16 * kt + j];
                    b[kt] = sb[(4 * q + g) * LDW + 16 * kt + j];
                }
#pragma unroll
                for (int it = 0; it < KT; ++it)
#pragma unroll
                    for (int jt = 0; jt < KT; ++jt) xacc[it][jt] = mfma_f64(a[it], b[jt], xacc[it][jt]);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    };
    const int64_t c = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * kReplayChunks + (j % kReplayChunks);
    const bool live = c < n_chunks;
    const bool first_copy = j < kReplayChunks;
    double aop[KT][KT][4];
    load_aop<KT>(a_tilde, K, /*transpose=*/false, K, aop);     // beta_{t-1} ~ A~ (rho'_t o beta_t)
    d4 be[KT];
#pragma unroll
    for (int it = 0; it < KT; ++it)
#pragma unroll
        for (int r = 0; r < 4; ++r) be[it][r] = live ? bend[c * Kp + 16 * it + g + 4 * r] : 0.0;
    const int64_t t0 = 1 + c * L;
    for (int64_t s = L - 1; s >= 0; --s) {
        const int64_t t = t0 + s;
        const bool on = live && t < T;
        // gamma_t, w_t from beta~_t (registers), alpha_t, rho'_t, c'_t
        d4 y[KT];
        double dot = 0.0;
        d4 al[KT], rho[KT];
#pragma unroll
        for (int it = 0; it < KT; ++it) {
            al[it] = d4{0.0, 0.0, 0.0, 0.0};
            rho[it] = d4{0.0, 0.0, 0.0, 0.0};
            if (on) {
                al[it] = *reinterpret_cast<const d4*>(alpha_tm + t * Kp + 16 * it + 4 * g);
                rho[it] = *reinterpret_cast<const d4*>(rho_tm + t * Kp + 16 * it + 4 * g);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) dot = fma(al[it][r], be[it][r], dot);
        }
        dot = sum_groups(dot);
        const double cp = on ? cprime[t] : 1.0;
        const double ginv = dot > 0.0 ? 1.0 / dot : 0.0;
        const double winv = (dot > 0.0 && cp > 0.0) ? 1.0 / (dot * cp) : 0.0;
        d4 wnow[XI ? KT : 1];
#pragma unroll
        for (int it = 0; it < KT; ++it) {
            d4 gm, ww;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                y[it][r] = rho[it][r] * be[it][r];
                gm[r] = al[it][r] * be[it][r] * ginv;
                ww[r] = y[it][r] * winv;
            }
            if (on && first_copy) {
                *reinterpret_cast<d4*>(gamma_tm + t * Kp + 16 * it + 4 * g) = gm;
                if constexpr (!XI) *reinterpret_cast<d4*>(w_tm + t * Kp + 16 * it + 4 * g) = ww;
            }
            if constexpr (XI) wnow[it] = ww;              // (zero where the column is off: rho' = 0 there)
        }
        if constexpr (XI) {
            if (s < L - 1) xi_add(al, wprev);             // the term of time t + 1: alpha_t (x) w_{t+1}
#pragma unroll
            for (int it = 0; it < KT; ++it) wprev[it] = wnow[it];
        }
        // beta~_{t-1} ~ A~ y, renormalised to sum 1 (columns that are off keep their start vector)
        d4 nb[KT];
        apply<KT>(aop, y, nb);
        double part = 0.0;
#pragma unroll
        for (int it = 0; it < KT; ++it)
#pragma unroll
            for (int r = 0; r < 4; ++r) part += nb[it][r];
        const double tot = sum_groups(part);
        const double inv = tot > 0.0 ? 1.0 / tot : 0.0;
        if (on) {
#pragma unroll
            for (int it = 0; it < KT; ++it)
#pragma unroll
                for (int r = 0; r < 4; ++r) be[it][r] = nb[it][r] * inv;
        }
    }
    if (bend_out != nullptr && live && first_copy && c >= 1) {
#pragma unroll
        for (int it = 0; it < KT; ++it)
#pragma unroll
            for (int r = 0; r < 4; ++r) bend_out[(c - 1) * Kp + 16 * it + g + 4 * r] = be[it][r];
    }
    if constexpr (XI) {                                         // the term of the chunk's first step: alpha_{t0 - 1} (x) w_{t0}
        d4 al0[KT];
#pragma unroll
        for (int it = 0; it < KT; ++it)
            al0[it] = (live && t0 < T) ? *reinterpret_cast<const d4*>(alpha_tm + (t0 - 1) * Kp + 16 * it + 4 * g)
                                       : d4{0.0, 0.0, 0.0, 0.0};
        xi_add(al0, wprev);
        double* out = xi_slabs + ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * Kp * Kp;
#pragma unroll
        for (int it = 0; it < KT; ++it)
#pragma unroll
            for (int jt = 0; jt < KT; ++jt)
#pragma unroll
                for (int r = 0; r < 4; ++r) out[(16 * it + g + 4 * r) * Kp + 16 * jt + j] = xacc[it][jt][r];
    }
    if (live && c == 0) {                                       // gamma_0 = alpha_0 o beta~_0, normalised
        double dot = 0.0;
        d4 al[KT];
#pragma unroll
        for (int it = 0; it < KT; ++it) {
            al[it] = *reinterpret_cast<const d4*>(alpha_tm + 16 * it + 4 * g);
#pragma unroll
            for (int r = 0; r < 4; ++r) dot = fma(al[it][r], be[it][r], dot);
        }
        // only the 4 lanes with j == 0 hold chunk 0: reduce over the lane groups of THIS column
        dot = sum_groups(dot);
        const double ginv = dot > 0.0 ? 1.0 / dot : 0.0;
#pragma unroll
        for (int it = 0; it < KT; ++it) {
            d4 gm;
#pragma unroll
            for (int r = 0; r < 4; ++r) gm[r] = al[it][r] * be[it][r] * ginv;
            *reinterpret_cast<d4*>(gamma_tm + 16 * it + 4 * g) = gm;
            if constexpr (!XI) *reinterpret_cast<d4*>(w_tm + 16 * it + 4 * g) = d4{0.0, 0.0, 0.0, 0.0};      // xi_0 = 0
        }
    }
}

// H6: raw[pi][pj] = sum_{t=1}^{T-1} alpha_tm[t-1][pi] * w_tm[t][pj] (lane-order positions), slabs per wave.
template <int KT>
__global__ __launch_bounds__(256) void hmm_xi_sum_kernel(const double* __restrict__ alpha_tm,
                                                         const double* __restrict__ w_tm, int64_t T,
                                                         int64_t steps_per_wave, double* __restrict__ slabs) {
    constexpr int Kp = 16 * KT;
    const int lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4;
    const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t lo = 1 + w * steps_per_wave;
    int64_t hi = lo + steps_per_wave;
    if (hi > T) hi = T;
    d4 acc[KT][KT];
#pragma unroll
    for (int it = 0; it < KT; ++it)
#pragma unroll
        for (int jt = 0; jt < KT; ++jt) acc[it][jt] = d4{0.0, 0.0, 0.0, 0.0};
    // the operands of step group t + 4 are requested before the MFMAs of group t (one group in flight per wave)
    double an[KT], bn[KT];
    auto fetch = [&](int64_t t) {
        const int64_t tt = t + g;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            an[kt] = tt < hi ? alpha_tm[(tt - 1) * Kp + 16 * kt + i] : 0.0;
            bn[kt] = tt < hi ? w_tm[tt * Kp + 16 * kt + i] : 0.0;
        }
    };
    fetch(lo);
    for (int64_t t = lo; t < hi; t += 4) {
        double a[KT], b[KT];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            a[kt] = an[kt];
            b[kt] = bn[kt];
        }
        fetch(t + 4);
#pragma unroll
        for (int it = 0; it < KT; ++it)
#pragma unroll
            for (int jt = 0; jt < KT; ++jt) acc[it][jt] = mfma_f64(a[it], b[jt], acc[it][jt]);
    }
    double* out = slabs + w * Kp * Kp;
#pragma unroll
    for (int it = 0; it < KT; ++it)
#pragma unroll
        for (int jt = 0; jt < KT; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) out[(16 * it + g + 4 * r) * Kp + 16 * jt + i] = acc[it][jt][r];
}

// partial[b] = sum over the b-th slice of t of (ln c'_t + mx_t); fixed partition and tree => deterministic
__global__ __launch_bounds__(256) void hmm_lnc_partial_kernel(const double* __restrict__ cprime,
                                                              const double* __restrict__ mx, int64_t T,
                                                              double* __restrict__ partial) {
    __shared__ double red[256];
    const int64_t per = (T + gridDim.x - 1) / gridDim.x;
    const int64_t lo = (int64_t)blockIdx.x * per;
    int64_t hi = lo + per;
    if (hi > T) hi = T;
    double s = 0.0;
    for (int64_t t = lo + threadIdx.x; t < hi; t += 256) s += log(cprime[t]) + mx[t];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

// ms[i][j] = a_tilde[i][j] * sum_w slabs[w][pos(i)][pos(j)];  also sum_t (ln c'_t + mx_t), gamma_first/last
// Entry e of ms = a~ o (sum of the xi slabs): 32 lanes per entry, eight entries per workgroup; fixed summation order.
__global__ __launch_bounds__(256) void hmm_finish_kernel(const double* __restrict__ slabs, int64_t n_waves,
                                                         const double* __restrict__ a_tilde, int K, int Kp,
                                                         const double* __restrict__ lnc_partial, int n_partial, int64_t T,
                                                         const double* __restrict__ gamma_tm,
                                                         double* __restrict__ out /*[K*K | K | K | 1]*/) {
    const int tid = threadIdx.x, sub = tid & 31;
    const int e = (int)blockIdx.x * 8 + (tid >> 5);
    double s = 0.0;
    if (e < K * K) {
        const int i = e / K, j = e - i * K;
        for (int64_t w = sub; w < n_waves; w += 32) s += slabs[w * Kp * Kp + hmm_pos(i) * Kp + hmm_pos(j)];
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (e < K * K && sub == 0) out[e] = a_tilde[e] * s;
    if (blockIdx.x == 0) {
        for (int k = tid; k < K; k += blockDim.x) {
            out[K * K + k] = gamma_tm[hmm_pos(k)];
            out[K * K + K + k] = gamma_tm[(T - 1) * Kp + hmm_pos(k)];
        }
        double p = 0.0;
        for (int b = tid; b < n_partial; b += 256) p += lnc_partial[b];
        __shared__ double red[4];
        p = sum_wave(p);
        if ((tid & 63) == 0) red[tid >> 6] = p;
        __syncthreads();
        if (tid == 0) out[K * K + 2 * K] = (red[0] + red[1]) + (red[2] + red[3]);
    }
}

// gamma_tm [T][Kp] lane order -> component-major [K][npad] (the M-step's responsibility buffer)
__global__ void hmm_gamma_to_cm_kernel(const double* __restrict__ gamma_tm, int64_t T, int K, int Kp, int64_t npad,
                                       double* __restrict__ gamma_cm) {
    __shared__ double tile[64][65];
    const int64_t t0 = (int64_t)blockIdx.x * 64;
    const int k0 = blockIdx.y * 64;
    for (int e = threadIdx.x; e < 64 * 64; e += blockDim.x) {
        const int tt = e >> 6, kk = e & 63;
        const int64_t t = t0 + tt;
        const int k = k0 + kk;
        tile[tt][kk] = (t < T && k < K) ? gamma_tm[t * Kp + hmm_pos(k)] : 0.0;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 64 * 64; e += blockDim.x) {
        const int kk = e >> 6, tt = e & 63;
        const int64_t t = t0 + tt;
        const int k = k0 + kk;
        if (t < T && k < K) gamma_cm[(int64_t)k * npad + t] = tile[tt][kk];
    }
}

// Row-range read-outs of the last pass in natural state order (the reference keeps alpha_vecs, beta_vecs [T, K] and
// xi_mats [T, K, K] as attributes, _hiddenmarkovnormal.py:1063-1069; here they are formed on demand):
//   what 0  alpha_t                       [n][K]
//   what 1  beta_t = gamma_t / alpha_t    [n][K]   (the reference's scaling: gamma = alpha o beta, :1013-1014; 0 where alpha = 0)
//   what 3  xi_t = (alpha_{t-1}^T w_t) o A~   [n][K][K], xi_0 = 0 (the reference's convention, :1068)
// w_tm == nullptr (the backward replay summed xi itself and wrote no w): w_t = rho'_t gamma_t / (alpha_t c'_t), 0 where alpha_t = 0
// (then every xi_t[.][j] is 0: alpha_t[j] = rho'_t[j] (alpha_{t-1} A~)[j] / c'_t is a sum of the non-negative terms in question).
__global__ void hmm_readout_kernel(const double* __restrict__ alpha_tm, const double* __restrict__ gamma_tm,
                                   const double* __restrict__ w_tm, const double* __restrict__ a_tilde, int K, int Kp,
                                   int what, int64_t row0, int64_t n_rows, double* __restrict__ out,
                                   const double* __restrict__ rho_tm = nullptr, const double* __restrict__ cprime = nullptr) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    auto pos = [](int s) { return (s & ~15) + 4 * ((s & 15) & 3) + ((s & 15) >> 2); };
    if (what == 3) {
        if (e >= n_rows * K * K) return;
        const int64_t t = row0 + e / ((int64_t)K * K);
        const int i = (int)((e / K) % K), j = (int)(e % K);
        double w = 0.0;
        if (t > 0) {
            if (w_tm) {
                w = w_tm[t * Kp + pos(j)];
            } else {
                const double al = alpha_tm[t * Kp + pos(j)];
                w = al > 0.0 ? rho_tm[t * Kp + pos(j)] * gamma_tm[t * Kp + pos(j)] / (al * cprime[t]) : 0.0;
            }
        }
        out[e] = t == 0 ? 0.0 : alpha_tm[(t - 1) * Kp + pos(i)] * a_tilde[i * K + j] * w;
        return;
    }
    if (e >= n_rows * K) return;
    const int64_t t = row0 + e / K;
    const int s = (int)(e % K);
    const double al = alpha_tm[t * Kp + pos(s)];
    if (what == 0) {
        out[e] = al;
    } else {
        const double g = gamma_tm[t * Kp + pos(s)];
        out[e] = al > 0.0 ? g / al : 0.0;
    }
}

// ---- Viterbi (estimate_latent_vars(loss="0-1", viterbi=True), _hiddenmarkovnormal.py:1465-1481) ----------
// Short sequences: the max-plus recursion is run sequentially by ONE wave (lane = state), 8 time steps of ln rho
// prefetched per lane; the back-pointers are chased through LDS-staged blocks.  It reproduces the reference's sums bit
// for bit.  Long sequences go through the chunked form further down (hmm_vit_*): at T = 1e7 the single wave needs ~10 s.
template <int KT>
__global__ __launch_bounds__(64) void hmm_viterbi_forward_kernel(const double* __restrict__ lnrho, int64_t npad,
                                                                 const double* __restrict__ ln_pi_tilde,
                                                                 const double* __restrict__ ln_a_tilde, int K, int64_t T,
                                                                 unsigned char* __restrict__ phi /*[T][Kp]*/,
                                                                 int* __restrict__ last_state) {
    constexpr int Kp = 16 * KT;
    const int j = threadIdx.x;
    const double NEG = -1.0e300;
    double col[Kp];
#pragma unroll
    for (int i = 0; i < Kp; ++i) col[i] = (i < K && j < K) ? ln_a_tilde[i * K + j] : NEG;
    const double* lr = lnrho + (int64_t)(j < K ? j : 0) * npad;
    double omega = j < K ? lr[0] + ln_pi_tilde[j] : NEG;
    for (int64_t t0 = 1; t0 < T; t0 += 8) {
        double e[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) e[u] = (t0 + u < T) ? lr[t0 + u] : 0.0;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (t0 + u < T) {
                double best = NEG * 2.0;
                int arg = 0;
#pragma unroll
                for (int i = 0; i < Kp; ++i) {
                    const double v = __shfl(omega, i) + col[i];
                    if (v > best) {          // strict: first maximiser, like numpy.argmax
                        best = v;
                        arg = i;
                    }
                }
                omega = j < K ? e[u] + best : NEG;
                if (j < Kp) phi[(t0 + u) * Kp + j] = (unsigned char)arg;
            }
        }
    }
    // first maximiser of omega_{T-1}
    double best = omega;
    int arg = j;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double ob = __shfl_xor(best, o);
        const int oa = __shfl_xor(arg, o);
        if (ob > best || (ob == best && oa < arg)) {
            best = ob;
            arg = oa;
        }
    }
    if (j == 0) *last_state = arg;
}

__global__ __launch_bounds__(256) void hmm_viterbi_backtrack_kernel(const unsigned char* __restrict__ phi, int Kp,
                                                                    int64_t T, const int* __restrict__ last_state,
                                                                    int32_t* __restrict__ z) {
    __shared__ unsigned char tile[512 * 64];
    const int BLK = 512 * 64 / Kp;             // chunks per LDS block (Kp <= 128)
    __shared__ int cur;
    if (threadIdx.x == 0) {
        cur = *last_state;
        z[T - 1] = cur;
    }
    __syncthreads();
    for (int64_t hi = T - 1; hi >= 1; hi -= BLK) {          // steps hi, hi-1, ..., lo use phi[t] to get state at t-1
        const int64_t lo = hi - BLK + 1 > 1 ? hi - BLK + 1 : 1;
        const int64_t n = (hi - lo + 1) * Kp;
        for (int64_t e = threadIdx.x; e < n; e += blockDim.x) tile[e] = phi[lo * Kp + e];
        __syncthreads();
        if (threadIdx.x == 0) {
            int k = cur;
            for (int64_t t = hi; t >= lo; --t) {
                k = tile[(t - lo) * Kp + k];
                z[t - 1] = k;
            }
            cur = k;
        }
        __syncthreads();
    }
}


// ---- chunked Viterbi -----------------------------------------------------------------------------------------------
// omega_t = ln rho_t + max_i (omega_{t-1}(i) + ln a~_ij) is a max-plus matrix-vector recursion, and max-plus products
// are associative, so the sequence is cut into chunks of L steps like the sum-product pass above:
//   hmm_vit_chunk_kernel    per chunk and start state i: the best score of reaching every state j at the chunk's end from
//                           state i at its start  ->  M_c (K x K), chunk- and start-state-parallel;
//   hmm_vit_scan_kernel     one workgroup: omega at every chunk start, omega_{c+1} = omega_c (x) M_c (the M_c staged
//                           through LDS a few chunks ahead of the wave that consumes them);
//   hmm_vit_replay_kernel   per chunk: the reference's recursion from the now known start vector, writing the
//                           back-pointers phi_t (first maximiser, like numpy.argmax) - chunk-parallel;
//   hmm_vit_backmap / backscan / fill   the backtrack the same way: per chunk the map end state -> start state, a
//                           sequential pass over the chunks' maps, then every chunk fills its own stretch of the path.
// The start vectors come out of re-associated sums, so they differ from the sequential recursion's by rounding (1e-16
// relative, ~1e-7 absolute at T = 1e7, where the scores reach -5e8): two paths whose scores agree to that level could be
// told apart differently than by the reference's strictly sequential sums.  Scores of competing paths differ by O(1) on
// anything but constructed ties; tests hold the chunked path to the sequential kernel's and to the reference fixtures'.
template <int KT>
__global__ __launch_bounds__(256) void hmm_vit_chunk_kernel(const double* __restrict__ lnrho, int64_t npad,
                                                            const double* __restrict__ ln_a_tilde, int K, int64_t T, int64_t L,
                                                            double* __restrict__ M /*[chunks][Kp][Kp]*/,
    const int* __restrict__ gate = nullptr /*the coalescence pass stands: nothing to do*/) {
    if (gate != nullptr && *gate == 0) return;
    constexpr int Kp = 16 * KT;
    const int j = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i0 = (int)blockIdx.y * 4 + wave;              // this wave's start state
    if (i0 >= K) return;
    const int64_t c = blockIdx.x;
    const int64_t t0 = 1 + c * L, t1 = (t0 + L < T) ? t0 + L : T;
    const double NEG = -1.0e300;
    double col[Kp];
#pragma unroll
    for (int i = 0; i < Kp; ++i) col[i] = (i < K && j < K) ? ln_a_tilde[i * K + j] : NEG;
    const double* lr = lnrho + (int64_t)(j < K ? j : 0) * npad;
    double v = (j == i0) ? 0.0 : NEG;
    for (int64_t tb = t0; tb < t1; tb += 8) {
        double e[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) e[u] = (tb + u < t1) ? lr[tb + u] : 0.0;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (tb + u < t1) {
                double best = NEG * 2.0;
#pragma unroll
                for (int i = 0; i < Kp; ++i) {
                    const double s = __shfl(v, i) + col[i];
                    best = s > best ? s : best;
                }
                v = j < K ? e[u] + best : NEG;
            }
        }
    }
    if (j < Kp) M[((int64_t)c * Kp + i0) * Kp + j] = v;
}

template <int KT>
__global__ __launch_bounds__(256) void hmm_vit_scan_kernel(const double* __restrict__ lnrho, int64_t npad,
                                                           const double* __restrict__ ln_pi_tilde,
                                                           const double* __restrict__ M, int K, int64_t chunks,
                                                           double* __restrict__ wstart /*[chunks][Kp]*/,
    const int* __restrict__ gate = nullptr /*the coalescence pass stands: nothing to do*/) {
    if (gate != nullptr && *gate == 0) return;
    constexpr int Kp = 16 * KT;
    constexpr int B = 8192 / (Kp * Kp) < 1 ? 1 : 8192 / (Kp * Kp);        // chunks per 64-KB LDS buffer
    constexpr int PER = (B * Kp * Kp + 255) / 256;                        // doubles per thread and group
    __shared__ double buf[2][B * Kp * Kp];
    const int tid = threadIdx.x, j = tid & 63;
    const bool lead = tid < 64;
    const double NEG = -1.0e300;
    double w = (lead && j < K) ? lnrho[(int64_t)j * npad] + ln_pi_tilde[j] : NEG;
    const int64_t groups = (chunks + B - 1) / B;
    double stage[PER];
    auto request = [&](int64_t g) {
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int64_t e = (int64_t)q * 256 + tid;
            const int64_t src = g * B * Kp * Kp + e;
            stage[q] = (e < (int64_t)B * Kp * Kp && src < chunks * Kp * Kp) ? M[src] : 0.0;
        }
    };
    auto deposit = [&](int b) {
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int e = q * 256 + tid;
            if (e < B * Kp * Kp) buf[b][e] = stage[q];
        }
    };
    if (groups > 0) {
        request(0);
        deposit(0);
    }
    __syncthreads();
    for (int64_t g = 0; g < groups; ++g) {
        if (g + 1 < groups) request(g + 1);                 // in flight while the leading wave works through group g
        if (lead) {
            const double* mb = buf[g & 1];
            for (int b = 0; b < B; ++b) {
                const int64_t c = g * B + b;
                if (c >= chunks) break;
                if (j < Kp) wstart[c * Kp + j] = w;
                double best = NEG * 2.0;
                for (int i = 0; i < K; ++i) {
                    const double s = __shfl(w, i) + mb[(b * Kp + i) * Kp + (j < Kp ? j : 0)];
                    best = s > best ? s : best;
                }
                w = j < K ? best : NEG;
            }
        }
        if (g + 1 < groups) deposit((int)((g + 1) & 1));
        __syncthreads();
    }
}

// omega_0 = ln rho_0 + ln pi~ (what the scan kernels start from), as row 0 of the chunk start vectors
__global__ void hmm_vit_omega0_kernel(const double* __restrict__ lnrho, int64_t npad, const double* __restrict__ ln_pi_tilde, int K,
                                      int Kp, double* __restrict__ wstart) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < Kp) wstart[j] = j < K ? lnrho[(int64_t)j * npad] + ln_pi_tilde[j] : -1.0e300;
}

template <int KT>
__global__ __launch_bounds__(64) void hmm_vit_replay_kernel(const double* __restrict__ lnrho, int64_t npad,
                                                            const double* __restrict__ ln_a_tilde,
                                                            const double* __restrict__ wstart, int K, int64_t T, int64_t L,
                                                            int64_t chunks, unsigned char* __restrict__ phi /*[T][Kp]*/,
                                                            int* __restrict__ last_state,
                                                            // The coalescence pass (hmm_capi.hip: hmmvb_viterbi).  sweep: chunks past the
                                                            // first start from the zero vector, no back-pointers are stored.  end_out:
                                                            // omega behind the chunk, minus its maximum, -> row c + 1.
                                                            int sweep = 0, double* __restrict__ end_out = nullptr,
                                                            const int* __restrict__ gate = nullptr) {
    if (gate != nullptr && *gate == 0) return;
    constexpr int Kp = 16 * KT;
    const int j = threadIdx.x;
    const int64_t c = blockIdx.x;
    const int64_t t0 = 1 + c * L, t1 = (t0 + L < T) ? t0 + L : T;
    const double NEG = -1.0e300;
    double col[Kp];
#pragma unroll
    for (int i = 0; i < Kp; ++i) col[i] = (i < K && j < K) ? ln_a_tilde[i * K + j] : NEG;
    const double* lr = lnrho + (int64_t)(j < K ? j : 0) * npad;
    double omega = (j < K) ? ((sweep && c > 0) ? 0.0 : wstart[c * Kp + j]) : NEG;
    for (int64_t tb = t0; tb < t1; tb += 8) {
        double e[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) e[u] = (tb + u < t1) ? lr[tb + u] : 0.0;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (tb + u < t1) {
                double best = NEG * 2.0;
                int arg = 0;
#pragma unroll
                for (int i = 0; i < Kp; ++i) {
                    const double v = __shfl(omega, i) + col[i];
                    if (v > best) {          // strict: first maximiser, like numpy.argmax
                        best = v;
                        arg = i;
                    }
                }
                omega = j < K ? e[u] + best : NEG;
                if (j < Kp && !sweep) phi[(tb + u) * Kp + j] = (unsigned char)arg;
            }
        }
    }
    if (end_out != nullptr && c + 1 < chunks) {
        const double m = max_wave(omega);
        if (j < Kp) end_out[(c + 1) * Kp + j] = j < K ? omega - m : NEG;
    }
    if (c == chunks - 1 && !sweep) {     // first maximiser of omega_{T-1}
        double best = omega;
        int arg = j;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double ob = __shfl_xor(best, o);
            const int oa = __shfl_xor(arg, o);
            if (ob > best || (ob == best && oa < arg)) {
                best = ob;
                arg = oa;
            }
        }
        if (j == 0) *last_state = arg;
    }
}

// map[c][j] = state at the chunk's start time (t0 - 1) of the survivor path that is in state j at its last time (t1 - 1)
__global__ __launch_bounds__(128) void hmm_vit_backmap_kernel(const unsigned char* __restrict__ phi, int Kp, int64_t T, int64_t L,
                                                             unsigned char* __restrict__ map /*[chunks][Kp]*/) {
    const int j = threadIdx.x;
    const int64_t c = blockIdx.x;
    const int64_t t0 = 1 + c * L, t1 = (t0 + L < T) ? t0 + L : T;
    if (j >= Kp) return;
    int s = j;
    for (int64_t t = t1 - 1; t >= t0; --t) s = phi[t * Kp + s];
    map[c * Kp + j] = (unsigned char)s;
}

// endst[c] = state of the best path at the last time of chunk c; sequential over the chunks' maps (LDS-staged blocks)
__global__ __launch_bounds__(256) void hmm_vit_backscan_kernel(const unsigned char* __restrict__ map, int Kp, int64_t chunks,
                                                               const int* __restrict__ last_state, int* __restrict__ endst) {
    __shared__ unsigned char tile[512 * 64];
    const int BLK = 512 * 64 / Kp;             // chunks per LDS block (Kp <= 128)
    __shared__ int cur;
    if (threadIdx.x == 0) cur = *last_state;
    __syncthreads();
    for (int64_t hi = chunks - 1; hi >= 0; hi -= BLK) {
        const int64_t lo = hi - BLK + 1 > 0 ? hi - BLK + 1 : 0;
        const int64_t n = (hi - lo + 1) * Kp;
        for (int64_t e = threadIdx.x; e < n; e += blockDim.x) tile[e] = map[lo * Kp + e];
        __syncthreads();
        if (threadIdx.x == 0) {
            int k = cur;
            for (int64_t c = hi; c >= lo; --c) {
                endst[c] = k;
                k = tile[(c - lo) * Kp + k];
            }
            cur = k;
        }
        __syncthreads();
    }
}

// every chunk writes its own stretch of the path, z[t0 - 1 .. t1 - 1] except z[t0 - 1] for c > 0 (the previous chunk's)
__global__ __launch_bounds__(64) void hmm_vit_fill_kernel(const unsigned char* __restrict__ phi, int Kp, int64_t T, int64_t L,
                                                          int64_t chunks, const int* __restrict__ endst, int32_t* __restrict__ z) {
    const int64_t c = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (c >= chunks) return;
    const int64_t t0 = 1 + c * L, t1 = (t0 + L < T) ? t0 + L : T;
    int s = endst[c];
    for (int64_t t = t1 - 1; t >= t0; --t) {
        z[t] = s;
        s = phi[t * Kp + s];
    }
    if (c == 0) z[0] = s;
}

// ---- chunk matrices with a lane per START state, and a two-level scan over them (round 4) ------------------------------
// hmm_vit_chunk_kernel above spends a wave per (chunk, start state) with the END state on the lane: 32 cross-lane
// broadcasts of omega per step, half of the lanes idle at K = 32 - 223 ms of the 320-ms Viterbi pass at T = 1e7, thirteen
// times the vector-ALU time of its K^2 additions and maxima.  Here the lane is the start state and keeps its whole omega
// vector in registers: every operand of  omega'(j) = e_t(j) + max_i (omega(i) + ln a~_ij)  is then either the lane's own
// register or uniform (ln a~_ij through scalar loads, e_t(j) a broadcast LDS read), the K x K loop is fully unrolled, and a
// wave serves 64 / KP chunks.  Same operations per entry as the kernel above: bit-identical M_c.  K <= 32.
template <int KP>
__global__ __launch_bounds__(64) void hmm_vit_chunk_lane_kernel(const double* __restrict__ lnrho, int64_t npad,
                                                                const double* __restrict__ a_pad /*[KP][KP], -1e300 padded*/,
                                                                int K, int64_t T, int64_t L, int64_t chunks,
                                                                double* __restrict__ M /*[chunks][KP][KP]*/,
    const int* __restrict__ gate = nullptr /*the coalescence pass stands: nothing to do*/) {
    if (gate != nullptr && *gate == 0) return;
    static_assert(KP == 16 || KP == 32, "the omega vector lives in registers");
    constexpr int CW = 64 / KP;                       // chunks per wave
    __shared__ double es[CW][8][KP];
    const int lane = threadIdx.x, h = lane / KP, i0 = lane % KP;
    const int64_t c = (int64_t)blockIdx.x * CW + h;
    const bool live = c < chunks;
    const int64_t cc = live ? c : chunks - 1;
    const int64_t t0 = 1 + cc * L, t1 = (t0 + L < T) ? t0 + L : T;
    const double NEG = -1.0e300;
    const double* lr = lnrho + (int64_t)(i0 < K ? i0 : 0) * npad;       // (staging: this lane fetches state i0's emissions)
    double w[KP];
#pragma unroll
    for (int j = 0; j < KP; ++j) w[j] = (j == i0 && i0 < K) ? 0.0 : NEG;
    // (all chunks of a launch but the last have L steps; the wave runs to the longest of its chunks)
    const int64_t steps = t1 - t0;
    int64_t wave_steps = steps;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const int64_t other = __shfl_xor(wave_steps, o);
        wave_steps = other > wave_steps ? other : wave_steps;
    }
    for (int64_t sb = 0; sb < wave_steps; sb += 8) {
        double e[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) e[u] = (sb + u < steps) ? lr[t0 + sb + u] : 0.0;
#pragma unroll
        for (int u = 0; u < 8; ++u) es[h][u][i0] = e[u];
#pragma unroll 1
        for (int u = 0; u < 8; ++u) {
            if (sb + u >= wave_steps) break;
            const bool on = sb + u < steps;
            double wn[KP];
#pragma unroll
            for (int j = 0; j < KP; ++j) {
                double best = w[0] + a_pad[j];
#pragma unroll
                for (int i = 1; i < KP; ++i) best = fmax(best, w[i] + a_pad[i * KP + j]);
                wn[j] = j < K ? es[h][u][j] + best : NEG;
            }
#pragma unroll
            for (int j = 0; j < KP; ++j) w[j] = on ? wn[j] : w[j];
        }
    }
    if (live) {
        double* out = M + ((int64_t)c * KP + i0) * KP;
#pragma unroll
        for (int j = 0; j < KP; ++j) out[j] = w[j];
    }
}

__global__ void hmm_vit_pad_kernel(const double* __restrict__ ln_a_tilde, int K, int KP, double* __restrict__ a_pad) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= KP * KP) return;
    const int i = e / KP, j = e - i * KP;
    a_pad[e] = (i < K && j < K) ? ln_a_tilde[i * K + j] : -1.0e300;
}

// P_s = M_c0 (x) M_c0+1 (x) ... over the kHmmSuper chunks of super-chunk s (max-plus products, K^3 each; a workgroup per
// super-chunk, the running product and the next factor in LDS)
template <int KP>
__global__ __launch_bounds__(256) void hmm_vit_super_kernel(const double* __restrict__ M, int K, int64_t chunks,
                                                            double* __restrict__ P /*[supers][KP][KP]*/,
    const int* __restrict__ gate = nullptr /*the coalescence pass stands: nothing to do*/) {
    if (gate != nullptr && *gate == 0) return;
    __shared__ double pa[KP * KP], pb[KP * KP], mc[KP * KP];
    const int tid = threadIdx.x;
    const int64_t s = blockIdx.x;
    const int64_t c0 = s * kHmmSuper, c1 = (c0 + kHmmSuper < chunks) ? c0 + kHmmSuper : chunks;
    for (int e = tid; e < KP * KP; e += 256) pa[e] = M[c0 * KP * KP + e];
    __syncthreads();
    double* cur = pa;
    double* nxt = pb;
    for (int64_t c = c0 + 1; c < c1; ++c) {
        for (int e = tid; e < KP * KP; e += 256) mc[e] = M[c * KP * KP + e];
        __syncthreads();
        for (int e = tid; e < KP * KP; e += 256) {
            const int i = e / KP, j = e - i * KP;
            double best = -2.0e300;
            if (i < K && j < K) {
                for (int k = 0; k < K; ++k) {
                    const double v = cur[i * KP + k] + mc[k * KP + j];
                    best = v > best ? v : best;
                }
            } else {
                best = -1.0e300;
            }
            nxt[e] = best;
        }
        __syncthreads();
        double* t = cur;
        cur = nxt;
        nxt = t;
    }
    for (int e = tid; e < KP * KP; e += 256) P[s * KP * KP + e] = cur[e];
}

// omega at every super-chunk start: one wave, omega_{s+1} = omega_s (x) P_s, the next product's column prefetched
template <int KP>
__global__ __launch_bounds__(64) void hmm_vit_scan2_kernel(const double* __restrict__ lnrho, int64_t npad,
                                                           const double* __restrict__ ln_pi_tilde, const double* __restrict__ P,
                                                           int K, int64_t supers, double* __restrict__ sstart /*[supers][KP]*/,
    const int* __restrict__ gate = nullptr /*the coalescence pass stands: nothing to do*/) {
    if (gate != nullptr && *gate == 0) return;
    const int j = threadIdx.x;
    const int jj = j < KP ? j : 0;
    const double NEG = -1.0e300;
    double w = j < K ? lnrho[(int64_t)j * npad] + ln_pi_tilde[j] : NEG;
    double col[KP], nxt[KP];
#pragma unroll
    for (int i = 0; i < KP; ++i) nxt[i] = supers > 0 ? P[i * KP + jj] : 0.0;
    for (int64_t s = 0; s < supers; ++s) {
#pragma unroll
        for (int i = 0; i < KP; ++i) col[i] = nxt[i];
        if (s + 1 < supers) {
#pragma unroll
            for (int i = 0; i < KP; ++i) nxt[i] = P[((s + 1) * KP + i) * KP + jj];
        }
        if (j < KP) sstart[s * KP + j] = w;
        double best = NEG * 2.0;
#pragma unroll
        for (int i = 0; i < KP; ++i) {
            const double v = __shfl(w, i) + col[i];
            best = (i < K && v > best) ? v : best;
        }
        w = j < K ? best : NEG;
    }
}

// omega at every chunk start of super-chunk s from its start vector: omega_{c+1} = omega_c (x) M_c (a wave per super-chunk)
template <int KP>
__global__ __launch_bounds__(64) void hmm_vit_fill2_kernel(const double* __restrict__ M, int K, int64_t chunks,
                                                           const double* __restrict__ sstart, double* __restrict__ wstart,
    const int* __restrict__ gate = nullptr /*the coalescence pass stands: nothing to do*/) {
    if (gate != nullptr && *gate == 0) return;
    const int j = threadIdx.x;
    const int jj = j < KP ? j : 0;
    const double NEG = -1.0e300;
    const int64_t s = blockIdx.x;
    const int64_t c0 = s * kHmmSuper, c1 = (c0 + kHmmSuper < chunks) ? c0 + kHmmSuper : chunks;
    double w = j < K ? sstart[s * KP + j] : NEG;
    double col[KP], nxt[KP];
#pragma unroll
    for (int i = 0; i < KP; ++i) nxt[i] = M[(c0 * KP + i) * KP + jj];
    for (int64_t c = c0; c < c1; ++c) {
#pragma unroll
        for (int i = 0; i < KP; ++i) col[i] = nxt[i];
        if (c + 1 < c1) {
#pragma unroll
            for (int i = 0; i < KP; ++i) nxt[i] = M[((c + 1) * KP + i) * KP + jj];
        }
        if (j < KP) wstart[c * KP + j] = w;
        double best = NEG * 2.0;
#pragma unroll
        for (int i = 0; i < KP; ++i) {
            const double v = __shfl(w, i) + col[i];
            best = (i < K && v > best) ? v : best;
        }
        w = j < K ? best : NEG;
    }
}

}  // namespace gmmvb
