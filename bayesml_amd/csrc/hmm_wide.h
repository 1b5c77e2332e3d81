// Chunk-parallel forward-backward for 65 .. 128 hidden states (round 4).
//
// hmm.h keeps the constant operand of every step (A~ or its transpose, K x K) and a chunk's K x K transfer product in one
// wave's registers, which ends at K = 64; beyond, hmm_generic.h walked the sequence in ONE workgroup (a matrix-vector product
// per step: 8 us at K = 128, 83 s for T = 1e7).  The recursions themselves (reference _hiddenmarkovnormal.py:999-1018) do
// not care: with
//   * the constant operand in LDS instead of registers, stored once per workgroup in the order the MFMA A operand is read
//     (fragment f = (it KT + kt) 4 + s, lane l: M[16 it + (l & 15)][16 kt + (l >> 4) + 4 s]; 128 KB at K = 128), and
//   * a chunk's transfer product spread over the four waves of a workgroup by column blocks of P^T (they are independent:
//     P^T <- diag(rho'_t) A~^T P^T acts on every column alone; only the rescaling needs the workgroup's common maximum),
// the same kernels run for KT = 5 .. 8 tiles of 16 states.  The boundary pass is a sequential pass over chunk products (a
// row-vector x matrix per chunk, 1024 threads) - over the products of 64 chunks each first, when there are many; the xi-sum is split over a workgroup's waves by row tiles; the state arrays, the read-outs and
// everything behind them are the ones of the other paths.  Cost: T 2 Kp^3 flop of chunk products on the f64 matrix pipe
// (0.6 s at K = 128, T = 1e7), everything else is small beside it.
#pragma once
#include "hmm.h"

namespace gmmvb {

template <int KT>
__host__ __device__ constexpr size_t hmm_wide_frag_bytes() { return (size_t)KT * KT * 4 * 64 * sizeof(double); }

// frag <- the A-operand fragments of M (M = a^T if transpose), zero beyond K; all threads of the workgroup, then a barrier
template <int KT>
__device__ __forceinline__ void fill_frags(const double* __restrict__ a, int K, bool transpose, double* __restrict__ frag) {
    for (int e = threadIdx.x; e < KT * KT * 4 * 64; e += blockDim.x) {
        const int l = e & 63, f = e >> 6;
        const int s = f & 3, kt = (f >> 2) % KT, it = (f >> 2) / KT;
        const int r = 16 * it + (l & 15), c = 16 * kt + (l >> 4) + 4 * s;
        double v = 0.0;
        if (r < K && c < K) v = transpose ? a[(int64_t)c * K + r] : a[(int64_t)r * K + c];
        frag[e] = v;
    }
    __syncthreads();
}

// out[it] = M . in for one 16-column block (hmm.h: apply), the A operand read from LDS
template <int KT>
__device__ __forceinline__ void apply_lds(const double* __restrict__ frag, const d4 (&in)[KT], d4 (&out)[KT]) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int it = 0; it < KT; ++it) {
        // (the fragments are loop-invariant over the time steps: left alone, the compiler hoists all KT^2 x 4 reads out of the
        // step loop - 512 registers at KT = 8, spilled to AGPRs and scratch.  The offset passes through an empty asm per
        // output tile, so the reads of one tile - 4 KT, 64 registers - are all that can be in flight)
        int off = it * KT * 256 + lane;
        asm volatile("" : "+v"(off));
        d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int s = 0; s < 4; ++s) acc = mfma_f64(frag[off + (kt * 4 + s) * 64], in[kt][s], acc);
        out[it] = acc;
    }
}

// H2 wide: P_c, one WORKGROUP per chunk; wave w keeps the column blocks jt = w JPW .. of P^T
template <int KT>
__global__ __launch_bounds__(256) void hmm_chunk_products_wide_kernel(const double* __restrict__ rho_tm,
                                                                      const double* __restrict__ a_tilde, int K, int64_t T,
                                                                      int64_t L, int64_t n_chunks,
                                                                      double* __restrict__ prod /*[n_chunks][Kp][Kp]*/,
                                                                      double* __restrict__ prod_t /*the transposes*/,
    const int* __restrict__ gate = nullptr /*hmm.h: the forgetting pass stands -> nothing to do*/) {
    if (gate != nullptr && *gate == 0) return;
    constexpr int Kp = 16 * KT, JPW = (KT + 3) / 4;
    extern __shared__ double frag[];                      // fragments, then [2][4] maxima
    double* smax = frag + KT * KT * 4 * 64;
    const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t c = blockIdx.x;
    fill_frags<KT>(a_tilde, K, /*transpose=*/true, frag);
    d4 pt[JPW][KT];
#pragma unroll
    for (int jj = 0; jj < JPW; ++jj)
#pragma unroll
        for (int it = 0; it < KT; ++it)
#pragma unroll
            for (int r = 0; r < 4; ++r) pt[jj][it][r] = (it == wave * JPW + jj && (g + 4 * r) == j) ? 1.0 : 0.0;
    const int64_t t0 = 1 + c * L;
    int64_t t1 = t0 + L;
    if (t1 > T) t1 = T;
    int par = 0;
    for (int64_t t = t0; t < t1; ++t) {
        d4 rho[KT];
#pragma unroll
        for (int it = 0; it < KT; ++it) rho[it] = *reinterpret_cast<const d4*>(rho_tm + t * Kp + 16 * it + 4 * g);
#pragma unroll
        for (int jj = 0; jj < JPW; ++jj) {
            if (wave * JPW + jj >= KT) continue;          // (wave-uniform)
            d4 nw[KT];
            apply_lds<KT>(frag, pt[jj], nw);
#pragma unroll
            for (int it = 0; it < KT; ++it)
#pragma unroll
                for (int r = 0; r < 4; ++r) pt[jj][it][r] = nw[it][r] * rho[it][r];
        }
        if (((t - t0) & 3) == 3) {                        // rescale by the workgroup's common maximum (any positive factor)
            double m = 0.0;
#pragma unroll
            for (int jj = 0; jj < JPW; ++jj)
#pragma unroll
                for (int it = 0; it < KT; ++it)
#pragma unroll
                    for (int r = 0; r < 4; ++r) m = fmax(m, pt[jj][it][r]);
            m = max_wave(m);
            if (lane == 0) smax[par * 4 + wave] = m;
            __syncthreads();
            m = fmax(fmax(smax[par * 4], smax[par * 4 + 1]), fmax(smax[par * 4 + 2], smax[par * 4 + 3]));
            par ^= 1;                                      // (the other set is written next: no second barrier)
            const double sc = m > 0.0 ? 1.0 / m : 1.0;
#pragma unroll
            for (int jj = 0; jj < JPW; ++jj)
#pragma unroll
                for (int it = 0; it < KT; ++it)
#pragma unroll
                    for (int r = 0; r < 4; ++r) pt[jj][it][r] *= sc;
        }
    }
    // P (row = column index of P^T) for the forward pass of the boundary scan, P^T for the backward one: either reads its
    // operand with the state on consecutive threads
    double* out = prod + c * Kp * Kp;
    double* out_t = prod_t + c * Kp * Kp;
#pragma unroll
    for (int jj = 0; jj < JPW; ++jj) {
        const int jt = wave * JPW + jj;
        if (jt >= KT) continue;
#pragma unroll
        for (int it = 0; it < KT; ++it)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                out[(16 * jt + j) * Kp + 16 * it + g + 4 * r] = pt[jj][it][r];
                out_t[(16 * it + g + 4 * r) * Kp + 16 * jt + j] = pt[jj][it][r];
            }
    }
}

// H3a wide: Q_s = product of the (up to) kHmmSuper chunk products of super-chunk s, as Q_s and Q_s^T - the chunk-products
// kernel with the step's operand changing: R^T <- P_c^T R^T, the fragments of P_c^T restaged into LDS for every chunk.
template <int KT>
__global__ __launch_bounds__(256) void hmm_super_products_wide_kernel(const double* __restrict__ prod_t, int64_t n_chunks,
                                                                      double* __restrict__ qprod, double* __restrict__ qprod_t,
    const int* __restrict__ gate = nullptr /*hmm.h: the forgetting pass stands -> nothing to do*/) {
    if (gate != nullptr && *gate == 0) return;
    constexpr int Kp = 16 * KT, JPW = (KT + 3) / 4;
    extern __shared__ double frag[];
    double* smax = frag + KT * KT * 4 * 64;
    const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t sblk = blockIdx.x;
    const int64_t c0 = sblk * kHmmSuper;
    const int64_t c1 = c0 + kHmmSuper < n_chunks ? c0 + kHmmSuper : n_chunks;
    d4 pt[JPW][KT];
#pragma unroll
    for (int jj = 0; jj < JPW; ++jj)
#pragma unroll
        for (int it = 0; it < KT; ++it)
#pragma unroll
            for (int r = 0; r < 4; ++r) pt[jj][it][r] = (it == wave * JPW + jj && (g + 4 * r) == j) ? 1.0 : 0.0;
    int par = 0;
    for (int64_t c = c0; c < c1; ++c) {
        __syncthreads();                                  // everybody is done with the previous chunk's fragments
        fill_frags<KT>(prod_t + c * Kp * Kp, Kp, /*transpose=*/false, frag);      // M = P_c^T
#pragma unroll
        for (int jj = 0; jj < JPW; ++jj) {
            if (wave * JPW + jj >= KT) continue;
            d4 nw[KT];
            apply_lds<KT>(frag, pt[jj], nw);
#pragma unroll
            for (int it = 0; it < KT; ++it) pt[jj][it] = nw[it];
        }
        double m = 0.0;
#pragma unroll
        for (int jj = 0; jj < JPW; ++jj)
#pragma unroll
            for (int it = 0; it < KT; ++it)
#pragma unroll
                for (int r = 0; r < 4; ++r) m = fmax(m, pt[jj][it][r]);
        m = max_wave(m);
        if (lane == 0) smax[par * 4 + wave] = m;
        __syncthreads();
        m = fmax(fmax(smax[par * 4], smax[par * 4 + 1]), fmax(smax[par * 4 + 2], smax[par * 4 + 3]));
        par ^= 1;
        const double sc = m > 0.0 ? 1.0 / m : 1.0;
#pragma unroll
        for (int jj = 0; jj < JPW; ++jj)
#pragma unroll
            for (int it = 0; it < KT; ++it)
#pragma unroll
                for (int r = 0; r < 4; ++r) pt[jj][it][r] *= sc;
    }
    double* out = qprod + sblk * Kp * Kp;
    double* out_t = qprod_t + sblk * Kp * Kp;
#pragma unroll
    for (int jj = 0; jj < JPW; ++jj) {
        const int jt = wave * JPW + jj;
        if (jt >= KT) continue;
#pragma unroll
        for (int it = 0; it < KT; ++it)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                out[(16 * jt + j) * Kp + 16 * it + g + 4 * r] = pt[jj][it][r];
                out_t[(16 * it + g + 4 * r) * Kp + 16 * jt + j] = pt[jj][it][r];
            }
    }
}

// H3 wide: the sequential pass over chunk products.  blockIdx.y = 0 forward (fstart[c + 1] ~ fstart[c] P_c), 1 backward
// (bend[c - 1] ~ P_c bend[c]); 1024 threads = (state i, eighth p of the contraction index), the next product's entries
// requested while the current ones are reduced.  Natural state order, like hmm_boundary_scan_kernel.
//   in_f == nullptr: the whole sequence of n_chunks products in one workgroup per direction (grid (1, 2)); the forward
//                    start vector is alpha_0 (made here, with c'_0), the backward one uniform;
//   in_f != nullptr: workgroup blockIdx.x fills in the chunks of ITS super-chunk (kHmmSuper products) from the super-chunk's
//                    start / end vector in_f[x] / in_b[x] - the second level of a two-level pass whose first level is this
//                    same kernel over the super-chunk products (hmm_super_products_wide_kernel).
constexpr int kHmmWideScanThreads = 1024;
template <int KT>
__global__ __launch_bounds__(kHmmWideScanThreads) void hmm_boundary_scan_wide_kernel(
    const double* __restrict__ rho_tm, const double* __restrict__ pi_tilde, const double* __restrict__ prod,
    const double* __restrict__ prod_t, int K, int64_t n_chunks, const double* __restrict__ in_f, const double* __restrict__ in_b,
    double* __restrict__ fstart, double* __restrict__ bend, double* __restrict__ cprime,
    double* __restrict__ alpha_tm, double* __restrict__ gamma_tm, double* __restrict__ w_tm,
    const int* __restrict__ gate = nullptr /*hmm.h: the forgetting pass stands -> nothing to do*/) {
    if (gate != nullptr && *gate == 0) return;
    constexpr int Kp = 16 * KT, PARTS = kHmmWideScanThreads / 128, JP = Kp / PARTS;      // JP contraction indices per thread
    static_assert(Kp <= 128 && Kp % PARTS == 0, "up to 128 states");
    __shared__ double sv[128];
    __shared__ double spart[PARTS][128];
    __shared__ double sred[2];
    const int tid = threadIdx.x, i = tid & 127, p = tid >> 7;
    const bool fwd = blockIdx.y == 0;
    const bool fill = in_f != nullptr;
    const int64_t c_lo = fill ? (int64_t)blockIdx.x * kHmmSuper : 0;
    const int64_t c_hi = fill ? (c_lo + kHmmSuper < n_chunks ? c_lo + kHmmSuper : n_chunks) : n_chunks;
    auto normalise = [&](double v) -> double {          // threads tid < 128 hold v_i; everybody gets v_i / sum (for i = tid & 127)
        __syncthreads();
        if (tid < 128) sv[tid] = v;
        __syncthreads();
        if (tid < 64) {
            double s = sum_wave(sv[tid] + sv[tid + 64]);
            if (tid == 0) sred[0] = s;
        }
        __syncthreads();
        const double tot = sred[0];
        return tot > 0.0 ? sv[i] / tot : 0.0;
    };
    double v;
    if (fill) {
        v = i < Kp ? (fwd ? in_f : in_b)[(int64_t)blockIdx.x * Kp + i] : 0.0;
        if (tid < Kp) {
            if (fwd) fstart[c_lo * Kp + tid] = v;
            else bend[(c_hi - 1) * Kp + tid] = v;
        }
    } else if (fwd) {
        double a0 = (tid < K) ? rho_tm[hmm_pos(tid)] * pi_tilde[tid] : 0.0;
        if (tid >= 128) a0 = 0.0;
        __syncthreads();
        if (tid < 128) sv[tid] = a0;
        __syncthreads();
        if (tid < 64) {
            double s = sum_wave(sv[tid] + sv[tid + 64]);
            if (tid == 0) {
                sred[0] = s;
                cprime[0] = s;
            }
        }
        __syncthreads();
        v = sred[0] > 0.0 ? sv[i] / sred[0] : 0.0;
        if (tid < Kp) {
            fstart[tid] = v;
            if (n_chunks == 0) {
                alpha_tm[hmm_pos(tid)] = v;
                gamma_tm[hmm_pos(tid)] = v;
                w_tm[hmm_pos(tid)] = 0.0;
            }
        }
    } else {
        v = i < K ? 1.0 / K : 0.0;
        if (tid < Kp && n_chunks > 0) bend[(n_chunks - 1) * Kp + tid] = v;
    }
    if (c_hi - c_lo < 2) return;
    // entries of a chunk product this thread multiplies: forward P[jj][i] (column i), backward P[i][jj] = P^T[jj][i] (row i),
    // jj in its part - consecutive threads read consecutive addresses either way (a first form read P[i][jj] from the one
    // array: 11 us per chunk)
    double cur[JP], nxt[JP];
    auto fetch = [&](int64_t c, double (&dst)[JP]) {
        const double* P = (fwd ? prod : prod_t) + c * Kp * Kp;
#pragma unroll
        for (int q = 0; q < JP; ++q) {
            const int jj = p * JP + q;
            dst[q] = i < Kp ? P[jj * Kp + i] : 0.0;
        }
    };
    if (fwd) fetch(c_lo, nxt);
    else fetch(c_hi - 1, nxt);
    const int64_t steps = c_hi - c_lo - 1;
    for (int64_t s = 0; s < steps; ++s) {
        const int64_t c = fwd ? c_lo + s : c_hi - 1 - s;          // the product applied in this step
#pragma unroll
        for (int q = 0; q < JP; ++q) cur[q] = nxt[q];
        if (s + 1 < steps) fetch(fwd ? c + 1 : c - 1, nxt);
        __syncthreads();
        if (tid < 128) sv[tid] = v;                                // (every part holds the same v_i)
        __syncthreads();
        double acc = 0.0;
#pragma unroll
        for (int q = 0; q < JP; ++q) acc = fma(sv[p * JP + q], cur[q], acc);
        spart[p][i] = acc;
        __syncthreads();
        double tot_i = 0.0;
#pragma unroll
        for (int q = 0; q < PARTS; ++q) tot_i += spart[q][i];
        if (i >= Kp) tot_i = 0.0;
        v = normalise(tot_i);
        if (tid < Kp) {
            if (fwd) fstart[(c + 1) * Kp + tid] = v;
            else bend[(c - 1) * Kp + tid] = v;
        }
    }
}

// H4 wide: forward replay, 16 chunks per wave, the four waves of a workgroup share A~^T in LDS
template <int KT>
__global__ __launch_bounds__(256) void hmm_forward_replay_wide_kernel(const double* __restrict__ rho_tm,
                                                                      const double* __restrict__ a_tilde, int K, int64_t T,
                                                                      int64_t L, int64_t n_chunks,
                                                                      const double* __restrict__ fstart,
                                                                      double* __restrict__ alpha_tm, double* __restrict__ cprime,
                                                                      int sweep = 0 /*hmm.h: hmm_forward_replay_kernel*/,
                                                                      double* __restrict__ end_out = nullptr,
                                                                      const int* __restrict__ gate = nullptr) {
    if (gate != nullptr && *gate == 0) return;
    constexpr int Kp = 16 * KT;
    extern __shared__ double frag[];
    const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
    const int64_t c = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 16 + j;
    const bool live = c < n_chunks;
    fill_frags<KT>(a_tilde, K, /*transpose=*/true, frag);        // alpha^T_new = A~^T alpha^T_old
    d4 al[KT];
#pragma unroll
    for (int it = 0; it < KT; ++it)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            al[it][r] = !live ? 0.0 : ((sweep && c > 0) ? (16 * it + g + 4 * r < K ? 1.0 / K : 0.0) : fstart[c * Kp + 16 * it + g + 4 * r]);
    if (live && c == 0 && !sweep) {
#pragma unroll
        for (int it = 0; it < KT; ++it) *reinterpret_cast<d4*>(alpha_tm + 16 * it + 4 * g) = al[it];
    }
    const int64_t t0 = 1 + c * L;
    for (int64_t s = 0; s < L; ++s) {
        const int64_t t = t0 + s;
        const bool on = live && t < T;
        d4 nw[KT];
        apply_lds<KT>(frag, al, nw);
        double part = 0.0;
#pragma unroll
        for (int it = 0; it < KT; ++it) {
            d4 rho = {0.0, 0.0, 0.0, 0.0};
            if (on) rho = *reinterpret_cast<const d4*>(rho_tm + t * Kp + 16 * it + 4 * g);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                nw[it][r] *= rho[r];
                part += nw[it][r];
            }
        }
        const double cp = sum_groups(part);
        const double inv = cp > 0.0 ? 1.0 / cp : 0.0;
#pragma unroll
        for (int it = 0; it < KT; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r) al[it][r] = nw[it][r] * inv;
            if (on && !sweep) *reinterpret_cast<d4*>(alpha_tm + t * Kp + 16 * it + 4 * g) = al[it];
        }
        if (on && !sweep && g == 0) cprime[t] = cp;
    }
    if (end_out != nullptr && live && c + 1 < n_chunks) {
#pragma unroll
        for (int it = 0; it < KT; ++it)
#pragma unroll
            for (int r = 0; r < 4; ++r) end_out[(c + 1) * Kp + 16 * it + g + 4 * r] = al[it][r];
    }
}

// H5 wide: backward replay (hmm.h: hmm_backward_replay_kernel), A~ in LDS
template <int KT>
__global__ __launch_bounds__(256) void hmm_backward_replay_wide_kernel(const double* __restrict__ rho_tm,
                                                                       const double* __restrict__ a_tilde, int K, int64_t T,
                                                                       int64_t L, int64_t n_chunks,
                                                                       const double* __restrict__ bend,
                                                                       const double* __restrict__ alpha_tm,
                                                                       const double* __restrict__ cprime,
                                                                       double* __restrict__ gamma_tm, double* __restrict__ w_tm,
                                                                       int sweep = 0 /*1: the recursion alone from uniform vectors, no loads of alpha, no stores*/,
                                                                       double* __restrict__ bend_out = nullptr /*beta~ in front of chunk c -> row c - 1*/,
                                                                       const int* __restrict__ gate = nullptr) {
    if (gate != nullptr && *gate == 0) return;
    constexpr int Kp = 16 * KT;
    extern __shared__ double frag[];
    const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
    const int64_t c = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 16 + j;
    const bool live = c < n_chunks;
    fill_frags<KT>(a_tilde, K, /*transpose=*/false, frag);       // beta_{t-1} ~ A~ (rho'_t o beta_t)
    d4 be[KT];
#pragma unroll
    for (int it = 0; it < KT; ++it)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            be[it][r] = !live ? 0.0 : (sweep ? (16 * it + g + 4 * r < K ? 1.0 / K : 0.0) : bend[c * Kp + 16 * it + g + 4 * r]);
    const int64_t t0 = 1 + c * L;
    for (int64_t s = L - 1; s >= 0; --s) {
        const int64_t t = t0 + s;
        const bool on = live && t < T;
        d4 y[KT];
        double dot = 0.0;
        d4 al[KT], rho[KT];
#pragma unroll
        for (int it = 0; it < KT; ++it) {
            al[it] = d4{0.0, 0.0, 0.0, 0.0};
            rho[it] = d4{0.0, 0.0, 0.0, 0.0};
            if (on) {
                if (!sweep) al[it] = *reinterpret_cast<const d4*>(alpha_tm + t * Kp + 16 * it + 4 * g);
                rho[it] = *reinterpret_cast<const d4*>(rho_tm + t * Kp + 16 * it + 4 * g);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) dot = fma(al[it][r], be[it][r], dot);
        }
        if (!sweep) dot = sum_groups(dot);
        const double cp = (on && !sweep) ? cprime[t] : 1.0;
        const double ginv = dot > 0.0 ? 1.0 / dot : 0.0;
        const double winv = (dot > 0.0 && cp > 0.0) ? 1.0 / (dot * cp) : 0.0;
#pragma unroll
        for (int it = 0; it < KT; ++it) {
            d4 gm, ww;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                y[it][r] = rho[it][r] * be[it][r];
                gm[r] = al[it][r] * be[it][r] * ginv;
                ww[r] = y[it][r] * winv;
            }
            if (on && !sweep) {
                *reinterpret_cast<d4*>(gamma_tm + t * Kp + 16 * it + 4 * g) = gm;
                *reinterpret_cast<d4*>(w_tm + t * Kp + 16 * it + 4 * g) = ww;
            }
        }
        d4 nb[KT];
        apply_lds<KT>(frag, y, nb);
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
    if (bend_out != nullptr && live && c >= 1) {
#pragma unroll
        for (int it = 0; it < KT; ++it)
#pragma unroll
            for (int r = 0; r < 4; ++r) bend_out[(c - 1) * Kp + 16 * it + g + 4 * r] = be[it][r];
    }
    if (live && c == 0 && !sweep) {                             // gamma_0 = alpha_0 o beta~_0, normalised
        double dot = 0.0;
        d4 al[KT];
#pragma unroll
        for (int it = 0; it < KT; ++it) {
            al[it] = *reinterpret_cast<const d4*>(alpha_tm + 16 * it + 4 * g);
#pragma unroll
            for (int r = 0; r < 4; ++r) dot = fma(al[it][r], be[it][r], dot);
        }
        dot = sum_groups(dot);
        const double ginv = dot > 0.0 ? 1.0 / dot : 0.0;
#pragma unroll
        for (int it = 0; it < KT; ++it) {
            d4 gm;
#pragma unroll
            for (int r = 0; r < 4; ++r) gm[r] = al[it][r] * be[it][r] * ginv;
            *reinterpret_cast<d4*>(gamma_tm + 16 * it + 4 * g) = gm;
            *reinterpret_cast<d4*>(w_tm + 16 * it + 4 * g) = d4{0.0, 0.0, 0.0, 0.0};      // xi_0 = 0
        }
    }
}

// H6 wide: raw[pi][pj] = sum_t alpha_tm[t-1][pi] w_tm[t][pj] (hmm.h: hmm_xi_sum_kernel), one slab per WORKGROUP: wave w
// accumulates the row tiles it = w RPW .. of the K x K sum over the workgroup's stretch of time steps
template <int KT>
__global__ __launch_bounds__(256) void hmm_xi_sum_wide_kernel(const double* __restrict__ alpha_tm, const double* __restrict__ w_tm,
                                                              int64_t T, int64_t steps_per_wg, double* __restrict__ slabs) {
    constexpr int Kp = 16 * KT, RPW = (KT + 3) / 4;
    const int lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t lo = 1 + (int64_t)blockIdx.x * steps_per_wg;
    int64_t hi = lo + steps_per_wg;
    if (hi > T) hi = T;
    d4 acc[RPW][KT];
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr)
#pragma unroll
        for (int jt = 0; jt < KT; ++jt) acc[rr][jt] = d4{0.0, 0.0, 0.0, 0.0};
    double an[RPW], bn[KT];
    auto fetch = [&](int64_t t) {
        const int64_t tt = t + g;
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) {
            const int it = wave * RPW + rr;
            an[rr] = (tt < hi && it < KT) ? alpha_tm[(tt - 1) * Kp + 16 * it + i] : 0.0;
        }
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) bn[kt] = tt < hi ? w_tm[tt * Kp + 16 * kt + i] : 0.0;
    };
    fetch(lo);
    for (int64_t t = lo; t < hi; t += 4) {
        double a[RPW], b[KT];
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) a[rr] = an[rr];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) b[kt] = bn[kt];
        fetch(t + 4);
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr)
#pragma unroll
            for (int jt = 0; jt < KT; ++jt) acc[rr][jt] = mfma_f64(a[rr], b[jt], acc[rr][jt]);
    }
    double* out = slabs + (int64_t)blockIdx.x * Kp * Kp;
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
        const int it = wave * RPW + rr;
        if (it >= KT) continue;
#pragma unroll
        for (int jt = 0; jt < KT; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) out[(16 * it + g + 4 * r) * Kp + 16 * jt + i] = acc[rr][jt][r];
    }
}

// ---- chunked Viterbi for 65 .. 128 states (hmm.h: hmm_vit_*) ------------------------------------------------------------------
// The lane is the END state - two of them per lane, j and j + 64 - and ln a~ sits in LDS (row i contiguous over j); a wave
// carries FOUR start states at once, so every pair of LDS reads feeds eight add / max pairs.
constexpr int kVitWideStarts = 4;          // start states per wave
__device__ __forceinline__ double readlane_f64(double v, int l) {          // l uniform
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(__double_as_longlong(v) & 0xffffffffll), l);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(__double_as_longlong(v) >> 32), l);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// M_c[i0][j] = best score of reaching j at the chunk's end from i0 at its start.  grid = (chunks, ceil(K / 32)), 8 waves (two
// per SIMD: the 128 KB of ln a~ allow one workgroup per CU, and a single wave per SIMD waits for every LDS read).
constexpr int kVitWideWaves = 8;
template <int KT>
__global__ __launch_bounds__(64 * kVitWideWaves) void hmm_vit_chunk_wide_kernel(const double* __restrict__ lnrho, int64_t npad,
                                                                 const double* __restrict__ ln_a_tilde, int K, int64_t T, int64_t L,
                                                                 double* __restrict__ M /*[chunks][Kp][Kp]*/,
    const int* __restrict__ gate = nullptr /*hmm.h: the coalescence pass stands*/) {
    if (gate != nullptr && *gate == 0) return;
    constexpr int Kp = 16 * KT, SB = kVitWideStarts;
    extern __shared__ double a_lds[];                  // [Kp][Kp]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const double NEG = -1.0e300;
    for (int e = threadIdx.x; e < Kp * Kp; e += 64 * kVitWideWaves) {
        const int i = e / Kp, j = e - i * Kp;
        a_lds[e] = (i < K && j < K) ? ln_a_tilde[i * K + j] : NEG;
    }
    __syncthreads();
    const int64_t c = blockIdx.x;
    const int i00 = ((int)blockIdx.y * kVitWideWaves + wave) * SB;        // this wave's first start state
    if (i00 >= K) return;
    const int64_t t0 = 1 + c * L, t1 = (t0 + L < T) ? t0 + L : T;
    const int j0 = lane, j1 = lane + 64;
    const double* lr0 = lnrho + (int64_t)(j0 < K ? j0 : 0) * npad;
    const double* lr1 = lnrho + (int64_t)(j1 < K ? j1 : 0) * npad;
    double v0[SB], v1[SB];
#pragma unroll
    for (int sb = 0; sb < SB; ++sb) {
        v0[sb] = (j0 == i00 + sb) ? 0.0 : NEG;
        v1[sb] = (j1 == i00 + sb) ? 0.0 : NEG;
    }
    for (int64_t tb = t0; tb < t1; tb += 8) {
        double e0[8], e1[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            e0[u] = (tb + u < t1) ? lr0[tb + u] : 0.0;
            e1[u] = (tb + u < t1) ? lr1[tb + u] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (tb + u >= t1) break;
            double b0[SB], b1[SB];
#pragma unroll
            for (int sb = 0; sb < SB; ++sb) b0[sb] = b1[sb] = NEG * 2.0;
#pragma unroll 8
            for (int i = 0; i < Kp; ++i) {
                const double a0 = a_lds[i * Kp + j0];
                const double a1 = Kp > 64 ? a_lds[i * Kp + (j1 < Kp ? j1 : 0)] : NEG;
#pragma unroll
                for (int sb = 0; sb < SB; ++sb) {
                    const double w = i < 64 ? readlane_f64(v0[sb], i) : readlane_f64(v1[sb], i - 64);
                    const double s0 = w + a0, s1 = w + a1;
                    b0[sb] = s0 > b0[sb] ? s0 : b0[sb];
                    b1[sb] = s1 > b1[sb] ? s1 : b1[sb];
                }
            }
#pragma unroll
            for (int sb = 0; sb < SB; ++sb) {
                v0[sb] = j0 < K ? e0[u] + b0[sb] : NEG;
                v1[sb] = j1 < K ? e1[u] + b1[sb] : NEG;
            }
        }
    }
#pragma unroll
    for (int sb = 0; sb < SB; ++sb) {
        const int i0 = i00 + sb;
        if (i0 >= Kp) break;
        double* row = M + ((int64_t)c * Kp + i0) * Kp;
        row[j0] = i0 < K ? v0[sb] : NEG;
        if (j1 < Kp) row[j1] = i0 < K ? v1[sb] : NEG;
    }
}

// wstart[c] = omega at the start of chunk c: omega_{c+1} = omega_c (x) M_c, one workgroup, 1024 threads = (end state j,
// eighth p of the start states); the next chunk's entries requested while the current ones are reduced
template <int KT>
__global__ __launch_bounds__(kHmmWideScanThreads) void hmm_vit_scan_wide_kernel(const double* __restrict__ lnrho, int64_t npad,
                                                                                const double* __restrict__ ln_pi_tilde,
                                                                                const double* __restrict__ M, int K, int64_t chunks,
                                                                                double* __restrict__ wstart /*[chunks][Kp]*/,
    const int* __restrict__ gate = nullptr /*hmm.h: the coalescence pass stands*/) {
    if (gate != nullptr && *gate == 0) return;
    constexpr int Kp = 16 * KT, PARTS = kHmmWideScanThreads / 128, IP = Kp / PARTS;
    __shared__ double sw[128];
    __shared__ double spart[PARTS][128];
    const int tid = threadIdx.x, j = tid & 127, p = tid >> 7;
    const double NEG = -1.0e300;
    double w = (j < K) ? lnrho[(int64_t)j * npad] + ln_pi_tilde[j] : NEG;
    double cur[IP], nxt[IP];
    auto fetch = [&](int64_t c, double (&dst)[IP]) {
        const double* P = M + c * Kp * Kp;
#pragma unroll
        for (int q = 0; q < IP; ++q) dst[q] = (j < Kp && p * IP + q < K) ? P[(p * IP + q) * Kp + j] : NEG;      // (rows >= K are not written)
    };
    if (chunks > 0) fetch(0, nxt);
    for (int64_t c = 0; c < chunks; ++c) {
        if (tid < Kp) wstart[c * Kp + tid] = w;
#pragma unroll
        for (int q = 0; q < IP; ++q) cur[q] = nxt[q];
        if (c + 1 < chunks) fetch(c + 1, nxt);
        __syncthreads();
        if (tid < 128) sw[tid] = w;
        __syncthreads();
        double best = NEG * 2.0;
#pragma unroll
        for (int q = 0; q < IP; ++q) {
            const int i = p * IP + q;
            const double s = (i < K ? sw[i] : NEG) + cur[q];
            best = s > best ? s : best;
        }
        spart[p][j] = best;
        __syncthreads();
        double b = spart[0][j];
#pragma unroll
        for (int q = 1; q < PARTS; ++q) b = spart[q][j] > b ? spart[q][j] : b;
        w = j < K ? b : NEG;
    }
}

// per chunk: the reference's recursion from the known start vector, back-pointers phi[t][Kp] (first maximiser).  Eight waves
// = eight chunks share ln a~ in LDS; lane = end states j and j + 64.
template <int KT>
__global__ __launch_bounds__(64 * kVitWideWaves) void hmm_vit_replay_wide_kernel(const double* __restrict__ lnrho, int64_t npad,
                                                                  const double* __restrict__ ln_a_tilde,
                                                                  const double* __restrict__ wstart, int K, int64_t T, int64_t L,
                                                                  int64_t chunks, unsigned char* __restrict__ phi /*[T][Kp]*/,
                                                                  int* __restrict__ last_state,
                                                                  int sweep = 0 /*hmm.h: hmm_vit_replay_kernel*/,
                                                                  double* __restrict__ end_out = nullptr,
                                                                  const int* __restrict__ gate = nullptr) {
    if (gate != nullptr && *gate == 0) return;
    constexpr int Kp = 16 * KT;
    extern __shared__ double a_lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const double NEG = -1.0e300;
    for (int e = threadIdx.x; e < Kp * Kp; e += 64 * kVitWideWaves) {
        const int i = e / Kp, j = e - i * Kp;
        a_lds[e] = (i < K && j < K) ? ln_a_tilde[i * K + j] : NEG;
    }
    __syncthreads();
    const int64_t c = (int64_t)blockIdx.x * kVitWideWaves + wave;
    if (c >= chunks) return;
    const int64_t t0 = 1 + c * L, t1 = (t0 + L < T) ? t0 + L : T;
    const int j0 = lane, j1 = lane + 64;
    const double* lr0 = lnrho + (int64_t)(j0 < K ? j0 : 0) * npad;
    const double* lr1 = lnrho + (int64_t)(j1 < K ? j1 : 0) * npad;
    const bool zero = sweep && c > 0;
    double om0 = j0 < K ? (zero ? 0.0 : wstart[c * Kp + j0]) : NEG;
    double om1 = j1 < K ? (zero ? 0.0 : wstart[c * Kp + j1]) : NEG;
    for (int64_t tb = t0; tb < t1; tb += 8) {
        double e0[8], e1[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            e0[u] = (tb + u < t1) ? lr0[tb + u] : 0.0;
            e1[u] = (tb + u < t1) ? lr1[tb + u] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (tb + u >= t1) break;
            double b0 = NEG * 2.0, b1 = NEG * 2.0;
            int g0 = 0, g1 = 0;
#pragma unroll 4
            for (int i = 0; i < Kp; ++i) {
                const double w = i < 64 ? readlane_f64(om0, i) : readlane_f64(om1, i - 64);
                const double s0 = w + a_lds[i * Kp + j0];
                const double s1 = w + a_lds[i * Kp + (j1 < Kp ? j1 : 0)];
                if (s0 > b0) {           // strict: first maximiser, like numpy.argmax
                    b0 = s0;
                    g0 = i;
                }
                if (s1 > b1) {
                    b1 = s1;
                    g1 = i;
                }
            }
            om0 = j0 < K ? e0[u] + b0 : NEG;
            om1 = j1 < K ? e1[u] + b1 : NEG;
            if (!sweep) {
                phi[(tb + u) * Kp + j0] = (unsigned char)g0;
                if (j1 < Kp) phi[(tb + u) * Kp + j1] = (unsigned char)g1;
            }
        }
    }
    if (end_out != nullptr && c + 1 < chunks) {
        const double m = max_wave(fmax(om0, om1));
        end_out[(c + 1) * Kp + j0] = j0 < K ? om0 - m : NEG;
        if (j1 < Kp) end_out[(c + 1) * Kp + j1] = j1 < K ? om1 - m : NEG;
    }
    if (c == chunks - 1 && !sweep) {     // first maximiser of omega_{T-1}
        double best = om0;
        int arg = j0;
        if (om1 > best) {
            best = om1;
            arg = j1;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double ob = __shfl_xor(best, o);
            const int oa = __shfl_xor(arg, o);
            if (ob > best || (ob == best && oa < arg)) {
                best = ob;
                arg = oa;
            }
        }
        if (lane == 0) *last_state = arg;
    }
}

}  // namespace gmmvb
