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
// the same kernels run for KT = 5 .. 8 tiles of 16 states.  The boundary pass is a sequential pass over the chunk products
// (a row-vector x matrix per chunk, 1024 threads); the xi-sum is split over a workgroup's waves by row tiles; the state arrays, the read-outs and
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
                                                                      double* __restrict__ prod_t /*the transposes*/) {
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

// H3 wide: the sequential pass over the chunk products.  Workgroup 0 forward (fstart[c + 1] ~ fstart[c] P_c), workgroup 1
// backward (bend[c - 1] ~ P_c bend[c]); 1024 threads = (state i, eighth p of the contraction index), the next chunk's
// entries requested while the current ones are reduced.  Natural state order, like hmm_boundary_scan_kernel.
constexpr int kHmmWideScanThreads = 1024;
template <int KT>
__global__ __launch_bounds__(kHmmWideScanThreads) void hmm_boundary_scan_wide_kernel(
    const double* __restrict__ rho_tm, const double* __restrict__ pi_tilde, const double* __restrict__ prod,
    const double* __restrict__ prod_t, int K, int64_t n_chunks, double* __restrict__ fstart, double* __restrict__ bend, double* __restrict__ cprime,
    double* __restrict__ alpha_tm, double* __restrict__ gamma_tm, double* __restrict__ w_tm) {
    constexpr int Kp = 16 * KT, PARTS = kHmmWideScanThreads / 128, JP = Kp / PARTS;      // JP contraction indices per thread
    static_assert(Kp <= 128 && Kp % PARTS == 0, "up to 128 states");
    __shared__ double sv[128];
    __shared__ double spart[PARTS][128];
    __shared__ double sred[2];
    const int tid = threadIdx.x, i = tid & 127, p = tid >> 7;
    const bool fwd = blockIdx.x == 0;
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
    if (fwd) {
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
    if (n_chunks < 2) return;
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
    if (fwd) fetch(0, nxt);
    else fetch(n_chunks - 1, nxt);
    const int64_t steps = n_chunks - 1;
    for (int64_t s = 0; s < steps; ++s) {
        const int64_t c = fwd ? s : n_chunks - 1 - s;            // the product applied in this step
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
                                                                      double* __restrict__ alpha_tm, double* __restrict__ cprime) {
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
        for (int r = 0; r < 4; ++r) al[it][r] = live ? fstart[c * Kp + 16 * it + g + 4 * r] : 0.0;
    if (live && c == 0) {
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
            if (on) *reinterpret_cast<d4*>(alpha_tm + t * Kp + 16 * it + 4 * g) = al[it];
        }
        if (on && g == 0) cprime[t] = cp;
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
                                                                       double* __restrict__ gamma_tm, double* __restrict__ w_tm) {
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
        for (int r = 0; r < 4; ++r) be[it][r] = live ? bend[c * Kp + 16 * it + g + 4 * r] : 0.0;
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
#pragma unroll
        for (int it = 0; it < KT; ++it) {
            d4 gm, ww;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                y[it][r] = rho[it][r] * be[it][r];
                gm[r] = al[it][r] * be[it][r] * ginv;
                ww[r] = y[it][r] * winv;
            }
            if (on) {
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
    if (live && c == 0) {                                       // gamma_0 = alpha_0 o beta~_0, normalised
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

}  // namespace gmmvb
