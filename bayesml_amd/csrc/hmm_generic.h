// Forward-backward and Viterbi for MORE than 64 hidden states: the plain form of the recursions of hmm.h.
//
// The chunk-parallel kernels of hmm.h keep K x K transfer products and the state vectors of 16 chunks in MFMA registers,
// which ends at K = 64 (KT = 4 tiles of 16 states).  The reference takes any number of classes
// (bayesml/hiddenmarkovnormal/_hiddenmarkovnormal.py:481 c_num_classes), so beyond 64 the same quantities - alpha_t, c'_t
// (:999-1006), gamma_t (:1013-1014), w_t and with it sum_t xi_t (:1016-1018, :839), the Viterbi path (:1465-1481) - are
// produced by ONE workgroup walking the sequence: a K x K matrix-vector product per time step, split over 1024 threads,
// the transition matrix in LDS while it fits (K <= 128) and in L2 beyond.  Same arrays, same layouts (time-major, "lane
// order" positions hmm_pos within blocks of 16 states, Kp = 16 ceil(K / 16)), so emission, M-step, read-outs and the
// K-side are the ones of the fast path.  A correctness path: ~0.3-1 us per time step.
#pragma once
#include "hmm.h"

namespace gmmvb {

constexpr int kHmmSeqThreads = 1024;

// ln rho [K][npad] -> rho' [T][Kp] lane order and mx[T] for any K: 64 time steps per workgroup, the components in tiles
// of 64 through LDS (first pass: the row maxima; second pass: exponentials, written time-major)
__global__ __launch_bounds__(256) void hmm_prep_generic_kernel(const double* __restrict__ lnrho, int64_t npad, int64_t T, int K,
                                                               int Kp, double* __restrict__ rho_tm, double* __restrict__ mx) {
    __shared__ double tile[64][65];
    __shared__ double smx[4][64];
    const int tid = threadIdx.x, tq = tid & 63, kk = tid >> 6;
    const int64_t t0 = (int64_t)blockIdx.x * 64;
    const int64_t tc = t0 + tq < T ? t0 + tq : T - 1;
    double m = -__builtin_huge_val();
    for (int k = kk; k < K; k += 4) m = fmax(m, lnrho[(int64_t)k * npad + tc]);
    smx[kk][tq] = m;
    __syncthreads();
    if (tid < 64) {
        const double mm = fmax(fmax(smx[0][tid], smx[1][tid]), fmax(smx[2][tid], smx[3][tid]));
        smx[0][tid] = mm;
        if (t0 + tid < T) mx[t0 + tid] = mm;
    }
    __syncthreads();
    for (int k0 = 0; k0 < Kp; k0 += 64) {
        for (int k = kk; k < 64; k += 4) tile[k][tq] = (k0 + k < K) ? lnrho[(int64_t)(k0 + k) * npad + tc] : 0.0;
        __syncthreads();
        for (int e = tid; e < 64 * 64; e += 256) {
            const int t = e >> 6, p = k0 + (e & 63);
            if (t0 + t >= T || p >= Kp) continue;
            const int k = hmm_state(p);
            rho_tm[(t0 + t) * Kp + p] = k < K ? exp(tile[k - k0][t] - smx[0][t]) : 0.0;
        }
        __syncthreads();
    }
}

// sum over the workgroup (every thread gets it); red: [16] doubles of LDS
__device__ __forceinline__ double block_sum_1024(double v, double* red) {
    v = sum_wave(v);
    __syncthreads();                                   // (red may still be read from the previous call)
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < kHmmSeqThreads / 64; ++w) s += red[w];
    return s;
}

// out[j] = sum_i vec[i] mat[i * K + j]  (j = 0 .. K-1) with the K x K matrix row-major: parts of the i range per thread,
// adjacent lanes take adjacent j (contiguous reads), partial sums through LDS in a fixed order.  P = parts, J = kHmmSeqThreads / P
// outputs per round.  `part_buf`: [P][J] doubles of LDS.  The result lands in out_lds[j] (natural state order); two barriers.
__device__ __forceinline__ void seq_matvec(const double* mat, int K, const double* vec_lds, int P, int J, double* part_buf,
                                           double* out_lds) {
    const int tid = threadIdx.x, part = tid / J, jl = tid - part * J;
    const int per = (K + P - 1) / P, i0 = part * per, i1 = i0 + per < K ? i0 + per : K;
    for (int j0 = 0; j0 < K; j0 += J) {
        const int j = j0 + jl;
        double acc = 0.0;
        if (j < K) {
#pragma unroll 8
            for (int i = i0; i < i1; ++i) acc = fma(vec_lds[i], mat[(int64_t)i * K + j], acc);
        }
        part_buf[part * J + jl] = acc;
        __syncthreads();
        if (tid < J && j0 + tid < K) {
            double s = 0.0;
            for (int p = 0; p < P; ++p) s += part_buf[p * J + tid];
            out_lds[j0 + tid] = s;
        }
        __syncthreads();
    }
}

struct HmmSeqShape {
    int P, J;            // parts of a dot product, outputs per round (P J = kHmmSeqThreads)
    int mat_in_lds;      // the K x K matrix is staged in LDS
    size_t lds_bytes;
};
inline HmmSeqShape hmm_seq_shape(int K) {
    HmmSeqShape s;
    int J = 64;
    while (J < K && J < kHmmSeqThreads) J *= 2;
    s.J = J;
    s.P = kHmmSeqThreads / J;
    const size_t vecs = (size_t)(3 * K + kHmmSeqThreads + 32) * sizeof(double);
    s.mat_in_lds = vecs + (size_t)K * K * sizeof(double) <= 150 * 1024 ? 1 : 0;
    s.lds_bytes = vecs + (s.mat_in_lds ? (size_t)K * K * sizeof(double) : 0);
    return s;
}

// Forward pass (:999-1006): alpha_0 = pi~ o rho'_0 / c'_0, alpha_t = rho'_t o (alpha_{t-1} A~) / c'_t.  Writes alpha_tm (lane
// order) and c'.  T == 1: gamma_0 = alpha_0, w_0 = 0 too.
__global__ __launch_bounds__(kHmmSeqThreads) void hmm_seq_forward_kernel(const double* __restrict__ rho_tm,
                                                                         const double* __restrict__ pi_tilde,
                                                                         const double* __restrict__ a_tilde, int K, int Kp,
                                                                         int64_t T, int P, int J, int mat_in_lds,
                                                                         double* __restrict__ alpha_tm,
                                                                         double* __restrict__ cprime,
                                                                         double* __restrict__ gamma_tm, double* __restrict__ w_tm,
                                                                         // chunk form (L > 0; hmm.h: the forgetting pass): workgroup c
                                                                         // walks steps 1 + c L .. from fstart[c] (natural order) - or,
                                                                         // sweep, from the uniform vector, storing nothing - and leaves
                                                                         // its last alpha in end_out[c + 1]
                                                                         int64_t L = 0, const double* __restrict__ fstart = nullptr,
                                                                         int sweep = 0, double* __restrict__ end_out = nullptr,
                                                                         const int* __restrict__ gate = nullptr) {
    if (gate != nullptr && *gate == 0) return;
    extern __shared__ double sh[];
    double* va = sh;                       // [K] alpha_{t-1}
    double* vb = va + K;                   // [K] alpha_{t-1} A~
    double* red = vb + 2 * K;              // [32]
    double* part_buf = red + 32;           // [kHmmSeqThreads]
    double* mat_l = part_buf + kHmmSeqThreads;
    const int tid = threadIdx.x;
    const double* mat = a_tilde;
    if (mat_in_lds) {
        for (int e = tid; e < K * K; e += kHmmSeqThreads) mat_l[e] = a_tilde[e];
        mat = mat_l;
    }
    const int64_t chunk = L > 0 ? (int64_t)blockIdx.x : 0;
    const int64_t t_lo = 1 + chunk * L, t_hi = (L > 0 && t_lo + L < T) ? t_lo + L : T;
    if (chunk == 0) {
        double v = 0.0;
        for (int k = tid; k < K; k += kHmmSeqThreads) v += rho_tm[hmm_pos(k)] * pi_tilde[k];
        const double s = block_sum_1024(v, red);
        for (int k = tid; k < K; k += kHmmSeqThreads) {
            const double a0 = rho_tm[hmm_pos(k)] * pi_tilde[k];
            const double an = s > 0.0 ? a0 / s : 0.0;
            va[k] = an;
            if (!sweep) {
                alpha_tm[hmm_pos(k)] = an;
                if (T == 1) {
                    gamma_tm[hmm_pos(k)] = an;
                    w_tm[hmm_pos(k)] = 0.0;
                }
            }
        }
        if (!sweep) {
            for (int k = K + tid; k < Kp; k += kHmmSeqThreads) {          // padding states of the last block of 16
                alpha_tm[hmm_pos(k)] = 0.0;
                if (T == 1) gamma_tm[hmm_pos(k)] = w_tm[hmm_pos(k)] = 0.0;
            }
            if (tid == 0) cprime[0] = s;
        }
    } else {
        for (int k = tid; k < K; k += kHmmSeqThreads) va[k] = sweep ? 1.0 / K : fstart[chunk * Kp + k];
    }
    __syncthreads();
    // (the time-major arrays are read one step ahead for this thread's first two states: their latency then lies beside the
    // matrix-vector product instead of in front of the step's dependent arithmetic)
    const int k0 = tid, k1 = tid + kHmmSeqThreads;
    const int p0 = hmm_pos(k0 < K ? k0 : 0), p1 = hmm_pos(k1 < K ? k1 : 0);
    double r0 = (t_lo < t_hi && k0 < K) ? rho_tm[t_lo * Kp + p0] : 0.0, r1 = (t_lo < t_hi && k1 < K) ? rho_tm[t_lo * Kp + p1] : 0.0;
    for (int64_t t = t_lo; t < t_hi; ++t) {
        const double c0 = r0, c1 = r1;
        if (t + 1 < t_hi) {
            if (k0 < K) r0 = rho_tm[(t + 1) * Kp + p0];
            if (k1 < K) r1 = rho_tm[(t + 1) * Kp + p1];
        }
        seq_matvec(mat, K, va, P, J, part_buf, vb);
        double part = 0.0;
        for (int k = tid; k < K; k += kHmmSeqThreads) {
            const double rr = k == k0 ? c0 : (k == k1 ? c1 : rho_tm[t * Kp + hmm_pos(k)]);
            const double nw = vb[k] * rr;
            vb[k] = nw;
            part += nw;
        }
        const double cp = block_sum_1024(part, red);
        const double inv = cp > 0.0 ? 1.0 / cp : 0.0;
        for (int k = tid; k < K; k += kHmmSeqThreads) {
            const double an = vb[k] * inv;
            va[k] = an;
            if (!sweep) alpha_tm[t * Kp + hmm_pos(k)] = an;
        }
        if (!sweep) {
            for (int k = K + tid; k < Kp; k += kHmmSeqThreads) alpha_tm[t * Kp + hmm_pos(k)] = 0.0;
            if (tid == 0) cprime[t] = cp;
        }
        __syncthreads();
    }
    if (end_out != nullptr && L > 0 && t_hi < T)                    // (a further chunk follows)
        for (int k = tid; k < K; k += kHmmSeqThreads) end_out[(chunk + 1) * Kp + k] = va[k];
}

// Backward pass (:1008-1014) with gamma_t and w_t = rho'_t o beta~_t / (c'_t (alpha_t . beta~_t)) (hmm.h): beta~_{T-1}
// uniform, beta~_{t-1} ~ A~ (rho'_t o beta~_t) renormalised to sum 1.  `a_tilde_t` is the TRANSPOSE of A~ (row-major), so
// that the matrix-vector product reads it like the forward pass reads A~.
__global__ __launch_bounds__(kHmmSeqThreads) void hmm_seq_backward_kernel(const double* __restrict__ rho_tm,
                                                                          const double* __restrict__ a_tilde_t, int K, int Kp,
                                                                          int64_t T, int P, int J, int mat_in_lds,
                                                                          const double* __restrict__ alpha_tm,
                                                                          const double* __restrict__ cprime,
                                                                          double* __restrict__ gamma_tm, double* __restrict__ w_tm,
                                                                          // chunk form (L > 0): workgroup c walks its steps downwards from
                                                                          // bend[c] (sweep: uniform, no alpha, no stores) and leaves the
                                                                          // beta~ in front of it in bend_out[c - 1]
                                                                          int64_t L = 0, const double* __restrict__ bend = nullptr,
                                                                          int sweep = 0, double* __restrict__ bend_out = nullptr,
                                                                          const int* __restrict__ gate = nullptr) {
    if (gate != nullptr && *gate == 0) return;
    extern __shared__ double sh[];
    double* be = sh;                       // [K] beta~_t
    double* y = be + K;                    // [K] rho'_t o beta~_t
    double* nb = y + K;                    // [K] A~ y
    double* red = nb + K;                  // [32]
    double* part_buf = red + 32;
    double* mat_l = part_buf + kHmmSeqThreads;
    const int tid = threadIdx.x;
    const double* mat = a_tilde_t;
    if (mat_in_lds) {
        for (int e = tid; e < K * K; e += kHmmSeqThreads) mat_l[e] = a_tilde_t[e];
        mat = mat_l;
    }
    const int64_t chunk = L > 0 ? (int64_t)blockIdx.x : 0;
    const int64_t t_lo = 1 + chunk * L, t_hi = (L > 0 && t_lo + L < T) ? t_lo + L : T;      // steps t_hi - 1 .. t_lo
    for (int k = tid; k < K; k += kHmmSeqThreads) be[k] = (L > 0 && !sweep && t_hi < T) ? bend[chunk * Kp + k] : 1.0 / K;
    __syncthreads();
    const int k0 = tid, k1 = tid + kHmmSeqThreads;
    const int p0 = hmm_pos(k0 < K ? k0 : 0), p1 = hmm_pos(k1 < K ? k1 : 0);
    double a0 = 0.0, a1 = 0.0, r0 = 0.0, r1 = 0.0, cpn = 1.0;         // alpha_t, rho'_t, c'_t one step ahead
    if (t_hi > t_lo) {
        if (k0 < K) { if (!sweep) a0 = alpha_tm[(t_hi - 1) * Kp + p0]; r0 = rho_tm[(t_hi - 1) * Kp + p0]; }
        if (k1 < K) { if (!sweep) a1 = alpha_tm[(t_hi - 1) * Kp + p1]; r1 = rho_tm[(t_hi - 1) * Kp + p1]; }
        if (!sweep) cpn = cprime[t_hi - 1];
    }
    for (int64_t t = t_hi - 1; t >= t_lo; --t) {
        const double ca0 = a0, ca1 = a1, cr0 = r0, cr1 = r1, cp = cpn;
        if (t - 1 >= t_lo) {
            if (k0 < K) { if (!sweep) a0 = alpha_tm[(t - 1) * Kp + p0]; r0 = rho_tm[(t - 1) * Kp + p0]; }
            if (k1 < K) { if (!sweep) a1 = alpha_tm[(t - 1) * Kp + p1]; r1 = rho_tm[(t - 1) * Kp + p1]; }
            if (!sweep) cpn = cprime[t - 1];
        }
        double ginv = 0.0, winv = 0.0;
        if (!sweep) {                                                  // (uniform)
            double dot = 0.0;
            for (int k = tid; k < K; k += kHmmSeqThreads) {
                const double al = k == k0 ? ca0 : (k == k1 ? ca1 : alpha_tm[t * Kp + hmm_pos(k)]);
                dot = fma(al, be[k], dot);
            }
            dot = block_sum_1024(dot, red);
            ginv = dot > 0.0 ? 1.0 / dot : 0.0;
            winv = (dot > 0.0 && cp > 0.0) ? 1.0 / (dot * cp) : 0.0;
        }
        for (int k = tid; k < K; k += kHmmSeqThreads) {
            const int p = hmm_pos(k);
            const double rr = k == k0 ? cr0 : (k == k1 ? cr1 : rho_tm[t * Kp + p]);
            const double yy = rr * be[k];
            y[k] = yy;
            if (!sweep) {
                const double al = k == k0 ? ca0 : (k == k1 ? ca1 : alpha_tm[t * Kp + p]);
                gamma_tm[t * Kp + p] = al * be[k] * ginv;
                w_tm[t * Kp + p] = yy * winv;
            }
        }
        if (!sweep)
            for (int k = K + tid; k < Kp; k += kHmmSeqThreads) gamma_tm[t * Kp + hmm_pos(k)] = w_tm[t * Kp + hmm_pos(k)] = 0.0;
        __syncthreads();
        seq_matvec(mat, K, y, P, J, part_buf, nb);
        double part = 0.0;
        for (int k = tid; k < K; k += kHmmSeqThreads) part += nb[k];
        const double tot = block_sum_1024(part, red);
        const double inv = tot > 0.0 ? 1.0 / tot : 0.0;
        for (int k = tid; k < K; k += kHmmSeqThreads) be[k] = nb[k] * inv;
        __syncthreads();
    }
    if (bend_out != nullptr && L > 0 && chunk >= 1)
        for (int k = tid; k < K; k += kHmmSeqThreads) bend_out[(chunk - 1) * Kp + k] = be[k];
    if (T > 1 && chunk == 0 && !sweep) {                           // gamma_0 = alpha_0 o beta~_0, normalised; xi_0 = 0
        double dot = 0.0;
        for (int k = tid; k < K; k += kHmmSeqThreads) dot = fma(alpha_tm[hmm_pos(k)], be[k], dot);
        dot = block_sum_1024(dot, red);
        const double ginv = dot > 0.0 ? 1.0 / dot : 0.0;
        for (int k = tid; k < K; k += kHmmSeqThreads) {
            gamma_tm[hmm_pos(k)] = alpha_tm[hmm_pos(k)] * be[k] * ginv;
            w_tm[hmm_pos(k)] = 0.0;
        }
        for (int k = K + tid; k < Kp; k += kHmmSeqThreads) gamma_tm[hmm_pos(k)] = w_tm[hmm_pos(k)] = 0.0;
    }
}

__global__ void hmm_transpose_kernel(const double* __restrict__ a, int K, double* __restrict__ at) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < K * K) at[(e % K) * K + e / K] = a[e];
}

// raw[pi][pj] = sum_{t in the slab} alpha_tm[t-1][pi] w_tm[t][pj] for one 16 x 16 tile of lane-order positions and one
// slab of time steps: slabs[slab][Kp][Kp], summed by hmm_finish_kernel in slab order.  grid = (slabs, (Kp / 16)^2).
__global__ __launch_bounds__(256) void hmm_xi_generic_kernel(const double* __restrict__ alpha_tm, const double* __restrict__ w_tm,
                                                             int Kp, int64_t T, int64_t steps_per_slab,
                                                             double* __restrict__ slabs) {
    const int tiles = Kp / 16;
    const int ti = blockIdx.y / tiles, tj = blockIdx.y - ti * tiles;
    const int i = threadIdx.x >> 4, j = threadIdx.x & 15;
    const int64_t lo = 1 + (int64_t)blockIdx.x * steps_per_slab;
    int64_t hi = lo + steps_per_slab;
    if (hi > T) hi = T;
    double acc = 0.0;
    const double* a = alpha_tm + 16 * ti + i;
    const double* w = w_tm + 16 * tj + j;
    for (int64_t t = lo; t < hi; ++t) acc = fma(a[(t - 1) * Kp], w[t * Kp], acc);
    slabs[(int64_t)blockIdx.x * Kp * Kp + (16 * ti + i) * Kp + 16 * tj + j] = acc;
}

// ---- Viterbi (:1465-1481) for any K: omega_t(j) = ln rho_t(j) + max_i (omega_{t-1}(i) + ln a~_ij), first maximiser like
// numpy.argmax; back-pointers as 16-bit integers [T][K] (natural order).  One workgroup; thread groups like seq_matvec, the
// parts' (best, arg) pairs combined in part order (ascending i: the first maximiser survives).
__global__ __launch_bounds__(kHmmSeqThreads) void hmm_seq_viterbi_kernel(const double* __restrict__ lnrho, int64_t npad,
                                                                         const double* __restrict__ ln_pi_tilde,
                                                                         const double* __restrict__ ln_a_tilde, int K, int64_t T,
                                                                         int P, int J, int mat_in_lds,
                                                                         unsigned short* __restrict__ phi,
                                                                         int* __restrict__ last_state,
                                                                         // chunk form (L > 0; hmm.h: the coalescence pass): workgroup c
                                                                         // walks steps 1 + c L .. from wstart[c] ([chunks][Kp], natural
                                                                         // order) - or, sweep, from zero, storing no back-pointers - and
                                                                         // leaves omega minus its maximum in end_out[c + 1]
                                                                         int64_t L = 0, int Kp = 0, const double* __restrict__ wstart = nullptr,
                                                                         int sweep = 0, double* __restrict__ end_out = nullptr,
                                                                         const int* __restrict__ gate = nullptr) {
    if (gate != nullptr && *gate == 0) return;
    extern __shared__ double sh[];
    double* om = sh;                       // [K] omega_{t-1}
    double* nw = om + K;                   // [K]
    double* best_buf = nw + 2 * K + 32;    // [kHmmSeqThreads]  (the slot of part_buf)
    double* mat_l = best_buf + kHmmSeqThreads;
    __shared__ int arg_buf[kHmmSeqThreads];
    const int tid = threadIdx.x, part = tid / J, jl = tid - part * J;
    const int per = (K + P - 1) / P, i0 = part * per, i1 = i0 + per < K ? i0 + per : K;
    const double* mat = ln_a_tilde;
    if (mat_in_lds) {
        for (int e = tid; e < K * K; e += kHmmSeqThreads) mat_l[e] = ln_a_tilde[e];
        mat = mat_l;
    }
    const int64_t chunk = L > 0 ? (int64_t)blockIdx.x : 0;
    const int64_t t_lo = 1 + chunk * L, t_hi = (L > 0 && t_lo + L < T) ? t_lo + L : T;
    for (int k = tid; k < K; k += kHmmSeqThreads)
        om[k] = chunk == 0 ? lnrho[(int64_t)k * npad] + ln_pi_tilde[k] : (sweep ? 0.0 : wstart[chunk * Kp + k]);
    __syncthreads();
    const double NEG = -__builtin_huge_val();
    for (int64_t t = t_lo; t < t_hi; ++t) {
        for (int j0 = 0; j0 < K; j0 += J) {
            const int j = j0 + jl;
            double best = NEG;
            int arg = i0 < K ? i0 : 0;
            if (j < K)
                for (int i = i0; i < i1; ++i) {
                    const double v = om[i] + mat[(int64_t)i * K + j];
                    if (v > best) {          // strict: first maximiser
                        best = v;
                        arg = i;
                    }
                }
            best_buf[part * J + jl] = best;
            arg_buf[part * J + jl] = arg;
            __syncthreads();
            if (tid < J && j0 + tid < K) {
                double b = best_buf[tid];
                int a = arg_buf[tid];
                for (int p = 1; p < P; ++p) {
                    const double ob = best_buf[p * J + tid];
                    if (ob > b) {
                        b = ob;
                        a = arg_buf[p * J + tid];
                    }
                }
                nw[j0 + tid] = lnrho[(int64_t)(j0 + tid) * npad + t] + b;
                if (!sweep) phi[t * K + j0 + tid] = (unsigned short)a;
            }
            __syncthreads();
        }
        for (int k = tid; k < K; k += kHmmSeqThreads) om[k] = nw[k];
        __syncthreads();
    }
    if (end_out != nullptr && L > 0 && t_hi < T) {       // (a further chunk follows)
        double m = NEG;
        for (int k = 0; k < K; ++k) m = fmax(m, om[k]);          // (every thread: K broadcast reads of LDS)
        for (int k = tid; k < K; k += kHmmSeqThreads) end_out[(chunk + 1) * Kp + k] = om[k] - m;
    }
    if (tid == 0 && t_hi == T && !sweep) {   // first maximiser of omega_{T-1}
        double b = om[0];
        int a = 0;
        for (int k = 1; k < K; ++k)
            if (om[k] > b) {
                b = om[k];
                a = k;
            }
        *last_state = a;
    }
}

__global__ void hmm_seq_backtrack_kernel(const unsigned short* __restrict__ phi, int K, int64_t T,
                                         const int* __restrict__ last_state, int32_t* __restrict__ z) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int k = *last_state;
    z[T - 1] = k;
    for (int64_t t = T - 1; t >= 1; --t) {
        k = phi[t * K + k];
        z[t - 1] = k;
    }
}

// The same path chunk-parallel (16-bit back-pointers, natural order): per chunk the map end state -> state in front of the
// chunk (a thread per end state follows its pointers), one thread walks the chunks' maps back from the last state, and every
// chunk fills in its stretch.  z[t] for t in [1 + c L - 1, t_hi) is written by chunk c.
__global__ __launch_bounds__(256) void hmm_seq_backmap_kernel(const unsigned short* __restrict__ phi, int K, int64_t T, int64_t L,
                                                              unsigned short* __restrict__ map /*[chunks][K]*/) {
    const int64_t c = blockIdx.x, t_lo = 1 + c * L, t_hi = t_lo + L < T ? t_lo + L : T;
    for (int k = threadIdx.x; k < K; k += 256) {
        int s = k;
        for (int64_t t = t_hi - 1; t >= t_lo; --t) s = phi[t * K + s];
        map[c * K + k] = (unsigned short)s;
    }
}
__global__ void hmm_seq_backscan_kernel(const unsigned short* __restrict__ map, int K, int64_t chunks, const int* __restrict__ last_state,
                                        int* __restrict__ endst /*[chunks]: the path's state at the chunk's last step*/) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int s = *last_state;
    for (int64_t c = chunks - 1; c >= 0; --c) {
        endst[c] = s;
        s = map[c * K + s];
    }
}
__global__ __launch_bounds__(64) void hmm_seq_fill_kernel(const unsigned short* __restrict__ phi, int K, int64_t T, int64_t L,
                                                          int64_t chunks, const int* __restrict__ endst, int32_t* __restrict__ z) {
    const int64_t c = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (c >= chunks) return;
    const int64_t t_lo = 1 + c * L, t_hi = t_lo + L < T ? t_lo + L : T;
    int s = endst[c];
    z[t_hi - 1] = s;
    for (int64_t t = t_hi - 1; t >= t_lo; --t) {
        s = phi[t * K + s];
        z[t - 1] = s;
    }
}

}  // namespace gmmvb
