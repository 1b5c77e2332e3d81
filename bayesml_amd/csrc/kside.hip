// K-sized linear algebra of the posterior update that the torch K-side would otherwise hand to MAGMA / rocSOLVER:
// one Cholesky factorisation and one triangular inverse per component, LDS-resident (D <= 128: 129 KB of the 160 KB).
//
// Replaces, per VB iteration, what the reference does with np.linalg.inv + two slogdet per component
// (bayesml/gaussianmixture/_gaussianmixture.py:746-756, 769): from W^-1 = G G^T it yields G (-> u^-1 = G / sqrt(nu)),
// G^-1 (-> u = sqrt(nu) G^-1, W = G^-T G^-1) and ln det W^-1 = 2 sum ln diag G.  Plain kernels on the caller's stream,
// no host synchronisation, no allocation: the whole K-side can be captured in a hipGraph.
#include "workspace.h"
#include "common.h"

#include <cstring>
#include <mutex>
#include <vector>

namespace gmmvb {

// One workgroup per matrix.  a [K][D][D] symmetric positive definite (row-major; only the lower triangle is read).
// g, g_inv [K][D][D] lower triangular (upper part zero filled); logdet [K] = 2 sum_j ln g_jj.
// A non-positive pivot gives NaN from sqrt and the NaNs spread (like the reference's inv() on a singular matrix
// this then shows up in the lower bound instead of raising).
__global__ __launch_bounds__(256) void chol_inv_kernel(const double* __restrict__ a, int D, double* __restrict__ g,
                                                       double* __restrict__ g_inv, double* __restrict__ logdet) {
    extern __shared__ double sm[];           // [D][D + 1]
    const int ld = D + 1;
    const int tid = threadIdx.x;
    const int64_t base = (int64_t)blockIdx.x * D * D;
    for (int e = tid; e < D * D; e += 256) {
        const int i = e / D, j = e - i * D;
        sm[i * ld + j] = j <= i ? a[base + e] : 0.0;
    }
    __syncthreads();
    // right-looking Cholesky, column by column; the trailing update runs on a 16 x 16 thread grid (rows ti + 16 a,
    // columns tj + 16 b of the trailing block: no index division, conflict-free column reads)
    const int ti = tid >> 4, tj = tid & 15;
    for (int j = 0; j < D; ++j) {
        if (tid == 0) sm[j * ld + j] = sqrt(sm[j * ld + j]);
        __syncthreads();
        const double inv = 1.0 / sm[j * ld + j];
        for (int i = j + 1 + tid; i < D; i += 256) sm[i * ld + j] *= inv;
        __syncthreads();
        for (int i = j + 1 + ti; i < D; i += 16) {
            const double lij = sm[i * ld + j];
            for (int c = j + 1 + tj; c <= i; c += 16) sm[i * ld + c] = fma(-lij, sm[c * ld + j], sm[i * ld + c]);
        }
        __syncthreads();
    }
    for (int e = tid; e < D * D; e += 256) {
        const int i = e / D, j = e - i * D;
        g[base + e] = sm[i * ld + j];
    }
    if (tid < 64) {          // ln det = 2 sum ln g_jj (one wave, fixed order)
        double s = 0.0;
        for (int j = tid; j < D; j += 64) s += log(sm[j * ld + j]);
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (tid == 0) logdet[blockIdx.x] = 2.0 * s;
    }
    __syncthreads();
    // in-place inverse of the lower-triangular factor, last column first (the LAPACK trti2 order):
    //   x_jj = 1 / g_jj,  x[j+1:, j] = -X[j+1:, j+1:] g[j+1:, j] x_jj   with X[j+1:, j+1:] already inverted
    for (int j = D - 1; j >= 0; --j) {
        const double xjj = 1.0 / sm[j * ld + j];
        double v = 0.0;
        const int i = j + 1 + tid;
        if (i < D) {
            double v1 = 0.0, v2 = 0.0, v3 = 0.0;          // four independent chains: the LDS latency is the cost here
            int p = j + 1;
            for (; p + 3 <= i; p += 4) {
                v = fma(sm[i * ld + p], sm[p * ld + j], v);
                v1 = fma(sm[i * ld + p + 1], sm[(p + 1) * ld + j], v1);
                v2 = fma(sm[i * ld + p + 2], sm[(p + 2) * ld + j], v2);
                v3 = fma(sm[i * ld + p + 3], sm[(p + 3) * ld + j], v3);
            }
            for (; p <= i; ++p) v = fma(sm[i * ld + p], sm[p * ld + j], v);
            v = -((v + v1) + (v2 + v3)) * xjj;
        }
        __syncthreads();
        if (i < D) sm[i * ld + j] = v;
        if (tid == 0) sm[j * ld + j] = xjj;
        __syncthreads();
    }
    for (int e = tid; e < D * D; e += 256) {
        const int i = e / D, j = e - i * D;
        g_inv[base + e] = sm[i * ld + j];
    }
}

// ---- drift hint (gmmvb_set_drift) in one launch -------------------------------------------------------------------
// Per component and direction (blockIdx.y = 0: A = u_old u_new^-1 -> gamma = 1 / ||A||_2;  1: A = u_new u_old^-1 ->
// big_gamma = ||A||_2):  G = A^T A, then `sq` times G <- G^2 / ||G^2||_F with the logarithms of the Frobenius norms
// summed with weights 2^-i:  ||A||_2^2 = lambda_max(G) <= ||G^(2^s)||_F^(1/2^s), a rigorous upper bound, at most
// D^(1/2^(s+1)) above the true norm.  The matrix lives in LDS ([128][129] doubles); a product is formed with every
// thread holding an 8 x 8 block of the result in registers (rows ty + 16 a, columns tx + 16 b: conflict-free LDS
// reads), then written back in place.  f64 FMA and f64 MFMA have the same peak on gfx950, so plain FMAs lose nothing.
// Direction 2 does the same for E = u_new u_old^-1 - I -> enorm >= ||E||_2: once the components hardly move, 1 -/+ enorm
// bounds the extreme singular values of u_new u_old^-1 far better than the two direct bounds (whose looseness factor
// D^(1/2^(s+1)) applies to a norm close to 1; here it applies to a norm close to 0).  The caller takes the better of the two.
// Block (k, 0) also computes delta[k] = || u_new (m_new - m_old) ||.
constexpr int kDriftLd = 129;

template <int NW>
__device__ __forceinline__ double block_sum_waves(double v, double* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = (red[0] + red[1]) + (red[2] + red[3]);      // fixed order
    if constexpr (NW == 8) t += (red[4] + red[5]) + (red[6] + red[7]);
    return t;
}

// PD = D rounded up to 32, 64 or 128: the products are PD x PD x PD (a D = 64 model does an eighth of the 128-wide work),
// every thread holds a (PD / 16)^2 block of the result.
// NW waves per workgroup: four, or eight for PD = 128 (round 4: with one wave per SIMD every MFMA group waited out the LDS
// latency of its operands; two waves per SIMD overlap them, and a wave's accumulator block halves to one tile row:
// 0.31 -> 0.21 ms per launch at the benchmark shape).
template <int PD, int NW>
__global__ __launch_bounds__(64 * NW) void drift_kernel(const double* __restrict__ u_old, const double* __restrict__ uinv_old,
                                                    const double* __restrict__ m_old, const double* __restrict__ u_new,
                                                    const double* __restrict__ uinv_new, const double* __restrict__ m_new,
                                                    int D, int sq_small, int sq_big, double* __restrict__ gamma,
                                                    double* __restrict__ delta, double* __restrict__ big,
                                                    double* __restrict__ enorm) {
    extern __shared__ double sm[];           // [PD][kDriftLd]
    __shared__ double red[NW];
    const int tid = threadIdx.x;
    const int k = blockIdx.x, dir = blockIdx.y;
    const int64_t base = (int64_t)k * D * D;
    const double* L = (dir == 0 ? u_old : u_new) + base;          // A = L R  (dir 2: A - I)
    const double* R = (dir == 0 ? uinv_new : uinv_old) + base;
    const int sq = dir == 1 ? sq_big : sq_small;
    const double tiny = 2.2250738585072014e-308;

    if (dir == 0) {          // delta
        // a wave per row: coalesced row reads, independent rows in flight (a thread per row walked its row with 128
        // dependent strided loads)
        const int lane = tid & 63, wv = tid >> 6;
        double dmv[2];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int i = lane + 64 * hh;
            dmv[hh] = i < D ? m_new[(int64_t)k * D + i] - m_old[(int64_t)k * D + i] : 0.0;
        }
        double part = 0.0;
        for (int r = wv; r < D; r += NW) {
            double y = 0.0;
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const int i = lane + 64 * hh;
                if (i < D) y = fma(u_new[base + (int64_t)r * D + i], dmv[hh], y);
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) y += __shfl_xor(y, o);
            part += lane == 0 ? y * y : 0.0;
        }
        const double t = block_sum_waves<NW>(part, red);
        if (tid == 0) delta[k] = sqrt(t) * (1.0 + 1e-9);
    }
    // stage L (zero padded to PD x PD)
    for (int e = tid; e < PD * PD; e += 64 * NW) {
        const int i = e / PD, j = e % PD;
        sm[i * kDriftLd + j] = (i < D && j < D) ? L[(int64_t)i * D + j] : 0.0;
    }
    __syncthreads();
    // Every product runs on v_mfma_f64_16x16x4_f64 (round 3; before: 64 FMAs per thread and step with both operands from
    // LDS, 35 us per 128^3 product where the pipe needs 14): the PD x PD result is TR x TR tiles of 16 x 16, wave w owns
    // the tile rows w, w + 4, ... (TR / 4 of them, at least one) and all TR tile columns; lane (i = l & 15, g = l >> 4)
    // supplies X[16 tr + i][p + g] and Y[p + g][16 tc + i] for the four steps p .. p + 3 of the contraction.
    constexpr int TR = PD / 16, RW = TR >= NW ? TR / NW : 1;
    const int lane = tid & 63, wv = tid >> 6, li = lane & 15, lg = lane >> 4;
    const bool wave_on = wv * RW < TR;                 // (PD = 32: two of the four waves have a tile row)
    d4 acc[RW][TR];
    auto zero_acc = [&]() {
#pragma unroll
        for (int a2 = 0; a2 < RW; ++a2)
#pragma unroll
            for (int b2 = 0; b2 < TR; ++b2) acc[a2][b2] = d4{0.0, 0.0, 0.0, 0.0};
    };
    // result tile (a2, b2), register r  <->  row 16 (RW wv + a2) + lg + 4 r, column 16 b2 + li
    // A = L R : R straight from global memory / L2 (rows of R are contiguous in j: lanes li read neighbours)
    zero_acc();
    if (wave_on) {
        for (int p = 0; p < PD; p += 4) {
            double av[RW], bv[TR];
#pragma unroll
            for (int a2 = 0; a2 < RW; ++a2) av[a2] = sm[(16 * (RW * wv + a2) + li) * kDriftLd + p + lg];
#pragma unroll
            for (int b2 = 0; b2 < TR; ++b2) {
                const int rr = p + lg, cc = 16 * b2 + li;
                bv[b2] = (rr < D && cc < D) ? R[(int64_t)rr * D + cc] : 0.0;
            }
#pragma unroll
            for (int a2 = 0; a2 < RW; ++a2)
#pragma unroll
                for (int b2 = 0; b2 < TR; ++b2) acc[a2][b2] = mfma_f64(av[a2], bv[b2], acc[a2][b2]);
        }
    }
    __syncthreads();
    if (wave_on) {
#pragma unroll
        for (int a2 = 0; a2 < RW; ++a2)
#pragma unroll
            for (int b2 = 0; b2 < TR; ++b2)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = 16 * (RW * wv + a2) + lg + 4 * r, j = 16 * b2 + li;
                    sm[i * kDriftLd + j] = j < D ? acc[a2][b2][r] - ((dir == 2 && i == j && i < D) ? 1.0 : 0.0) : 0.0;
                }
    }
    __syncthreads();
    // G = A^T A, then the squarings; every product is followed by its Frobenius norm
    double log_lmax = 0.0, w = 1.0;
    for (int it = 0; it <= sq; ++it) {
        zero_acc();
        if (wave_on) {
            for (int p = 0; p < PD; p += 4) {
                double av[RW], bv[TR];
                // it == 0: X = A^T, i.e. X[row][p] = A[p][row]
#pragma unroll
                for (int a2 = 0; a2 < RW; ++a2) {
                    const int row = 16 * (RW * wv + a2) + li;
                    av[a2] = it == 0 ? sm[(p + lg) * kDriftLd + row] : sm[row * kDriftLd + p + lg];
                }
#pragma unroll
                for (int b2 = 0; b2 < TR; ++b2) bv[b2] = sm[(p + lg) * kDriftLd + 16 * b2 + li];
#pragma unroll
                for (int a2 = 0; a2 < RW; ++a2)
#pragma unroll
                    for (int b2 = 0; b2 < TR; ++b2) acc[a2][b2] = mfma_f64(av[a2], bv[b2], acc[a2][b2]);
            }
        }
        double ss = 0.0;
#pragma unroll
        for (int a2 = 0; a2 < RW; ++a2)
#pragma unroll
            for (int b2 = 0; b2 < TR; ++b2)
#pragma unroll
                for (int r = 0; r < 4; ++r) ss = fma(acc[a2][b2][r], acc[a2][b2][r], ss);
        double f = sqrt(block_sum_waves<NW>(ss, red));          // (the barrier inside also ends every read of the old matrix)
        f = f > tiny ? f : tiny;                         // also NaN -> tiny: the NaNs then show up in log_lmax below
        log_lmax += w * log(f);
        w *= 0.5;
        const double inv = 1.0 / f;
        if (wave_on) {
#pragma unroll
            for (int a2 = 0; a2 < RW; ++a2)
#pragma unroll
                for (int b2 = 0; b2 < TR; ++b2)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        sm[(16 * (RW * wv + a2) + lg + 4 * r) * kDriftLd + 16 * b2 + li] = acc[a2][b2][r] * inv;
        }
        __syncthreads();
    }
    if (tid == 0) {
        if (dir == 0) {
            double g = exp(-0.5 * log_lmax) * (1.0 - 1e-9);
            gamma[k] = (g == g && g < __builtin_huge_val()) ? g : 0.0;          // non-finite: no information
        } else {
            double G = exp(0.5 * log_lmax) * (1.0 + 1e-9);
            (dir == 1 ? big : enorm)[k] = (G == G) ? G : __builtin_huge_val();
        }
    }
}

// ---- the whole K-side of one VB iteration in one launch per component ---------------------------------------------------
// What the reference does between two data passes (bayesml/gaussianmixture/_gaussianmixture.py):
//   _calc_n_x_bar_s's finishing (:729-732)  x_bar = p + a/ns,  S = B/ns - (a/ns)(a/ns)^T   (ns > 0; else 0 / the previous S)
//   _calc_vl (:671-723)                      the lower bound's K-sized terms under the CURRENT posterior q
//   _update_q_mu_lambda, _update_q_pi (:741-770)   q' from the prior and (ns, x_bar, S)
//   _calc_q_*_features (:738-756)            E[ln pi], E[ln det Lambda], ln B(W, nu) of q', and what the E-step kernel
//                                            consumes: u' = sqrt(nu') G^-1 (W'^-1 = G G^T), c'
// One workgroup per component: W'^-1 is assembled straight into LDS, factorised and inverted there (as in
// chol_inv_kernel), W' = G^-T G^-1 formed from it; the traces / quadratic forms of the lower bound are one pass over
// the D x D entries with a fixed-order block reduction.  kside_finish_kernel adds the per-component partial terms in
// component order (deterministic) and forms the eight lower-bound terms, their sum, and the drift summary.
struct PriorView {
    const double *alpha, *m, *kappa, *nu, *w_inv, *ln_b_w_nu;
    double ln_c_alpha;
};
struct PostView {
    double *alpha, *m, *kappa, *nu, *w_inv, *w, *u, *u_inv, *e_ln_pi, *e_ln_lambda_det, *ln_b_w_nu, *c;
};
constexpr int kPartials = 12;      // per component: p_x, p_z, (a0-1)E[ln pi], p_mu_lambda, h, lgamma(a), (a-1)psi(a), a, q_mu_lambda

// (lgamma out of line: inlined into kside_step_kernel<1024> - 128 registers per lane at sixteen waves - the library routine's
// temporaries pushed 67 registers of the factorisation's loops into scratch)
__device__ __attribute__((noinline)) double lgamma_call(double x) { return lgamma(x); }

// psi(x), x > 0: recurrence up to x >= 10, then the asymptotic series to x^-14 (remainder < 2e-16 there)
__device__ inline double digamma_pos(double x) {
    double r = 0.0;
    while (x < 10.0) {
        r -= 1.0 / x;
        x += 1.0;
    }
    const double i2 = 1.0 / (x * x);
    const double ser = i2 * (1.0 / 12.0 - i2 * (1.0 / 120.0 - i2 * (1.0 / 252.0 - i2 * (1.0 / 240.0 - i2 * (1.0 / 132.0 -
                       i2 * (691.0 / 32760.0 - i2 * (1.0 / 12.0)))))));
    return r + log(x) - 0.5 / x - ser;
}

// deterministic block sum of NV values per thread (NT threads): wave shuffles, then the wave results in a fixed tree
template <int NV, int NT>
__device__ __forceinline__ void block_sum_n(double (&v)[NV], double* red /*[NT / 64 * NV]*/) {
#pragma unroll
    for (int q = 0; q < NV; ++q)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v[q] += __shfl_xor(v[q], o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int q = 0; q < NV; ++q) red[(threadIdx.x >> 6) * NV + q] = v[q];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NV; ++q) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < NT / 64; w += 4)
            t += (red[w * NV + q] + red[(w + 1) * NV + q]) + (red[(w + 2) * NV + q] + red[(w + 3) * NV + q]);
        v[q] = t;
    }
}

// NT threads per component: 1024 for D > 32 (sixteen waves hide the LDS latency of the factorisation's short dependent
// steps: 0.61 -> see DESIGN.md 4d), 256 below.
template <int NT>
__global__ __launch_bounds__(NT) void kside_step_kernel(int K, int D, PriorView pr, PostView q, PostView qn,
                                                         const double* __restrict__ stats, const double* __restrict__ pivot,
                                                         double* __restrict__ s_prev, double* __restrict__ ns_out,
                                                         double* __restrict__ x_bar_out, double* __restrict__ s_out,
                                                         double* __restrict__ partials /*[K][kPartials]*/) {
    extern __shared__ double sm[];           // mat [D][D + 1] | abar, xbar, dev0, dq, dm [D] each | red [32]
    const int ld = D + 1;
    double* mat = sm;
    double* abar = sm + (size_t)D * ld;
    double* xbar = abar + D;
    double* dev0 = xbar + D;
    double* dq = dev0 + D;
    double* dm = dq + D;
    double* red = dm + D;          // [NT / 64 * 4]
    const int tid = threadIdx.x, k = blockIdx.x;
    const int64_t vb = (int64_t)k * D, mb = (int64_t)k * D * D;
    const double LN_2PI = 1.8378770664093454835606594728112, LN_2 = 0.69314718055994530941723212145818,
                 LN_PI = 1.1447298858494001741434273513531;
    const double ns = stats[k], h = stats[K + k];
    const double* a = stats + 2 * (int64_t)K + vb;
    const double* B = stats + 2 * (int64_t)K + (int64_t)K * D + mb;
    const bool pos = ns > 0.0;
    const double safe = pos ? ns : 1.0;
    const double kap0 = pr.kappa[k], nu0 = pr.nu[k], al0 = pr.alpha[k];
    const double kapq = q.kappa[k], nuq = q.nu[k], alq = q.alpha[k], elpq = q.e_ln_pi[k], eldq = q.e_ln_lambda_det[k];
    for (int i = tid; i < D; i += NT) {
        const double ab = a[i] / safe;
        const double xb = pos ? pivot[i] + ab : 0.0;
        abar[i] = ab;
        xbar[i] = xb;
        dev0[i] = xb - pr.m[vb + i];
        dq[i] = xb - q.m[vb + i];
        dm[i] = q.m[vb + i] - pr.m[vb + i];
        x_bar_out[vb + i] = xb;
    }
    __syncthreads();
    const double kapn = kap0 + ns, nun = nu0 + ns, aln = al0 + ns;
    const double coef = kap0 * ns / kapn;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};          // tr(S nu W), dq' nu W dq, dm' nu W dm, tr(W0^-1 nu W)
    for (int e = tid; e < D * D; e += NT) {
        const int i = e / D, j = e - i * D;
        const double sij = pos ? B[e] / safe - abar[i] * abar[j] : s_prev[mb + e];
        s_out[mb + e] = sij;
        s_prev[mb + e] = sij;
        const double ew = nuq * q.w[mb + e];
        const double w0 = pr.w_inv[mb + e];
        acc[0] = fma(sij, ew, acc[0]);
        acc[1] = fma(dq[i] * dq[j], ew, acc[1]);
        acc[2] = fma(dm[i] * dm[j], ew, acc[2]);
        acc[3] = fma(w0, ew, acc[3]);
        const double wn = w0 + ns * sij + coef * (dev0[i] * dev0[j]);
        qn.w_inv[mb + e] = wn;
        mat[i * ld + j] = j <= i ? wn : 0.0;
    }
    block_sum_n<4, NT>(acc, red);
    if (tid == 0) {
        double* pt = partials + (int64_t)k * kPartials;
        pt[0] = 0.5 * ns * (eldq - D / kapq - acc[0] - acc[1] - D * LN_2PI);                                   // p_x
        pt[1] = ns * elpq;                                                                                     // p_z
        pt[2] = (al0 - 1.0) * elpq;                                                                            // -> p_pi
        pt[3] = 0.5 * (D * (log(kap0) - LN_2PI - kap0 / kapq) - kap0 * acc[2] + 2.0 * pr.ln_b_w_nu[k] +
                       (nu0 - D) * eldq - acc[3]);                                                             // p_mu_lambda
        pt[4] = h;                                                                                             // -> q_z
        pt[5] = lgamma_call(alq);
        pt[6] = (alq - 1.0) * digamma_pos(alq);
        pt[7] = alq;
        pt[8] = 0.5 * (D * (1.0 + LN_2PI - log(kapq)) - 2.0 * q.ln_b_w_nu[k] - (nuq - D) * eldq + nuq * D);    // q_mu_lambda
        ns_out[k] = ns;
        qn.alpha[k] = aln;
        qn.kappa[k] = kapn;
        qn.nu[k] = nun;
    }
    for (int i = tid; i < D; i += NT) qn.m[vb + i] = (kap0 * pr.m[vb + i] + ns * xbar[i]) / kapn;
    __syncthreads();
    // ---- Cholesky of W'^-1 in LDS (as in chol_inv_kernel).  (Round 4 measured a version blocked by 16 columns - diagonal
    // block in one wave's registers, panel solve with a thread per row, trailing update from 16-term dot products, and the
    // inverse block column by block column: 24 + 32 barriers instead of 384 + 256 - at 0.312 ms per launch against 0.298:
    // the 128 barrier-separated columns are not what the kernel's time is made of.  Not kept.)
    constexpr int TG = NT == 1024 ? 32 : 16;                  // the trailing update runs on a TG x TG thread grid
    const int ti = tid / TG, tj = tid % TG;
    for (int j = 0; j < D; ++j) {
        if (tid == 0) mat[j * ld + j] = sqrt(mat[j * ld + j]);
        __syncthreads();
        const double inv = 1.0 / mat[j * ld + j];
        for (int i = j + 1 + tid; i < D; i += NT) mat[i * ld + j] *= inv;
        __syncthreads();
        for (int i = j + 1 + ti; i < D; i += TG) {
            const double lij = mat[i * ld + j];
            for (int c = j + 1 + tj; c <= i; c += TG) mat[i * ld + c] = fma(-lij, mat[c * ld + j], mat[i * ld + c]);
        }
        __syncthreads();
    }
    const double sq = sqrt(nun), isq = 1.0 / sq;
    for (int e = tid; e < D * D; e += NT) {
        const int i = e / D, j = e - i * D;
        qn.u_inv[mb + e] = mat[i * ld + j] * isq;             // u'^-1 = G / sqrt(nu')
    }
    // log det W'^-1 and the digamma / lgamma sums over d < D
    double f[3] = {0.0, 0.0, 0.0};
    for (int d = tid; d < D; d += NT) {
        f[0] += log(mat[d * ld + d]);
        f[1] += digamma_pos(0.5 * (nun - d));
        f[2] += lgamma_call(0.5 * (nun - d));
    }
    // (sum of the new alphas over ALL components with the block, K / NT terms per thread: one thread walking K dependent
    // global loads was 40 us of this kernel at K = 64 and most of it at K = 256 - round 6)
    double f4[4] = {f[0], f[1], f[2], 0.0};
    for (int c = tid; c < K; c += NT) f4[3] += pr.alpha[c] + stats[c];
    block_sum_n<4, NT>(f4, red);
    f[0] = f4[0];
    f[1] = f4[1];
    f[2] = f4[2];
    const double logdet = 2.0 * f[0];
    const double eld = f[1] + D * LN_2 - logdet;
    if (tid == 0) {
        const double asum = f4[3];
        const double elp = digamma_pos(aln) - digamma_pos(asum);
        qn.e_ln_pi[k] = elp;
        qn.e_ln_lambda_det[k] = eld;
        qn.ln_b_w_nu[k] = 0.5 * (nun * logdet - nun * D * LN_2 - 0.5 * D * (D - 1) * LN_PI - 2.0 * f[2]);
        qn.c[k] = elp + 0.5 * (eld - D * LN_2PI - D / kapn);
    }
    __syncthreads();
    // ---- in-place inverse of the factor: column j from the columns right of it; SEG threads share a row's dot product
    constexpr int SEG = NT / 128;                             // D <= 128 rows
    const int rl = tid / SEG, seg = tid % SEG;
    for (int j = D - 1; j >= 0; --j) {
        const double xjj = 1.0 / mat[j * ld + j];
        const int i = j + 1 + rl;
        double v = 0.0;
        if (i < D)
            for (int p = j + 1 + seg; p <= i; p += SEG) v = fma(mat[i * ld + p], mat[p * ld + j], v);
#pragma unroll
        for (int o = 1; o < SEG; o <<= 1) v += __shfl_xor(v, o);
        v = -v * xjj;
        __syncthreads();
        if (i < D && seg == 0) mat[i * ld + j] = v;
        if (tid == 0) mat[j * ld + j] = xjj;
        __syncthreads();
    }
    // u' = sqrt(nu') G^-1 and W' = G^-T G^-1 (upper triangle computed, mirrored: exactly symmetric)
    for (int e = tid; e < D * D; e += NT) {
        const int i = e / D, j = e - i * D;
        qn.u[mb + e] = mat[i * ld + j] * sq;
        if (i <= j) {
            double w = 0.0;
            for (int p = j; p < D; ++p) w = fma(mat[p * ld + i], mat[p * ld + j], w);
            qn.w[mb + e] = w;
            qn.w[mb + (int64_t)j * D + i] = w;
        }
    }
}

// scal = [p_x, p_z, p_pi, p_mu_lambda, q_z, q_pi, q_mu_lambda, vl, drift summary min_k (gamma_k - delta_k / 30)]
__global__ void kside_finish_kernel(int K, const double* __restrict__ partials, double ln_c_alpha,
                                    const double* __restrict__ gamma, const double* __restrict__ delta,
                                    double* __restrict__ scal) {
    // one workgroup of 64 (one thread's 9 K dependent loads were 0.1 ms at K = 256); thread 0 forms the lower bound from the
    // nine sums
    __shared__ double ts[kPartials];
    if (blockIdx.x != 0) return;
    {
        // nine sums over the components in a FIXED order: lane l takes k = l, l + 64, ... in turn, then the lanes' sums meet
        // in a fixed butterfly (round 5; before, one thread added all K in component order - the result differs in the last
        // bits from that, not from run to run)
        double acc[9];
#pragma unroll
        for (int q = 0; q < 9; ++q) acc[q] = 0.0;
        for (int k = threadIdx.x; k < K; k += 64) {
#pragma unroll
            for (int q = 0; q < 9; ++q) acc[q] += partials[(int64_t)k * kPartials + q];
        }
#pragma unroll
        for (int q = 0; q < 9; ++q) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) acc[q] += __shfl_xor(acc[q], o);
            if (threadIdx.x == 0) ts[q] = acc[q];
        }
    }
    // drift summary min_k (gamma_k - delta_k / 30), a NaN sticks: a minimum, whatever order it is taken in
    double g = 0.0;
    if (gamma) {
        g = __builtin_huge_val();
        bool nan = false;
        for (int k = threadIdx.x; k < K; k += 64) {
            const double v = gamma[k] - delta[k] / 30.0;
            nan = nan || v != v;
            g = v < g ? v : g;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double w = __shfl_xor(g, o);
            g = w < g ? w : g;
            nan = nan || (bool)__shfl_xor((int)nan, o);
        }
        if (nan) g = __builtin_nan("");
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    double t[kPartials];
    for (int q = 0; q < kPartials; ++q) t[q] = q < 9 ? ts[q] : 0.0;
    const double a0 = t[7];
    const double p_x = t[0], p_z = t[1], p_pi = ln_c_alpha + t[2], p_ml = t[3], q_z = -t[4];
    const double q_pi = (t[5] - lgamma(a0)) + (a0 - K) * digamma_pos(a0) - t[6];      // entropy of Dirichlet(alpha)
    const double q_ml = t[8];
    scal[0] = p_x;
    scal[1] = p_z;
    scal[2] = p_pi;
    scal[3] = p_ml;
    scal[4] = q_z;
    scal[5] = q_pi;
    scal[6] = q_ml;
    scal[7] = p_x + p_z + p_pi + p_ml + q_z + q_pi + q_ml;
    scal[8] = g;
}

// gamma = max(gamma, 1 - e), big = min(big, 1 + e) (drift_kernel's third direction)
__global__ void drift_combine_kernel(int K, double* __restrict__ gamma, double* __restrict__ big, const double* __restrict__ enorm) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    const double e = enorm[k] * (1.0 + 1e-9);
    const double g2 = (1.0 - e) * (1.0 - 1e-9), b2 = (1.0 + e) * (1.0 + 1e-9);
    gamma[k] = g2 > gamma[k] ? g2 : gamma[k];
    big[k] = b2 < big[k] ? b2 : big[k];
}

}  // namespace gmmvb

using namespace gmmvb;

// hipFuncAttributeMaxDynamicSharedMemorySize is a property of (kernel, device): remembered per device ordinal, under a
// lock (a process may drive several GPUs, or create models from several threads)
namespace {
struct LdsAttr {
    const void* fn;
    int dev;
    size_t bytes;
};
std::mutex g_lds_mu;
std::vector<LdsAttr> g_lds_set;

hipError_t ensure_dynamic_lds(const void* fn, size_t bytes) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(g_lds_mu);
    for (LdsAttr& a : g_lds_set) {
        if (a.fn != fn || a.dev != dev) continue;
        if (a.bytes >= bytes) return hipSuccess;
        e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e == hipSuccess) a.bytes = bytes;
        return e;
    }
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) g_lds_set.push_back({fn, dev, bytes});
    return e;
}
}  // namespace

extern "C" int gmmvb_kside_factor(int K, int D, const double* w_inv_dev, double* g_dev, double* g_inv_dev,
                                  double* logdet_dev, void* stream) {
    if (K < 1 || D < 1) return fail(GMMVB_EINVAL, "K and D must be positive");
    if (D > 128) return fail(GMMVB_EUNSUPPORTED, "gmmvb_kside_factor: D > 128 (the factor is kept in LDS)");
    if (!w_inv_dev || !g_dev || !g_inv_dev || !logdet_dev) return fail(GMMVB_EINVAL, "null argument");
    const size_t lds = (size_t)D * (D + 1) * sizeof(double);
    {
        hipError_t e = ensure_dynamic_lds((const void*)chol_inv_kernel, (size_t)128 * 129 * sizeof(double));
        if (e != hipSuccess) return fail(GMMVB_EHIP, "hipFuncSetAttribute(chol_inv_kernel)", e);
    }
    hipLaunchKernelGGL(chol_inv_kernel, dim3(K), dim3(256), lds, (hipStream_t)stream, w_inv_dev, D, g_dev, g_inv_dev,
                       logdet_dev);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(GMMVB_EHIP, "chol_inv_kernel launch", e);
    return GMMVB_OK;
}

extern "C" int gmmvb_kside_drift(int K, int D, const double* u_old_dev, const double* uinv_old_dev, const double* m_old_dev,
                                 const double* u_new_dev, const double* uinv_new_dev, const double* m_new_dev,
                                 int squarings, int squarings_big, double* gamma_dev, double* delta_dev,
                                 double* big_gamma_dev, double* enorm_dev, void* stream) {
    if (K < 1 || D < 1 || squarings < 0 || squarings_big < 0) return fail(GMMVB_EINVAL, "bad argument");
    if (D > 128) return fail(GMMVB_EUNSUPPORTED, "gmmvb_kside_drift: D > 128 (the matrix is kept in LDS)");
    if (!u_old_dev || !uinv_old_dev || !m_old_dev || !u_new_dev || !uinv_new_dev || !m_new_dev || !gamma_dev || !delta_dev ||
        !big_gamma_dev || !enorm_dev)
        return fail(GMMVB_EINVAL, "null argument");
#define GMMVB_DRIFT(PD)                                                                                                       \
    {                                                                                                                          \
        const size_t lds = (size_t)PD * kDriftLd * sizeof(double);                                                             \
        constexpr int NWV = PD == 128 ? 8 : 4;                                                                                 \
        hipError_t e0 = ensure_dynamic_lds((const void*)drift_kernel<PD, NWV>, lds);                                           \
        if (e0 != hipSuccess) return fail(GMMVB_EHIP, "hipFuncSetAttribute(drift_kernel)", e0);                               \
        hipLaunchKernelGGL((drift_kernel<PD, NWV>), dim3(K, 3), dim3(64 * NWV), lds, (hipStream_t)stream, u_old_dev, uinv_old_dev, \
                           m_old_dev, u_new_dev, uinv_new_dev, m_new_dev, D, squarings, squarings_big, gamma_dev, delta_dev,  \
                           big_gamma_dev, enorm_dev);                                                                          \
    }
    if (D <= 32) GMMVB_DRIFT(32)
    else if (D <= 64) GMMVB_DRIFT(64)
    else GMMVB_DRIFT(128)
#undef GMMVB_DRIFT
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(GMMVB_EHIP, "drift_kernel launch", e);
    return GMMVB_OK;
}

extern "C" int gmmvb_kside_step(int K, int D, const gmmvb_prior_view* prior, const gmmvb_post_view* q, const gmmvb_post_view* q_next,
                                const double* stats_dev, const double* pivot_dev, double* s_prev_dev, double* ns_dev,
                                double* x_bar_dev, double* s_dev, int want_drift, double* gamma_dev, double* delta_dev,
                                double* big_gamma_dev, double* scal_dev, double* scratch_dev, void* stream) {
    if (K < 1 || D < 1) return fail(GMMVB_EINVAL, "K and D must be positive");
    if (D > 128) return fail(GMMVB_EUNSUPPORTED, "gmmvb_kside_step: D > 128 (the matrices are kept in LDS)");
    if (!prior || !q || !q_next || !stats_dev || !pivot_dev || !s_prev_dev || !ns_dev || !x_bar_dev || !s_dev || !scal_dev ||
        !scratch_dev || (want_drift && (!gamma_dev || !delta_dev || !big_gamma_dev)))
        return fail(GMMVB_EINVAL, "null argument");
    static_assert(sizeof(gmmvb_prior_view) == sizeof(PriorView) && sizeof(gmmvb_post_view) == sizeof(PostView), "view layouts");
    PriorView pr;
    PostView qa, qb;
    std::memcpy(&pr, prior, sizeof(pr));
    std::memcpy(&qa, q, sizeof(qa));
    std::memcpy(&qb, q_next, sizeof(qb));
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = ((size_t)D * (D + 1) + 5 * (size_t)D + 64) * sizeof(double);
    const bool wide = D > 32;
    {
        hipError_t e = ensure_dynamic_lds(wide ? (const void*)kside_step_kernel<1024> : (const void*)kside_step_kernel<256>, lds);
        if (e != hipSuccess) return fail(GMMVB_EHIP, "hipFuncSetAttribute(kside_step_kernel)", e);
    }
    double* partials = scratch_dev;                       // [K][kPartials]
    double* enorm = scratch_dev + (size_t)K * kPartials;  // [K]
    if (wide)
        hipLaunchKernelGGL(kside_step_kernel<1024>, dim3(K), dim3(1024), lds, st, K, D, pr, qa, qb, stats_dev, pivot_dev, s_prev_dev,
                           ns_dev, x_bar_dev, s_dev, partials);
    else
        hipLaunchKernelGGL(kside_step_kernel<256>, dim3(K), dim3(256), lds, st, K, D, pr, qa, qb, stats_dev, pivot_dev, s_prev_dev,
                           ns_dev, x_bar_dev, s_dev, partials);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(GMMVB_EHIP, "kside_step_kernel launch", e);
    if (want_drift) {
        int rc = gmmvb_kside_drift(K, D, qa.u, qa.u_inv, qa.m, qb.u, qb.u_inv, qb.m, 8, 6, gamma_dev, delta_dev, big_gamma_dev, enorm,
                                   stream);
        if (rc) return rc;
        hipLaunchKernelGGL(drift_combine_kernel, dim3((K + 255) / 256), dim3(256), 0, st, K, gamma_dev, big_gamma_dev, enorm);
    }
    hipLaunchKernelGGL(kside_finish_kernel, dim3(1), dim3(64), 0, st, K, partials, pr.ln_c_alpha, want_drift ? gamma_dev : nullptr,
                       want_drift ? delta_dev : nullptr, scal_dev);
    e = hipGetLastError();
    if (e != hipSuccess) return fail(GMMVB_EHIP, "kside_finish_kernel launch", e);
    return GMMVB_OK;
}

// ---- the Dirichlet half of the HMM's K-side in one launch (round 5) ------------------------------------------------------
// What hiddenmarkovnormal.LearnModel does with eta (initial-state) and zeta (transition rows) between two data passes
// (bayesml/hiddenmarkovnormal/_hiddenmarkovnormal.py):
//   _calc_vl (:883-943)        the lower bound's terms that involve them, under the CURRENT posterior q: E[ln p(z)], E[ln p(pi)],
//                              E[ln p(A)], -E[ln q(z)], -E[ln q(pi)], -E[ln q(A)] - and the total with the three Normal-Wishart
//                              terms gmmvb_kside_step has left in scal_nw ([0] p_x, [3] p_mu_lambda, [6] q_mu_lambda)
//   _update_q_pi, _update_q_a (:980-986)      eta' = eta_0 + ns, zeta' = zeta_0 + ms
//   _calc_q_pi_features, _calc_q_a_features (:861-868)   ln pi~', pi~' = exp(ln pi~' - max), ln a~', a~' = exp(ln a~' - ONE global
//                              max), ln C(zeta') summed over the rows
// One workgroup (K^2 <= 16384 entries), every sum a fixed-order block reduction.  The torch functions of _kside.py
// (hmm_lower_bound, hmm_update_q, hmm_features) remain the specification; tests/test_gpu_hmm.py holds this kernel to them.
namespace {
constexpr int kDirThreads = 1024;
__device__ __forceinline__ double block_sum_1(double v, double* red) {
    double a[1] = {v};
    block_sum_n<1, kDirThreads>(a, red);
    __syncthreads();
    return a[0];
}
__device__ __forceinline__ double block_max_1(double v, double* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = red[0];
    for (int w = 1; w < kDirThreads / 64; ++w) t = fmax(t, red[w]);
    __syncthreads();
    return t;
}
struct HmmDirArgs {
    const double *eta0, *zeta0;                  // prior [K], [K][K]
    double ln_c_eta0, ln_c_zeta0;                // ln C(eta_0), sum_i ln C(zeta_0[i])
    const double *eta, *zeta, *ln_pi, *ln_a, *ln_c_zeta;      // current q: [K], [K][K], [K], [K][K], [1]
    const double *ms, *g0, *sum_ln_c;            // forward-backward summary: [K][K], [K], [1]
    const double *ns;                            // [K]
    const double *scal_nw;                       // gmmvb_kside_step's nine doubles
    const double *h_scale;                       // [1]: 0 when the pass had no emission (random-responsibility start)
    double *eta_n, *zeta_n, *ln_pi_n, *pi_n, *ln_a_n, *a_n, *ln_c_zeta_n;      // q'
    double *scal;                                // [10] p_x p_z p_pi p_a p_mu_lambda q_z q_pi q_a q_mu_lambda vl
};
__global__ __launch_bounds__(kDirThreads) void hmm_kside_dirichlet_kernel(int K, HmmDirArgs a) {
    __shared__ double red[kDirThreads / 64];
    __shared__ double rowsum[256];               // K <= 256 here (the launcher checks)
    const int tid = threadIdx.x;
    const int KK = K * K;
    // ---- the lower bound's Dirichlet terms under q
    double amax = -__builtin_huge_val(), pmax = -__builtin_huge_val();
    for (int e = tid; e < KK; e += kDirThreads) amax = fmax(amax, a.ln_a[e]);
    for (int k = tid; k < K; k += kDirThreads) pmax = fmax(pmax, a.ln_pi[k]);
    amax = block_max_1(amax, red);
    pmax = block_max_1(pmax, red);
    double s_ms_lna = 0.0, s_z0_lna = 0.0, s_ms_sh = 0.0, s_z_lna = 0.0;
    for (int e = tid; e < KK; e += kDirThreads) {
        const double la = a.ln_a[e], m = a.ms[e];
        s_ms_lna = fma(m, la, s_ms_lna);
        s_z0_lna = fma(a.zeta0[e] - 1.0, la, s_z0_lna);
        s_ms_sh = fma(m, la - amax, s_ms_sh);
        s_z_lna = fma(a.zeta[e] - 1.0, la, s_z_lna);
    }
    double s_g_lnp = 0.0, s_e0_lnp = 0.0, s_g_sh = 0.0, s_lg = 0.0, s_e_psi = 0.0, s_eta = 0.0;
    for (int k = tid; k < K; k += kDirThreads) {
        const double lp = a.ln_pi[k], g = a.g0[k], et = a.eta[k];
        s_g_lnp = fma(g, lp, s_g_lnp);
        s_e0_lnp = fma(a.eta0[k] - 1.0, lp, s_e0_lnp);
        s_g_sh = fma(g, lp - pmax, s_g_sh);
        s_lg += lgamma(et);
        s_e_psi = fma(et - 1.0, digamma_pos(et), s_e_psi);
        s_eta += et;
    }
    s_ms_lna = block_sum_1(s_ms_lna, red);
    s_z0_lna = block_sum_1(s_z0_lna, red);
    s_ms_sh = block_sum_1(s_ms_sh, red);
    s_z_lna = block_sum_1(s_z_lna, red);
    s_g_lnp = block_sum_1(s_g_lnp, red);
    s_e0_lnp = block_sum_1(s_e0_lnp, red);
    s_g_sh = block_sum_1(s_g_sh, red);
    s_lg = block_sum_1(s_lg, red);
    s_e_psi = block_sum_1(s_e_psi, red);
    s_eta = block_sum_1(s_eta, red);
    if (tid == 0) {
        const double p_x = a.scal_nw[0], p_ml = a.scal_nw[3], q_ml = a.scal_nw[6];
        const double sg = p_x * a.h_scale[0];        // sum gamma ln rho = E[ln p(x|z)] in closed form (ref:905 via :871-877)
        const double p_z = s_g_lnp + s_ms_lna;
        const double p_pi = a.ln_c_eta0 + s_e0_lnp;
        const double p_a = a.ln_c_zeta0 + s_z0_lna;
        const double q_z = -sg - s_ms_sh - s_g_sh + a.sum_ln_c[0];
        const double q_pi = (s_lg - lgamma(s_eta)) + (s_eta - K) * digamma_pos(s_eta) - s_e_psi;      // entropy of Dirichlet(eta)
        const double q_a = -a.ln_c_zeta[0] - s_z_lna;
        a.scal[0] = p_x;
        a.scal[1] = p_z;
        a.scal[2] = p_pi;
        a.scal[3] = p_a;
        a.scal[4] = p_ml;
        a.scal[5] = q_z;
        a.scal[6] = q_pi;
        a.scal[7] = q_a;
        a.scal[8] = q_ml;
        a.scal[9] = p_x + p_z + p_pi + p_a + p_ml + q_z + q_pi + q_a + q_ml;
    }
    // ---- q': eta', zeta' and their features
    double se = 0.0;
    for (int k = tid; k < K; k += kDirThreads) {
        const double v = a.eta0[k] + a.ns[k];
        a.eta_n[k] = v;
        se += v;
    }
    se = block_sum_1(se, red);
    const double psi_se = digamma_pos(se);
    double pm = -__builtin_huge_val();
    for (int k = tid; k < K; k += kDirThreads) {
        const double v = digamma_pos(a.eta0[k] + a.ns[k]) - psi_se;
        a.ln_pi_n[k] = v;
        pm = fmax(pm, v);
    }
    pm = block_max_1(pm, red);
    for (int k = tid; k < K; k += kDirThreads) a.pi_n[k] = exp(a.ln_pi_n[k] - pm);      // (the thread's own stores)
    // row sums of zeta' in row order (a thread per row: K <= 256 rows of K <= 256 entries)
    for (int i = tid; i < K; i += kDirThreads) {
        double r = 0.0;
        for (int j = 0; j < K; ++j) r += a.zeta0[i * K + j] + a.ms[i * K + j];
        rowsum[i] = r;
    }
    __syncthreads();
    double am = -__builtin_huge_val(), lgz = 0.0;
    for (int e = tid; e < KK; e += kDirThreads) {
        const int i = e / K;
        const double z = a.zeta0[e] + a.ms[e];
        a.zeta_n[e] = z;
        const double v = digamma_pos(z) - digamma_pos(rowsum[i]);
        a.ln_a_n[e] = v;
        am = fmax(am, v);
        lgz += lgamma(z);
    }
    am = block_max_1(am, red);
    for (int e = tid; e < KK; e += kDirThreads) a.a_n[e] = exp(a.ln_a_n[e] - am);
    double lgr = 0.0;
    for (int i = tid; i < K; i += kDirThreads) lgr += lgamma(rowsum[i]);
    lgz = block_sum_1(lgz, red);
    lgr = block_sum_1(lgr, red);
    if (tid == 0) a.ln_c_zeta_n[0] = lgr - lgz;
}
}  // namespace

extern "C" int hmmvb_kside_dirichlet(int K, const double* eta0_dev, const double* zeta0_dev, double ln_c_eta0, double ln_c_zeta0,
                                     const double* eta_dev, const double* zeta_dev, const double* ln_pi_dev, const double* ln_a_dev,
                                     const double* ln_c_zeta_dev, const double* fb_dev, const double* ns_dev,
                                     const double* scal_nw_dev, const double* h_scale_dev, double* eta_next_dev,
                                     double* zeta_next_dev, double* ln_pi_next_dev, double* pi_next_dev, double* ln_a_next_dev,
                                     double* a_next_dev, double* ln_c_zeta_next_dev, double* scal_dev, void* stream) {
    if (K < 1 || K > 256) return fail(GMMVB_EUNSUPPORTED, "hmmvb_kside_dirichlet: 1 <= K <= 256");
    if (!eta0_dev || !zeta0_dev || !eta_dev || !zeta_dev || !ln_pi_dev || !ln_a_dev || !ln_c_zeta_dev || !fb_dev || !ns_dev ||
        !scal_nw_dev || !h_scale_dev || !eta_next_dev || !zeta_next_dev || !ln_pi_next_dev || !pi_next_dev || !ln_a_next_dev ||
        !a_next_dev || !ln_c_zeta_next_dev || !scal_dev)
        return fail(GMMVB_EINVAL, "null argument");
    HmmDirArgs a{eta0_dev, zeta0_dev, ln_c_eta0, ln_c_zeta0, eta_dev, zeta_dev, ln_pi_dev, ln_a_dev, ln_c_zeta_dev,
                 fb_dev, fb_dev + (size_t)K * K, fb_dev + (size_t)K * K + 2 * (size_t)K, ns_dev, scal_nw_dev, h_scale_dev,
                 eta_next_dev, zeta_next_dev, ln_pi_next_dev, pi_next_dev, ln_a_next_dev, a_next_dev, ln_c_zeta_next_dev, scal_dev};
    hipLaunchKernelGGL(hmm_kside_dirichlet_kernel, dim3(1), dim3(kDirThreads), 0, (hipStream_t)stream, K, a);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(GMMVB_EHIP, "hmm_kside_dirichlet_kernel launch", e);
    return GMMVB_OK;
}
