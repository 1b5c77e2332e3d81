// Small problems: ALL restarts and ALL VB iterations of gaussianmixture.LearnModel.update_posterior in ONE launch.
//
// Reference: bayesml/gaussianmixture/_gaussianmixture.py:846-872 - `for i in range(num_init)` x `for t in range(max_itr)`
// around _update_q_mu_lambda / _update_q_pi (:741-770), _update_q_z (:772-784), _calc_vl (:671-723) and the convergence test
// (:869).  At the sizes BayesML's own tutorials use (K = 3, D = 2, N = 1000) one VB iteration is a few thousand flops: the
// general engine (workspace, ~10 launches and one host synchronisation per iteration) spends its time on launches.  Here
// workgroup r runs restart r from its initial posterior to convergence: E-step with a thread per row, statistics through
// LDS with a thread per (statistic, row split), the K-sized closed forms with a thread per component in registers
// (c_degree is a template parameter), the lower bound and the convergence test on the device.  The restarts are
// independent given their initial states (the host draws them in the reference's order), so they run side by side; the
// host replays the reference's winner rule (:873) and progress lines from the traces.  One launch, one device-to-host copy.
//
// Same formulation as the large kernels: ln rho = c_k - |U_k (x - m_k)|^2 / 2 with W^-1 = G G^T,
// U = sqrt(nu) G^-1; moments about the pivot; every reduction in a fixed order (run-to-run identical).
#include "workspace.h"
#include "common.h"

#include <algorithm>

namespace gmmvb {

constexpr int kSmallThreads = 512;
constexpr int kSmallMaxD = 8;
constexpr int kSmallMaxK = 32;
constexpr int kSmallMaxElems = 256;        // K (1 + D + D (D + 1) / 2): at least two row splits per statistic
constexpr int64_t kSmallMaxRows = 16384;
constexpr int kSmallTerms = 8;             // p_x, p_z, p_pi, p_mu_lambda, q_z, q_pi, q_mu_lambda, vl

struct SmallShape {
    int EK, E, S, C;          // statistics per component / in all, row splits, rows per chunk
    size_t lds_doubles;
};

__host__ __device__ inline int small_pow2_floor(int v) {
    int p = 1;
    while (2 * p <= v) p *= 2;
    return p;
}

inline SmallShape small_shape(int K, int D) {
    SmallShape s;
    s.EK = 1 + D + D * (D + 1) / 2;
    s.E = K * s.EK;
    s.S = kSmallThreads / s.E;
    s.C = std::min(kSmallThreads, small_pow2_floor(8192 / (K + D)));
    s.lds_doubles = (size_t)8 * K + 3 * (size_t)K * D + 5 * (size_t)K * D * D + 12 * (size_t)K + (size_t)s.S * s.E +
                    (size_t)s.C * (K + D) + 32;
    return s;
}

inline int64_t small_post_len(int K, int D) { return 6 * (int64_t)K + 2 * (int64_t)K * D + 3 * (int64_t)K * D * D + K; }

struct SmallArgs {
    const void* x;
    int64_t ldx, n_rows;
    const double* pivot;       // [D]
    const double* prior;       // alpha K | m KD | kappa K | nu K | w_inv KDD | ln_b_w_nu K | ln_c_alpha 1
    const double* init;        // init_type 0: per restart [m KD | w_inv KDD];  1: per restart r [N][K]
    int K, init_type, max_itr, EK, E, S, C;
    double tol;
    double* out;               // [R][out_len]
    int64_t out_len;
    double* r_out;             // [R][N][K] or null
};

__device__ inline double small_digamma(double x) {          // psi(x), x > 0 (as kside.hip)
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

template <int D, typename XT>
__global__ __launch_bounds__(kSmallThreads) void small_fit_kernel(SmallArgs a) {
    extern __shared__ double sm[];
    const int K = a.K, tid = threadIdx.x, rs = blockIdx.x;
    constexpr int DD = D * D;
    const double LN_2PI = 1.8378770664093454835606594728112, LN_2 = 0.69314718055994530941723212145818,
                 LN_PI = 1.1447298858494001741434273513531;
    // ---- LDS carve-up
    double* alpha = sm;
    double* kappa = alpha + K;
    double* nu = kappa + K;
    double* elp = nu + K;
    double* eld = elp + K;
    double* lnb = eld + K;
    double* cc = lnb + K;
    double* ns = cc + K;
    double* mm = ns + K;               // [K][D] posterior means
    double* xbar = mm + K * D;
    double* av = xbar + K * D;         // [K][D] sum r (x - pivot)
    double* winv = av + K * D;         // [K][D][D]
    double* wm = winv + K * DD;        // W
    double* um = wm + K * DD;          // U = sqrt(nu) G^-1, lower triangular
    double* Bm = um + K * DD;          // sum r (x - pivot)(x - pivot)^T
    double* sv = Bm + K * DD;          // S of the last data pass (kept for components with ns == 0: ref :729)
    double* pt = sv + K * DD;          // [K][12] lower-bound partials
    double* part = pt + 12 * K;        // [S][E]
    double* rbuf = part + a.S * a.E;   // [C][K]
    double* xbuf = rbuf + a.C * K;     // [C][D]
    double* red = xbuf + a.C * D;      // [16] wave partials | [16..] scalars: vl, stop flag, h
    // ---- prior (global, read-only)
    const double* p_alpha = a.prior;
    const double* p_m = p_alpha + K;
    const double* p_kappa = p_m + K * D;
    const double* p_nu = p_kappa + K;
    const double* p_winv = p_nu + K;
    const double* p_lnb = p_winv + K * DD;
    const double ln_c_alpha = p_lnb[K];
    double* out = a.out + (int64_t)rs * a.out_len;
    double* trace = out + 2 + kSmallTerms;
    const XT* x = (const XT*)a.x;
    const int64_t N = a.n_rows;

    // ---- initial posterior: prior hyper-parameters, means / precisions of the restart's sub-samples (ref :786-796)
#pragma unroll 1
    for (int e = tid; e < K; e += kSmallThreads) {
        alpha[e] = p_alpha[e];
        kappa[e] = p_kappa[e];
        nu[e] = p_nu[e];
    }
    {
        const double* im = a.init_type == 0 ? a.init + (int64_t)rs * (K * D + K * DD) : p_m;
        const double* iw = a.init_type == 0 ? im + K * D : p_winv;
#pragma unroll 1
        for (int e = tid; e < K * D; e += kSmallThreads) mm[e] = im[e];
#pragma unroll 1
        for (int e = tid; e < K * DD; e += kSmallThreads) {
            winv[e] = iw[e];
            sv[e] = 0.0;
        }
    }
    __syncthreads();

    // this thread's statistic (M phase): component, kind and feature indices
    const bool m_thread = tid < a.E * a.S;
    const int my_e = tid % a.E, my_s = tid / a.E;
    const int my_k = my_e / a.EK, my_j = my_e % a.EK;
    int i1 = 0, i2 = 0, kind = 0;            // 0: ns, 1: a[i1], 2: B[i1][i2]
    if (my_j >= 1 && my_j <= D) {
        kind = 1;
        i1 = my_j - 1;
    } else if (my_j > D) {
        kind = 2;
        int p = my_j - 1 - D;
        for (int r0 = 0; r0 < D; ++r0) {
            if (p < D - r0) {
                i1 = r0;
                i2 = r0 + p;
                break;
            }
            p -= D - r0;
        }
    }
    const int rps = (a.C + a.S - 1) / a.S;          // rows of a chunk per split

    // ---- features of the posterior in LDS (ref :738-739, :745-756) - thread k, matrices in registers
    auto features = [&]() {
        if (tid < K) {
            const int k = tid;
            double g[D][D];
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
                for (int j = 0; j < D; ++j) g[i][j] = j <= i ? winv[k * DD + i * D + j] : 0.0;
            // Cholesky W^-1 = G G^T (a non-positive pivot gives NaN, which spreads into the lower bound like inv() would)
#pragma unroll
            for (int j = 0; j < D; ++j) {
                double d = g[j][j];
#pragma unroll
                for (int p = 0; p < j; ++p) d = fma(-g[j][p], g[j][p], d);
                d = sqrt(d);
                g[j][j] = d;
                const double inv = 1.0 / d;
#pragma unroll
                for (int i = j + 1; i < D; ++i) {
                    double v = g[i][j];
#pragma unroll
                    for (int p = 0; p < j; ++p) v = fma(-g[i][p], g[j][p], v);
                    g[i][j] = v * inv;
                }
            }
            const double nuk = nu[k], kap = kappa[k];
            double logdet = 0.0, dig = 0.0, lg = 0.0;
#pragma unroll
            for (int d = 0; d < D; ++d) {
                logdet += log(g[d][d]);
                dig += small_digamma(0.5 * (nuk - d));
                lg += lgamma(0.5 * (nuk - d));
            }
            logdet *= 2.0;
            // in-place inverse of the factor, last column first (X[j+1:, j] = -X[j+1:, j+1:] g[j+1:, j] / g_jj)
#pragma unroll
            for (int j = D - 1; j >= 0; --j) {
                const double xjj = 1.0 / g[j][j];
                double col[D];
#pragma unroll
                for (int i = 0; i < D; ++i) col[i] = i > j ? g[i][j] : 0.0;
#pragma unroll
                for (int i = j + 1; i < D; ++i) {
                    double v = 0.0;
#pragma unroll
                    for (int p = j + 1; p <= i; ++p) v = fma(g[i][p], col[p], v);
                    g[i][j] = -v * xjj;
                }
                g[j][j] = xjj;
            }
            const double sq = sqrt(nuk);
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
                for (int j = 0; j < D; ++j) {
                    um[k * DD + i * D + j] = j <= i ? g[i][j] * sq : 0.0;
                    if (i <= j) {
                        double w = 0.0;
#pragma unroll
                        for (int p = j; p < D; ++p) w = fma(g[p][i], g[p][j], w);
                        wm[k * DD + i * D + j] = w;
                        wm[k * DD + j * D + i] = w;
                    }
                }
            double asum = 0.0;
#pragma unroll 1
            for (int c = 0; c < K; ++c) asum += alpha[c];
            const double e_lp = small_digamma(alpha[k]) - small_digamma(asum);
            const double e_ld = dig + D * LN_2 - logdet;
            elp[k] = e_lp;
            eld[k] = e_ld;
            lnb[k] = 0.5 * (nuk * logdet - nuk * D * LN_2 - 0.5 * D * (D - 1) * LN_PI - 2.0 * lg);
            cc[k] = e_lp + 0.5 * (e_ld - D * LN_2PI - D / kap);
        }
        __syncthreads();
    };

    // ---- one data pass: E-step (ref :772-783; skipped when `loaded`: r comes from the restart's Dirichlet draws,
    // ref :734-736) and the statistics about the pivot (ref :725-732), h = sum r ln r (ref :704)
    auto data_pass = [&](bool loaded) {
        double acc = 0.0, h = 0.0;
        double* r_glob = a.r_out ? a.r_out + (int64_t)rs * N * K : nullptr;
        const double* r_in = loaded ? a.init + (int64_t)rs * N * K : nullptr;
#pragma unroll 1
        for (int64_t n0 = 0; n0 < N; n0 += a.C) {
            if (tid < a.C) {
                const int64_t n = n0 + tid;
                double* rr = rbuf + tid * K;
                if (n < N) {
                    double xv[D];
#pragma unroll
                    for (int i = 0; i < D; ++i) {
                        xv[i] = (double)x[n * a.ldx + i];
                        xbuf[tid * D + i] = xv[i] - a.pivot[i];
                    }
                    if (loaded) {
#pragma unroll 1
                        for (int k = 0; k < K; ++k) {
                            const double v = r_in[n * K + k];
                            rr[k] = v;
                            if (v > 0.0) h = fma(v, log(v), h);
                        }
                    } else {
                        double best = -__builtin_huge_val();
#pragma unroll 1
                        for (int k = 0; k < K; ++k) {
                            double df[D];
#pragma unroll
                            for (int i = 0; i < D; ++i) df[i] = xv[i] - mm[k * D + i];
                            double q = 0.0;
#pragma unroll
                            for (int i = 0; i < D; ++i) {
                                double y = 0.0;
#pragma unroll
                                for (int j = 0; j <= i; ++j) y = fma(um[k * DD + i * D + j], df[j], y);
                                q = fma(y, y, q);
                            }
                            const double v = cc[k] - 0.5 * q;
                            rr[k] = v;
                            best = v > best ? v : best;
                        }
                        double ssum = 0.0;
#pragma unroll 1
                        for (int k = 0; k < K; ++k) ssum += exp(rr[k] - best);
                        const double lse = best + log(ssum);
#pragma unroll 1
                        for (int k = 0; k < K; ++k) {
                            const double t = rr[k] - lse;
                            const double r = exp(t);
                            rr[k] = r;
                            if (r > 0.0) h = fma(r, t, h);
                        }
                    }
                    if (r_glob)
#pragma unroll 1
                        for (int k = 0; k < K; ++k) r_glob[n * K + k] = rr[k];
                } else {
#pragma unroll 1
                    for (int k = 0; k < K; ++k) rr[k] = 0.0;
#pragma unroll
                    for (int i = 0; i < D; ++i) xbuf[tid * D + i] = 0.0;
                }
            }
            __syncthreads();
            if (m_thread) {
                const int lo = my_s * rps, hi = min(a.C, lo + rps);
                if (kind == 0)
#pragma unroll 1
                    for (int n = lo; n < hi; ++n) acc += rbuf[n * K + my_k];
                else if (kind == 1)
#pragma unroll 1
                    for (int n = lo; n < hi; ++n) acc = fma(rbuf[n * K + my_k], xbuf[n * D + i1], acc);
                else
#pragma unroll 1
                    for (int n = lo; n < hi; ++n) acc = fma(rbuf[n * K + my_k] * xbuf[n * D + i1], xbuf[n * D + i2], acc);
            }
            __syncthreads();
        }
        if (m_thread) part[my_s * a.E + my_e] = acc;
        h = sum_wave(h);
        if ((tid & 63) == 0) red[tid >> 6] = h;
        __syncthreads();
        if (tid < a.E) {
            double t = 0.0;
#pragma unroll 1
            for (int s = 0; s < a.S; ++s) t += part[s * a.E + tid];
            if (kind == 0) ns[my_k] = t;
            else if (kind == 1) av[my_k * D + i1] = t;
            else {
                Bm[my_k * DD + i1 * D + i2] = t;
                Bm[my_k * DD + i2 * D + i1] = t;
            }
        }
        if (tid == 0) {
            double t = 0.0;
            for (int w = 0; w < kSmallThreads / 64; ++w) t += red[w];
            red[18] = t;
        }
        __syncthreads();
    };

    // ---- lower bound under the posterior in LDS with the statistics of the pass (ref :671-723); moments (ref :729-732)
    auto lower_bound = [&]() -> double {
        if (tid < K) {
            const int k = tid;
            const double nsk = ns[k];
            const bool pos = nsk > 0.0;
            const double safe = pos ? nsk : 1.0;
            double ab[D], dq[D], dm[D];
#pragma unroll
            for (int i = 0; i < D; ++i) {
                ab[i] = av[k * D + i] / safe;
                const double xb = pos ? a.pivot[i] + ab[i] : 0.0;
                xbar[k * D + i] = xb;
                dq[i] = xb - mm[k * D + i];
                dm[i] = mm[k * D + i] - p_m[k * D + i];
            }
            const double nuq = nu[k], kapq = kappa[k], alq = alpha[k], kap0 = p_kappa[k], nu0 = p_nu[k], al0 = p_alpha[k];
            double t0 = 0.0, t1 = 0.0, t2 = 0.0, t3 = 0.0;
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
                for (int j = 0; j < D; ++j) {
                    const int e = k * DD + i * D + j;
                    const double sij = pos ? Bm[e] / safe - ab[i] * ab[j] : sv[e];
                    sv[e] = sij;
                    const double ew = nuq * wm[e];
                    t0 = fma(sij, ew, t0);
                    t1 = fma(dq[i] * dq[j], ew, t1);
                    t2 = fma(dm[i] * dm[j], ew, t2);
                    t3 = fma(p_winv[e], ew, t3);
                }
            double* p = pt + 12 * k;
            p[0] = 0.5 * nsk * (eld[k] - D / kapq - t0 - t1 - D * LN_2PI);
            p[1] = nsk * elp[k];
            p[2] = (al0 - 1.0) * elp[k];
            p[3] = 0.5 * (D * (log(kap0) - LN_2PI - kap0 / kapq) - kap0 * t2 + 2.0 * p_lnb[k] + (nu0 - D) * eld[k] - t3);
            p[5] = lgamma(alq);
            p[6] = (alq - 1.0) * small_digamma(alq);
            p[7] = alq;
            p[8] = 0.5 * (D * (1.0 + LN_2PI - log(kapq)) - 2.0 * lnb[k] - (nuq - D) * eld[k] + nuq * D);
        }
        __syncthreads();
        if (tid == 0) {
            double t[9] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll 1
            for (int k = 0; k < K; ++k)
                for (int q = 0; q < 9; ++q)
                    if (q != 4) t[q] += pt[12 * k + q];
            const double a0 = t[7];
            const double p_x = t[0], p_z = t[1], p_pi = ln_c_alpha + t[2], p_ml = t[3], q_z = -red[18];
            const double q_pi = (t[5] - lgamma(a0)) + (a0 - K) * small_digamma(a0) - t[6];
            const double q_ml = t[8];
            const double vl = p_x + p_z + p_pi + p_ml + q_z + q_pi + q_ml;
            out[2] = p_x;
            out[3] = p_z;
            out[4] = p_pi;
            out[5] = p_ml;
            out[6] = q_z;
            out[7] = q_pi;
            out[8] = q_ml;
            out[9] = vl;
            red[16] = vl;
        }
        __syncthreads();
        return red[16];
    };

    // ---- closed-form update from the statistics (ref :741-743, :758-770); its features follow
    auto update_hyper = [&]() {
        if (tid < K) {
            const int k = tid;
            const double nsk = ns[k], kap0 = p_kappa[k];
            const double kapn = kap0 + nsk;
            const double coef = kap0 * nsk / kapn;
            double dev0[D];
#pragma unroll
            for (int i = 0; i < D; ++i) {
                const double xb = xbar[k * D + i];
                dev0[i] = xb - p_m[k * D + i];
                mm[k * D + i] = (kap0 * p_m[k * D + i] + nsk * xb) / kapn;
            }
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
                for (int j = 0; j < D; ++j) {
                    const int e = k * DD + i * D + j;
                    winv[e] = p_winv[e] + nsk * sv[e] + coef * (dev0[i] * dev0[j]);
                }
            alpha[k] = p_alpha[k] + nsk;
            kappa[k] = kapn;
            nu[k] = p_nu[k] + nsk;
        }
        __syncthreads();
    };

    // (one copy of every phase in the code: t = -1 is the pass under the initial posterior)
    double vl = 0.0;
    int n_vl = 0, converged = 0;
#pragma unroll 1
    for (int t = -1; t < a.max_itr; ++t) {
        const double vl_before = vl;
        if (t >= 0) update_hyper();
        features();
        data_pass(t < 0 && a.init_type == 1);
        vl = lower_bound();
        if (tid == 0) trace[n_vl] = vl;
        ++n_vl;
        if (t >= 0 && fabs((vl - vl_before) / vl_before) < a.tol) {        // (uniform: every thread read the same LDS word)
            converged = 1;
            break;
        }
    }
    // ---- the restart's result: the posterior that produced the last data pass, its features, that pass's moments
    if (tid == 0) {
        out[0] = (double)n_vl;
        out[1] = (double)converged;
    }
    double* o = trace + (a.max_itr + 1);
#pragma unroll 1
    for (int e = tid; e < K; e += kSmallThreads) {
        o[e] = alpha[e];
        o[K + K * D + e] = kappa[e];
        o[2 * K + K * D + e] = nu[e];
    }
#pragma unroll 1
    for (int e = tid; e < K * D; e += kSmallThreads) o[K + e] = mm[e];
    double* o2 = o + 3 * K + K * D;
#pragma unroll 1
    for (int e = tid; e < K * DD; e += kSmallThreads) {
        o2[e] = winv[e];
        o2[K * DD + e] = wm[e];
    }
    double* o3 = o2 + 2 * K * DD;
#pragma unroll 1
    for (int e = tid; e < K; e += kSmallThreads) {
        o3[e] = elp[e];
        o3[K + e] = eld[e];
        o3[2 * K + e] = lnb[e];
        o3[3 * K + e] = ns[e];
    }
    double* o4 = o3 + 4 * K;
#pragma unroll 1
    for (int e = tid; e < K * D; e += kSmallThreads) o4[e] = xbar[e];
#pragma unroll 1
    for (int e = tid; e < K * DD; e += kSmallThreads) o4[K * D + e] = sv[e];
}

template <typename XT>
static hipError_t launch_small(int D, int R, size_t lds, hipStream_t st, const SmallArgs& a) {
#define GMMVB_SMALL_CASE(DV)                                                                                          \
    case DV: {                                                                                                        \
        hipError_t e = hipFuncSetAttribute((const void*)small_fit_kernel<DV, XT>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                           (int)lds);                                                                 \
        if (e != hipSuccess) return e;                                                                                \
        hipLaunchKernelGGL((small_fit_kernel<DV, XT>), dim3(R), dim3(kSmallThreads), lds, st, a);                     \
        return hipGetLastError();                                                                                     \
    }
    switch (D) {
        GMMVB_SMALL_CASE(1)
        GMMVB_SMALL_CASE(2)
        GMMVB_SMALL_CASE(3)
        GMMVB_SMALL_CASE(4)
        GMMVB_SMALL_CASE(5)
        GMMVB_SMALL_CASE(6)
        GMMVB_SMALL_CASE(7)
        GMMVB_SMALL_CASE(8)
        default: return hipErrorInvalidValue;
    }
#undef GMMVB_SMALL_CASE
}

}  // namespace gmmvb

using namespace gmmvb;

extern "C" {

int gmmvb_small_supported(int K, int D, int64_t n_rows) {
    if (K < 1 || D < 1 || n_rows < 1) return 0;
    if (D > kSmallMaxD || K > kSmallMaxK || n_rows > kSmallMaxRows) return 0;
    return K * (1 + D + D * (D + 1) / 2) <= kSmallMaxElems ? 1 : 0;
}

int64_t gmmvb_small_out_len(int K, int D, int max_itr) {
    if (K < 1 || D < 1 || max_itr < 0) return -1;
    // n_vl, converged | the lower bound's terms of the last pass | trace | alpha, m, kappa, nu | w_inv, w | e_ln_pi,
    // e_ln_lambda_det, ln_b_w_nu, ns | x_bar | s
    return 2 + kSmallTerms + (int64_t)(max_itr + 1) + small_post_len(K, D);
}

int gmmvb_small_fit(int K, int D, int x_dtype, const void* x_dev, int64_t ldx, int64_t n_rows, const double* pivot_dev,
                    const double* prior_dev, int n_restarts, int init_type, const double* init_dev, int max_itr,
                    double tolerance, double* out_dev, double* r_dev, void* stream) {
    if (!x_dev || !pivot_dev || !prior_dev || !out_dev) return fail(GMMVB_EINVAL, "null argument");
    if (x_dtype != GMMVB_F32 && x_dtype != GMMVB_F64) return fail(GMMVB_EINVAL, "x_dtype must be GMMVB_F32 or GMMVB_F64");
    if (n_restarts < 1 || max_itr < 0 || ldx < D) return fail(GMMVB_EINVAL, "bad argument");
    if (init_type != 0 && init_type != 1) return fail(GMMVB_EINVAL, "init_type must be 0 (sub-sample moments) or 1 (responsibilities)");
    if (!init_dev) return fail(GMMVB_EINVAL, "init_dev is null");
    if (!gmmvb_small_supported(K, D, n_rows)) return fail(GMMVB_EUNSUPPORTED, "shape outside the small-problem kernel's range");
    const SmallShape sh = small_shape(K, D);
    SmallArgs a{x_dev, ldx, n_rows, pivot_dev, prior_dev, init_dev, K, init_type, max_itr, sh.EK, sh.E, sh.S, sh.C, tolerance,
                out_dev, gmmvb_small_out_len(K, D, max_itr), r_dev};
    const size_t lds = sh.lds_doubles * sizeof(double);
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = x_dtype == GMMVB_F64 ? launch_small<double>(D, n_restarts, lds, st, a) : launch_small<float>(D, n_restarts, lds, st, a);
    if (e != hipSuccess) return fail(GMMVB_EHIP, "small_fit_kernel launch", e);
    return GMMVB_OK;
}

}  // extern "C"
