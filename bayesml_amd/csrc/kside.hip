// K-sized linear algebra of the posterior update that the torch K-side would otherwise hand to MAGMA / rocSOLVER:
// one Cholesky factorisation and one triangular inverse per component, LDS-resident (D <= 128: 129 KB of the 160 KB).
//
// Replaces, per VB iteration, what the reference does with np.linalg.inv + two slogdet per component
// (bayesml/gaussianmixture/_gaussianmixture.py:746-756, 769): from W^-1 = G G^T it yields G (-> u^-1 = G / sqrt(nu)),
// G^-1 (-> u = sqrt(nu) G^-1, W = G^-T G^-1) and ln det W^-1 = 2 sum ln diag G.  Plain kernels on the caller's stream,
// no host synchronisation, no allocation: the whole K-side can be captured in a hipGraph.
#include "workspace.h"

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

__device__ __forceinline__ double block_sum256(double v, double* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);          // fixed order
}

__global__ __launch_bounds__(256) void drift_kernel(const double* __restrict__ u_old, const double* __restrict__ uinv_old,
                                                    const double* __restrict__ m_old, const double* __restrict__ u_new,
                                                    const double* __restrict__ uinv_new, const double* __restrict__ m_new,
                                                    int D, int sq_small, int sq_big, double* __restrict__ gamma,
                                                    double* __restrict__ delta, double* __restrict__ big,
                                                    double* __restrict__ enorm) {
    extern __shared__ double sm[];           // [128][kDriftLd]
    __shared__ double red[4];
    const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
    const int k = blockIdx.x, dir = blockIdx.y;
    const int64_t base = (int64_t)k * D * D;
    const double* L = (dir == 0 ? u_old : u_new) + base;          // A = L R  (dir 2: A - I)
    const double* R = (dir == 0 ? uinv_new : uinv_old) + base;
    const int sq = dir == 1 ? sq_big : sq_small;
    const double tiny = 2.2250738585072014e-308;
    double acc[8][8];

    if (dir == 0) {          // delta
        double part = 0.0;
        if (tid < D) {
            double y = 0.0;
            for (int i = 0; i < D; ++i) y = fma(u_new[base + (int64_t)tid * D + i], m_new[(int64_t)k * D + i] - m_old[(int64_t)k * D + i], y);
            part = y * y;
        }
        const double t = block_sum256(part, red);
        if (tid == 0) delta[k] = sqrt(t) * (1.0 + 1e-9);
    }
    // stage L (zero padded to 128 x 128)
    for (int e = tid; e < 128 * 128; e += 256) {
        const int i = e >> 7, j = e & 127;
        sm[i * kDriftLd + j] = (i < D && j < D) ? L[(int64_t)i * D + j] : 0.0;
    }
    __syncthreads();
    // A = L R : A[i][j] = sum_p L[i][p] R[p][j]   (R straight from global memory / L2: consecutive tx read consecutive j)
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b) acc[a][b] = 0.0;
    for (int p = 0; p < D; ++p) {
        double lv[8], rv[8];
#pragma unroll
        for (int a = 0; a < 8; ++a) lv[a] = sm[(ty + 16 * a) * kDriftLd + p];
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const int j = tx + 16 * b;
            rv[b] = j < D ? R[(int64_t)p * D + j] : 0.0;
        }
#pragma unroll
        for (int a = 0; a < 8; ++a)
#pragma unroll
            for (int b = 0; b < 8; ++b) acc[a][b] = fma(lv[a], rv[b], acc[a][b]);
    }
    __syncthreads();
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const int i = ty + 16 * a, j = tx + 16 * b;
            sm[i * kDriftLd + j] = acc[a][b] - ((dir == 2 && i == j && i < D) ? 1.0 : 0.0);
        }
    __syncthreads();
    // G = A^T A, then the squarings; every product is followed by its Frobenius norm
    double log_lmax = 0.0, w = 1.0;
    for (int it = 0; it <= sq; ++it) {
#pragma unroll
        for (int a = 0; a < 8; ++a)
#pragma unroll
            for (int b = 0; b < 8; ++b) acc[a][b] = 0.0;
        for (int p = 0; p < 128; ++p) {
            double lv[8], rv[8];
#pragma unroll
            for (int a = 0; a < 8; ++a) lv[a] = it == 0 ? sm[p * kDriftLd + ty + 16 * a] : sm[(ty + 16 * a) * kDriftLd + p];
#pragma unroll
            for (int b = 0; b < 8; ++b) rv[b] = sm[p * kDriftLd + tx + 16 * b];
#pragma unroll
            for (int a = 0; a < 8; ++a)
#pragma unroll
                for (int b = 0; b < 8; ++b) acc[a][b] = fma(lv[a], rv[b], acc[a][b]);
        }
        double ss = 0.0;
#pragma unroll
        for (int a = 0; a < 8; ++a)
#pragma unroll
            for (int b = 0; b < 8; ++b) ss = fma(acc[a][b], acc[a][b], ss);
        double f = sqrt(block_sum256(ss, red));          // (the barrier inside also ends every read of the old matrix)
        f = f > tiny ? f : tiny;                         // also NaN -> tiny: the NaNs then show up in log_lmax below
        log_lmax += w * log(f);
        w *= 0.5;
        const double inv = 1.0 / f;
#pragma unroll
        for (int a = 0; a < 8; ++a)
#pragma unroll
            for (int b = 0; b < 8; ++b) sm[(ty + 16 * a) * kDriftLd + tx + 16 * b] = acc[a][b] * inv;
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

}  // namespace gmmvb

using namespace gmmvb;

extern "C" int gmmvb_kside_factor(int K, int D, const double* w_inv_dev, double* g_dev, double* g_inv_dev,
                                  double* logdet_dev, void* stream) {
    if (K < 1 || D < 1) return fail(GMMVB_EINVAL, "K and D must be positive");
    if (D > 128) return fail(GMMVB_EUNSUPPORTED, "gmmvb_kside_factor: D > 128 (the factor is kept in LDS)");
    if (!w_inv_dev || !g_dev || !g_inv_dev || !logdet_dev) return fail(GMMVB_EINVAL, "null argument");
    const size_t lds = (size_t)D * (D + 1) * sizeof(double);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)chol_inv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           128 * 129 * (int)sizeof(double));
        if (e != hipSuccess) return fail(GMMVB_EHIP, "hipFuncSetAttribute(chol_inv_kernel)", e);
        attr_set = true;
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
    const size_t lds = (size_t)128 * kDriftLd * sizeof(double);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)drift_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return fail(GMMVB_EHIP, "hipFuncSetAttribute(drift_kernel)", e);
        attr_set = true;
    }
    hipLaunchKernelGGL(drift_kernel, dim3(K, 3), dim3(256), lds, (hipStream_t)stream, u_old_dev, uinv_old_dev, m_old_dev,
                       u_new_dev, uinv_new_dev, m_new_dev, D, squarings, squarings_big, gamma_dev, delta_dev, big_gamma_dev,
                       enorm_dev);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(GMMVB_EHIP, "drift_kernel launch", e);
    return GMMVB_OK;
}
