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
    // right-looking Cholesky, column by column
    for (int j = 0; j < D; ++j) {
        if (tid == 0) sm[j * ld + j] = sqrt(sm[j * ld + j]);
        __syncthreads();
        const double inv = 1.0 / sm[j * ld + j];
        for (int i = j + 1 + tid; i < D; i += 256) sm[i * ld + j] *= inv;
        __syncthreads();
        const int m = D - j - 1;
        for (int e = tid; e < m * m; e += 256) {
            const int i = e / m, c = e - i * m;
            if (c <= i) sm[(j + 1 + i) * ld + j + 1 + c] -= sm[(j + 1 + i) * ld + j] * sm[(j + 1 + c) * ld + j];
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
            for (int p = j + 1; p <= i; ++p) v = fma(sm[i * ld + p], sm[p * ld + j], v);
            v = -v * xjj;
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
                                           160 * 1024);
        if (e != hipSuccess) return fail(GMMVB_EHIP, "hipFuncSetAttribute(chol_inv_kernel)", e);
        attr_set = true;
    }
    hipLaunchKernelGGL(chol_inv_kernel, dim3(K), dim3(256), lds, (hipStream_t)stream, w_inv_dev, D, g_dev, g_inv_dev,
                       logdet_dev);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(GMMVB_EHIP, "chol_inv_kernel launch", e);
    return GMMVB_OK;
}
