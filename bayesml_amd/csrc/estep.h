// E-step kernel: ln rho_nk = c_k - 0.5 * || U_k (x_n - m_k) ||^2 on f64 MFMA.
//
// Replaces the per-component loop of the reference's _update_q_z
// (bayesml/gaussianmixture/_gaussianmixture.py:773-781).  The reference forms
// sum((diff @ Lambda_k) * diff); here Lambda_k = U_k^T U_k with U_k lower triangular, so the
// quadratic form is the squared norm of y = U_k x - U_k m_k: a [16T x 16T] x [16T x samples]
// product whose upper-triangular tile pairs are skipped, with -U_k m_k as the initial accumulator.
//
// Mapping (one wave = 16*NB samples, all K components, no LDS, no cross-wave traffic):
//   MFMA D[j][n] += A[j][i] B[i][n]:  A = 16x16 tile of U_k (rows j), B = x^T (samples on columns).
//   Lane l = (n = l & 15, g = l >> 4) keeps, for each of its NB samples and each 16-feature block b,
//   the four features 16b + 4g + s (s = 0..3) in registers for the whole k loop: x is read from HBM
//   exactly once.  U_k tiles stream from L2 (K * P * 2 KB, shared by every wave on the chip).
//   The accumulator has the sample on the lane and the output row on (register, lane group), so
//   ||y||^2 is a per-lane sum of squares plus two cross-group adds.
#pragma once
#include "common.h"

namespace gmmvb {

template <typename XT>
__host__ __device__ constexpr int estep_nb(int t) {
    // register budget: accumulators 8*T*NB VGPRs, x registers T*NB*4*(sizeof(XT)/4)
    const int cap = (sizeof(XT) == 4 ? 16 : 8) / t;
    return cap < 1 ? 1 : (cap > 4 ? 4 : cap);
}

template <int T, typename XT, bool VEC>
__global__ __launch_bounds__(256) void estep_mfma_f64(
    const XT* __restrict__ x, int64_t ldx, int64_t n_rows, int D,
    const double* __restrict__ upack,   // [K][P][16][16]
    const double* __restrict__ bpack,   // [K][T][4][4]  = -(U_k m_k)[16 jt + g + 4 r]
    const double* __restrict__ cvec,    // [K]
    int K, double* __restrict__ lnrho /*[K][npad]*/, int64_t npad) {
    constexpr int NB = estep_nb<XT>(T);
    constexpr int P = tri_pairs(T);
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int n = lane & 15;
    const int g = lane >> 4;
    const int64_t rows_per_wave = 16 * NB;
    const int64_t n_tiles = (n_rows + rows_per_wave - 1) / rows_per_wave;

    for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < n_tiles; tile += (int64_t)gridDim.x * 4) {
        const int64_t n0 = tile * rows_per_wave;
        // ---- x tile -> registers (kept in storage dtype, widened at each use)
        XT xr[NB][T][4];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            int64_t row = n0 + 16 * nb + n;
            if (row >= n_rows) row = n_rows - 1;      // clamp: padded samples are never stored
            const XT* xp = x + row * ldx + 4 * g;
#pragma unroll
            for (int b = 0; b < T; ++b) {
                if constexpr (VEC) {
                    typedef XT v4 __attribute__((ext_vector_type(4)));
                    const v4 v = *reinterpret_cast<const v4*>(xp + 16 * b);
#pragma unroll
                    for (int s = 0; s < 4; ++s) xr[nb][b][s] = v[s];
                } else {
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        const int f = 16 * b + 4 * g + s;
                        xr[nb][b][s] = f < D ? xp[16 * b + s] : XT(0);
                    }
                }
            }
        }
        // ---- all components
        for (int k = 0; k < K; ++k) {
            const double* up = upack + (int64_t)k * P * 256 + n * 16 + g * 4;
            const double* bp = bpack + ((int64_t)k * T * 4 + g) * 4;
            d4 acc[T][NB];
#pragma unroll
            for (int jt = 0; jt < T; ++jt) {
                const d4 bias = *reinterpret_cast<const d4*>(bp + jt * 16);
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) acc[jt][nb] = bias;
            }
#pragma unroll
            for (int jt = 0; jt < T; ++jt) {
#pragma unroll
                for (int b = 0; b <= jt; ++b) {
                    const d4 a = *reinterpret_cast<const d4*>(up + pair_index(jt, b) * 256);
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb)
                            acc[jt][nb] = mfma_f64(a[s], (double)xr[nb][b][s], acc[jt][nb]);
                    }
                }
            }
            const double ck = cvec[k];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                double q = 0.0;
#pragma unroll
                for (int jt = 0; jt < T; ++jt) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) q = fma(acc[jt][nb][r], acc[jt][nb][r], q);
                }
                q = sum_groups(q);
                const int64_t row = n0 + 16 * nb + n;
                if (g == 0 && row < n_rows) lnrho[(int64_t)k * npad + row] = ck - 0.5 * q;
            }
        }
    }
}

}  // namespace gmmvb
