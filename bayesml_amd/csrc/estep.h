// E-step kernels: ln rho_nk = c_k - 0.5 * || U_k (x_n - m_k) ||^2 on f64 MFMA.
//
// Replaces the per-component loop of the reference's _update_q_z
// (bayesml/gaussianmixture/_gaussianmixture.py:773-781).  The reference forms
// sum((diff @ Lambda_k) * diff); here Lambda_k = U_k^T U_k with U_k lower triangular, so the
// quadratic form is the squared norm of y = U_k x - U_k m_k: a [16T x 16T] x [16T x samples]
// product whose upper-triangular tile pairs are skipped, with -U_k m_k as the initial accumulator.
//
// Mapping (one wave = 16*NB samples, all K components):
//   MFMA D[j][n] += A[j][i] B[i][n]:  A = 16x16 tile of U_k (rows j), B = x^T (samples on columns).
//   Lane l = (n = l & 15, g = l >> 4) keeps, for each of its NB samples and each 16-feature block b,
//   the four features 16b + 4g + s (s = 0..3) in registers for the whole k loop: x is read from HBM
//   exactly once.  The accumulator has the sample on the lane and the output row on (register,
//   lane group), so ||y||^2 is a per-lane sum of squares plus two cross-group adds.
//
// Parameter image (written by pack_params_kernel, one per component, IMG doubles, 1 KB granules):
//   [ P tile pairs ][ half h ][ lane 0..63 ][ 2 ]   element = U[16 jt + (lane & 15)][16 b + 4 (lane >> 4) + 2 h + e]
//   [ T ][ g ][ r ]                                 bias    = -(U m)[16 jt + g + 4 r]
// so that a wave reads its A fragments as two lane-linear 16-byte accesses (conflict-free in LDS,
// fully coalesced from L2) and the image can be copied global -> LDS by 1-KB LDS-DMA pieces.
//
// Two variants share that image:
//   estep_lds_f64   U_k images double-buffered in LDS by global_load_lds (one L2 read per workgroup,
//                   latency hidden a whole component block ahead); the default.
//   estep_mfma_f64  no LDS: every wave streams the image from L2 straight into registers.
#pragma once
#include "common.h"

namespace gmmvb {

template <typename XT>
__host__ __device__ constexpr int estep_nb(int t) {
    // register budget: accumulators 8*T*NB VGPRs, x registers T*NB*4*(sizeof(XT)/4)
    const int cap = (sizeof(XT) == 4 ? 16 : 8) / t;
    return cap < 1 ? 1 : (cap > 4 ? 4 : cap);
}

// doubles per component image, rounded up to 1 KB (= one wave-wide 16-byte LDS-DMA piece)
__host__ __device__ constexpr int img_doubles(int t) { return (tri_pairs(t) * 256 + t * 16 + 127) / 128 * 128; }
// components staged per LDS buffer: fill ~64 KB per buffer (2 buffers <= 160 KB)
__host__ __device__ constexpr int estep_kb(int t) {
    const int kb = (64 * 1024) / (img_doubles(t) * 8);
    return kb < 1 ? 1 : (kb > 16 ? 16 : kb);
}

template <int T, int NB, typename XT, bool VEC>
__device__ __forceinline__ void load_x_tile(const XT* __restrict__ x, int64_t ldx, int64_t n_rows, int D, int64_t n0,
                                            int n, int g, XT (&xr)[NB][T][4]) {
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
}

// One component for one wave tile: image pointer `im` (global or LDS) -> ln rho stores.
template <int T, int NB, typename XT, typename ImgPtr>
__device__ __forceinline__ void estep_component(ImgPtr im, const XT (&xr)[NB][T][4], double ck, int lane, int n, int g,
                                                int64_t n0, int64_t n_rows, double* __restrict__ lnrho_k) {
    constexpr int P = tri_pairs(T);
    typedef double d2 __attribute__((ext_vector_type(2)));
    d4 acc[T][NB];
#pragma unroll
    for (int jt = 0; jt < T; ++jt) {
        const d2 b01 = *reinterpret_cast<const d2*>(im + P * 256 + (jt * 4 + g) * 4);
        const d2 b23 = *reinterpret_cast<const d2*>(im + P * 256 + (jt * 4 + g) * 4 + 2);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[jt][nb] = d4{b01[0], b01[1], b23[0], b23[1]};
    }
#pragma unroll
    for (int jt = 0; jt < T; ++jt) {
#pragma unroll
        for (int b = 0; b <= jt; ++b) {
            const int p = pair_index(jt, b);
            const d2 a01 = *reinterpret_cast<const d2*>(im + p * 256 + lane * 2);
            const d2 a23 = *reinterpret_cast<const d2*>(im + p * 256 + 128 + lane * 2);
            const double a[4] = {a01[0], a01[1], a23[0], a23[1]};
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) acc[jt][nb] = mfma_f64(a[s], (double)xr[nb][b][s], acc[jt][nb]);
            }
        }
    }
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
        if (g == 0 && row < n_rows) lnrho_k[row] = ck - 0.5 * q;
    }
}

// ---- variant without LDS -------------------------------------------------------------------
template <int T, typename XT, bool VEC>
__global__ __launch_bounds__(256) void estep_mfma_f64(const XT* __restrict__ x, int64_t ldx, int64_t n_rows, int D,
                                                      const double* __restrict__ img /*[K][IMG]*/,
                                                      const double* __restrict__ cvec, int K,
                                                      double* __restrict__ lnrho /*[K][npad]*/, int64_t npad) {
    constexpr int NB = estep_nb<XT>(T);
    constexpr int IMG = img_doubles(T);
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int n = lane & 15, g = lane >> 4;
    const int64_t rows_per_wave = 16 * NB;
    const int64_t n_tiles = (n_rows + rows_per_wave - 1) / rows_per_wave;
    for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < n_tiles; tile += (int64_t)gridDim.x * 4) {
        const int64_t n0 = tile * rows_per_wave;
        XT xr[NB][T][4];
        load_x_tile<T, NB, XT, VEC>(x, ldx, n_rows, D, n0, n, g, xr);
        for (int k = 0; k < K; ++k)
            estep_component<T, NB, XT>(img + (int64_t)k * IMG, xr, cvec[k], lane, n, g, n0, n_rows,
                                       lnrho + (int64_t)k * npad);
    }
}

// ---- LDS-staged variant ----------------------------------------------------------------------
// NW = 4: one wave per SIMD, 16*NB samples per wave.  NW = 8: two waves per SIMD with half the samples
// each (same samples per workgroup and per LDS fill), so one wave's epilogue / LDS waits / barrier
// arrival overlap the other wave's MFMAs.
template <typename XT>
__host__ __device__ constexpr int estep_nb_w(int t, int nw) {
    return nw == 8 ? (estep_nb<XT>(t) > 1 ? estep_nb<XT>(t) / 2 : 1) : estep_nb<XT>(t);
}

template <int T, typename XT, bool VEC, int NW>
__global__ __launch_bounds__(64 * NW) void estep_lds_f64(const XT* __restrict__ x, int64_t ldx, int64_t n_rows, int D,
                                                         const double* __restrict__ img /*[K][IMG]*/,
                                                         const double* __restrict__ cvec, int K,
                                                         double* __restrict__ lnrho /*[K][npad]*/, int64_t npad) {
    constexpr int NB = estep_nb_w<XT>(T, NW);
    constexpr int IMG = img_doubles(T);
    constexpr int KB = estep_kb(T);
    __shared__ __attribute__((aligned(16))) double smem[2][KB * IMG];   // the ONLY LDS object of the kernel
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = lane & 15, g = lane >> 4;
    const int64_t rows_per_wg = NW * 16 * NB;
    const int64_t n_wg_tiles = (n_rows + rows_per_wg - 1) / rows_per_wg;
    const int n_blocks = (K + KB - 1) / KB;

    // global -> LDS copy of component block kb: 1-KB pieces (64 lanes x 16 B), lane-linear on both sides
    auto stage = [&](int kb, int buf) {
        const int k0 = kb * KB;
        const int kcount = (K - k0 < KB) ? (K - k0) : KB;
        const int pieces = kcount * (IMG / 128);
        const double* src = img + (int64_t)k0 * IMG + lane * 2;
        for (int piece = wave; piece < pieces; piece += NW)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + piece * 128),
                                             (__attribute__((address_space(3))) void*)(&smem[buf][piece * 128]), 16, 0,
                                             0);
    };

    for (int64_t wt = blockIdx.x; wt < n_wg_tiles; wt += gridDim.x) {
        const int64_t n0 = wt * rows_per_wg + (int64_t)wave * 16 * NB;   // may lie past n_rows: rows clamp, stores mask
        XT xr[NB][T][4];
        load_x_tile<T, NB, XT, VEC>(x, ldx, n_rows, D, n0, n, g, xr);
        stage(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int kb = 0; kb < n_blocks; ++kb) {
            if (kb + 1 < n_blocks) stage(kb + 1, (kb + 1) & 1);      // prefetch a whole block ahead
            const double* buf = smem[kb & 1];
            const int k0 = kb * KB;
#pragma unroll 1
            for (int kk = 0; kk < KB; ++kk) {
                const int k = k0 + kk;
                if (k >= K) break;
                estep_component<T, NB, XT>(buf + kk * IMG, xr, cvec[k], lane, n, g, n0, n_rows,
                                           lnrho + (int64_t)k * npad);
            }
            // the prefetched block must have landed, and every wave must be done reading this one
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }
}

}  // namespace gmmvb
