// Data pass for c_degree > 128 (more than 8 feature tiles): plain f64 vector arithmetic, no MFMA tiling.
//
// The MFMA kernels of estep.h / mstep.h keep a whole 16T x 16T factor image in LDS and a sample's T feature tiles in
// registers; that stops at T = 8.  The reference takes any positive c_degree (_gaussianmixture.py:433), so beyond 128 the
// same two computations - ln rho_nk = c_k - 0.5 || U_k (x_n - m_k) ||^2 (:773-781) and the weighted moments ns, h, a, B
// (:725-732, :704) - run here at the vector ALU's f64 rate: same formulation, same statistics block, same read-outs (they
// only see the [K][npad] ln rho array), no pruning, no lists.  Slow path by construction: N K D^2 flops at ~10 TFLOP/s.
#pragma once
#include "common.h"

namespace gmmvb {

// rows of one E-step workgroup (one wave): the centred rows d = x - m_k sit in LDS feature-major, [D][rows]
__host__ __device__ constexpr int generic_rows(int D) {
    int r = 64;
    while (r > 1 && (int64_t)r * D > 19200) r >>= 1;        // 150 KB of doubles
    return r;
}

// ln rho for the rows [blockIdx.x * R, ...) and component blockIdx.y.  Lane = row; U_k[j][i] is read through scalar loads
// (the address does not depend on the lane).
template <typename XT>
__global__ __launch_bounds__(64) void estep_generic_kernel(const XT* __restrict__ x, int64_t ldx, int64_t n_rows, int D,
                                                           const double* __restrict__ u /*[K][D][D] lower triangular*/,
                                                           const double* __restrict__ m /*[K][D]*/,
                                                           const double* __restrict__ cvec, int R,
                                                           double* __restrict__ lnrho, int64_t npad) {
    extern __shared__ double dsh[];                           // [D][R]
    const int k = blockIdx.y;
    const int lane = threadIdx.x;
    const int64_t row = (int64_t)blockIdx.x * R + lane;
    const bool live = lane < R && row < n_rows;
    const double* mk = m + (int64_t)k * D;
    const double* uk = u + (int64_t)k * D * D;
    if (lane < R) {
        const XT* xr = x + (live ? row : n_rows - 1) * ldx;
        for (int i = 0; i < D; ++i) dsh[(int64_t)i * R + lane] = (double)xr[i] - mk[i];
    }
    __syncthreads();
    double q = 0.0;
    for (int j = 0; j < D; ++j) {
        const double* uj = uk + (int64_t)j * D;
        double y = 0.0;
        for (int i = 0; i <= j; ++i) y = fma(uj[i], dsh[(int64_t)i * R + (lane < R ? lane : 0)], y);
        q = fma(y, y, q);
    }
    if (live) lnrho[(int64_t)k * npad + row] = cvec[k] - 0.5 * q;
}

// responsibility of (row n, component k) and its contribution to h, per mode (mstep.h: direct_r)
__device__ __forceinline__ double generic_weight(const double* __restrict__ lr, const double* __restrict__ lse,
                                                 const double* __restrict__ aux, int64_t n, int direct_r, double& h) {
    const double v = lr[n];
    if (direct_r == 2) {                  // HMM: r = gamma, h accumulates sum gamma * ln rho (aux)
        if (v > 0.0 && aux) h = fma(v, aux[n], h);           // (no aux: h stays 0, hmmvb_skip_h)
        return v;
    }
    if (direct_r) {                       // responsibilities given directly
        if (v > 0.0) h = fma(v, log(v), h);
        return v;
    }
    const double t = v - lse[n];
    const double r = exp(t);
    h = fma(r, t, h);
    return r;
}

// ns, h, a of component blockIdx.x over the rows of split blockIdx.y  ->  first[(split K + k)(D + 2)] = [ns | h | a[D]]
template <typename XT>
__global__ __launch_bounds__(256) void mstep_generic_first_kernel(const XT* __restrict__ x, int64_t ldx, int64_t n_rows, int D,
                                                                  const double* __restrict__ pivot,
                                                                  const double* __restrict__ lnrho, const double* __restrict__ lse,
                                                                  const double* __restrict__ aux, int64_t npad, int K,
                                                                  int64_t rows_per_split, int direct_r,
                                                                  double* __restrict__ first) {
    __shared__ double wsh[256];
    __shared__ double red[2][256];
    const int k = blockIdx.x, tid = threadIdx.x;
    const int64_t lo = (int64_t)blockIdx.y * rows_per_split;
    const int64_t hi = lo + rows_per_split < n_rows ? lo + rows_per_split : n_rows;
    const double* lr = lnrho + (int64_t)k * npad;
    const double* ax = aux ? aux + (int64_t)k * npad : nullptr;
    double ns = 0.0, h = 0.0;
    double* out = first + ((int64_t)blockIdx.y * K + k) * (D + 2);
    // features in panels of 256 (one per thread); the weights of 256 rows at a time through LDS
    for (int f0 = 0; f0 < D; f0 += 256) {
        const int f = f0 + tid;
        const double pv = f < D ? pivot[f] : 0.0;
        double a = 0.0;
        for (int64_t c0 = lo; c0 < hi; c0 += 256) {
            const int64_t n = c0 + tid;
            double hh = 0.0;
            const double r = n < hi ? generic_weight(lr, lse, ax, n, direct_r, hh) : 0.0;
            __syncthreads();
            wsh[tid] = r;
            if (f0 == 0) {
                ns += r;
                h += hh;
            }
            __syncthreads();
            const int cnt = hi - c0 < 256 ? (int)(hi - c0) : 256;
            if (f < D)
                for (int c = 0; c < cnt; ++c) a = fma(wsh[c], (double)x[(c0 + c) * ldx + f] - pv, a);
        }
        if (f < D) out[2 + f] = a;
    }
    red[0][tid] = ns;
    red[1][tid] = h;
    __syncthreads();
    if (tid == 0) {
        double s0 = 0.0, s1 = 0.0;
        for (int i = 0; i < 256; ++i) {         // fixed order
            s0 += red[0][i];
            s1 += red[1][i];
        }
        out[0] = s0;
        out[1] = s1;
    }
}

// One 16 x 16 tile (i0 <= j0) of B_k over the rows of a split  ->  second[((split K + k) tiles + tile) 256 + ti 16 + tj]
template <typename XT>
__global__ __launch_bounds__(256) void mstep_generic_second_kernel(const XT* __restrict__ x, int64_t ldx, int64_t n_rows, int D,
                                                                   const double* __restrict__ pivot,
                                                                   const double* __restrict__ lnrho, const double* __restrict__ lse,
                                                                   const double* __restrict__ aux, int64_t npad, int K,
                                                                   int64_t rows_per_split, int direct_r, int TD /*ceil(D/16)*/,
                                                                   double* __restrict__ second) {
    __shared__ double wsh[64];
    __shared__ double xi[64][17], xj[64][17];
    const int tile = blockIdx.x, k = blockIdx.y, tid = threadIdx.x;
    int tj0 = 0;
    while (tri_pairs(tj0 + 1) <= tile) ++tj0;                  // tile = pair_index(tj0, ti0), ti0 <= tj0
    const int ti0 = tile - tri_pairs(tj0);
    const int ti = tid >> 4, tj = tid & 15;
    const int64_t lo = (int64_t)blockIdx.z * rows_per_split;
    const int64_t hi = lo + rows_per_split < n_rows ? lo + rows_per_split : n_rows;
    const double* lr = lnrho + (int64_t)k * npad;
    const double* ax = aux ? aux + (int64_t)k * npad : nullptr;
    double acc = 0.0;
    for (int64_t c0 = lo; c0 < hi; c0 += 64) {
        __syncthreads();
        if (tid < 64) {
            const int64_t n = c0 + tid;
            double hh = 0.0;
            wsh[tid] = n < hi ? generic_weight(lr, lse, ax, n, direct_r, hh) : 0.0;
        }
        for (int e = tid; e < 64 * 32; e += 256) {
            const int c = e >> 5, q = e & 31;
            const int f = (q < 16 ? 16 * ti0 + q : 16 * tj0 + q - 16);
            const int64_t n = c0 + c;
            const double v = (n < hi && f < D) ? (double)x[n * ldx + f] - pivot[f] : 0.0;
            if (q < 16) xi[c][q] = v;
            else xj[c][q - 16] = v;
        }
        __syncthreads();
#pragma unroll 8
        for (int c = 0; c < 64; ++c) acc = fma(wsh[c] * xi[c][ti], xj[c][tj], acc);
    }
    const int tiles = tri_pairs(TD);
    second[(((int64_t)blockIdx.z * K + k) * tiles + tile) * 256 + tid] = acc;
}

// sums over the splits in split order and scatters to stats = [ns | h | a | B] (B mirrored exactly)
static __global__ void reduce_generic_kernel(const double* __restrict__ first, const double* __restrict__ second, int S, int K, int D,
                                      int TD, double* __restrict__ stats) {
    const int k = blockIdx.y;
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int tiles = tri_pairs(TD);
    const int64_t n_first = D + 2, n_second = (int64_t)tiles * 256;
    if (e >= n_first + n_second) return;
    double v = 0.0;
    if (e < n_first) {
        for (int s = 0; s < S; ++s) v += first[((int64_t)s * K + k) * (D + 2) + e];
        if (e == 0) stats[k] = v;
        else if (e == 1) stats[K + k] = v;
        else stats[2 * (int64_t)K + (int64_t)k * D + (e - 2)] = v;
        return;
    }
    const int64_t t = e - n_first;
    const int tile = (int)(t >> 8), ti = (int)((t >> 4) & 15), tj = (int)(t & 15);
    int tj0 = 0;
    while (tri_pairs(tj0 + 1) <= tile) ++tj0;
    const int ti0 = tile - tri_pairs(tj0);
    const int f1 = 16 * ti0 + ti, f2 = 16 * tj0 + tj;
    if (f1 >= D || f2 >= D) return;
    if (ti0 == tj0 && ti > tj) return;       // diagonal tiles: one triangle, mirrored
    for (int s = 0; s < S; ++s) v += second[(((int64_t)s * K + k) * tiles + tile) * 256 + (t & 255)];
    double* B = stats + 2 * (int64_t)K + (int64_t)K * D;
    B[((int64_t)k * D + f1) * D + f2] = v;
    B[((int64_t)k * D + f2) * D + f1] = v;
}

}  // namespace gmmvb
