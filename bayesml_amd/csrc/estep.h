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
#include <type_traits>
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

// rows[nb] = row of x this lane's sample (n, nb) is read from (always a valid row)
template <int T, int NB, typename XT, bool VEC>      // T = feature blocks to load (the first 16 T features)
__device__ __forceinline__ void load_x_tile(const XT* __restrict__ x, int64_t ldx, int D, const int64_t (&rows)[NB],
                                            int g, XT (&xr)[NB][T][4]) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const XT* xp = x + rows[nb] * ldx + 4 * g;
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

// rows of a contiguous wave tile starting at n0: load rows clamp to the last valid row, store rows are -1 past the end
template <int NB>
__device__ __forceinline__ void tile_rows(int64_t n0, int n, int64_t n_rows, int64_t (&ld)[NB], int64_t (&stv)[NB]) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int64_t row = n0 + 16 * nb + n;
        ld[nb] = row < n_rows ? row : n_rows - 1;
        stv[nb] = row < n_rows ? row : -1;
    }
}

// One component for one wave tile: image pointer `im` (global or LDS) -> ln rho stores.
// JB < T evaluates only the first JB output blocks (16 JB rows of y): the stored value c_k - q_JB / 2 is then an
// upper bound of ln rho (q_JB <= q), which the pruned E-step (below) uses to discard components.
// `rows[nb]` = this lane's sample row (or -1: no store).
// The tile pairs of blocks 0 .. JB-1 are the first tri_pairs(JB) of the image; BOFF = offset of the bias block.
template <int NB, typename XT, int JB, int BOFF, typename ImgPtr>
__device__ __forceinline__ void estep_component(ImgPtr im, const XT (&xr)[NB][JB][4], double ck, int lane, int g,
                                                const int64_t (&rows)[NB], double* __restrict__ lnrho_k) {
    constexpr int P = BOFF / 256;
    typedef double d2 __attribute__((ext_vector_type(2)));
    d4 acc[JB][NB];
#pragma unroll
    for (int jt = 0; jt < JB; ++jt) {
        const d2 b01 = *reinterpret_cast<const d2*>(im + P * 256 + (jt * 4 + g) * 4);
        const d2 b23 = *reinterpret_cast<const d2*>(im + P * 256 + (jt * 4 + g) * 4 + 2);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[jt][nb] = d4{b01[0], b01[1], b23[0], b23[1]};
    }
#pragma unroll
    for (int jt = 0; jt < JB; ++jt) {
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
        for (int jt = 0; jt < JB; ++jt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) q = fma(acc[jt][nb][r], acc[jt][nb][r], q);
        }
        q = sum_groups(q);
        if (g == 0 && rows[nb] >= 0) lnrho_k[rows[nb]] = ck - 0.5 * q;
    }
}

// One feature tile (JB = 1): the quadratic form itself, summed over the lane groups (every lane of sample n gets it) - for
// callers that keep the values in registers (hmm.h: hmm_emission_mfma16_kernel).  Same operations as estep_component; the
// operands of a component are loaded apart from their use so that the caller can request a component ahead.
struct Comp16 {
    double a[4], b[4];
};
__device__ __forceinline__ Comp16 load_component16(const double* __restrict__ im, int lane, int g) {
    typedef double d2 __attribute__((ext_vector_type(2)));
    const d2 a01 = *reinterpret_cast<const d2*>(im + lane * 2);
    const d2 a23 = *reinterpret_cast<const d2*>(im + 128 + lane * 2);
    const d2 b01 = *reinterpret_cast<const d2*>(im + 256 + g * 4);
    const d2 b23 = *reinterpret_cast<const d2*>(im + 256 + g * 4 + 2);
    return Comp16{{a01[0], a01[1], a23[0], a23[1]}, {b01[0], b01[1], b23[0], b23[1]}};
}
template <int NB, typename XT>
__device__ __forceinline__ void component16_mfma(const Comp16& c, const XT (&xr)[NB][1][4], d4 (&acc)[NB]) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[nb] = d4{c.b[0], c.b[1], c.b[2], c.b[3]};
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[nb] = mfma_f64(c.a[s], (double)xr[nb][0][s], acc[nb]);
}
template <int NB>
__device__ __forceinline__ void component16_q(const d4 (&acc)[NB], double (&q)[NB]) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        double t = 0.0;
#pragma unroll
        for (int r = 0; r < 4; ++r) t = fma(acc[nb][r], acc[nb][r], t);
        q[nb] = sum_groups(t);
    }
}

// The same with an early way out for candidates that turn out irrelevant: after the first J1 output blocks (J1 (J1 + 1) / 2
// of the JB (JB + 1) / 2 tile pairs) the partial sum q_J1 <= q already bounds ln rho from above; if for EVERY row of the wave
// tile  c_k - q_J1 / 2 < thr[row]  (the row's relevance threshold: its best exact value - 80 ln 2, written by the
// selection kernels), the remaining blocks are skipped and the bound is stored instead of the value.  Whoever reads the
// array treats a stored value below thr[row] as a bound (records.h, rec_finish_kernel).  Pairs that are evaluated in
// full go through exactly the same operations as in estep_component.
template <int NB, typename XT, int JB, int J1, int BOFF, typename ImgPtr>
__device__ __forceinline__ bool estep_component_exit(ImgPtr im, const XT (&xr)[NB][JB][4], double ck, int lane, int g,
                                                     const int64_t (&rows)[NB], double* __restrict__ lnrho_k,
                                                     const float (&thv)[NB] /*thr[row] of the tile's rows*/, float margin) {
    static_assert(J1 >= 1 && J1 < JB, "the way out lies strictly inside the block loop");
    constexpr int P = BOFF / 256;
    typedef double d2 __attribute__((ext_vector_type(2)));
    float th[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) th[nb] = rows[nb] >= 0 ? thv[nb] - margin : __builtin_huge_valf();
    d4 acc[JB][NB];
#pragma unroll
    for (int jt = 0; jt < JB; ++jt) {
        const d2 b01 = *reinterpret_cast<const d2*>(im + P * 256 + (jt * 4 + g) * 4);
        const d2 b23 = *reinterpret_cast<const d2*>(im + P * 256 + (jt * 4 + g) * 4 + 2);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[jt][nb] = d4{b01[0], b01[1], b23[0], b23[1]};
    }
    auto blocks = [&](auto lo, auto hi) {
#pragma unroll
        for (int jt = decltype(lo)::value; jt < decltype(hi)::value; ++jt) {
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
    };
    blocks(std::integral_constant<int, 0>{}, std::integral_constant<int, J1>{});
    double q[NB];
    bool out = true;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        q[nb] = 0.0;
#pragma unroll
        for (int jt = 0; jt < J1; ++jt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) q[nb] = fma(acc[jt][nb][r], acc[jt][nb][r], q[nb]);
        }
        out = out && (ck - 0.5 * sum_groups(q[nb]) < (double)th[nb]);          // NaN: stays in
    }
    if (__all(out)) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const double qs = sum_groups(q[nb]);
            if (g == 0 && rows[nb] >= 0) lnrho_k[rows[nb]] = ck - 0.5 * qs;
        }
        return true;                   // (the whole wave tile took the way out)
    }
    blocks(std::integral_constant<int, J1>{}, std::integral_constant<int, JB>{});
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        double qq = q[nb];
#pragma unroll
        for (int jt = J1; jt < JB; ++jt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) qq = fma(acc[jt][nb][r], acc[jt][nb][r], qq);
        }
        qq = sum_groups(qq);
        if (g == 0 && rows[nb] >= 0) lnrho_k[rows[nb]] = ck - 0.5 * qq;
    }
    return false;
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
        int64_t ld[NB], stv[NB];
        tile_rows<NB>(n0, n, n_rows, ld, stv);
        XT xr[NB][T][4];
        load_x_tile<T, NB, XT, VEC>(x, ldx, D, ld, g, xr);
        for (int k = 0; k < K; ++k)
            estep_component<NB, XT, T, tri_pairs(T) * 256>(img + (int64_t)k * IMG, xr, cvec[k], lane, g, stv,
                                                           lnrho + (int64_t)k * npad);
    }
}

// ---- one feature tile (D <= 16) on the vector ALU (round 4) -------------------------------------------------------------
// With a single 16 x 16 tile the matrix pipe cannot skip the upper triangle of U_k: estep_mfma_f64 spends 4 MFMAs = 8192
// flop per 16 rows and component where 2 x 136 multiply-adds are needed, and it runs AT the f64 matrix peak (HMM config 5:
// 3.1 ms for K = 32, T = 1e7).  The vector ALU has the same f64 peak on gfx950 (one v_fma_f64 per lane every 4 cycles) and
// takes the triangle as it is: a lane is a row, x_n sits in 16 registers, and every multiplier - U_k[j][i], the bias
// -(U_k m_k)[j], c_k - is uniform over the wave (scalar loads from a packed lower-triangular image, 160 doubles per
// component: row j at j (j + 1) / 2, the bias at 136).  152 v_fma_f64 per row and component instead of the MFMAs'
// 256-equivalent, and 512-byte coalesced stores (a lane per row) instead of 128-byte ones.  Same formulation as the MFMA
// kernels (y = U x - U m, q = sum y^2); the order of the additions inside a row of U differs from the matrix pipe's, so the
// values differ from estep_mfma_f64's by rounding (the parity tests hold both to the oracle).
constexpr int kTriImg = 160;                 // doubles per component of the packed image (136 + 16, padded)

static __global__ void pack_tri16_kernel(const double* __restrict__ u, const double* __restrict__ m, int K, int D,
                                         double* __restrict__ tri /*[K][kTriImg]*/) {
    const int k = blockIdx.x;
    const double* uk = u + (int64_t)k * D * D;
    const double* mk = m + (int64_t)k * D;
    double* out = tri + (int64_t)k * kTriImg;
    for (int e = threadIdx.x; e < kTriImg; e += blockDim.x) {
        double v = 0.0;
        if (e < 136) {
            int j = 0;
            while ((j + 1) * (j + 2) / 2 <= e) ++j;
            const int i = e - j * (j + 1) / 2;
            if (j < D && i < D) v = uk[(int64_t)j * D + i];
        } else if (e < 152) {
            const int j = e - 136;
            double sacc = 0.0;
            if (j < D)
                for (int i = 0; i <= j; ++i) sacc = fma(uk[(int64_t)j * D + i], mk[i], sacc);      // (as pack_params_kernel)
            v = -sacc;
        }
        out[e] = v;
    }
}

// One row per lane.  What bounds the kernel is the delivery of the multipliers, not the multiply-adds: a uniform multiplier
// is 8 bytes per v_fma_f64 and SIMD, and the scalar cache hands a CU about 2 bytes per cycle (19 s_load_dwordx16 per
// component and wave: 6 GB through the scalar caches per pass at HMM config 5) - 2.7 ms where the multiply-adds alone would
// take 1.3.  Measured and dropped (profiles/r4_experiments.md): several rows per lane (the compiler then requests all of a
// component's multipliers at its top - 300 SGPRs, spilled to vector lanes and read back with a v_readlane per multiply-add),
// the same with hand-placed s_load / s_waitcnt (scalar loads return out of order, so one unit of look-ahead is all
// lgkmcnt(0) allows: 3.5 ms), and multipliers broadcast from LDS (the same 2 useful bytes per cycle).
// || U_k (x - m_k) ||^2 of one row from a packed component image (pack_tri16_kernel); t uniform: scalar loads
__device__ __forceinline__ double rows16_quadratic(const double* __restrict__ t, const double (&xr)[16]) {
    double q = 0.0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        double y = t[136 + j];
#pragma unroll
        for (int i = 0; i <= j; ++i) y = fma(t[j * (j + 1) / 2 + i], xr[i], y);
        q = fma(y, y, q);
    }
    return q;
}

// a row of x (16 features, zero-padded) as doubles
template <typename XT, bool VEC>
__device__ __forceinline__ void rows16_load(const XT* __restrict__ xp, int D, double (&xr)[16]) {
    if constexpr (VEC) {
        typedef XT v4 __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const v4 v = *reinterpret_cast<const v4*>(xp + 4 * b);
#pragma unroll
            for (int e = 0; e < 4; ++e) xr[4 * b + e] = (double)v[e];
        }
    } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) xr[i] = i < D ? (double)xp[i] : 0.0;
    }
}

template <typename XT, bool VEC>
__global__ __launch_bounds__(256) void estep_rows16_f64(const XT* __restrict__ x, int64_t ldx, int64_t n_rows, int D,
                                                        const double* __restrict__ tri /*[K][kTriImg]*/,
                                                        const double* __restrict__ cvec, int K,
                                                        double* __restrict__ lnrho /*[K][npad]*/, int64_t npad) {
    const int64_t n_tiles = (n_rows + 255) / 256;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t n = tile * 256 + threadIdx.x;
        const int64_t row = n < n_rows ? n : n_rows - 1;
        double xr[16];
        rows16_load<XT, VEC>(x + row * ldx, D, xr);
        for (int k = 0; k < K; ++k) {
            const double q = rows16_quadratic(tri + (int64_t)k * kTriImg, xr);
            if (n < n_rows) lnrho[(int64_t)k * npad + n] = cvec[k] - 0.5 * q;
        }
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
        int64_t ld[NB], stv[NB];
        tile_rows<NB>(n0, n, n_rows, ld, stv);
        XT xr[NB][T][4];
        load_x_tile<T, NB, XT, VEC>(x, ldx, D, ld, g, xr);
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
                estep_component<NB, XT, T, tri_pairs(T) * 256>(buf + kk * IMG, xr, cvec[k], lane, g, stv,
                                                               lnrho + (int64_t)k * npad);
            }
            // the prefetched block must have landed, and every wave must be done reading this one
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }
}

// ---- pruned E-step -----------------------------------------------------------------------------------------------
// Once the responsibilities are sparse, almost every (sample, component) pair only has to be shown irrelevant
// (r_nk < 2^-80: it changes neither lse_n nor, mstep.h, any statistic beyond the last bit).  The bounds that show it come
// from the int8 matrix pipe (estep_i8.h: a bound pass over all pairs, the proof round over listed pairs) and are carried
// from pass to pass (records.h); what cannot be shown irrelevant is evaluated exactly by the gather kernel below
// (component fixed per workgroup, sample rows gathered through per-component lists).  The results of the exact pairs do not
// depend on list order (one MFMA column per sample); the lists are filled without atomics, in a fixed order (aux_kernels.h).
// Exact ln rho for listed (sample, component) pairs: a component's image is staged once and kept while a workgroup works
// through that component's list, 8 waves x 16 NB x kGatherTiles entries per chunk.
constexpr int kGatherTiles = 8;

// The work distribution is read from the device (no host knowledge of the list lengths):
// plan[k] = index of component k's first chunk of kGatherTiles x 8 waves x 16 NB list entries, plan[K] = total
// (gather_plan_kernel, records.h).  A fixed grid of persistent workgroups takes contiguous runs of chunks, restaging
// the component image only when the component changes.
// EXIT: candidates of a selection round - rows carry a relevance threshold thr[row], see estep_component_exit.
template <int T, typename XT, bool VEC, bool EXIT>
__global__ __launch_bounds__(512) void estep_gather_dev_f64(const XT* __restrict__ x, int64_t ldx, int D,
                                                            const double* __restrict__ img, const double* __restrict__ cvec,
                                                            int K, const int* __restrict__ lists /*[K][cap]*/, int64_t cap,
                                                            const int* __restrict__ counts /*[K]*/,
                                                            const int* __restrict__ plan /*[K + 1]*/,
                                                            double* __restrict__ lnrho, int64_t npad,
                                                            const float* __restrict__ thr,
                                                            unsigned long long* __restrict__ exits /*pairs that took the way out*/,
                                                            float exit_margin /*nats the partial bound must lie below thr*/) {
    constexpr int NW = 8;
    constexpr int NB = estep_nb_w<XT>(T, NW);
    constexpr int IMG = img_doubles(T);
    constexpr int CHUNK = NW * 16 * NB * kGatherTiles;
    __shared__ __attribute__((aligned(16))) double smem[IMG];
    __shared__ int s_first[257];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = lane & 15, g = lane >> 4;
    for (int k = threadIdx.x; k <= K; k += 512) s_first[k] = plan[k];
    __syncthreads();
    const int total = s_first[K];
    const int per = (total + (int)gridDim.x - 1) / (int)gridDim.x;
    const int c0 = (int)blockIdx.x * per;
    const int c1 = c0 + per < total ? c0 + per : total;
    int k = 0, kcur = -1;
    unsigned long long my_exits = 0;                     // pairs of this wave that took the early way out (statistics)
    for (int c = c0; c < c1; ++c) {
        while (s_first[k + 1] <= c) ++k;                 // component of chunk c (empty components are skipped)
        if (k != kcur) {
            __syncthreads();                             // every wave is done with the previous image
            const double* src = img + (int64_t)k * IMG + lane * 2;
            for (int piece = wave; piece < IMG / 128; piece += NW)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + piece * 128),
                                                 (__attribute__((address_space(3))) void*)(&smem[piece * 128]), 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            kcur = k;
        }
        const int count = counts[k];
        const int64_t chunk0 = (int64_t)(c - s_first[k]) * CHUNK;
        const int* list = lists + (int64_t)k * cap;
        const double ck = cvec[k];
        double* out = lnrho + (int64_t)k * npad;
        // Software pipeline over the chunk's tiles (round 4): while tile t is computed, the rows of tile t + 1 are in flight
        // (second register set) and the list entries of tile t + 2 have been requested - before, a wave's list -> rows -> MFMA
        // chain was serial and only the other waves of the SIMD covered it.  Every iteration issues the same loads (tiles
        // past the list's end repeat its last entry and skip the arithmetic), so the compiler's in-order vmcnt stays exact.
        auto rows_of = [&](int t, int64_t (&ld)[NB], int64_t (&stv)[NB]) {
            const int64_t e0 = chunk0 + ((int64_t)t * NW + wave) * 16 * NB;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const int64_t e = e0 + 16 * nb + n;
                const int row = list[e < count ? e : count - 1];
                ld[nb] = row;
                stv[nb] = e < count ? row : -1;
            }
        };
        auto live = [&](int t) { return chunk0 + ((int64_t)t * NW + wave) * 16 * NB < count; };
        auto thr_of = [&](const int64_t (&ld)[NB], float (&tv)[NB]) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) tv[nb] = EXIT ? thr[ld[nb]] : 0.0f;
        };
        auto compute = [&](int t, const XT (&xr)[NB][T][4], const int64_t (&stv)[NB], const float (&tv)[NB]) {
            if (!live(t)) return;                      // (wave-uniform)
            if constexpr (EXIT) {
                if (estep_component_exit<NB, XT, T, T / 2, tri_pairs(T) * 256>(smem, xr, ck, lane, g, stv, out, tv, exit_margin)) {
                    const int64_t left = count - (chunk0 + ((int64_t)t * NW + wave) * 16 * NB);
                    my_exits += (unsigned long long)(left < 16 * NB ? left : 16 * NB);
                }
            } else
                estep_component<NB, XT, T, tri_pairs(T) * 256>(smem, xr, ck, lane, g, stv, out);
        };
        static_assert(kGatherTiles % 2 == 0, "the tile loop is unrolled by two register sets");
        int64_t ld0[NB], st0[NB], ld1[NB], st1[NB], ld2[NB], st2[NB], ld3[NB], st3[NB];
        XT xa[NB][T][4], xb[NB][T][4];
        float ta[NB], tb[NB];
        rows_of(0, ld0, st0);
        rows_of(1, ld1, st1);
        load_x_tile<T, NB, XT, VEC>(x, ldx, D, ld0, g, xa);
        thr_of(ld0, ta);
        for (int t = 0; t < kGatherTiles; t += 2) {
            rows_of(t + 2, ld2, st2);
            load_x_tile<T, NB, XT, VEC>(x, ldx, D, ld1, g, xb);
            thr_of(ld1, tb);
            compute(t, xa, st0, ta);
            rows_of(t + 3, ld3, st3);
            load_x_tile<T, NB, XT, VEC>(x, ldx, D, ld2, g, xa);
            thr_of(ld2, ta);
            compute(t + 1, xb, st1, tb);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                ld0[nb] = ld2[nb];
                st0[nb] = st2[nb];
                ld1[nb] = ld3[nb];
                st1[nb] = st3[nb];
            }
        }
    }
    // one atomic per wave (an integer counter: order-free; one per exit would serialise 5e5 of them on one address)
    if (EXIT && lane == 0 && my_exits != 0ull) atomicAdd(exits, my_exits);
}

}  // namespace gmmvb
