// Dense E-step for 128 < D <= 256 (9 .. 16 feature tiles): ln rho_nk = c_k - 0.5 || U_k (x_n - m_k) ||^2 on f64 MFMA with the
// parameter image streamed through LDS one group of output block rows at a time.
//
// Reference: bayesml/gaussianmixture/_gaussianmixture.py:773-781 (the reference accepts any c_degree, :433).  estep_lds_f64
// (estep.h) keeps a component's whole image in LDS and all T accumulator tiles of a sample tile in registers: at T = 16 the
// image is 274 KB (LDS: 160 KB) and the accumulators 128 registers.  Neither is needed at once: output block row jt of
// y = U x - U m only needs U's tile pairs (jt, 0 .. jt) - contiguous in the image - and its 16 x 16 accumulator can be squared
// and added to ||y||^2 as soon as it is complete.  So the image goes through LDS in pieces of whole block rows, at most 32
// tile pairs (64 KB) each, double-buffered across pieces and components by LDS-DMA (a piece ahead), and a wave keeps ONE
// accumulator tile plus its x tile (16 T floats per lane).  Same image layout, same MFMA order per block row, same
// reduction as estep_component: the values differ from a hypothetical T > 8 instance of that kernel by nothing.
// Eight waves per workgroup share a piece; one barrier per piece.
#pragma once
#include <type_traits>
#include <utility>
#include "estep.h"

namespace gmmvb {

constexpr int kRowsPiecePairs = 32;       // tile pairs per LDS piece: 64 KB, two pieces resident

// first block row of piece `pc` (pieces are maximal runs of whole block rows with at most kRowsPiecePairs tile pairs)
__host__ __device__ constexpr int rows_piece_begin(int T, int pc) {
    int r = 0;
    for (int q = 0; q < pc && r < T; ++q) {
        int pairs = 0;
        while (r < T && pairs + (r + 1) <= kRowsPiecePairs) {
            pairs += r + 1;
            ++r;
        }
    }
    return r;
}
__host__ __device__ constexpr int rows_pieces(int T) {
    int n = 0;
    while (rows_piece_begin(T, n) < T) ++n;
    return n;
}

template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

template <int T, typename XT, bool VEC>
__global__ __launch_bounds__(512) void estep_rows_f64(const XT* __restrict__ x, int64_t ldx, int64_t n_rows, int D,
                                                      const double* __restrict__ img /*[K][IMG]*/,
                                                      const double* __restrict__ cvec, int K,
                                                      double* __restrict__ lnrho /*[K][npad]*/, int64_t npad) {
    static_assert(T > 8 && T <= 16, "up to eight tiles: estep_lds_f64");
    constexpr int NW = 8;
    constexpr int IMG = img_doubles(T);
    constexpr int P = tri_pairs(T);
    constexpr int NP = rows_pieces(T);
    typedef double d2 __attribute__((ext_vector_type(2)));
    __shared__ __attribute__((aligned(16))) double smem[2][kRowsPiecePairs * 256];      // the ONLY LDS object of the kernel
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = lane & 15, g = lane >> 4;
    const int64_t rows_per_wg = NW * 16;
    const int64_t n_wg_tiles = (n_rows + rows_per_wg - 1) / rows_per_wg;

    // global -> LDS copy of piece pc of component k: 1-KB pieces (64 lanes x 16 B), lane-linear on both sides
    auto stage = [&](int k, int pc_first_pair, int pc_pairs, int buf) {
        const int pieces = pc_pairs * 2;
        const double* src = img + (int64_t)k * IMG + (int64_t)pc_first_pair * 256 + lane * 2;
        for (int piece = wave; piece < pieces; piece += NW)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + piece * 128),
                                             (__attribute__((address_space(3))) void*)(&smem[buf][piece * 128]), 16, 0, 0);
    };
    constexpr int kFirstPairs = pair_index(rows_piece_begin(T, 1), 0);                   // tile pairs of piece 0

    for (int64_t wt = blockIdx.x; wt < n_wg_tiles; wt += gridDim.x) {
        const int64_t n0 = wt * rows_per_wg + (int64_t)wave * 16;      // may lie past n_rows: rows clamp, stores mask
        int64_t ld[1], stv[1];
        tile_rows<1>(n0, n, n_rows, ld, stv);
        XT xr[1][T][4];
        load_x_tile<T, 1, XT, VEC>(x, ldx, D, ld, g, xr);
        int u = 0;                                                     // pieces gone through: the parity picks the buffer
        stage(0, 0, kFirstPairs, 0);
        for (int k = 0; k < K; ++k) {
            double q = 0.0;
            const double* bias = img + (int64_t)k * IMG + P * 256;
            static_for<NP>([&](auto PC) {
                constexpr int pc = decltype(PC)::value;
                constexpr int j0 = rows_piece_begin(T, pc), j1 = rows_piece_begin(T, pc + 1);
                constexpr int p0 = pair_index(j0, 0);
                // this piece has landed (every wave's share), and every wave is done reading the other buffer
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if constexpr (pc + 1 < NP) {
                    constexpr int q0 = pair_index(j1, 0), q1 = pair_index(rows_piece_begin(T, pc + 2), 0);
                    stage(k, q0, q1 - q0, (u + 1) & 1);
                } else {
                    if (k + 1 < K) stage(k + 1, 0, kFirstPairs, (u + 1) & 1);
                }
                const double* buf = smem[u & 1];
#pragma unroll
                for (int jt = j0; jt < j1; ++jt) {
                    const d2 b01 = *reinterpret_cast<const d2*>(bias + (jt * 4 + g) * 4);
                    const d2 b23 = *reinterpret_cast<const d2*>(bias + (jt * 4 + g) * 4 + 2);
                    d4 acc = d4{b01[0], b01[1], b23[0], b23[1]};
#pragma unroll
                    for (int b = 0; b <= jt; ++b) {
                        const int p = pair_index(jt, b) - p0;
                        const d2 a01 = *reinterpret_cast<const d2*>(buf + p * 256 + lane * 2);
                        const d2 a23 = *reinterpret_cast<const d2*>(buf + p * 256 + 128 + lane * 2);
                        const double a[4] = {a01[0], a01[1], a23[0], a23[1]};
#pragma unroll
                        for (int s = 0; s < 4; ++s) acc = mfma_f64(a[s], (double)xr[0][b][s], acc);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) q = fma(acc[r], acc[r], q);
                }
                ++u;
            });
            q = sum_groups(q);
            if (g == 0 && stv[0] >= 0) lnrho[(int64_t)k * npad + stv[0]] = cvec[k] - 0.5 * q;
        }
        __syncthreads();              // (the next tile's first piece overwrites a buffer the last piece may still be read from)
    }
}

}  // namespace gmmvb
