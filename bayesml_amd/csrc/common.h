// Shared device helpers for the gfx950 GMM-VB kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gmmvb {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

// The relevance line: a (row, component) pair with r_nk < 2^-kRelevanceBits is invisible in every f64 sum the reference
// forms over fewer than 2^(kRelevanceBits - 53) terms (the row's log-normaliser over K components, the per-component
// statistics over the rows): such pairs are proven, not evaluated (records.h) and not accumulated (mstep.h).
// 80 bits: 2^27 = 1.3e8 rows - config 4's whole job - before the dropped mass of a component, at most N 2^-80, reaches
// half an ulp of a sum of size one; the f64 sums themselves carry a rounding error of about sqrt(N) 2^-53 (1e-12 at N = 1e8),
// ten thousand times that.  Rounds 1-3 drew the line at 2^-100; measured on one box (profiles/r4_experiments.md): the
// benchmark step 6.94 -> 6.71 ms, config 4's shard 17.8 -> 15.3 ms, cluster spread 1.0 (where 62-70 % of the non-dominant
// active pairs sat between the two lines) 23.1 -> 9.5 ms.  -DGMMVB_RELEVANCE_BITS=n builds a variant (tools/build_variant.sh).
#ifndef GMMVB_RELEVANCE_BITS
#define GMMVB_RELEVANCE_BITS 80
#endif
static_assert(GMMVB_RELEVANCE_BITS >= 64 && GMMVB_RELEVANCE_BITS <= 400, "below 2^-64 a pair can reach the sums' own rounding");
constexpr int kRelevanceBits = GMMVB_RELEVANCE_BITS;
constexpr double kRelevanceNats = kRelevanceBits * 0.69314718055994530942;

constexpr int kTile = 16;      // v_mfma_f64_16x16x4_f64 output tile
constexpr int kWave = 64;      // CDNA wavefront
constexpr int kMaxTiles = 8;   // D <= 128

__host__ __device__ constexpr int tri_pairs(int t) { return t * (t + 1) / 2; }
// index of the tile pair (hi, lo) with lo <= hi in the packed lower triangle
__host__ __device__ constexpr int pair_index(int hi, int lo) { return hi * (hi + 1) / 2 + lo; }

// D = A(16x4) * B(4x16) + C, f64.  Lane l supplies A[l&15][l>>4] and B[l>>4][l&15]; it receives
// C[(l>>4) + 4*r][l&15] in element r (guide: "f64 MFMA does NOT use the f32 row map").
__device__ __forceinline__ d4 mfma_f64(double a, double b, d4 c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// sum over the four 16-lane groups (lanes l, l^16, l^32, l^48); every lane gets the total
__device__ __forceinline__ double sum_groups(double v) {
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}

__device__ __forceinline__ double sum_wave(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// Slab written by one M-step wave for one (row split, component):
//   [ P tile pairs x (4 regs x 64 lanes) | a[16 T] | ns | h | pad ]
__host__ __device__ constexpr int slab_len(int t) { return tri_pairs(t) * 256 + 16 * t + 16; }

}  // namespace gmmvb
