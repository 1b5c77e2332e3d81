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

// Partner values across the wave's 16-lane groups without LDS: gfx950's v_permlane16_swap exchanges the odd groups of its
// first operand with the even groups of its second, v_permlane32_swap the upper half of the first with the lower half of
// the second.  Swapping a value WITH ITSELF leaves (own, partner) pairs: after the 16-swap, first = the even group's value
// and second = the odd group's in both groups of a pair; after the 32-swap, first = the lower half's and second = the
// upper half's in both halves.  A reduction over the four groups is then two swaps per dword and one operation per
// level, all on the vector ALU (__shfl_xor is a ds_bpermute: an LDS round trip per level and dword; every MFMA epilogue
// of the E-step kernels ends in this reduction).  a + b = b + a exactly, so the sums are the bits the shuffles gave.
#ifndef GMMVB_GROUPS_BY_SHUFFLE
template <bool HALVES>
__device__ __forceinline__ void swap_pairs(double v, double& first, double& second) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = (unsigned)b, hi = (unsigned)(b >> 32);
    unsigned f0, f1, s0, s1;
    if constexpr (HALVES) {
        const auto x = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
        const auto y = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
        f0 = x[0], s0 = x[1], f1 = y[0], s1 = y[1];
    } else {
        const auto x = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
        const auto y = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        f0 = x[0], s0 = x[1], f1 = y[0], s1 = y[1];
    }
    first = __longlong_as_double((long long)(((unsigned long long)f1 << 32) | f0));
    second = __longlong_as_double((long long)(((unsigned long long)s1 << 32) | s0));
}
// the hardware exchange itself on two doubles: (odd groups of `first`) <-> (even groups of `second`), or with HALVES
// (upper half of `first`) <-> (lower half of `second`)
template <bool HALVES>
__device__ __forceinline__ void swap_two(double& first, double& second) {
    const unsigned long long a = (unsigned long long)__double_as_longlong(first), b = (unsigned long long)__double_as_longlong(second);
    unsigned f0, f1, s0, s1;
    if constexpr (HALVES) {
        const auto x = __builtin_amdgcn_permlane32_swap((unsigned)a, (unsigned)b, false, false);
        const auto y = __builtin_amdgcn_permlane32_swap((unsigned)(a >> 32), (unsigned)(b >> 32), false, false);
        f0 = x[0], s0 = x[1], f1 = y[0], s1 = y[1];
    } else {
        const auto x = __builtin_amdgcn_permlane16_swap((unsigned)a, (unsigned)b, false, false);
        const auto y = __builtin_amdgcn_permlane16_swap((unsigned)(a >> 32), (unsigned)(b >> 32), false, false);
        f0 = x[0], s0 = x[1], f1 = y[0], s1 = y[1];
    }
    first = __longlong_as_double((long long)(((unsigned long long)f1 << 32) | f0));
    second = __longlong_as_double((long long)(((unsigned long long)s1 << 32) | s0));
}
// Four values per lane, each to be summed over the four lane groups, the total of p_g wanted in group g only: a butterfly
// of three exchanges and three additions (the four separate sum_groups take eight of each), in sum_groups' order
// ((g0 + g1) + (g2 + g3)), so the totals are the same bits.
__device__ __forceinline__ double sum_groups_scatter4(double p0, double p1, double p2, double p3) {
    swap_two<false>(p0, p1);          // even groups: p0 of the pair, odd groups: p1 of the pair
    swap_two<false>(p2, p3);
    double a = p0 + p1, b = p2 + p3;
    swap_two<true>(a, b);             // lower half: both halves' a, upper half: both halves' b
    return a + b;
}
#endif

// sum over the four 16-lane groups (lanes l, l^16, l^32, l^48); every lane gets the total
__device__ __forceinline__ double sum_groups(double v) {
#ifdef GMMVB_GROUPS_BY_SHUFFLE
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
#else
    double a, b;
    swap_pairs<false>(v, a, b);
    swap_pairs<true>(a + b, a, b);
    return a + b;
#endif
}
// largest value over the four groups, in all of them
__device__ __forceinline__ double max_groups(double v) {
#ifdef GMMVB_GROUPS_BY_SHUFFLE
    v = fmax(v, __shfl_xor(v, 16));
    return fmax(v, __shfl_xor(v, 32));
#else
    double a, b;
    swap_pairs<false>(v, a, b);
    swap_pairs<true>(fmax(a, b), a, b);
    return fmax(a, b);
#endif
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
