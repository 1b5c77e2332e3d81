// Shared device helpers for the gfx950 GMM-VB kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gmmvb {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

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
