// What the record kernels (records.h), the selection kernels (aux_kernels.h) and the stateless sweep (project.h) share:
// the record layout, directed roundings, the slot-selection keys and the per-block counting idiom.  Only inline device
// functions and types - includable from several translation units.
#pragma once
#include "common.h"

namespace gmmvb {

constexpr int kSelRows = 256;
// the per-block count / base arrays are block-major (see the scan below)
__device__ __forceinline__ int64_t blk_at(int k, int64_t b, int K) { return b * K + k; }

// OR over the wave's 64 lanes (all of them active), as a wave-UNIFORM value in scalar registers: four v_or_b32 with a DPP
// operand inside the rows of 16 lanes, the four row results through v_readlane.  (Round 5.  The shuffle butterfly it replaces
// cost twelve LDS permutes and left the result in vector registers - the loops over the set bits that follow then ran their
// control flow, their bit scans and the list addresses that depend on the component on the vector ALU, lane by lane the same.)
__device__ __forceinline__ unsigned wave_or32(unsigned v) {
    asm(
        "s_nop 1\n\t"
        "v_or_b32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_or_b32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_or_b32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_or_b32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(v));
    return (unsigned)__builtin_amdgcn_readlane((int)v, 0) | (unsigned)__builtin_amdgcn_readlane((int)v, 16) |
           (unsigned)__builtin_amdgcn_readlane((int)v, 32) | (unsigned)__builtin_amdgcn_readlane((int)v, 48);
}
__device__ __forceinline__ unsigned long long wave_or(unsigned long long v) {
    return ((unsigned long long)wave_or32((unsigned)(v >> 32)) << 32) | wave_or32((unsigned)v);
}
// the shuffle butterfly (result in vector registers): rec_finish_kernel is faster with it (measured: +13 % with the form above)
__device__ __forceinline__ unsigned long long wave_or_shfl(unsigned long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v |= __shfl_xor(v, o);
    return v;
}

constexpr int kRecSlots = 8;
constexpr unsigned short kRecEmpty = 0xFFFF;
constexpr unsigned short kRecListed = 0x4000, kRecExactBit = 0x8000, kRecCompMask = 0x3FFF;
constexpr double kRelNats = kRelevanceNats;      // kRelevanceBits ln 2 (common.h)

struct RecArrays {
    unsigned short* k;      // [C][npad] component of slot j (kRecEmpty: unused)
    float* d;               // [C][npad] lower bound of the whitened distance, rounded towards zero
    float* B;               // [npad] upper bound of ln rho for every component without a slot (-inf: there is none)
    unsigned char* exact;   // [npad] bit j: d_j is the distance itself: d_j <= dist <= d_j (1 + 2^-22)
    unsigned char* sel;     // [npad] bit j: slot j was listed for exact evaluation in the current pass
    unsigned char* flags;   // [npad] bit 0: overflow row of the current pass (all K pairs evaluated exactly);
                            //        bit 1: refreshed row (components without a slot were listed too)
    int64_t npad;
};

__device__ __forceinline__ float f32_down(double v) { return __double2float_rd(v); }      // v >= 0: towards zero
__device__ __forceinline__ float f32_up(double v) { return __double2float_ru(v); }

// whitened distance (lower bound if v is an upper bound of ln rho) from a stored value; NaN -> 0 (always a candidate)
__device__ __forceinline__ double dist_of(double c, double v) {
    const double q = 2.0 * (c - v);
    return q > 0.0 ? sqrt(q) : 0.0;
}
// the same as a LOWER bound in f32 (directed rounding all the way: 1e-7 relative looseness), for values that are
// bounds anyway - the f64 square root is a quarter-rate instruction and the sweep does K of them per row
__device__ __forceinline__ float dist_lower_f32(double c, double v) {
    const float q = __double2float_rd(2.0 * (c - v));
    return q > 0.0f ? sqrtf(q) * (1.0f - 2.4e-7f) : 0.0f;       // two ulps below whatever rounding sqrtf has
}

// per-block component counts of a 64-bit mask word (as in select_mask_kernel)
template <bool UNIFORM = true>
__device__ __forceinline__ void count_word(unsigned long long mk, int w, int wave, int (*wcnt)[256]) {
    unsigned long long present = UNIFORM ? wave_or(mk) : wave_or_shfl(mk);
    while (present) {
        const int b = __builtin_ctzll(present);
        present &= present - 1;
        const int c = __builtin_popcountll(__ballot((mk >> b) & 1ull));
        if ((threadIdx.x & 63) == 0) wcnt[wave][64 * w + b] = c;
    }
}

// (the slot-selection keys of the sweeps: records.h, "The carried E-step")
__device__ __forceinline__ unsigned sweep_key(float ub, unsigned k) {
    const unsigned bits = __float_as_uint(ub);
    const unsigned inv = bits ^ (~(unsigned)((int)bits >> 31) & 0x7FFFFFFFu);        // descending in ub, exact
    return (inv & 0xFFFFFF00u) | k;
}
__device__ __forceinline__ float sweep_key_bound(unsigned key) {                     // >= the bound the key was made of
    const unsigned inv = key & 0xFFFFFF00u;
    return __uint_as_float(inv ^ (~(unsigned)((int)inv >> 31) & 0x7FFFFFFFu));
}
__device__ __forceinline__ void sweep_chain(unsigned (&s)[kRecSlots + 1], unsigned key) {
#pragma unroll
    for (int j = 0; j < kRecSlots; ++j) {
        const unsigned lo = min(s[j], key);
        key = max(s[j], key);
        s[j] = lo;
    }
    s[kRecSlots] = min(s[kRecSlots], key);
}

// (records.h, "A settled row's reference": components that moved by more than this get a fresh own-pair bound first)
__device__ __forceinline__ bool own_first(double big_gamma, double delta) { return big_gamma > 1.004 || delta > 0.04; }

}  // namespace gmmvb
