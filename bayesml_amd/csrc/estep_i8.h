// E-step on the int8 matrix pipe: ln rho_nk = c_k - 0.5 * || U_k (x_n - m_k) ||^2 with both operands cut into
// 7-bit signed digits and the products summed exactly in int32 (v_mfma_i32_32x32x32_i8, 32 cycles per
// 32x32x32 block = 8x the f64 MFMA's multiply rate).  Same function, same outputs as estep.h; the arithmetic is
// fixed point per row of U_k and per sample:
//
//   U_k[j][:]      = 2^ej * sum_a dU_a[j][:] 2^(-6-7a)      a = 0..5, dU_a in [-64, 64]  (ej: exponent of the row max)
//   x_n - pivot    = 2^en * sum_b dX_b[n][:] 2^(-6-7b)      b = 0..5                       (en: exponent of the sample max)
//   (U_k (x_n - pivot))_j = 2^(ej+en-12) * sum_w 2^(-7w) * [ sum_{a+b=w} dU_a[j][:] . dX_b[n][:] ]      w = 0..5
//
// The bracket is an exact integer (<= 24 MFMAs x 32 x 64 x 64 < 2^22); digit pairs with a + b > 5 are dropped, which
// together with the 42-bit truncation of each operand is an error of about 20 x 2^-42 relative to
// (row max of U) x (sample max of |x - pivot|) per term - 1e-9 absolute on ln rho at the benchmark's scale, 5e-8 on the
// posterior after 10 VB iterations (f64 path: 1e-9), against the 1e-5 contract.  21 int8 MFMAs (672 cycles) replace
// the 64 f64 MFMAs (4096 cycles) of a 32x32x32 block; the lower-triangular U_k skips 6 of the 16 block pairs at D = 128.
//
// Mapping: one wave = 32 samples (columns of the MFMA), all K components.  Lane l = (c = l & 31, h = l >> 5) holds
// the sample's digits for features 32 it + 16 h + (0..15) in byte order for the whole k loop (6 x T32 x 4 VGPRs).
// Output rows land on (register g, h): row = (g & 3) + 8 (g >> 2) + 4 h, so ||y||^2 is a per-lane sum plus one
// cross-half add.
//
// Component image (bytes; 1-KB granules for the LDS-DMA, written by pack_params_i8_kernel):
//   [ pair p = (jt, it <= jt) ][ digit a ][ lane ][ 16 ]   byte e of lane (r, h) = dU_a[32 jt + r][32 it + 16 h + e]
//   [ jt ][ h ][ g ][ 2 ] doubles                         (2^ej, (U_k (m_k - pivot))_j) for row j = 32 jt + (g&3) + 8(g>>2) + 4h
#pragma once
#include "common.h"

#include <type_traits>

namespace gmmvb {

typedef int i4v __attribute__((ext_vector_type(4)));
typedef int i16v __attribute__((ext_vector_type(16)));

constexpr int kDigits = 6;        // the E-step proper: 42-bit operands
constexpr int kBoundDigits = 3;   // the pruned E-step's bound pass: 21-bit operands and a rigorous error term
// Image tails.  ND = 6: [jt][h][g][2] doubles (2^ej, b_j).  ND = 3 (bound pass): one exponent per 32-row output block
// (rows are digitised against the block's largest entry - coarser for small rows, and the error term says so), then
//   [jt][h][g] floats b_j   and   [jt][2] floats (2^e_jt, beta_jt)
// so that a lane reads 4 bytes per output row instead of 16 and no per-row scale.

__host__ __device__ constexpr int i8_blocks(int D) { return (D + 31) / 32; }
__host__ __device__ constexpr int i8_img_bytes(int nd, int t32) {
    return (tri_pairs(t32) * nd * 1024 + (nd == kBoundDigits ? t32 * 128 + 32 : t32 * 512) + 1023) / 1024 * 1024;
}
__host__ __device__ constexpr int i8_kb(int nd, int t32) {
    const int kb = (64 * 1024) / i8_img_bytes(nd, t32);
    return kb < 1 ? 1 : (kb > 8 ? 8 : kb);
}
// With ND digits per operand and the digit pairs a + b <= ND - 1 kept, every term U~ x~ (both in (-1, 1)) is off by
// at most 2^(-7 ND) (|x~| + |U~| + 1.02 (ND - 1)): two operand truncations and the dropped pairs.  Over the at most
// 32 T32 terms of a row, times 2^(ej + en): |y_j - y^_j| <= 2^ej 2^en i8_err(ND, T32).
__host__ __device__ constexpr double i8_err(int nd, int t32) {
    double w = 1.0;
    for (int i = 0; i < 7 * nd; ++i) w *= 0.5;
    return w * (2.0 + 1.02 * (nd - 1)) * 32.0 * t32 * 1.000001;
}

// ND balanced base-128 digits of t (|t| <= 64): t = d0 + d1/128 + d2/128^2 + ...; returns them as bytes
template <int ND>
__device__ __forceinline__ void digits_of(double t, int (&d)[ND]) {
#pragma unroll
    for (int a = 0; a < ND; ++a) {
        const double r = __builtin_rint(t);
        d[a] = (int)r;
        t = (t - r) * 128.0;
    }
}

// K-side: digits of u, row exponents, bias.  One block per component.
template <int ND>
__global__ void pack_params_i8_kernel(const double* __restrict__ u, const double* __restrict__ m,
                                      const double* __restrict__ pivot, int K, int D, int T32, int img_bytes,
                                      unsigned char* __restrict__ img) {
    __shared__ double row_scale[128];   // 2^(6 - ej)  (0 for padding rows)
    const int k = blockIdx.x;
    const int P = tri_pairs(T32);
    const double* uk = u + (int64_t)k * D * D;
    const double* mk = m + (int64_t)k * D;
    unsigned char* im = img + (int64_t)k * img_bytes;
    double* consts = reinterpret_cast<double*>(im + P * ND * 1024);
    __shared__ double row_max[128];
    __shared__ double row_beta[128];
    __shared__ int row_bad[128];
    for (int j = threadIdx.x; j < 32 * T32; j += blockDim.x) {
        double mx = 0.0, bias = 0.0, abias = 0.0;
        bool bad = false;
        if (j < D) {
            for (int i = 0; i <= j; ++i) {
                const double v = uk[(int64_t)j * D + i];
                bad |= !(fabs(v) <= 1.7976931348623157e308);
                mx = fmax(mx, fabs(v));
                bias = fma(v, mk[i] - pivot[i], bias);
                abias = fma(fabs(v), fabs(mk[i] - pivot[i]), abias);
            }
        }
        int e = 0;
        if (mx > 0.0) (void)frexp(mx, &e);          // mx = f 2^e, f in [0.5, 1)
        const int jt = j >> 5, w = j & 31;
        const int h = (w >> 2) & 1, g = (w & 3) + 4 * (w >> 3);      // w = (g & 3) + 8 (g >> 2) + 4 h
        if constexpr (ND == kBoundDigits) {
            // b_j in f32; beta_j >= 2^-20 |b_j| + the f64 rounding of b_j covers every f32 rounding of the epilogue
            // that scales with the bias (i8_rows_bound) - the block keeps the largest of its rows' beta_j;
            // +inf = "no bound" (non-finite row or bias)
            const bool wide = bad || !(fabs(bias) < 1e30) || !(abias < 1e30);
            reinterpret_cast<float*>(consts)[(jt * 2 + h) * 16 + g] = wide ? 0.0f : (float)bias;
            row_beta[j] = wide ? __builtin_huge_val() : (fabs(bias) * 9.5367431640625e-7 + abias * 2.9e-14) * 1.0001 + 1e-37;
            row_max[j] = mx;
            row_bad[j] = bad;
        } else {
            row_scale[j] = (mx > 0.0 && !bad) ? ldexp(1.0, 6 - e) : 0.0;
            double* cp = consts + ((jt * 2 + h) * 16 + g) * 2;
            cp[0] = bad ? __builtin_nan("") : (mx > 0.0 ? ldexp(1.0, e) : 0.0);
            cp[1] = bias;
        }
    }
    if constexpr (ND == kBoundDigits) {
        __syncthreads();
        for (int j = threadIdx.x; j < 32 * T32; j += blockDim.x) {
            const int jt = j >> 5;
            double mx = 0.0, beta = 0.0;
            bool bad = false;
            for (int r = 0; r < 32; ++r) {
                mx = fmax(mx, row_max[32 * jt + r]);
                beta = fmax(beta, row_beta[32 * jt + r]);
                bad |= row_bad[32 * jt + r] != 0;
            }
            int e = 0;
            if (mx > 0.0) (void)frexp(mx, &e);
            // f32 products of powers of two stay exact for |e| <= 45; outside (or with a non-finite row) the block
            // gives no bound: scale 0 (all digits 0) and beta +inf
            const bool wide = bad || e < -45 || e > 45;
            row_scale[j] = (mx > 0.0 && !wide) ? ldexp(1.0, 6 - e) : 0.0;
            float* tail = reinterpret_cast<float*>(consts) + T32 * 32;
            if ((j & 31) == 0) {
                tail[2 * jt] = (mx > 0.0 && !wide) ? (float)ldexp(1.0, e) : 0.0f;
                tail[2 * jt + 1] = wide ? __builtin_huge_valf() : (float)(beta * 1.0001);
            }
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < P * 1024; e += blockDim.x) {
        const int p = e >> 10, lane = (e >> 4) & 63, b = e & 15;
        int jt = 0;
        while (tri_pairs(jt + 1) <= p) ++jt;
        const int it = p - tri_pairs(jt);
        const int jj = 32 * jt + (lane & 31), ii = 32 * it + 16 * (lane >> 5) + b;
        const double v = (jj < D && ii <= jj) ? uk[(int64_t)jj * D + ii] * row_scale[jj] : 0.0;
        int d[ND];
        digits_of<ND>(v, d);
#pragma unroll
        for (int a = 0; a < ND; ++a) im[((p * ND + a) * 64 + lane) * 16 + b] = (unsigned char)(d[a] & 0xff);
    }
    for (int e = P * ND * 1024 + (ND == kBoundDigits ? T32 * 128 + 8 * T32 : T32 * 512) + threadIdx.x; e < img_bytes;
         e += blockDim.x)
        im[e] = 0;
}

// Sample digits for one wave tile.  xd[a][it] = 16 bytes = digit a of features 32 it + 16 h + (0..15);
// c2 = 2^(en - 12 - 7 (ND - 1)), the weight of the last kept digit pair class (NaN when the sample holds a non-finite
// value, so that its ln rho comes out NaN like the f64 path's); sn = 2^en.
template <int ND, int T32, typename XT, bool VEC>
__device__ __forceinline__ void load_x_digits_row(const XT* __restrict__ x, int64_t ldx, int D,
                                                  const double* __restrict__ pivot, int64_t row, int h,
                                                  i4v (&xd)[ND][T32], double& c2, double& sn) {
    const XT* xp = x + row * ldx + 16 * h;
    // 16 features of block `it` as centred doubles (zero past D); two passes over x (the second hits L1/L2) keep
    // the conversion's live registers at one block instead of the whole row
    auto load_block = [&](int it, double (&v)[16]) {
        const int f0 = 32 * it + 16 * h;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if constexpr (VEC) {
                typedef XT v4 __attribute__((ext_vector_type(4)));
                const bool in = f0 < D;             // D % 16 == 0 here: a 16-feature block is all in or all out
                v4 t = {0, 0, 0, 0};
                if (in) t = *reinterpret_cast<const v4*>(xp + 32 * it + 4 * q);
#pragma unroll
                for (int s = 0; s < 4; ++s) v[4 * q + s] = in ? (double)t[s] - pivot[f0 + 4 * q + s] : 0.0;
            } else {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int f = f0 + 4 * q + s;
                    v[4 * q + s] = f < D ? (double)xp[32 * it + 4 * q + s] - pivot[f] : 0.0;
                }
            }
        }
    };
    double mx = 0.0;
    bool bad = false;
#pragma unroll
    for (int it = 0; it < T32; ++it) {
        double v[16];
        load_block(it, v);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const double a = fabs(v[e]);
            bad |= !(a <= 1.7976931348623157e308);
            mx = fmax(mx, a);
        }
    }
    mx = fmax(mx, __shfl_xor(mx, 32));
    bad |= (bool)__shfl_xor((int)bad, 32);
    int en = 0;
    if (mx > 0.0) (void)frexp(mx, &en);
    const double scale = bad ? 0.0 : ldexp(1.0, 6 - en);
    c2 = bad ? __builtin_nan("") : ldexp(1.0, en - 12 - 7 * (ND - 1));
    sn = ldexp(1.0, en);
#pragma unroll
    for (int it = 0; it < T32; ++it) {
        double v[16];
        load_block(it, v);
        unsigned w[ND][4];
#pragma unroll
        for (int a = 0; a < ND; ++a)
#pragma unroll
            for (int q = 0; q < 4; ++q) w[a][q] = 0u;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            int d[ND];
            digits_of<ND>(bad ? 0.0 : v[e] * scale, d);
#pragma unroll
            for (int a = 0; a < ND; ++a) w[a][e >> 2] |= (unsigned)(d[a] & 0xff) << (8 * (e & 3));
        }
#pragma unroll
        for (int a = 0; a < ND; ++a) xd[a][it] = i4v{(int)w[a][0], (int)w[a][1], (int)w[a][2], (int)w[a][3]};
    }
}

template <int ND, int T32, typename XT, bool VEC>
__device__ __forceinline__ void load_x_digits(const XT* __restrict__ x, int64_t ldx, int64_t n_rows, int D,
                                              const double* __restrict__ pivot, int64_t n0, int c, int h,
                                              i4v (&xd)[ND][T32], double& c2, double& sn) {
    int64_t row = n0 + c;
    if (row >= n_rows) row = n_rows - 1;          // clamp: padded samples are never stored
    load_x_digits_row<ND, T32, XT, VEC>(x, ldx, D, pivot, row, h, xd, c2, sn);
}

// LDS reads whose completion is awaited by hand.  The compiler only ever emits s_waitcnt lgkmcnt(0) in this
// kernel, which would drain the prefetches together with the operand it needs; these reads are invisible to its
// counter model and are fenced by lds_wait<N>() (LDS returns in order: "at most N younger reads still in flight").
// The waited-for registers pass through the fence as in/out operands so that no consumer can be moved above it.
template <int OFF>
__device__ __forceinline__ void lds_read16(i4v& dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
template <int N>
__device__ __forceinline__ void lds_wait(i4v& a) {
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a) : "n"(N));
}
template <int N>
__device__ __forceinline__ void lds_wait(i4v& a, i4v& b) {
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N));
}

union pair_bits {
    i4v v;
    double d[2];
};

// MFMAs of step (JT, IT, A) with U digit fragment `ua`
template <int ND, int T32, int IT, int A>
__device__ __forceinline__ void i8_step_mfma(const i4v& ua, const i4v (&xd)[ND][T32], i16v (&acc)[ND]) {
#pragma unroll
    for (int b = 0; b + A < ND; ++b)
        acc[A + b] = __builtin_amdgcn_mfma_i32_32x32x32_i8(ua, xd[b][IT], acc[A + b], 0, 0, 0);
}

// Execution order of the steps.  Output blocks run in ascending order (REV = false) or descending order (REV = true):
// the two waves that share a SIMD take opposite orders, so that one's epilogues (vector ALU) fall into the other's
// long MFMA runs instead of both leaving the matrix pipe idle at the same time.  Step e (execution index) of a
// component -> its output block, and its position in the image (layout step = ND pair + digit).
template <int ND, int T32, bool REV>
struct i8_order {
    static constexpr int block_of(int e) {
        int o = 0, base = 0;
        while (true) {
            const int jt = REV ? T32 - 1 - o : o;
            const int n = (jt + 1) * ND;
            if (e < base + n) return jt;
            base += n;
            ++o;
        }
    }
    static constexpr int first_of(int jt) {      // execution index of the block's first step
        int base = 0;
        for (int o = 0; o < T32; ++o) {
            const int b = REV ? T32 - 1 - o : o;
            if (b == jt) return base;
            base += (b + 1) * ND;
        }
        return base;
    }
    static constexpr int layout_of(int e) {
        const int jt = block_of(e);
        return tri_pairs(jt) * ND + (e - first_of(jt));
    }
};

// Steps E .. end of its output block (recursion over the compile-time execution index: offsets and wait counts are
// immediates).  ua[e % 3] holds step e; step e + 2 is requested before step e runs.
template <int ND, int T32, bool REV, int E>
__device__ __forceinline__ void i8_steps(unsigned frag_addr, unsigned const_addr, i4v (&ua)[3], const i4v (&xd)[ND][T32],
                                         i16v (&acc)[ND], i4v (&sb)[2][2]) {
    constexpr bool BND = ND == kBoundDigits;
    constexpr int CSTRIDE = BND ? 128 : 512;                      // bytes of row constants per output block
    using ord = i8_order<ND, T32, REV>;
    constexpr int NS = tri_pairs(T32) * ND;
    constexpr int JT = ord::block_of(E), S = ord::layout_of(E);
    constexpr int IT = S / ND - tri_pairs(JT), A = S % ND;
    constexpr bool last = (IT == JT && A == ND - 1);
    if constexpr (E + 2 < NS) lds_read16<1024 * ord::layout_of(E + 2 < NS ? E + 2 : 0)>(ua[(E + 2) % 3], frag_addr);
    if constexpr (last) {
        lds_read16<JT * CSTRIDE + 0>(sb[0][0], const_addr);
        if constexpr (!BND) lds_read16<JT * CSTRIDE + 16>(sb[0][1], const_addr);
    }
    lds_wait<(E + 1 < NS) + (E + 2 < NS) + (last ? (BND ? 1 : 2) : 0)>(ua[E % 3]);
    i8_step_mfma<ND, T32, IT, A>(ua[E % 3], xd, acc);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (!last) i8_steps<ND, T32, REV, E + 1>(frag_addr, const_addr, ua, xd, acc, sb);
}

// per-lane constants of the epilogue: digit-class weights (c[0] for the heaviest pair of classes) and, for the bound
// pass, ce = 2^en i8_err
struct i8_lane_consts {
    double c[3];
    float c2f;      // bound pass: 2^(en - 26)
    float cef;      // bound pass: 2^en i8_err (rounded up), +inf = no bound for this sample
};

// Epilogue of output block JT, rows 2 GP and 2 GP + 1 of this lane: integer digit sums -> y -> q (f64, ND = 6).
template <int ND, int JT, int GP>
__device__ __forceinline__ void i8_rows(unsigned const_addr, const i16v (&acc)[ND], i4v (&sb)[2][2],
                                        const i8_lane_consts& lc, double& q) {
    static_assert(ND == 6, "the f64 epilogue combines six digit classes");
    if constexpr (GP + 1 < 8) {
        lds_read16<JT * 512 + 32 * (GP + 1)>(sb[(GP + 1) & 1][0], const_addr);
        lds_read16<JT * 512 + 32 * (GP + 1) + 16>(sb[(GP + 1) & 1][1], const_addr);
    }
    lds_wait<(GP + 1 < 8) ? 2 : 0>(sb[GP & 1][0], sb[GP & 1][1]);
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int g = 2 * GP + e;
        const int t0 = (acc[0][g] << 7) + acc[1][g];
        const int t1 = (acc[2][g] << 7) + acc[3][g];
        const int t2 = (acc[4][g] << 7) + acc[5][g];
        const double z = fma((double)t0, lc.c[0], fma((double)t1, lc.c[1], (double)t2 * lc.c[2]));
        pair_bits u;
        u.v = sb[GP & 1][e];
        const double y = fma(z, u.d[0], -u.d[1]);
        q = fma(y, y, q);
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (GP + 1 < 8) i8_rows<ND, JT, GP + 1>(const_addr, acc, sb, lc, q);
}

// Bound-pass epilogue (ND = 3) in plain f32 (measured with tools/overlap_probe.hip: next to int8 MFMAs, f32, f64 and
// integer vector instructions overlap with the matrix pipe, packed-f32 ones do not).  With I = 128 t0 + acc2 (the exact
// integer digit sum), m = 2^(en - 26) 2^e_jt and the row constant b of the image, y = I m - b satisfies
//   |y_exact - y| <= c := 2^e_jt 2^en i8_err  (digits)  + 2^-22 (|y| + |b|)  (f32: I to 24 bits, b to 24 bits, one fma)
//                          + the f64 rounding of b   <=   2^-20 |y| + beta_jt + 2^e_jt cef,
// and with u = |y| (1 - 2^-20), e = beta_jt + 2^e_jt cef:   y_exact^2 >= (u - e)_+^2 >= u^2 - 2 e u.
// So a block only accumulates Y2 = sum y^2 (and used to accumulate A = sum |y|: 7 instructions per row), and the caller forms
// q >= (1 - 2^-19) Y2 - 2 (beta_jt + 2^e_jt cef) A.  A NaN y poisons Y2; an infinite cef / beta makes q = -inf, i.e.
// the bound +inf ("candidate"): it never lies.  Step GP covers this lane's rows 4 GP .. 4 GP + 3 of block JT.
// (round 5) A is not accumulated any more: over the lane's 16 rows of the block, A = sum |y| <= 4 sqrt(sum y^2) (Cauchy-
// Schwarz), which the caller forms from Y2 - one instruction per row less in an epilogue that is as long as the block's
// MFMAs at D = 128 and twice as long at D = 64; the error term it multiplies is 1e-5 of q, a quarter more of it is nothing.
template <int JT, int GP>
__device__ __forceinline__ void i8_rows_bound(unsigned const_addr, const i16v (&acc)[kBoundDigits], i4v (&sb)[2][2],
                                              float m, float& y2) {
    if constexpr (GP + 1 < 4) lds_read16<JT * 128 + 16 * (GP + 1)>(sb[(GP + 1) & 1][0], const_addr);
    lds_wait<(GP + 1 < 4) ? 1 : 0>(sb[GP & 1][0]);
    union { i4v v; float f[4]; } r;
    r.v = sb[GP & 1][0];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int g = 4 * GP + e;
        const float tf = (float)((acc[0][g] << 7) + acc[1][g]);
        const float I = __builtin_fmaf(tf, 128.0f, (float)acc[2][g]);
        const float y = __builtin_fmaf(I, m, -r.f[e]);
        y2 = __builtin_fmaf(y, y, y2);
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (GP + 1 < 4) i8_rows_bound<JT, GP + 1>(const_addr, acc, sb, m, y2);
}

// TWO: also an UPPER bound of sum y_exact^2 in qu:  y_exact^2 <= (|y| (1 + 2^-20) + e)^2, e = beta_jt + 2^e_jt cef, i.e. per
// block and lane  (1 + 2^-19) Y2 + 2 (1 + 2^-20) e A + 16 e^2  (16 rows per lane and block).
template <int ND, bool BOUND, int T32, bool REV, int O, typename QT, bool TWO = false>
__device__ __forceinline__ void i8_blocks_from(unsigned frag_addr, unsigned const_addr, i4v (&ua)[3],
                                               const i4v (&xd)[ND][T32], const i8_lane_consts& lc, const i4v& scales,
                                               const i4v& scales2, QT& q, QT* qu = nullptr) {
    constexpr int JT = REV ? T32 - 1 - O : O;
    i16v acc[ND];
#pragma unroll
    for (int w = 0; w < ND; ++w) acc[w] = i16v{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    i4v sb[2][2];
    i8_steps<ND, T32, REV, i8_order<ND, T32, REV>::first_of(JT)>(frag_addr, const_addr, ua, xd, acc, sb);
    if constexpr (BOUND) {
        // (2^e_jt, beta_jt) of the component's output blocks: `scales` holds blocks 0, 1; `scales2` blocks 2, 3
        union { i4v v; float f[4]; } sc;
        sc.v = JT < 2 ? scales : scales2;
        const float sj = sc.f[2 * (JT & 1)], beta = sc.f[2 * (JT & 1) + 1];
        float y2 = 0.0f;
        i8_rows_bound<JT, 0>(const_addr, acc, sb, sj * lc.c2f, y2);
        // sum |y| over the lane's 16 rows <= sqrt(16 sum y^2); 2e-6 covers the f32 sum (16 x 2^-24) and v_sqrt_f32's ulp
        // (a NaN y2 stays NaN: the bound says nothing, as before)
        // (the raw v_sqrt_f32 may flush a subnormal y2 to zero while sum |y| is still ~4e-19: below FLT_MIN the bound of the
        // sum is taken from FLT_MIN itself, so that the error term never vanishes.  The factor 4 = sqrt(16) belongs to the
        // 16 rows a lane holds per output block - kBoundDigits' epilogue; a change of rows per lane changes it)
        // (one v_max_f32: a compare-and-select in this epilogue cost the proof round 0.3 ns per pair - 6.02 -> 6.36 ms per
        // benchmark step; a NaN y2 still reaches q through the fma below)
        float ysafe;
        asm("v_max_f32 %0, %1, %2" : "=v"(ysafe) : "v"(y2), "v"(1.17549435e-38f));
        const float ya = 4.000008f * __builtin_amdgcn_sqrtf(ysafe);
        const float eb = __builtin_fmaf(sj, lc.cef, beta);
        q = __builtin_fmaf(-2.0f * ya, eb, __builtin_fmaf(y2, 0.9999980926513671875f, q));
        if constexpr (TWO)
            *qu = __builtin_fmaf(16.001f * eb, eb, __builtin_fmaf(2.00001f * ya, eb, __builtin_fmaf(y2, 1.0000019073486328125f, *qu)));
    } else {
        i8_rows<ND, JT, 0>(const_addr, acc, sb, lc, q);
    }
    if constexpr (O + 1 < T32)
        i8_blocks_from<ND, BOUND, T32, REV, O + 1, QT, TWO>(frag_addr, const_addr, ua, xd, lc, scales, scales2, q, qu);
}

// One component for one wave tile; `im_lds` is the LDS byte address of the component's image.
// The U digits are requested two steps (one step = one digit a of one block pair = ND - a MFMAs) ahead of their
// use and the epilogue's row constants one pair of rows ahead, so that with only two waves per SIMD the LDS
// latency sits behind MFMAs instead of in front of them.
// T32 = output blocks evaluated (and x blocks held), TI >= T32 = blocks of the image (its leading tri_pairs(T32) block
// pairs are exactly the ones needed).
template <int ND, bool BOUND, int T32, int TI, bool REV>
__device__ __forceinline__ double estep_i8_component(unsigned im_lds, const i4v (&xd)[ND][T32], double ck,
                                                   const i8_lane_consts& lc, int lane, int c, int h, int64_t n0,
                                                   int64_t n_rows, double* __restrict__ lnrho_k,
                                                   float* __restrict__ ub_k /*BOUND: the f32 bound array's row instead*/) {
    using ord = i8_order<ND, T32, REV>;
    constexpr int P = tri_pairs(TI);
    const unsigned frag_addr = im_lds + lane * 16;                              // layout step s at + 1024 s
    // row constants: ND = 6 (jt, g) at + 512 jt + 256 h + 16 g; bound pass (jt, g) at + 128 jt + 64 h + 4 g
    const unsigned const_addr = im_lds + P * ND * 1024 + h * (BOUND ? 64 : 256);
    typename std::conditional<BOUND, float, double>::type q = 0;
    i4v scales = {0, 0, 0, 0}, scales2 = {0, 0, 0, 0};
    if constexpr (BOUND) {
        lds_read16<0>(scales, im_lds + P * ND * 1024 + TI * 128);
        lds_read16<16>(scales2, im_lds + P * ND * 1024 + TI * 128);
        lds_wait<0>(scales, scales2);
    }
    i4v ua[3];
    lds_read16<1024 * ord::layout_of(0)>(ua[0], frag_addr);
    lds_read16<1024 * ord::layout_of(1)>(ua[1], frag_addr);
    i8_blocks_from<ND, BOUND, T32, REV, 0>(frag_addr, const_addr, ua, xd, lc, scales, scales2, q);
    q += __shfl_xor(q, 32);
    if constexpr (BOUND) q = (q < 0) ? 0 : q;       // a negative lower bound of a square says nothing (NaN passes)
    const int64_t row = n0 + c;
    // BOUND: an upper bound of ln rho; 2^-16 of q covers the f32 rounding of its 128 squares and additions
    const double v = BOUND ? fma(-0.5 * (double)q, 1.0 - 1.52587890625e-05, ck + 1e-12 * fabs(ck)) : ck - 0.5 * (double)q;
    if (h == 0 && row < n_rows) {
        if constexpr (BOUND) ub_k[row] = __double2float_ru(v);       // what the sweeps carry (records.h): rounded up
        else lnrho_k[row] = v;
    }
    return v;
}

// ND = 6: the E-step.  ND = 3, BOUND: upper bounds of ln rho for the pruned E-step (estep.h) from three digits per
// operand at 6 instead of 21 MFMAs per block pair; the error term keeps the bound rigorous whatever the
// conditioning.  The bound pass may stop after TB < T32 output blocks (32 TB rows of y and the leading
// tri_pairs(TB) block pairs): dropping rows only loosens the bound - more candidates for the exact pass, less work in
// the pass every pair goes through; gmmvb_estep moves TB up and down with the candidate / active ratio it observes.
template <int ND, bool BOUND, int T32, int TB, typename XT, bool VEC, int NW>
__global__ __launch_bounds__(64 * NW) void estep_i8(const XT* __restrict__ x, int64_t ldx, int64_t n_rows, int D,
                                                    const unsigned char* __restrict__ img /*[K][IMGB]*/,
                                                    const double* __restrict__ pivot, const double* __restrict__ cvec,
                                                    int K, double* __restrict__ lnrho /*[K][npad]*/, int64_t npad,
                                                    int* __restrict__ khat /*[n_rows] first maximiser over k, or null*/,
                                                    float* __restrict__ ub /*[K][npad] BOUND: the bounds go here, in f32*/) {
    constexpr int IMGB = i8_img_bytes(ND, T32);
    constexpr int KB = i8_kb(ND, T32);
    __shared__ __attribute__((aligned(16))) unsigned char smem[2][KB * IMGB];   // the ONLY LDS object of the kernel
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 31, h = lane >> 5;
    const int64_t rows_per_wg = NW * 32;
    const int64_t n_wg_tiles = (n_rows + rows_per_wg - 1) / rows_per_wg;
    const int n_blocks = (K + KB - 1) / KB;

    auto stage = [&](int kb, int buf) {
        const int k0 = kb * KB;
        const int kcount = (K - k0 < KB) ? (K - k0) : KB;
        const int pieces = kcount * (IMGB / 1024);
        const unsigned char* src = img + (int64_t)k0 * IMGB + lane * 16;
        for (int piece = wave; piece < pieces; piece += NW)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + piece * 1024),
                                             (__attribute__((address_space(3))) void*)(&smem[buf][piece * 1024]), 16, 0,
                                             0);
    };

    for (int64_t wt = blockIdx.x; wt < n_wg_tiles; wt += gridDim.x) {
        const int64_t n0 = wt * rows_per_wg + (int64_t)wave * 32;
        stage(0, 0);
        static_assert(TB >= 1 && TB <= T32 && (BOUND || TB == T32), "only the bound pass may stop early");
        i4v xd[ND][TB];
        double c2, sn;
        load_x_digits<ND, TB, XT, VEC>(x, ldx, n_rows, D, pivot, n0, c, h, xd, c2, sn);
        i8_lane_consts lc;
        lc.c[2] = c2;
        lc.c[1] = c2 * (ND == 6 ? 16384.0 : 128.0);      // weight of t1 (ND = 6) / of t0 = 128 acc0 + acc1 (ND = 3)
        lc.c[0] = c2 * 268435456.0;
        {
            // bound pass: f32 products of powers of two stay exact for |en| <= 45 (and the image keeps |ej| <= 45);
            // outside, or for a non-finite sample, cef = +inf turns the bound into the trivial one
            int en = 0;
            (void)frexp(sn, &en);
            const bool ok = (c2 == c2) && en >= -44 && en <= 46;
            lc.c2f = ok ? (float)c2 : 0.0f;
            lc.cef = ok ? (float)(sn * i8_err(ND, TB) * 1.0001) : __builtin_huge_valf();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        double best = -__builtin_huge_val();
        int arg = 0;
        for (int kb = 0; kb < n_blocks; ++kb) {
            if (kb + 1 < n_blocks) stage(kb + 1, (kb + 1) & 1);
            const unsigned buf = (unsigned)(uintptr_t)((__attribute__((address_space(3))) unsigned char*)smem[kb & 1]);
            const int k0 = kb * KB;
#pragma unroll 1
            for (int kk = 0; kk < KB; ++kk) {
                const int k = k0 + kk;
                if (k >= K) break;
                double v;
                if (T32 > 1 && wave >= NW / 2)      // the second wave of each SIMD (wave uniform: no divergence)
                    v = estep_i8_component<ND, BOUND, TB, T32, true>(buf + kk * IMGB, xd, cvec[k], lc, lane, c, h, n0,
                                                                     n_rows, lnrho + (int64_t)k * npad,
                                                                     BOUND ? ub + (int64_t)k * npad : nullptr);
                else
                    v = estep_i8_component<ND, BOUND, TB, T32, false>(buf + kk * IMGB, xd, cvec[k], lc, lane, c, h, n0,
                                                                      n_rows, lnrho + (int64_t)k * npad,
                                                                      BOUND ? ub + (int64_t)k * npad : nullptr);
                if (v > best) {
                    best = v;
                    arg = k;
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        if (khat && h == 0 && n0 + c < n_rows) khat[n0 + c] = arg;
    }
}

// ---- sample digits kept in HBM ---------------------------------------------------------------------------------------------
// The list-driven kernels below evaluate ONE component per sample, so the f64 digit extraction of load_x_digits_row
// (3400 vector instructions per 32-sample tile) is not amortised over K components as in estep_i8: measured in round 2,
// it made a 3-digit pair as slow as its f64 evaluation.  The three digit planes of every row are therefore made once per
// sample matrix (and per regrouping of the rows) and kept next to x:
//   xq [row][a = 0..2][32 T32] int8   digit a of feature f of (x - pivot) 2^(6 - en), the bytes load_x_digits_row<3> forms
//   xqe[row]                   int8   en, the exponent of the row's largest |x - pivot| (kNoDigits: non-finite row, or
//                                     an exponent outside the range in which the f32 epilogue is exact - no bound)
// 3 bytes per feature instead of 4: a pair costs 96 T32 bytes of HBM traffic and no vector arithmetic before its MFMAs.
constexpr signed char kNoDigits = 127;
__host__ __device__ constexpr int64_t i8_digit_row_bytes(int t32) { return (int64_t)kBoundDigits * 32 * t32; }

// 8 threads per row, thread g owns features 16 g .. 16 g + 15 (g < 2 T32); 32 rows per 256-thread block.
template <typename XT>
__global__ __launch_bounds__(256) void x_digits_kernel(const XT* __restrict__ x, int64_t ldx, int64_t n_rows, int D, int T32,
                                                       const double* __restrict__ pivot, unsigned char* __restrict__ xq,
                                                       signed char* __restrict__ xqe) {
    constexpr int ND = kBoundDigits;
    const int64_t row = (int64_t)blockIdx.x * 32 + (threadIdx.x >> 3);
    const int g = threadIdx.x & 7;
    const bool in_row = row < n_rows;
    const bool has = in_row && g < 2 * T32;
    double v[16];
    double mx = 0.0;
    bool bad = false;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int f = 16 * g + e;
        v[e] = (has && f < D) ? (double)x[row * ldx + f] - pivot[f] : 0.0;
        const double a = fabs(v[e]);
        bad |= !(a <= 1.7976931348623157e308);
        mx = fmax(mx, a);
    }
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) {
        mx = fmax(mx, __shfl_xor(mx, o));
        bad |= (bool)__shfl_xor((int)bad, o);
    }
    int en = 0;
    if (mx > 0.0) (void)frexp(mx, &en);
    const bool ok = !bad && en >= -44 && en <= 46;          // the range in which estep_i8's f32 epilogue stays exact
    const double scale = ok ? ldexp(1.0, 6 - en) : 0.0;
    unsigned w[ND][4];
#pragma unroll
    for (int a = 0; a < ND; ++a)
#pragma unroll
        for (int q = 0; q < 4; ++q) w[a][q] = 0u;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        int d[ND];
        digits_of<ND>(v[e] * scale, d);
#pragma unroll
        for (int a = 0; a < ND; ++a) w[a][e >> 2] |= (unsigned)(d[a] & 0xff) << (8 * (e & 3));
    }
    if (has) {
        unsigned char* dst = xq + row * i8_digit_row_bytes(T32) + 16 * g;
#pragma unroll
        for (int a = 0; a < ND; ++a)
            *reinterpret_cast<i4v*>(dst + a * 32 * T32) = i4v{(int)w[a][0], (int)w[a][1], (int)w[a][2], (int)w[a][3]};
    }
    if (in_row && g == 0) xqe[row] = ok ? (signed char)en : kNoDigits;
}

// ---- two-sided bounds for listed (sample, component) pairs: the proof round of the pruned E-step ---------------------------
// A row with a single active component has r = 1.0 exactly whatever the value of its ln rho (records.h, "settled rows"):
// all the E-step owes such a row is the PROOF that it still has one - a lower bound of its component's ln rho and upper
// bounds of the components whose carried bound no longer clears it.  Three digits on the int8 pipe give both to about
// 1e-5 relative (same images, digits and error terms as the bound pass; the error term is added for the lower bound).
// Returns (lower, upper) bound of q = || U (x - m) ||^2.
template <int T32, bool REV>
__device__ __forceinline__ void estep_i8_two_sided(unsigned im_lds, const i4v (&xd)[kBoundDigits][T32],
                                                   const i8_lane_consts& lc, int lane, int h, float& q_lo, float& q_hi) {
    constexpr int ND = kBoundDigits;
    using ord = i8_order<ND, T32, REV>;
    constexpr int P = tri_pairs(T32);
    const unsigned frag_addr = im_lds + lane * 16;
    const unsigned const_addr = im_lds + P * ND * 1024 + h * 64;
    float q = 0.0f, qu = 0.0f;
    i4v scales = {0, 0, 0, 0}, scales2 = {0, 0, 0, 0};
    lds_read16<0>(scales, im_lds + P * ND * 1024 + T32 * 128);
    lds_read16<16>(scales2, im_lds + P * ND * 1024 + T32 * 128);
    lds_wait<0>(scales, scales2);
    i4v ua[3];
    lds_read16<1024 * ord::layout_of(0)>(ua[0], frag_addr);
    lds_read16<1024 * ord::layout_of(1)>(ua[1], frag_addr);
    i8_blocks_from<ND, true, T32, REV, 0, float, true>(frag_addr, const_addr, ua, xd, lc, scales, scales2, q, &qu);
    q += __shfl_xor(q, 32);
    qu += __shfl_xor(qu, 32);
    q_lo = (q < 0) ? 0 : q;
    q_hi = qu;
}

constexpr int kI8PairTiles = 8;      // wave tiles (32 list entries each) per wave and chunk
__host__ __device__ constexpr int i8_pairs_per_chunk() { return 8 * 32 * kI8PairTiles; }

// plan[k] = component k's first chunk of i8_pairs_per_chunk() list entries, plan[K] = total (gather_plan_kernel); a fixed
// grid of persistent workgroups takes contiguous runs of chunks and restages the 31-KB digit image only when the
// component changes.  For every listed pair (row, k):
//   ub[k][row] <- an upper bound of ln rho_{row,k}, rounded up to f32 (the array the sweeps carry, records.h) - unless
//                 ub is null (a round that only serves lower bounds);
//   lb[k][row] <- a lower bound of it (f64; -inf when the image or the sample admits no bound - the caller then
//                 treats the row as unproven and evaluates it exactly).
template <int T32>
__global__ __launch_bounds__(512) void estep_i8_proof(const unsigned char* __restrict__ xq, const signed char* __restrict__ xqe,
                                                      const unsigned char* __restrict__ img /*[K][IMGB], 3 digits*/,
                                                      const double* __restrict__ cvec, int K,
                                                      const int* __restrict__ lists /*[K][cap]*/, int64_t cap,
                                                      const int* __restrict__ counts, const int* __restrict__ plan,
                                                      float* __restrict__ ub /*[K][npad]*/, double* __restrict__ lb /*[K][npad]*/,
                                                      int64_t npad) {
    constexpr int ND = kBoundDigits, NW = 8;
    constexpr int IMGB = i8_img_bytes(ND, T32);
    constexpr int CHUNK = i8_pairs_per_chunk();
    constexpr int64_t RS = i8_digit_row_bytes(T32);
    __shared__ __attribute__((aligned(16))) unsigned char smem[IMGB];
    __shared__ int s_first[257];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 31, h = lane >> 5;
    for (int k = threadIdx.x; k <= K; k += 512) s_first[k] = plan[k];
    __syncthreads();
    const int total = s_first[K];
    const int per = (total + (int)gridDim.x - 1) / (int)gridDim.x;
    const int c0 = (int)blockIdx.x * per;
    const int c1 = c0 + per < total ? c0 + per : total;
    const unsigned im_lds = (unsigned)(uintptr_t)((__attribute__((address_space(3))) unsigned char*)smem);
    int k = 0, kcur = -1;
    for (int ch = c0; ch < c1; ++ch) {
        while (s_first[k + 1] <= ch) ++k;
        if (k != kcur) {
            __syncthreads();
            const unsigned char* src = img + (int64_t)k * IMGB + lane * 16;
            for (int piece = wave; piece < IMGB / 1024; piece += NW)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + piece * 1024),
                                                 (__attribute__((address_space(3))) void*)(&smem[piece * 1024]), 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            kcur = k;
        }
        const int count = counts[k];
        const int64_t chunk0 = (int64_t)(ch - s_first[k]) * CHUNK;
        const int* list = lists + (int64_t)k * cap;
        const double ck = cvec[k];
        // tiles of this wave in the chunk: t = 0 .. nt - 1 at list entries chunk0 + (t NW + wave) 32; the digits of tile
        // t + 1 are requested before tile t is computed (two register sets, the loop is unrolled by two)
        int nt = 0;
        while (nt < kI8PairTiles && chunk0 + ((int64_t)nt * NW + wave) * 32 < count) ++nt;
        auto entry = [&](int t) -> int {                    // list entry of this lane's pair in tile t
            const int64_t e = chunk0 + ((int64_t)t * NW + wave) * 32 + c;
            return list[e < count ? e : count - 1];
        };
        auto request = [&](int idx, i4v (&xd)[ND][T32], int64_t& row, int& en) {
            row = idx;
            const unsigned char* src = xq + row * RS + 16 * h;
#pragma unroll
            for (int a = 0; a < ND; ++a)
#pragma unroll
                for (int it = 0; it < T32; ++it) xd[a][it] = *reinterpret_cast<const i4v*>(src + a * 32 * T32 + 32 * it);
            en = xqe[row];
        };
        auto evaluate = [&](int t, const i4v (&xd)[ND][T32], int64_t row, int en) {
            const bool ok = en != kNoDigits;
            i8_lane_consts lc;
            lc.c[0] = lc.c[1] = lc.c[2] = 0.0;        // (the f64 weights belong to the six-digit epilogue)
            lc.c2f = ok ? (float)ldexp(1.0, en - 12 - 7 * (ND - 1)) : 0.0f;
            lc.cef = ok ? (float)(ldexp(1.0, en) * i8_err(ND, T32) * 1.0001) : __builtin_huge_valf();
            float q_lo, q_hi;
            if (T32 > 1 && wave >= NW / 2)
                estep_i8_two_sided<T32, true>(im_lds, xd, lc, lane, h, q_lo, q_hi);
            else
                estep_i8_two_sided<T32, false>(im_lds, xd, lc, lane, h, q_lo, q_hi);
            const int64_t e = chunk0 + ((int64_t)t * NW + wave) * 32 + c;
            if (h == 0 && e < count) {
                // 2^-16 of q covers the f32 rounding of its 32 T32 squares and additions (as in estep_i8_component)
                const double up = fma(-0.5 * (double)q_lo, 1.0 - 1.52587890625e-05, ck + 1e-12 * fabs(ck));
                const double lo = fma(-0.5 * (double)q_hi, 1.0 + 1.52587890625e-05, ck - 1e-12 * fabs(ck));
                if (ub) ub[(int64_t)k * npad + row] = __double2float_ru(up);
                lb[(int64_t)k * npad + row] = (lo == lo) ? lo : -__builtin_huge_val();
            }
        };
        // Two register sets: the digits of tile t + 1 are requested before tile t is computed, and the list entry of tile
        // t + 2 before that - a tile's digit loads then wait for nothing but addresses that arrived a tile ago.
        i4v xa[ND][T32], xb[ND][T32];
        int64_t ra = 0, rb = 0;
        int ea = 0, eb = 0;
        int i1 = nt > 1 ? entry(1) : 0;
        if (nt > 0) request(entry(0), xa, ra, ea);
        for (int t = 0; t < nt; t += 2) {
            const int i2 = t + 2 < nt ? entry(t + 2) : 0;
            if (t + 1 < nt) request(i1, xb, rb, eb);
            evaluate(t, xa, ra, ea);
            if (t + 1 >= nt) break;
            const int i3 = t + 3 < nt ? entry(t + 3) : 0;
            if (t + 2 < nt) request(i2, xa, ra, ea);
            evaluate(t + 1, xb, rb, eb);
            i1 = i3;
        }
    }
}


// ---- the proof round over ROW SUPERBLOCKS (round 4) -------------------------------------------------------------------------
// estep_i8_proof walks one component's list after the other, so a row with p proof pairs has its digit planes fetched from
// HBM p times - at K = 256, D = 64 (config 4) p is 6 - 11, and the kernel is bound by exactly those gathered 192-byte reads
// (45 GB at ~4.8 TB/s in a 9-ms launch).  Here the pairs are regrouped into ITEMS of at most kProofItem entries of one
// component's list that lie in one superblock of kProofSuperRows rows (the lists are ascending, so that is a contiguous
// stretch of the list; its ends come from the selection blocks' bases), and the items of a superblock are handed to the
// workgroups of ONE XCD (workgroup b runs on XCD b mod 8), interleaved so that all of them work on the same one or two
// superblocks at any time: 0.8 - 1.5 MB of digit planes, which stay in that XCD's 4-MB L2 while every component's items
// gather from them.  HBM sees every touched row about once.  Same arithmetic per pair, same outputs.
constexpr int kProofSuperRows = 4096;
constexpr int kProofItem = 256;            // one 32-entry tile for each of the eight waves
constexpr int kProofXcds = 8;

__device__ __forceinline__ void proof_unit(const int* __restrict__ blk_base, const int* __restrict__ counts, int nblk, int s,
                                           int k, int K, int& lo, int& hi) {
    constexpr int BS = kProofSuperRows / 256;
    const int b0 = s * BS, b1 = b0 + BS;
    lo = blk_base[(int64_t)b0 * K + k];                 // (block-major: aux_kernels.h blk_at)
    hi = b1 < nblk ? blk_base[(int64_t)b1 * K + k] : counts[k];
}

// cum[s][k] = items of superblock s that belong to components < k (cum[s][K] = tot[s] = all of them).  One workgroup per
// superblock, one thread per component.
__global__ __launch_bounds__(256) void proof_units_kernel(const int* __restrict__ blk_base /*[nblk][K] exclusive bases*/,
                                                          const int* __restrict__ counts, int K, int nblk,
                                                          int* __restrict__ cum /*[n_super][K + 1]*/, int* __restrict__ tot) {
    __shared__ int sc[256];
    const int s = blockIdx.x, k = threadIdx.x;
    int n = 0;
    if (k < K) {
        int lo, hi;
        proof_unit(blk_base, counts, nblk, s, k, K, lo, hi);
        n = (hi - lo + kProofItem - 1) / kProofItem;
    }
    sc[k] = n;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
        const int v = k >= o ? sc[k - o] : 0;
        __syncthreads();
        sc[k] += v;
        __syncthreads();
    }
    if (k < K) cum[(int64_t)s * (K + 1) + k] = sc[k] - n;
    if (k == 255) {
        cum[(int64_t)s * (K + 1) + K] = sc[255];
        tot[s] = sc[255];
    }
}

// Workgroup x of kProofXcds: xb[s] = items of the superblocks s' < s with s' = s = x (mod 8); xtot[x] = their total
__global__ __launch_bounds__(1024) void proof_order_kernel(const int* __restrict__ tot, int n_super, int* __restrict__ xb,
                                                           int* __restrict__ xtot) {
    __shared__ int sc[1024];
    const int x = blockIdx.x, t = threadIdx.x;
    const int nx = n_super > x ? (n_super - x + kProofXcds - 1) / kProofXcds : 0;
    const int per = (nx + 1023) / 1024;
    const int j0 = t * per, j1 = j0 + per < nx ? j0 + per : nx;
    int sum = 0;
    for (int j = j0; j < j1; ++j) sum += tot[x + kProofXcds * j];
    sc[t] = sum;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const int v = t >= o ? sc[t - o] : 0;
        __syncthreads();
        sc[t] += v;
        __syncthreads();
    }
    int run = sc[t] - sum;
    for (int j = j0; j < j1; ++j) {
        xb[x + kProofXcds * j] = run;
        run += tot[x + kProofXcds * j];
    }
    if (t == 1023) xtot[x] = sc[1023];
}

// items[first(x) + xb[s] + cum[s][k] + c] = (k, first list entry, entries, s) for chunk c of unit (s, k)
__global__ __launch_bounds__(256) void proof_items_kernel(const int* __restrict__ blk_base, const int* __restrict__ counts, int K,
                                                          int nblk, const int* __restrict__ cum, const int* __restrict__ xb,
                                                          const int* __restrict__ xtot, i4v* __restrict__ items) {
    const int s = blockIdx.x, k = threadIdx.x;
    if (k >= K) return;
    int first = 0;
    for (int q = 0; q < s % kProofXcds; ++q) first += xtot[q];
    int lo, hi;
    proof_unit(blk_base, counts, nblk, s, k, K, lo, hi);
    i4v* dst = items + first + xb[s] + cum[(int64_t)s * (K + 1) + k];
    for (int e = lo; e < hi; e += kProofItem) *dst++ = i4v{k, e, hi - e < kProofItem ? hi - e : kProofItem, s};
}

template <int T32>
__global__ __launch_bounds__(512) void estep_i8_proof_blocked(const unsigned char* __restrict__ xq, const signed char* __restrict__ xqe,
                                                              const unsigned char* __restrict__ img /*[K][IMGB], 3 digits*/,
                                                              const double* __restrict__ cvec, int K,
                                                              const int* __restrict__ lists /*[K][cap]*/, int64_t cap,
                                                              const i4v* __restrict__ items, const int* __restrict__ xtot,
                                                              float* __restrict__ ub /*[K][npad]*/, double* __restrict__ lb /*[K][npad]*/,
                                                              int64_t npad) {
    constexpr int ND = kBoundDigits, NW = 8;
    constexpr int IMGB = i8_img_bytes(ND, T32);
    constexpr int64_t RS = i8_digit_row_bytes(T32);
    constexpr int NS = T32 <= 3 ? 3 : 2;             // digit register sets = items whose digit planes are in flight + 1
    constexpr int NI = (IMGB + 512 * 16 - 1) / (512 * 16);      // 16-byte pieces of an image per thread
    __shared__ __attribute__((aligned(16))) unsigned char smem[2][IMGB];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 31, h = lane >> 5;
    const int x = (int)(blockIdx.x % kProofXcds), j = (int)(blockIdx.x / kProofXcds), W = (int)(gridDim.x / kProofXcds);
    int first = 0;
    for (int q = 0; q < x; ++q) first += xtot[q];
    const int n_items = xtot[x];
    const i4v* my = items + first;
    if (n_items <= j) return;
    // Every iteration issues the SAME memory operations in the same order (items past the end repeat the workgroup's last
    // item and only skip the arithmetic and the stores): the compiler's in-order vmcnt bookkeeping then stays exact, and a
    // wait for the oldest request leaves the younger ones in flight.
    auto desc = [&](int t) -> i4v {                  // item t of this workgroup: (component, first entry, entries, superblock)
        const int i = j + t * W;
        i4v d = my[i < n_items ? i : n_items - 1];
        if (i >= n_items) d[3] = -1;
        return d;
    };
    auto entry = [&](const i4v& d) -> int {           // row of this lane's pair (the item's last row beyond its end)
        const int o = wave * 32 + c;
        return lists[(int64_t)d[0] * cap + d[1] + (o < d[2] ? o : d[2] - 1)];
    };
    // The image goes global -> registers -> LDS with ordinary loads (not LDS-DMA): every wait is then the compiler's own,
    // exact, in-order vmcnt - a blanket vmcnt(0) per item made every item wait for the list entries requested for the
    // item after next, an HBM round trip (first form of this kernel: no faster than estep_i8_proof).
    auto image_load = [&](int k, i4v (&ir)[NI]) {
        const unsigned char* src = img + (int64_t)k * IMGB;
#pragma unroll
        for (int q = 0; q < NI; ++q) {
            const int o = (q * 512 + tid) * 16;
            ir[q] = *reinterpret_cast<const i4v*>(src + (o < IMGB ? o : IMGB - 16));
        }
    };
    auto image_store = [&](int buf, const i4v (&ir)[NI]) {
#pragma unroll
        for (int q = 0; q < NI; ++q)
            if ((q * 512 + tid) * 16 < IMGB) *reinterpret_cast<i4v*>(&smem[buf][(q * 512 + tid) * 16]) = ir[q];
    };
    auto request = [&](int idx, i4v (&xd)[ND][T32], int64_t& row, int& en) {
        row = idx;
        const unsigned char* src = xq + row * RS + 16 * h;
#pragma unroll
        for (int a = 0; a < ND; ++a)
#pragma unroll
            for (int it = 0; it < T32; ++it) xd[a][it] = *reinterpret_cast<const i4v*>(src + a * 32 * T32 + 32 * it);
        en = xqe[row];
    };
    auto evaluate = [&](const i4v& d, int buf, const i4v (&xd)[ND][T32], int64_t row, int en) {
        if (d[3] < 0 || wave * 32 >= d[2]) return;    // (wave-uniform: past the end, or this wave's tile lies beyond the item)
        const unsigned im_lds = (unsigned)(uintptr_t)((__attribute__((address_space(3))) unsigned char*)smem[buf]);
        const bool ok = en != kNoDigits;
        i8_lane_consts lc;
        lc.c[0] = lc.c[1] = lc.c[2] = 0.0;
        lc.c2f = ok ? (float)ldexp(1.0, en - 12 - 7 * (ND - 1)) : 0.0f;
        lc.cef = ok ? (float)(ldexp(1.0, en) * i8_err(ND, T32) * 1.0001) : __builtin_huge_valf();
        float q_lo, q_hi;
        if (T32 > 1 && wave >= NW / 2)
            estep_i8_two_sided<T32, true>(im_lds, xd, lc, lane, h, q_lo, q_hi);
        else
            estep_i8_two_sided<T32, false>(im_lds, xd, lc, lane, h, q_lo, q_hi);
        if (h == 0 && wave * 32 + c < d[2]) {
            const int k = d[0];
            const double ck = cvec[k];
            const double up = fma(-0.5 * (double)q_lo, 1.0 - 1.52587890625e-05, ck + 1e-12 * fabs(ck));
            const double lo = fma(-0.5 * (double)q_hi, 1.0 + 1.52587890625e-05, ck - 1e-12 * fabs(ck));
            if (ub) ub[(int64_t)k * npad + row] = __double2float_ru(up);
            lb[(int64_t)k * npad + row] = (lo == lo) ? lo : -__builtin_huge_val();
        }
    };
    // Software pipeline over the workgroup's items t = 0, 1, ... (dd[i] = descriptor of item t + i).  In iteration t:
    //   the image of item t (in registers since iteration t - 1) goes to LDS buffer t & 1; barrier;
    //   requests, in this order: the image of item t + 1; the list entries of item t + NS; the digit planes of item
    //             t + NS - 1 (into the register set item t - 1 has left); the descriptor of item t + NS + 2 - what the next
    //             iteration needs first (image, then entries) is requested first, so waiting for it (vmcnt counts in order)
    //             leaves the digit planes behind it in flight;
    //   item t is computed from register set t % NS.
    // One barrier per item: behind it item t's image is complete, and everybody has left the buffer item t + 1's image will
    // be written to in the next iteration.
    i4v xs[NS][ND][T32];
    int64_t rows[NS];
    int ens[NS];
    i4v dd[NS + 2];
#pragma unroll
    for (int i = 0; i < NS + 2; ++i) dd[i] = desc(i);
    i4v ir[NI];
    image_load(dd[0][0], ir);
#pragma unroll
    for (int q = 0; q < NS; ++q) {
        rows[q] = 0;
        ens[q] = 0;
        if (q + 1 < NS) request(entry(dd[q]), xs[q], rows[q], ens[q]);
    }
    int e_next = entry(dd[NS - 1]);
    // (at the loop header the compiler merges the prologue's and the back edge's pending loads and waits for all of them:
    // the body is unrolled over several rounds of the register sets so that this happens once in UNR items)
    constexpr int UNR = (T32 <= 2 ? 4 : 2) * NS;
    for (int t = 0; dd[0][3] >= 0; t += UNR) {
#pragma unroll
        for (int p = 0; p < UNR; ++p) {
            const int buf = (t + p) & 1;
            image_store(buf, ir);
            __syncthreads();
            image_load(dd[1][0], ir);
            const int e_new = entry(dd[NS]);
            request(e_next, xs[(p + NS - 1) % NS], rows[(p + NS - 1) % NS], ens[(p + NS - 1) % NS]);
            e_next = e_new;
            const i4v dn = desc(t + p + NS + 2);
            evaluate(dd[0], buf, xs[p % NS], rows[p % NS], ens[p % NS]);
#pragma unroll
            for (int i = 0; i < NS + 1; ++i) dd[i] = dd[i + 1];
            dd[NS + 1] = dn;
        }
    }
}

}  // namespace gmmvb
