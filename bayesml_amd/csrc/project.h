// The stateless sweep of the pruned E-step: per-pair upper bounds of ln rho from the CURRENT parameters and the row itself,
// instead of f32 bounds carried (and eroded) from pass to pass (records.h: rec_sweep_kernel).
//
// A tile of 256 rows (= a selection block) of the regrouped row order belongs - apart from the few tiles on a group border -
// to one dominant component j, the tile's reference.  For ANY reference j and every component k, with dx = x_n - m_j:
//
//   || U_k (x_n - m_k) ||^2  =  || U_k dx ||^2  +  2 g_jk . dx  +  s_jk^2   >=   2 g_jk . dx  +  s_jk^2,
//                                g_jk = U_k^T U_k (m_j - m_k),  s_jk = || U_k (m_j - m_k) ||
//
// (an identity, then the square dropped: no triangle inequality, the cross term keeps the pair's direction - which is what a
// bound has to know at K = 256, D = 64, where the centre-to-centre distance s_jk - sigma d_j alone leaves a hundred components
// per row undecided; tools/probe_projection.py).  A first form kept lmin_k || dx ||^2 with a Gershgorin bound of
// lambda_min(U_k^T U_k): on the fits it is 0 for most components (and 0.01 where it is not) - it bought nothing and cost a
// kernel and a GEMM column, so it went.  The K x K table (g_jk as two base-128 int8 digits, s_jk, the constants) is made from
// the parameters in force by proj_table_kernel - nothing is carried, nothing erodes -, and the row-dependent part is ONE int8
// GEMM per tile on the digit planes that the proof round keeps in HBM (estep_i8.h: xq):
//
//   p_nk = g_jk . (x_n - pivot)      [256 rows] x [D] x [K columns]        v_mfma_i32_32x32x32_i8, three MFMAs per block
//   ln rho_nk  <=  A_jk - p_nk + err_nk,     A_jk = c_k - s_jk^2 / 2 + g_jk . (m_j - pivot)
//
// with every rounding on the safe side (digit truncation, dropped digit class, f32 epilogue; a row or column without digits
// gives no bound: it never lies).  Two uses (GMMVB_PROJECT, capi_estep.hip):
//   filter   proj_filter_kernel: the carried sweep (records.h) stays; the pairs it lists for the int8 proof round - most of
//            them far pairs whose carried bound has eroded - are taken off the lists where the table clears them;
//   only     rec_project_kernel: the sweep itself is stateless (outputs of rec_sweep_kernel<PREV>); what the table leaves goes
//            the way the carried sweep's candidates went.  The settled rows' own distance bound (dlock) is still carried
//            through (Gamma, delta): one float per row.
// Measured in profiles/r6_experiments.md.
#pragma once
#include "rec_common.h"

namespace gmmvb {

// (the pieces of estep_i8.h this file relies on, restated so that it can live in a translation unit of its own: the digit
// planes' layout [row][3 digits][32 T32] with one exponent byte per row, 127 = no digits)
typedef int pj_i4v __attribute__((ext_vector_type(4)));
typedef int pj_i16v __attribute__((ext_vector_type(16)));
constexpr signed char kProjNoDigits = 127;                                 // = kNoDigits
constexpr int kProjPlaneDigits = 3;                                        // = kBoundDigits: digits per feature in xq
__host__ __device__ constexpr int proj_blocks(int D) { return (D + 31) / 32; }
__host__ __device__ constexpr int64_t proj_digit_row_bytes(int t32) { return (int64_t)kProjPlaneDigits * 32 * t32; }

constexpr int kProjDigits = 2;
// column blocks of 32 components
__host__ __device__ constexpr int proj_kblocks(int K) { return (K + 31) / 32; }
__host__ __device__ constexpr int64_t proj_image_bytes(int K, int t32) { return (int64_t)proj_kblocks(K) * t32 * kProjDigits * 1024; }
__host__ __device__ constexpr int proj_tri_len(int D) { return D * (D + 1) / 2; }

// sums of a, b, c and the maximum of d over the workgroup (256 threads), in every thread
__device__ __forceinline__ void proj_reduce(double& a, double& b, double& c, double& d, double (*scr)[4]) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        a += __shfl_xor(a, o);
        b += __shfl_xor(b, o);
        c += __shfl_xor(c, o);
        const double t = __shfl_xor(d, o);
        d = (t > d || t != t) ? t : d;
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
        scr[threadIdx.x >> 6][0] = a;
        scr[threadIdx.x >> 6][1] = b;
        scr[threadIdx.x >> 6][2] = c;
        scr[threadIdx.x >> 6][3] = d;
    }
    __syncthreads();
    a = scr[0][0] + scr[1][0] + scr[2][0] + scr[3][0];
    b = scr[0][1] + scr[1][1] + scr[2][1] + scr[3][1];
    c = scr[0][2] + scr[1][2] + scr[2][2] + scr[3][2];
    d = scr[0][3];
#pragma unroll
    for (int w = 1; w < 4; ++w) d = (scr[w][3] > d || scr[w][3] != scr[w][3]) ? scr[w][3] : d;
}

// The table.  Workgroup (k, y): component k against the references j = y, y + gridDim.y, ...
//   gimg   [K][proj_kblocks][T32][2][64 lanes][16]   B operands of v_mfma_i32_32x32x32_i8: lane (c, h) of block (kb, it) holds digit b
//                                                     of features 32 it + 16 h + (0..15) of column 32 kb + c, for reference j
//   gconst [K][32 proj_kblocks] float4               x = C = 2^(eg - 19): p^ = 2^en C t for the integer digit sum t = 128 acc0 + acc1
//                                                     y = E: |p - p^| <= 2^en E   (both truncations, the dropped digit class, f64 rounding of g)
//                                                     z = 0;  w = A_jk rounded up (+inf: no bound for this column)
__global__ __launch_bounds__(256) void proj_table_kernel(const double* __restrict__ u, const double* __restrict__ m,
                                                         const double* __restrict__ cvec, const double* __restrict__ pivot,
                                                         int K, int D, int T32, unsigned char* __restrict__ gimg,
                                                         float4* __restrict__ gconst) {
    extern __shared__ double s_tri[];                 // packed lower triangle of U_k: row r at r (r + 1) / 2
    __shared__ double s_dm[128], s_v[128], scr[4][4];
    const int k = blockIdx.x, tid = threadIdx.x;
    const int KB = proj_kblocks(K), Dp = 32 * T32;
    {
        const double* uk = u + (int64_t)k * D * D;
        for (int e = tid; e < D * D; e += 256) {
            const int r = e / D, i = e - r * D;
            if (i <= r) s_tri[r * (r + 1) / 2 + i] = uk[e];
        }
    }
    const double ck = cvec[k];
    for (int j = blockIdx.y; j < K; j += gridDim.y) {
        __syncthreads();
        double off_j = 0.0;                               // (m_j - pivot)_i
        if (tid < D) {
            off_j = m[(int64_t)j * D + tid] - pivot[tid];
            s_dm[tid] = m[(int64_t)j * D + tid] - m[(int64_t)k * D + tid];
        }
        __syncthreads();
        double g = 0.0, s2 = 0.0, b = 0.0, babs = 0.0;
        if (tid < D) {
            double v = 0.0;
            const double* row = s_tri + tid * (tid + 1) / 2;
            for (int i = 0; i <= tid; ++i) v = fma(row[i], s_dm[i], v);
            s_v[tid] = v;
            s2 = v * v;
        }
        __syncthreads();
        if (tid < D) {
            for (int r = tid; r < D; ++r) g = fma(s_tri[r * (r + 1) / 2 + tid], s_v[r], g);
            b = g * off_j;
            babs = fabs(b);
        }
        double gmax = tid < D ? fabs(g) : 0.0;
        proj_reduce(s2, b, babs, gmax, scr);
        int eg = 0;
        if (gmax > 0.0) (void)frexp(gmax, &eg);
        const bool wide = !(gmax <= 1.7976931348623157e308) || !(s2 <= 1e300) || eg < -60 || eg > 60;
        const double scale = wide ? 0.0 : ldexp(1.0, 6 - eg);
        int d[kProjDigits] = {0, 0};
        if (tid < D && !wide) {                           // two balanced base-128 digits of g 2^(6 - eg) in [-64, 64]
            const double t = g * scale, r0 = __builtin_rint(t);
            d[0] = (int)r0;
            d[1] = (int)__builtin_rint((t - r0) * 128.0);
        }
        double s1 = tid < D ? fabs((double)d[0] + (double)d[1] * 0.0078125) : 0.0, s11 = tid < D ? fabs((double)d[1]) : 0.0;
        double z0 = 0.0, z1 = 0.0;
        proj_reduce(s1, s11, z0, z1, scr);
        if (tid < Dp) {
            const int it = tid >> 5, h = (tid >> 4) & 1, byte = tid & 15, kb = k >> 5, c = k & 31;
#pragma unroll
            for (int dg = 0; dg < kProjDigits; ++dg)
                gimg[(((((int64_t)j * KB + kb) * T32 + it) * kProjDigits + dg) * 64 + (32 * h + c)) * 16 + byte] =
                    (unsigned char)((tid < D ? d[dg] : 0) & 0xff);
        }
        if (tid == 0) {
            // digit-product units: 2^-8 (sum |X^| + sum |G^|) + D 2^-16 for the two truncations, sum |dX1| |dG1| / 2^14 for the
            // dropped class (|X^| <= 64.5, |dX1| <= 64), 1e-9 x 64.5 D for the f64 rounding of g itself
            const double units = 0.00390625 * (64.5 * Dp + s1) + Dp * 1.52587890625e-05 + 64.0 * s11 * 6.103515625e-05 +
                                 1e-9 * 64.5 * Dp;
            const double a = ck - 0.5 * s2 + b;
            float4 o;
            o.x = wide ? 0.0f : (float)ldexp(1.0, eg - 19);
            o.y = wide ? 0.0f : __double2float_ru(ldexp(1.0, eg - 12) * units * 1.0001);
            o.z = 0.0f;
            o.w = wide ? __builtin_huge_valf() : __double2float_ru(a + 1e-9 * (fabs(ck) + s2 + babs) + 1e-30);
            gconst[(int64_t)j * KB * 32 + k] = o;
        }
    }
    // columns K .. 32 KB - 1 of the last block: no component (their image bytes are never written: the buffer is zeroed once at
    // allocation; their constants say "no bound" and the sweep masks their bits anyway)
    if (k == 0 && K + tid < 32 * KB)
        for (int j = blockIdx.y; j < K; j += gridDim.y)
            gconst[(int64_t)j * KB * 32 + K + tid] = make_float4(0.0f, 0.0f, 0.0f, __builtin_huge_valf());
}

// The reference component of every tile of kSelRows rows of the regrouped order: the group (list of regroup_rows, lengths in
// counts) that holds most of the tile's rows.  One workgroup.
__global__ __launch_bounds__(256) void proj_tile_ref_kernel(const int* __restrict__ counts, int K, int64_t n_tiles,
                                                            int* __restrict__ tile_ref) {
    __shared__ long long s_end[1025];
    if (threadIdx.x == 0) {
        long long run = 0;
        for (int k = 0; k < K; ++k) {
            run += counts[k];
            s_end[k] = run;
        }
    }
    __syncthreads();
    for (int64_t t = threadIdx.x; t < n_tiles; t += 256) {
        const long long lo = t * kSelRows, hi = lo + kSelRows;
        int a = 0, b = K - 1;                      // first group that ends behind lo
        while (a < b) {
            const int mid = (a + b) >> 1;
            if (s_end[mid] > lo) b = mid; else a = mid + 1;
        }
        int best = a;
        long long most = -1;
        for (int k = a; k < K; ++k) {
            const long long start = k == 0 ? 0 : s_end[k - 1];
            if (start >= hi) break;
            const long long ov = (s_end[k] < hi ? s_end[k] : hi) - (start > lo ? start : lo);
            if (ov > most) {
                most = ov;
                best = k;
            }
        }
        tile_ref[t] = best;
    }
}

// ---- the GEMM's block and its comparison, shared by the two kernels below ------------------------------------------------------
// Rows on the MFMA's M axis (A operand: a row's digits, lane (r, h) holds features 32 it + 16 h + (0..15) of row r), components on
// its N axis (B operand: straight from the reference's image in global memory - 1 KB per wave and load, L2 resident: the
// tiles of a component follow one another).  Accumulator register g of lane (c, h) is row (g & 3) + 8 (g >> 2) + 4 h of the
// block, column c: a ballot per register is the word of two rows (low half: h = 0, high half: h = 1) for the block's 32 components.
template <int T32>
__device__ __forceinline__ void proj_load_rows(const unsigned char* __restrict__ xq, int64_t row, int h, pj_i4v (&xd)[kProjDigits][T32]) {
    const unsigned char* xr = xq + row * proj_digit_row_bytes(T32) + 16 * h;
#pragma unroll
    for (int a = 0; a < kProjDigits; ++a)
#pragma unroll
        for (int it = 0; it < T32; ++it) xd[a][it] = *reinterpret_cast<const pj_i4v*>(xr + a * 32 * T32 + 32 * it);
}
template <int T32>
__device__ __forceinline__ void proj_block(const unsigned char* __restrict__ gimg_j, int kb, int lane,
                                           const pj_i4v (&xd)[kProjDigits][T32], pj_i16v& acc0, pj_i16v& acc1) {
    pj_i4v g0[T32], g1[T32];
    const unsigned char* gp = gimg_j + (int64_t)kb * T32 * kProjDigits * 1024 + lane * 16;
#pragma unroll
    for (int it = 0; it < T32; ++it) {
        g0[it] = *reinterpret_cast<const pj_i4v*>(gp + (it * kProjDigits) * 1024);
        g1[it] = *reinterpret_cast<const pj_i4v*>(gp + (it * kProjDigits + 1) * 1024);
    }
    acc0 = pj_i16v{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    acc1 = acc0;
#pragma unroll
    for (int it = 0; it < T32; ++it) {
        acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(xd[0][it], g0[it], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(xd[0][it], g1[it], acc1, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(xd[1][it], g0[it], acc1, 0, 0, 0);
    }
}
// |t| = |128 acc0 + acc1| stays below this (digits in [-64.5, 64.5], 32 T32 features, two kept classes)
template <int T32>
__device__ __forceinline__ constexpr float proj_tmax() { return 64.5f * 64.5f * 128.0f * (float)(32 * T32) * 1.01f; }
// not cleared: ln rho_nk <= A - R (C t - E) is not below the row's threshold w, i.e. NOT  R (C t - E) + w > A'  where A' carries
// 2^-20 of every magnitude that goes into the comparison (two fmas and the conversion of t; NaN: not cleared)
__device__ __forceinline__ bool proj_left(int a0, int a1, float4 cc, float R, float w, float A) {
    const float tf = (float)((a0 << 7) + a1);
    const float v1 = fmaf(cc.x, tf, -cc.y);
    return !(fmaf(R, v1, w) > A);
}

// The sweep.  Thread = row before and after the GEMM (like rec_sweep_kernel: reference value, threshold, record, masks);
// in between wave w takes rows 64 w .. 64 w + 63 of the tile as two blocks of 32 MFMA rows against the tile's image in LDS.
// Accumulator register g of lane (c, h) is row (g & 3) + 8 (g >> 2) + 4 h of the block, column c: a ballot per register is
// the candidate word of two rows (low half: h = 0, high half: h = 1) for the block's 32 components.
template <int T32>
__global__ __launch_bounds__(kSelRows) void rec_project_kernel(
    const unsigned char* __restrict__ xq, const signed char* __restrict__ xqe,
    const unsigned char* __restrict__ gimg, const float4* __restrict__ gconst, const int* __restrict__ tile_ref,
    const double* __restrict__ u /*ln rho: exact values of the pairs evaluated before the sweep, lower bounds of own pairs*/,
    int64_t npad, int64_t n_rows, int K, const double* __restrict__ drift, const double* __restrict__ c_new, RecArrays rec,
    unsigned long long* __restrict__ masks, int* __restrict__ blk_cnt, double* __restrict__ epart, double* __restrict__ opart,
    unsigned char* __restrict__ lock, float* __restrict__ dlock, float* __restrict__ rthr, const unsigned char* __restrict__ lcomp,
    unsigned long long* __restrict__ pmask, int* __restrict__ pblk, int proof_all, int own_fresh,
    unsigned long long* __restrict__ cand_ctr /*+= pairs the table did not clear (diagnostics), or null*/) {
    __shared__ int wcnt[4][256];
    __shared__ int pcnt[4][256];
    __shared__ float2 sq[256];          // Gamma (1 + 1e-6) up, c' down (settled rows)
    __shared__ double sc[256];
    __shared__ unsigned char sfirst[256];
    __shared__ float s_delta[256];
    __shared__ int wsum[3][4];
    __shared__ __attribute__((aligned(16))) float s_R[256], s_w[256];
    __shared__ unsigned s_m32[256][9];                   // candidate words per row and column block (K <= 256: 8 blocks)
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int W = (K + 63) / 64, KB = proj_kblocks(K);
    const int64_t tile0 = (int64_t)blockIdx.x * kSelRows;
    const int64_t n = tile0 + tid;
    const bool valid = n < n_rows;
    const int jref = tile_ref[blockIdx.x];
    // ---- per row: what rec_sweep_kernel<PREV> reads before its first barrier ----------------------------------------------
    unsigned long long fresh[4] = {0ull, 0ull, 0ull, 0ull};
    unsigned lk_pre = 0u;
    int kset_pre = 0;
    float dl_pre = 0.0f;
    double lb_pre = 0.0;
    const int64_t nn = valid ? n : n_rows - 1;
    if (valid) {
        for (int w = 0; w < W; ++w) fresh[w] = masks[(int64_t)w * npad + n];
        if (lock != nullptr) {
            lk_pre = lock[n];
            kset_pre = lcomp[n];
            dl_pre = dlock[n];
            if (own_fresh && lk_pre == 1u) lb_pre = u[(int64_t)kset_pre * npad + n];
        }
    }
    const int en_row = xqe[nn];
    for (int k = lane; k < K; k += 64) wcnt[wave][k] = pcnt[wave][k] = 0;
    for (int k = tid; k < K; k += kSelRows) {
        sfirst[k] = (own_fresh && own_first(drift[3 * K + k], drift[K + k])) ? 1 : 0;
        s_delta[k] = f32_up(drift[K + k]);
        sq[k] = make_float2(f32_up(drift[3 * K + k] * (1.0 + 1e-6)), f32_down(c_new[k]));
        sc[k] = c_new[k];
    }
#pragma unroll
    for (int q = 0; q < 9; ++q) s_m32[tid][q] = 0u;
    const double ninf = -__builtin_huge_val();
    double vb = ninf;
    if (valid) {
        for (int w = 0; w < W; ++w) {
            unsigned long long mm = fresh[w];
            while (mm) {
                const int b = __builtin_ctzll(mm);
                mm &= mm - 1;
                const double v = u[(int64_t)(64 * w + b) * npad + n];
                vb = (v > vb || v != v) ? v : vb;
            }
        }
    }
    __syncthreads();
    // ---- reference value and threshold (as in rec_sweep_kernel) ----------------------------------------------------------
    unsigned long long mk[4] = {0ull, 0ull, 0ull, 0ull}, pm[4] = {0ull, 0ull, 0ull, 0ull};
    unsigned long long nocand[4] = {0ull, 0ull, 0ull, 0ull};
    int listed = 0, over_i = 0, left_i = 0;
    bool by_bound = false, over = false;
    int kset = -1;
    float d_set = 0.0f, thr_f = __builtin_huge_valf();
    if (valid) {
        const double thr = vb - kRelNats;
        by_bound = lock != nullptr && lk_pre == 1u && (fresh[0] | fresh[1] | fresh[2] | fresh[3]) == 0ull;
        float thr_set = 0.0f;
        if (by_bound) {
            kset = kset_pre;
            float dn;
            if (sfirst[kset]) {
                dn = f32_up(dist_of(sc[kset], lb_pre) * (1.0 + 1e-9));
            } else {
                dn = fmaf(sq[kset].x, dl_pre, s_delta[kset]) * (1.0f + 2.4e-7f);
            }
            const float lb = sq[kset].y - dn * dn * 0.5000005f;
            d_set = dn;
            thr_set = (lb - fabsf(lb) * 2.4e-7f) - ((float)kRelevanceNats + 0.2f);
        }
        over = !by_bound && !(thr > ninf);
        thr_f = by_bound ? thr_set : (over ? -__builtin_huge_valf() : f32_down(thr));
        if (over) fresh[0] = fresh[1] = fresh[2] = fresh[3] = 0ull;
        rthr[n] = thr_f;
#pragma unroll
        for (int w = 0; w < 4; ++w) nocand[w] = fresh[w];
        if (by_bound) nocand[kset >> 6] |= 1ull << (kset & 63);
    }
    {
        // what the GEMM's epilogue needs per row, and the tile's largest magnitudes for its f32 slack
        const bool digits = en_row != (int)kProjNoDigits;
        const float R = digits ? ldexpf(1.0f, en_row) : __builtin_nanf("");      // NaN: no bound for this row
        const float wv = thr_f == thr_f ? thr_f : -__builtin_huge_valf();
        s_R[tid] = valid ? R : 0.0f;
        s_w[tid] = valid ? wv : __builtin_huge_valf();                            // (rows past the end: never a candidate)
    }
    __syncthreads();
    // ---- the GEMM ----------------------------------------------------------------------------------------------------------
    {
        const int c = lane & 31, h = lane >> 5;
        const float4* gc = gconst + (int64_t)jref * KB * 32;
        const unsigned char* gimg_j = gimg + (int64_t)jref * KB * T32 * kProjDigits * 1024;
        for (int rb = 0; rb < 2; ++rb) {
            const int r0 = 64 * wave + 32 * rb;
            if (tile0 + r0 >= n_rows) break;                                            // (wave uniform)
            int64_t row_a = tile0 + r0 + c;
            if (row_a >= n_rows) row_a = n_rows - 1;
            pj_i4v xd[kProjDigits][T32];
            proj_load_rows<T32>(xq, row_a, h, xd);
            // this lane's sixteen rows: rows r0 + 4 h + 8 q + (0..3), q = 0..3
            float Rg[16], wg[16];
            float Rm = 0.0f, wm = 0.0f;               // their largest magnitudes, for the f32 slack (NaN / inf rows are candidates anyway)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f4 a = *reinterpret_cast<const f4*>(&s_R[r0 + 4 * h + 8 * q]);
                const f4 b = *reinterpret_cast<const f4*>(&s_w[r0 + 4 * h + 8 * q]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    Rg[4 * q + e] = a[e];
                    wg[4 * q + e] = b[e];
                    Rm = fmaxf(Rm, a[e]);
                    wm = fmaxf(wm, fabsf(b[e]) < 3.0e38f ? fabsf(b[e]) : 0.0f);
                }
            }
            for (int kb = 0; kb < KB; ++kb) {
                const float4 cc = gc[32 * kb + c];
                pj_i16v acc0, acc1;
                proj_block<T32>(gimg_j, kb, lane, xd, acc0, acc1);
                const float A = cc.w + 9.6e-7f * (Rm * fmaf(cc.x, proj_tmax<T32>(), cc.y) + wm + fabsf(cc.w));
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    const unsigned long long bal = __builtin_amdgcn_ballot_w64(proj_left(acc0[g], acc1[g], cc, Rg[g], wg[g], A));
                    if (bal != 0ull && lane == 0) {
                        const int row = r0 + (g & 3) + 8 * (g >> 2);
                        s_m32[row][kb] = (unsigned)bal;
                        s_m32[row + 4][kb] = (unsigned)(bal >> 32);
                    }
                }
            }
        }
    }
    __syncthreads();
    // ---- per row again: candidates, proof masks, record, counts ---------------------------------------------------------------
    bool stays = false;
    if (valid) {
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            if (w < W) {
                unsigned long long word = (unsigned long long)s_m32[tid][2 * w] | ((unsigned long long)s_m32[tid][2 * w + 1] << 32);
                const int left = K - 64 * w;
                if (left < 64) word &= (1ull << left) - 1ull;
                mk[w] = word & ~nocand[w];
                left_i += __builtin_popcountll(mk[w]);
            }
        }
        bool proof_row = false, proof_cand = false;
        if (pmask != nullptr && proof_all && !by_bound && !over && (mk[0] | mk[1] | mk[2] | mk[3]) != 0ull) {
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                pm[w] = mk[w];
                mk[w] = 0ull;
            }
            proof_cand = true;
        }
        if (by_bound) {
            stays = (mk[0] | mk[1] | mk[2] | mk[3]) == 0ull;
            if (stays) {
                dlock[n] = d_set;
            } else if (pmask != nullptr) {
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    pm[w] = mk[w];
                    mk[w] = 0ull;
                }
                if (!sfirst[kset]) pm[kset >> 6] |= 1ull << (kset & 63);
                proof_row = true;
            } else {
                mk[kset >> 6] |= 1ull << (kset & 63);
                rthr[n] = -__builtin_huge_valf();
            }
        }
        int in_slots = 0;
        if (!stays) {
            // the record: the exact pairs take the slots (largest value first); every pair that is neither exact nor listed lies
            // below the row's threshold - that IS the rest bound.  Candidates get no slot: rec_finish_kernel rebuilds the
            // record of a row with listed pairs outside its slots from their exact values ("refreshed row").
            unsigned s[kRecSlots + 1];
#pragma unroll
            for (int j = 0; j <= kRecSlots; ++j) s[j] = 0xFFFFFFFFu;
            if (!by_bound) {
                for (int w = 0; w < W; ++w) {
                    unsigned long long mm = fresh[w];
                    while (mm) {
                        const int b = __builtin_ctzll(mm);
                        mm &= mm - 1;
                        const int k = 64 * w + b;
                        sweep_chain(s, sweep_key(f32_up(u[(int64_t)k * npad + n]), (unsigned)k));
                    }
                }
            }
            unsigned ex = 0;
            unsigned long long in_slot[4] = {0ull, 0ull, 0ull, 0ull};
            float ds[kRecSlots];
            unsigned short ks[kRecSlots];
#pragma unroll
            for (int j = 0; j < kRecSlots; ++j) {
                ks[j] = kRecEmpty;
                ds[j] = __builtin_huge_valf();
                if (s[j] == 0xFFFFFFFFu) continue;
                const int k = (int)(s[j] & 0xFFu);
                ks[j] = (unsigned short)k;
                in_slot[k >> 6] |= 1ull << (k & 63);
                ex |= 1u << j;
                ds[j] = f32_down(dist_of(sc[k], u[(int64_t)k * npad + n]));
            }
            for (int w = 0; w < W; ++w) {
                mk[w] |= fresh[w] & ~in_slot[w];              // exact pairs without a slot are listed again
                listed += __builtin_popcountll(mk[w]);
            }
#pragma unroll
            for (int j = 0; j < kRecSlots; ++j) {
                rec.k[(int64_t)j * rec.npad + n] = ks[j];
                rec.d[(int64_t)j * rec.npad + n] = ds[j];
            }
            rec.B[n] = over ? -__builtin_huge_valf() : thr_f;
            rec.exact[n] = (unsigned char)(over ? 0 : ex);
            rec.sel[n] = 0;
        }
        if (over) {
            for (int w = 0; w < W; ++w) mk[w] = (K - 64 * w >= 64) ? ~0ull : ((1ull << (K - 64 * w)) - 1ull);
            listed = K;
        }
        rec.flags[n] = (unsigned char)(stays ? 4 : (proof_row ? 8 : (over ? 1 : ((listed > in_slots ? 2 : 0) | (proof_cand ? 16 : 0)))));
        over_i = over ? 1 : 0;
        for (int w = 0; w < W; ++w) {
            masks[(int64_t)w * npad + n] = mk[w];
            if (pmask) pmask[(int64_t)w * npad + n] = pm[w];
        }
    }
    for (int w = 0; w < W; ++w) {
        count_word(mk[w], w, wave, wcnt);
        if (pmask) count_word(pm[w], w, wave, pcnt);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        listed += __shfl_xor(listed, o);
        over_i += __shfl_xor(over_i, o);
        left_i += __shfl_xor(left_i, o);
    }
    if (lane == 0) {
        wsum[0][wave] = listed;
        wsum[1][wave] = over_i;
        wsum[2][wave] = left_i;
    }
    __syncthreads();
    for (int k = tid; k < K; k += kSelRows) {
        blk_cnt[blk_at(k, blockIdx.x, K)] = wcnt[0][k] + wcnt[1][k] + wcnt[2][k] + wcnt[3][k];
        if (pmask) pblk[blk_at(k, blockIdx.x, K)] = pcnt[0][k] + pcnt[1][k] + pcnt[2][k] + pcnt[3][k];
    }
    if (tid == 0) {
        epart[blockIdx.x] = (double)(wsum[0][0] + wsum[0][1] + wsum[0][2] + wsum[0][3]);
        opart[blockIdx.x] = (double)(wsum[1][0] + wsum[1][1] + wsum[1][2] + wsum[1][3]);
        const int left = wsum[2][0] + wsum[2][1] + wsum[2][2] + wsum[2][3];
        if (cand_ctr && left) atomicAdd(cand_ctr, (unsigned long long)left);
    }
}

// The table as a FILTER in front of the proof round.  The carried sweep (records.h) has just listed, for the
// int8 proof round, the pairs whose carried bound no longer clears the row's threshold - most of them far pairs whose bound
// has eroded over the parameter updates.  A pair that the table of the parameters in force clears needs no proof: it is
// taken off the proof lists (a settled row that loses all its candidates stays settled as it is).  Work only where the
// sweep left proof pairs: a block of 32 rows x 32 components runs its MFMAs only if it holds one.  (Measured and lost,
// profiles/r6_experiments.md: the image staged in LDS per tile; waves owning column blocks with the tile's digit planes in
// LDS; a thread per row walking its listed pairs with v_dot4_i32_i8 - scattered 16-byte loads of the g_jk digits per pair.)
// Reads rthr / flags / pmask as the sweep wrote them (the sweep stores the settled rows' new distance bound in dlock for
// proof rows too); rewrites pmask, pblk and the flags of rows whose candidates are all gone.
template <int T32>
__global__ __launch_bounds__(kSelRows) void proj_filter_kernel(
    const unsigned char* __restrict__ xq, const signed char* __restrict__ xqe,
    const unsigned char* __restrict__ gimg, const float4* __restrict__ gconst, const int* __restrict__ tile_ref, int64_t npad,
    int64_t n_rows, int K, const float* __restrict__ rthr, RecArrays rec, const unsigned long long* __restrict__ masks,
    unsigned long long* __restrict__ pmask, int* __restrict__ pblk, const unsigned char* __restrict__ lcomp,
    unsigned long long* __restrict__ removed_ctr /*+= proof pairs the table made unnecessary, or null*/) {
    __shared__ int pcnt[4][256];
    __shared__ __attribute__((aligned(16))) float s_R[256], s_w[256];
    __shared__ unsigned s_c32[256][9];                   // the sweep's candidates per row and column block
    __shared__ unsigned s_m32[256][9];                   // ... of which the table does not clear
    __shared__ int s_any[4], s_rem[4];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int W = (K + 63) / 64, KB = proj_kblocks(K);
    const int64_t tile0 = (int64_t)blockIdx.x * kSelRows;
    const int64_t n = tile0 + tid;
    const bool valid = n < n_rows;
    const int64_t nn = valid ? n : n_rows - 1;
    // ---- per row: its proof candidates ---------------------------------------------------------------------------------
    unsigned long long pm[4] = {0ull, 0ull, 0ull, 0ull}, cand[4] = {0ull, 0ull, 0ull, 0ull};
    unsigned fl = 0u;
    int kset = -1;
    if (valid) {
        fl = rec.flags[n];
        if (fl == 8u || (fl & 16u)) {
            for (int w = 0; w < W; ++w) cand[w] = pm[w] = pmask[(int64_t)w * npad + n];
            if (fl == 8u) {
                kset = lcomp[n];
                cand[kset >> 6] &= ~(1ull << (kset & 63));               // (the row's own pair is no candidate)
            }
        }
    }
    const bool has = (cand[0] | cand[1] | cand[2] | cand[3]) != 0ull;
    {
        const int en_row = xqe[nn];
        const bool digits = en_row != (int)kProjNoDigits;
        s_R[tid] = (has && digits) ? ldexpf(1.0f, en_row) : (has ? __builtin_nanf("") : 0.0f);
        s_w[tid] = has ? rthr[n] : __builtin_huge_valf();                 // (rows without candidates: cleared whatever comes)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            s_c32[tid][2 * q] = (unsigned)cand[q];
            s_c32[tid][2 * q + 1] = (unsigned)(cand[q] >> 32);
        }
        s_c32[tid][8] = 0u;
#pragma unroll
        for (int q = 0; q < 9; ++q) s_m32[tid][q] = 0xFFFFFFFFu;          // (a block that is not computed clears nothing)
    }
    for (int k = lane; k < K; k += 64) pcnt[wave][k] = 0;
    const unsigned long long anyb = __builtin_amdgcn_ballot_w64(has);
    if (lane == 0) s_any[wave] = anyb != 0ull;
    __syncthreads();
    const bool tile_any = (s_any[0] | s_any[1] | s_any[2] | s_any[3]) != 0;
    if (!tile_any) return;                                   // nothing listed in this tile: pmask / pblk stay as the sweep wrote them
    const int jref = tile_ref[blockIdx.x];
    // ---- the GEMM, block by block where the sweep left candidates ----------------------------------------------------------
    {
        const int c = lane & 31, h = lane >> 5;
        const float4* gc = gconst + (int64_t)jref * KB * 32;
        const unsigned char* gimg_j = gimg + (int64_t)jref * KB * T32 * kProjDigits * 1024;
        for (int rb = 0; rb < 2; ++rb) {
            const int r0 = 64 * wave + 32 * rb;
            if (tile0 + r0 >= n_rows) break;
            // column blocks in which one of the block's 32 rows has a candidate (lanes c and c + 32 look at row r0 + c)
            unsigned need = 0u;
            for (int kb = 0; kb < KB; ++kb)
                if (__builtin_amdgcn_ballot_w64(s_c32[r0 + c][kb] != 0u) != 0ull) need |= 1u << kb;
            if (need == 0u) continue;
            int64_t row_a = tile0 + r0 + c;
            if (row_a >= n_rows) row_a = n_rows - 1;
            pj_i4v xd[kProjDigits][T32];
            proj_load_rows<T32>(xq, row_a, h, xd);
            float Rg[16], wg[16];
            float Rm = 0.0f, wm = 0.0f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f4 a = *reinterpret_cast<const f4*>(&s_R[r0 + 4 * h + 8 * q]);
                const f4 b = *reinterpret_cast<const f4*>(&s_w[r0 + 4 * h + 8 * q]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    Rg[4 * q + e] = a[e];
                    wg[4 * q + e] = b[e];
                    Rm = fmaxf(Rm, a[e]);
                    wm = fmaxf(wm, fabsf(b[e]) < 3.0e38f ? fabsf(b[e]) : 0.0f);
                }
            }
            for (int kb = 0; kb < KB; ++kb) {
                if (!((need >> kb) & 1u)) continue;                                       // (wave uniform)
                const float4 cc = gc[32 * kb + c];
                pj_i16v acc0, acc1;
                proj_block<T32>(gimg_j, kb, lane, xd, acc0, acc1);
                const float A = cc.w + 9.6e-7f * (Rm * fmaf(cc.x, proj_tmax<T32>(), cc.y) + wm + fabsf(cc.w));
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    const unsigned long long bal = __builtin_amdgcn_ballot_w64(proj_left(acc0[g], acc1[g], cc, Rg[g], wg[g], A));
                    if (lane == 0) {
                        const int row = r0 + (g & 3) + 8 * (g >> 2);
                        s_m32[row][kb] = (unsigned)bal;
                        s_m32[row + 4][kb] = (unsigned)(bal >> 32);
                    }
                }
            }
        }
    }
    __syncthreads();
    // ---- per row: what is left of its proof candidates -----------------------------------------------------------------------
    int removed = 0;
    if (valid && has) {
        unsigned long long left[4] = {0ull, 0ull, 0ull, 0ull};
        bool any = false;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            if (w < W) {
                const unsigned long long word = (unsigned long long)s_m32[tid][2 * w] | ((unsigned long long)s_m32[tid][2 * w + 1] << 32);
                left[w] = cand[w] & word;
                removed += __builtin_popcountll(cand[w]) - __builtin_popcountll(left[w]);
                any = any || left[w] != 0ull;
            }
        }
        if (removed > 0) {
            if (!any) {
                if (fl == 8u) {
                    removed += __builtin_popcountll(pm[kset >> 6] & (1ull << (kset & 63)));      // (its own pair needs no bound either)
                    rec.flags[n] = 4;                                 // stays settled; dlock holds the sweep's new distance bound
                } else {
                    int total = 0;
                    for (int w = 0; w < W; ++w) total += __builtin_popcountll(masks[(int64_t)w * npad + n]);
                    rec.flags[n] = (unsigned char)(total > __builtin_popcount((unsigned)rec.sel[n]) ? 2 : 0);
                }
                for (int w = 0; w < W; ++w) pm[w] = 0ull;
            } else {
                for (int w = 0; w < W; ++w) pm[w] = left[w] | (pm[w] & ~cand[w]);      // (a settled row's own pair stays listed)
            }
            for (int w = 0; w < W; ++w) pmask[(int64_t)w * npad + n] = pm[w];
        }
    }
    for (int w = 0; w < W; ++w) count_word(pm[w], w, wave, pcnt);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) removed += __shfl_xor(removed, o);
    if (lane == 0) s_rem[wave] = removed;
    __syncthreads();
    for (int k = tid; k < K; k += kSelRows) pblk[blk_at(k, blockIdx.x, K)] = pcnt[0][k] + pcnt[1][k] + pcnt[2][k] + pcnt[3][k];
    if (tid == 0 && removed_ctr) {
        const int r = s_rem[0] + s_rem[1] + s_rem[2] + s_rem[3];
        if (r) atomicAdd(removed_ctr, (unsigned long long)r);
    }
}

}  // namespace gmmvb
