// K-sized packing, row log-normaliser, slab reduction and read-out kernels (all f64, bandwidth-trivial).
// (The non-template kernels here have internal linkage: several translation units of the C ABI - capi*.hip - include this file.)
#pragma once
#include "common.h"
#include "rec_common.h"

namespace gmmvb {

// u [K][D][D] lower triangular (y = u d)  ->  per-component parameter image (layout: estep.h)
//   [P][half][lane][2] tiles of u (zero padded to 16T) | [T][g][r] bias = -(u m) | zero pad to 1 KB
// (also copies c -> cvec and, if asked, the pivot the int8 images are packed about: two device-to-device copies less per
// parameter hand-over, each of which was a dispatch on the iteration's critical path)
static __global__ void pack_params_kernel(const double* __restrict__ u, const double* __restrict__ m, int K, int D,
                                   int T, int img_len, double* __restrict__ img, const double* __restrict__ c_src = nullptr,
                                   double* __restrict__ c_dst = nullptr, const double* __restrict__ pivot_src = nullptr,
                                   double* __restrict__ pivot_dst = nullptr) {
    const int k = blockIdx.x;
    if (c_dst && threadIdx.x == 0) c_dst[k] = c_src[k];
    if (pivot_dst && k == 0)
        for (int f = threadIdx.x; f < D; f += blockDim.x) pivot_dst[f] = pivot_src[f];
    const int P = tri_pairs(T);
    const double* uk = u + (int64_t)k * D * D;
    double* im = img + (int64_t)k * img_len;
    for (int e = threadIdx.x; e < P * 256; e += blockDim.x) {
        const int p = e >> 8, h = (e >> 7) & 1, lane = (e >> 1) & 63, ee = e & 1;
        int jt = 0;
        while (tri_pairs(jt + 1) <= p) ++jt;
        const int b = p - tri_pairs(jt);
        const int jj = 16 * jt + (lane & 15), ii = 16 * b + 4 * (lane >> 4) + 2 * h + ee;
        im[e] = (jj < D && ii < D && ii <= jj) ? uk[(int64_t)jj * D + ii] : 0.0;
    }
    const double* mk = m + (int64_t)k * D;
    for (int jj = threadIdx.x; jj < 16 * T; jj += blockDim.x) {
        double s = 0.0;
        if (jj < D)
            for (int ii = 0; ii <= jj; ++ii) s = fma(uk[(int64_t)jj * D + ii], mk[ii], s);
        const int jt = jj >> 4, w = jj & 15, g = w & 3, r = w >> 2;   // accumulator row = g + 4 r
        im[P * 256 + (jt * 4 + g) * 4 + r] = -s;
    }
    for (int e = P * 256 + 16 * T + threadIdx.x; e < img_len; e += blockDim.x) im[e] = 0.0;
}

// xc[n][f] = (double)x[n][f] - pivot[f] for f < D, 0 for D <= f < Dp (Dp = 16T): the M-step's operand,
// made once per sample matrix so its inner loop carries no convert / subtract / mask work.
// Rows n_rows .. pad_rows-1 are zero filled: the M-step prefetches a few rows past its range without a clamp.
template <typename XT>
__global__ void center_rows_kernel(const XT* __restrict__ x, int64_t ldx, int64_t n_rows, int64_t pad_rows, int D,
                                   int Dp, const double* __restrict__ pivot, double* __restrict__ xc) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= pad_rows * Dp) return;
    const int64_t n = e / Dp;
    const int f = (int)(e - n * Dp);
    xc[e] = (n < n_rows && f < D) ? (double)x[n * ldx + f] - pivot[f] : 0.0;
}

// lse[n] = ln sum_k exp(lnrho[k][n]) (single pass, running max), and per block of kLseRows rows the largest
// ln r_nk = lnrho[k][n] - lse[n] of every component (dpart[block][k]); thr_kernel turns those into the M-step's
// skip thresholds.
constexpr int kLseRows = 1024;      // rows per block (256 threads x 4)
// apart[block] = number of (row, component) pairs of the block with ln r >= -80 ln 2 (how sparse r is: the next
// E-step prunes only when that fraction is small).
// `stride` > 1: only every stride-th block of rows is visited (a sample of the rows: its maxima are lower bounds of the
// true ones, which is all a skip threshold needs; see lse_mask_kernel).
static __global__ __launch_bounds__(256) void row_lse_kernel(const double* __restrict__ lnrho, int64_t npad, int64_t n_rows,
                                                      int K, double* __restrict__ lse, double* __restrict__ dpart,
                                                      double* __restrict__ apart, int stride) {
    __shared__ double wmax[4];
    __shared__ int wcnt[4];
    int active = 0;
    const int64_t base = (int64_t)blockIdx.x * stride * kLseRows + threadIdx.x;
    double l[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int64_t n = base + 256 * q;
        l[q] = 0.0;
        if (n >= n_rows) continue;
        double mx = lnrho[n], s = 1.0;
        for (int k = 1; k < K; ++k) {
            const double v = lnrho[(int64_t)k * npad + n];
            if (v > mx) {
                s = fma(s, exp(mx - v), 1.0);
                mx = v;
            } else {
                s += exp(v - mx);
            }
        }
        l[q] = mx + log(s);
        lse[n] = l[q];
    }
    if (!dpart) return;
    const double ninf = -__builtin_huge_val();
    for (int k = 0; k < K; ++k) {
        double d = ninf;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t n = base + 256 * q;
            if (n < n_rows) {
                const double v = lnrho[(int64_t)k * npad + n] - l[q];
                d = (v > d || v != v) ? v : d;          // NaN wins: the threshold becomes NaN = nothing is skipped
                active += !(v < -kRelevanceNats);
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double v = __shfl_xor(d, o);
            d = (v > d || v != v) ? v : d;
        }
        if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = d;
        __syncthreads();
        if (threadIdx.x == 0) {
#pragma unroll
            for (int w = 1; w < 4; ++w) d = (wmax[w] > d || wmax[w] != wmax[w]) ? wmax[w] : d;
            dpart[(int64_t)blockIdx.x * K + k] = d;
        }
        __syncthreads();
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) active += __shfl_xor(active, o);
    if ((threadIdx.x & 63) == 0) wcnt[threadIdx.x >> 6] = active;
    __syncthreads();
    if (threadIdx.x == 0 && apart) apart[blockIdx.x] = (double)(wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3]);
}

// thr[k] = max over blocks of dpart[.][k] - 80 ln 2 (mstep.h, sparse responsibilities); act_total = sum of apart.
// One 256-thread workgroup per component (workgroup K sums apart).
static __global__ __launch_bounds__(256) void thr_kernel(const double* __restrict__ dpart, const double* __restrict__ apart,
                                                  int blocks, int K, double* __restrict__ thr,
                                                  double* __restrict__ act_total) {
    __shared__ double part[256];
    const int k = blockIdx.x;
    if (k == K) {
        if (!apart) return;
        double a = 0.0;
        for (int b = threadIdx.x; b < blocks; b += 256) a += apart[b];
        part[threadIdx.x] = a;
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = 0.0;
            for (int i = 0; i < 256; ++i) t += part[i];      // fixed order
            *act_total = t;
        }
        return;
    }
    if (!dpart) return;
    double d = -__builtin_huge_val();
    for (int b = threadIdx.x; b < blocks; b += 256) {
        const double v = dpart[(int64_t)b * K + k];
        d = (v > d || v != v) ? v : d;
    }
    part[threadIdx.x] = d;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 256; ++i) d = (part[i] > d || part[i] != part[i]) ? part[i] : d;
        thr[k] = d - kRelevanceNats;
    }
}

// ---- pruned E-step (estep.h): candidate selection --------------------------------------------------------------
// Three small passes build the per-component sample lists without atomics (and in a fixed order):
//   select_mask_kernel   one thread per sample: which components are candidates (bit mask, 64 components per word),
//                        and how many candidates each 256-sample block has per component;
//   scan_parts / scan_apply   exclusive scan of those block counts per component -> block bases and list lengths;
//   fill_lists_kernel    every block writes its candidates at its bases.
// MODE 0 (best)  : the mask holds khat_n = first maximiser of the bounds u[k][n] (also stored in khat).
// MODE 3 (given) : like MODE 0 with khat already written (by the bound kernel): u is not read.
// (Every further candidate is selected from the per-row records, records.h.)
template <int MODE>
__global__ __launch_bounds__(kSelRows) void select_mask_kernel(const double* __restrict__ u, int64_t npad, int64_t n_rows,
                                                               int K, int* __restrict__ khat,
                                                               unsigned long long* __restrict__ masks /*[W][npad]*/,
                                                               int* __restrict__ blk_cnt /*[blocks][K]*/) {
    static_assert(MODE == 0 || MODE == 3, "only the best-component modes exist");
    __shared__ int wcnt[4][256];
    const int64_t n = (int64_t)blockIdx.x * kSelRows + threadIdx.x;
    const bool valid = n < n_rows;
    const int W = (K + 63) / 64;
    const int wave = threadIdx.x >> 6;
    for (int k = threadIdx.x & 63; k < K; k += 64) wcnt[wave][k] = 0;
    int kh = -1;
    if (MODE == 3 && valid) kh = khat[n];
    if (MODE == 0) {
        int arg = 0;
        if (valid) {
            double best = u[n];
            for (int k = 1; k < K; ++k) {
                const double v = u[(int64_t)k * npad + n];
                const bool up = v > best;
                best = up ? v : best;
                arg = up ? k : arg;
            }
            khat[n] = arg;
        }
        kh = arg;
    }
    for (int w = 0; w < W; ++w) {
        unsigned long long mk = 0;
        if (valid) {
            if ((kh >> 6) == w) mk = 1ull << (kh & 63);
            masks[(int64_t)w * npad + n] = mk;
        }
        unsigned long long present = wave_or(mk);
        while (present) {
            const int b = __builtin_ctzll(present);
            present &= present - 1;
            const int c = __builtin_popcountll(__ballot((mk >> b) & 1ull));
            if ((threadIdx.x & 63) == 0) wcnt[wave][64 * w + b] = c;
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < K; k += kSelRows)
        blk_cnt[blk_at(k, blockIdx.x, K)] = wcnt[0][k] + wcnt[1][k] + wcnt[2][k] + wcnt[3][k];
}

// Exclusive scan of the block counts per component.  blk is BLOCK-major, blk[b][k] at b K + k (round 5; it was
// component-major): a selection kernel's workgroup then writes its K counts as one contiguous run instead of K scattered
// 4-byte stores a whole array row apart - at K = 256 those were 0.8 GB of 64-byte sectors per array and kernel, several
// arrays per pass -, fill_lists_kernel reads its K bases in one piece, and the scan streams whole rows.
// blk[b][k] <- sum of blk[b'][k] over b' < b; counts[k] = the total.  Three launches: sums of kScanParts parts (every part a
// contiguous range of ~40 blocks, all components at once), the exclusive scan of those sums per component, then every part
// scans its blocks from its base.  Thread t of a part's workgroup: component t % K, row t / K of 256 / K block rows.
constexpr int kScanParts = 1024;
constexpr int kScanChunk = 32;            // parts per thread of the middle kernel (kScanParts / kScanChunk chunks per component)

// sums of the parts: part p = blocks [p per, (p + 1) per), all components
static __global__ __launch_bounds__(256) void scan_parts_kernel(const int* __restrict__ blk, int blocks, int K, int* __restrict__ parts) {
    __shared__ int sh[256];
    const int p = blockIdx.x;
    const int R = K <= 256 ? 256 / K : 1;
    const int t = threadIdx.x, k = t % K, r = t / K;
    const int per = (blocks + kScanParts - 1) / kScanParts;
    const int b0 = p * per, b1 = b0 + per < blocks ? b0 + per : blocks;
    int s = 0;
    if (t < R * K)
        for (int b = b0 + r; b < b1; b += R) s += blk[blk_at(k, b, K)];
    sh[t] = s;
    __syncthreads();
    if (t < K) {
        int tot = 0;
        for (int q = 0; q < R; ++q) tot += sh[q * K + t];
        parts[p * K + t] = tot;
    }
}

// parts[p][k] <- sum over p' < p (in place), counts[k] = the total.  One workgroup per eight components: thread
// (component t % 8, chunk t / 8) takes kScanChunk consecutive parts.
static __global__ __launch_bounds__(256) void scan_mid_kernel(int* __restrict__ parts, int K, int* __restrict__ counts) {
    static_assert(kScanParts == 32 * kScanChunk, "256 threads = 8 components x 32 chunks");
    __shared__ int sh[32][8];
    const int t = threadIdx.x, kk = t & 7, c = t >> 3;
    const int k = blockIdx.x * 8 + kk;
    int v[kScanChunk];
    int s = 0;
#pragma unroll
    for (int i = 0; i < kScanChunk; ++i) {
        v[i] = k < K ? parts[(int64_t)(c * kScanChunk + i) * K + k] : 0;
        s += v[i];
    }
    sh[c][kk] = s;
    __syncthreads();
    int run = 0;
    for (int q = 0; q < c; ++q) run += sh[q][kk];
    if (k < K) {
#pragma unroll
        for (int i = 0; i < kScanChunk; ++i) {
            parts[(int64_t)(c * kScanChunk + i) * K + k] = run;
            run += v[i];
        }
        if (c == 31) counts[k] = run;
    }
}

// every part scans its blocks from its base; rows b0 + i R + r, r = 0 .. R - 1, go through the workgroup together
static __global__ __launch_bounds__(256) void scan_apply_kernel(int* __restrict__ blk, int blocks, int K, const int* __restrict__ parts) {
    __shared__ int sh[256];
    const int p = blockIdx.x;
    const int R = K <= 256 ? 256 / K : 1;
    const int t = threadIdx.x, k = t % K, r = t / K;
    const bool on = t < R * K;
    const int per = (blocks + kScanParts - 1) / kScanParts;
    const int b0 = p * per, b1 = b0 + per < blocks ? b0 + per : blocks;
    int run = on ? parts[p * K + k] : 0;
    if (R == 1) {
        // one thread per component walks the part's blocks: eight loads at a time, then their eight stores (the same
        // array: a load behind a store would wait for it)
        for (int c0 = b0; c0 < b1; c0 += 8) {
            int v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = (on && c0 + i < b1) ? blk[blk_at(k, c0 + i, K)] : 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (on && c0 + i < b1) blk[blk_at(k, c0 + i, K)] = run;
                run += v[i];
            }
        }
        return;
    }
    for (int c0 = b0; c0 < b1; c0 += R) {                 // (workgroup-uniform bounds)
        const int b = c0 + r;
        const int v = (on && b < b1) ? blk[blk_at(k, b, K)] : 0;
        __syncthreads();                                  // (sh of the previous round has been read)
        sh[t] = v;
        __syncthreads();
        if (on) {
            int before = 0, all = 0;
            for (int q = 0; q < R; ++q) {
                const int w = sh[q * K + k];
                before += q < r ? w : 0;
                all += w;
            }
            if (b < b1) blk[blk_at(k, b, K)] = run + before;
            run += all;
        }
    }
}

inline void launch_scan_counts(hipStream_t st, int* blk, int blocks, int K, int* counts, int* parts /*[kScanParts * K]*/) {
    hipLaunchKernelGGL(scan_parts_kernel, dim3(kScanParts), dim3(256), 0, st, blk, blocks, K, parts);
    hipLaunchKernelGGL(scan_mid_kernel, dim3((K + 7) / 8), dim3(256), 0, st, parts, K, counts);
    hipLaunchKernelGGL(scan_apply_kernel, dim3(kScanParts), dim3(256), 0, st, blk, blocks, K, parts);
}

// lock (delta lists of the cache of single-component rows, records.h rec_finish_kernel): a row that leaves its
// component's cache (state 2, or 4 in the list of a component that is not its best any more) is listed with the sign
// bit set; one that enters (state 3, or 4 in the list of its new component lcomp) without.  State afterwards: 0 / 1.
static __global__ __launch_bounds__(kSelRows) void fill_lists_kernel(const unsigned long long* __restrict__ masks, int64_t npad,
                                                              int64_t n_rows, int K, const int* __restrict__ blk_base,
                                                              int* __restrict__ lists, int64_t cap,
                                                              unsigned char* __restrict__ lock = nullptr,
                                                              const unsigned char* __restrict__ lcomp = nullptr,
                                                              const double* __restrict__ block_total = nullptr
                                                              /*[blocks] bits set in this block's masks, if the producer counted
                                                                them: a block without any has nothing to fill (round 6: most
                                                                blocks of a converged pass - settled rows - are such)*/) {
    if (block_total != nullptr && block_total[blockIdx.x] == 0.0) return;
    __shared__ int wcnt[4][256];
    const int64_t n = (int64_t)blockIdx.x * kSelRows + threadIdx.x;
    const bool valid = n < n_rows;
    const int W = (K + 63) / 64;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int k = lane; k < K; k += 64) wcnt[wave][k] = 0;
    // this block's base in component threadIdx.x's list (one contiguous run of K ints): requested together with the masks -
    // behind the count it was a second memory round trip per workgroup, and the kernel is 39 000 workgroups of two round trips
    const int base_pre = (int)threadIdx.x < K ? blk_base[blk_at((int)threadIdx.x, blockIdx.x, K)] : 0;
    unsigned long long mkw[4] = {0ull, 0ull, 0ull, 0ull}, prw[4] = {0ull, 0ull, 0ull, 0ull};      // (kept for the second pass)
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        if (w >= W) break;
        const unsigned long long mk = valid ? masks[(int64_t)w * npad + n] : 0ull;
        mkw[w] = mk;
        unsigned long long present = wave_or(mk);
        prw[w] = present;
        while (present) {
            const int b = __builtin_ctzll(present);
            present &= present - 1;
            const int c = __builtin_popcountll(__ballot((mk >> b) & 1ull));
            if (lane == 0) wcnt[wave][64 * w + b] = c;
        }
    }
    __syncthreads();
    // every component's base of this block, for each of the four waves: ONE batch of parallel loads per workgroup (a load
    // inside the loop below is a dependent memory round trip per component and wave: at K = 256 a wave sees dozens)
    for (int k = threadIdx.x; k < K; k += kSelRows) {
        const int c0 = wcnt[0][k], c1 = wcnt[1][k], c2 = wcnt[2][k], c3 = wcnt[3][k];
        if (c0 + c1 + c2 + c3 > 0) {
            const int base = k == (int)threadIdx.x ? base_pre : blk_base[blk_at(k, blockIdx.x, K)];
            wcnt[0][k] = base;
            wcnt[1][k] = base + c0;
            wcnt[2][k] = base + c0 + c1;
            wcnt[3][k] = base + c0 + c1 + c2;
        }
    }
    __syncthreads();
    const unsigned lk0 = (lock && valid) ? lock[n] : 0u;
    const int kh0 = (lock && valid) ? (int)lcomp[n] : 0;
    if (lk0 >= 2u) lock[n] = lk0 == 2u ? 0 : 1;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        if (w >= W) break;
        const unsigned long long mk = mkw[w];
        unsigned long long present = prw[w];
        while (present) {
            const int b = __builtin_ctzll(present);
            present &= present - 1;
            const int k = 64 * w + b;
            const unsigned long long bal = __ballot((mk >> b) & 1ull);
            const int off = wcnt[wave][k];
            if ((mk >> b) & 1ull) {
                int entry = (int)n;
                if (lock) {
                    const bool leaves = lk0 == 2u || (lk0 == 4u && kh0 != k);
                    entry = leaves ? (int)((unsigned)entry | 0x80000000u) : entry;
                }
                lists[(int64_t)k * cap + off + __builtin_popcountll(bal & ((1ull << lane) - 1ull))] = entry;
            }
        }
    }
}

// lse[n] for every row, and in the same pass the M-step's active mask (ln r_nk = lnrho[k][n] - lse[n] >= thr[k], mstep.h),
// its block counts, and the number of pairs with ln r >= -80 ln 2 per block (apart).  thr comes from a sample of
// the rows (row_lse_kernel with a stride): a maximum over fewer rows is smaller, the threshold lower, the lists at
// worst a little longer - never a relevant sample dropped.  One thread per row, 256 rows per block.
static __global__ __launch_bounds__(kSelRows) void lse_mask_kernel(const double* __restrict__ lnrho, int64_t npad, int64_t n_rows,
                                                            int K, const double* __restrict__ thr, double* __restrict__ lse,
                                                            unsigned long long* __restrict__ masks,
                                                            int* __restrict__ blk_cnt, double* __restrict__ apart,
                                                            int* __restrict__ khat /*first maximiser, or null*/) {
    __shared__ int wcnt[4][256];
    __shared__ int wact[4];
    const int64_t n = (int64_t)blockIdx.x * kSelRows + threadIdx.x;
    const bool valid = n < n_rows;
    const int W = (K + 63) / 64;
    const int wave = threadIdx.x >> 6;
    for (int k = threadIdx.x & 63; k < K; k += 64) wcnt[wave][k] = 0;
    double l = 0.0;
    if (valid) {
        double mx = lnrho[n], s = 1.0;
        int arg = 0;
        for (int k = 1; k < K; ++k) {
            const double v = lnrho[(int64_t)k * npad + n];
            if (v > mx) {
                s = fma(s, exp(mx - v), 1.0);
                mx = v;
                arg = k;
            } else {
                s += exp(v - mx);
            }
        }
        l = mx + log(s);
        lse[n] = l;
        if (khat) khat[n] = arg;
    }
    int active = 0;
    for (int w = 0; w < W; ++w) {
        unsigned long long mk = 0;
        if (valid) {
            const int kend = K - 64 * w < 64 ? K - 64 * w : 64;
            for (int b = 0; b < kend; ++b) {
                const double t = lnrho[(int64_t)(64 * w + b) * npad + n] - l;
                mk |= (unsigned long long)(!(t < thr[64 * w + b])) << b;   // NaN stays active
                active += !(t < -kRelevanceNats);
            }
            masks[(int64_t)w * npad + n] = mk;
        }
        unsigned long long present = wave_or(mk);
        while (present) {
            const int b = __builtin_ctzll(present);
            present &= present - 1;
            const int c = __builtin_popcountll(__ballot((mk >> b) & 1ull));
            if ((threadIdx.x & 63) == 0) wcnt[wave][64 * w + b] = c;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) active += __shfl_xor(active, o);
    if ((threadIdx.x & 63) == 0) wact[wave] = active;
    __syncthreads();
    for (int k = threadIdx.x; k < K; k += kSelRows)
        blk_cnt[blk_at(k, blockIdx.x, K)] = wcnt[0][k] + wcnt[1][k] + wcnt[2][k] + wcnt[3][k];
    if (threadIdx.x == 0) apart[blockIdx.x] = (double)(wact[0] + wact[1] + wact[2] + wact[3]);
}

// r row-major [n][K] -> workspace [K][npad] (direct mode), lse = 0
// (iperm: the caller's row -> internal row, null = identity; see "rows grouped by their dominant component")
static __global__ void load_r_kernel(const double* __restrict__ r, int64_t n_rows, int K, double* __restrict__ buf,
                              int64_t npad, double* __restrict__ lse, const int* __restrict__ iperm) {
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= n_rows) return;
    const int64_t m = iperm ? iperm[n] : n;
    for (int k = 0; k < K; ++k) buf[(int64_t)k * npad + m] = r[n * K + k];
    lse[m] = 0.0;
}

// mode 0: ln rho; mode 1: r = exp(ln rho - lse) (or the stored r in direct mode)
static __global__ void readout_kernel(const double* __restrict__ buf, const double* __restrict__ lse, int64_t npad,
                               int64_t row0, int64_t n_rows, int K, int mode, int direct_r,
                               double* __restrict__ out, const int* __restrict__ iperm) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_rows * K) return;
    const int64_t n = e / K;
    const int k = (int)(e - n * K);
    const int64_t m = iperm ? iperm[row0 + n] : row0 + n;
    const double v = buf[(int64_t)k * npad + m];
    out[e] = (mode == 0 || direct_r) ? v : exp(v - lse[m]);
}

// z[i] = src[iperm[row0 + i]] (the best components rec_finish_kernel left, in the caller's row order)
static __global__ void gather_int_kernel(const int* __restrict__ src, int64_t row0, int64_t n_rows, const int* __restrict__ iperm,
                                  int32_t* __restrict__ z) {
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n < n_rows) z[n] = src[iperm ? iperm[row0 + n] : row0 + n];
}

// first maximiser over k, like numpy.argmax on the reference's r_vecs (_gaussianmixture.py:1191)
static __global__ void argmax_kernel(const double* __restrict__ buf, int64_t npad, int64_t row0, int64_t n_rows, int K,
                              int32_t* __restrict__ z, const int* __restrict__ iperm) {
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= n_rows) return;
    const int64_t m = iperm ? iperm[row0 + n] : row0 + n;
    double best = buf[m];
    int arg = 0;
    for (int k = 1; k < K; ++k) {
        const double v = buf[(int64_t)k * npad + m];
        if (v > best) {
            best = v;
            arg = k;
        }
    }
    z[n] = arg;
}

// Sum slabs over row splits (fixed order => run-to-run identical) and scatter to
// stats = [ ns[K] | h[K] | a[K][D] | B[K][D][D] ].
static __global__ void reduce_stats_kernel(const double* __restrict__ slabs, int S, int K, int D, int T,
                                    double* __restrict__ stats) {
    const int k = blockIdx.y;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    const int P = tri_pairs(T);
    const int L = slab_len(T);
    if (e >= P * 256 + 16 * T + 2) return;
    double v = 0.0;
    int s = 0;
    for (; s + 16 <= S; s += 16) {              // sixteen loads in flight, added in split order (the same bits as one by one)
        double t[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) t[u] = slabs[((int64_t)(s + u) * K + k) * L + e];
#pragma unroll
        for (int u = 0; u < 16; ++u) v += t[u];
    }
    for (; s < S; ++s) v += slabs[((int64_t)s * K + k) * L + e];
    double* ns = stats;
    double* h = stats + K;
    double* a = stats + 2 * (int64_t)K;
    double* B = a + (int64_t)K * D;
    if (e < P * 256) {
        const int p = e >> 8, r = (e >> 6) & 3, lane = e & 63;
        int t2 = 0;
        while (tri_pairs(t2 + 1) <= p) ++t2;
        const int t1 = p - tri_pairs(t2);
        const int col = lane & 15, row = (lane >> 4) + 4 * r;     // f64 MFMA C/D map
        const int f1 = T * row + t1, f2 = T * col + t2;
        if (f1 >= D || f2 >= D) return;
        if (t1 == t2 && row > col) return;   // diagonal tiles: keep one triangle, mirror it (exact symmetry)
        B[((int64_t)k * D + f1) * D + f2] = v;
        B[((int64_t)k * D + f2) * D + f1] = v;
    } else if (e < P * 256 + 16 * T) {
        const int f = e - P * 256;
        if (f < D) a[(int64_t)k * D + f] = v;
    } else if (e == P * 256 + 16 * T) {
        ns[k] = v;
    } else {
        h[k] = v;
    }
}

// The same for the list M-step's chunk slabs (mstep.h): component k owns slabs plan[k] .. plan[k + 1] - 1.
// accumulate: stats += the sum (the settled-row cache taking in a pass's delta lists); add: stats = the sum + add (the
// statistics of a pass = its lists + the cache), add in the layout of stats.
static __global__ void reduce_chunks_kernel(const double* __restrict__ slabs, const int* __restrict__ plan, int K, int D, int T,
                                     double* __restrict__ stats, int accumulate = 0, const double* __restrict__ add = nullptr) {
    const int k = blockIdx.y;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    const int P = tri_pairs(T);
    const int L = slab_len(T);
    if (e >= P * 256 + 16 * T + 2) return;
    double v = 0.0;
    const int c0 = plan[k], c1 = plan[k + 1];
    // (a component has ~70 slabs at the benchmark shape and the loop was a chain of dependent round trips to HBM: eight
    // loads in flight, added in the same chunk order - the same bits)
    int c = c0;
    for (; c + 8 <= c1; c += 8) {
        double t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = slabs[(int64_t)(c + u) * L + e];
#pragma unroll
        for (int u = 0; u < 8; ++u) v += t[u];
    }
    for (; c < c1; ++c) v += slabs[(int64_t)c * L + e];
    auto put = [&](int64_t idx) {
        double o = v;
        if (accumulate) o += stats[idx];
        if (add) o += add[idx];
        stats[idx] = o;
    };
    const int64_t ns0 = 0, h0 = K, a0 = 2 * (int64_t)K, B0 = a0 + (int64_t)K * D;
    if (e < P * 256) {
        const int p = e >> 8, r = (e >> 6) & 3, lane = e & 63;
        int t2 = 0;
        while (tri_pairs(t2 + 1) <= p) ++t2;
        const int t1 = p - tri_pairs(t2);
        const int col = lane & 15, row = (lane >> 4) + 4 * r;     // f64 MFMA C/D map
        const int f1 = T * row + t1, f2 = T * col + t2;
        if (f1 >= D || f2 >= D) return;
        if (t1 == t2 && row > col) return;   // diagonal tiles: keep one triangle, mirror it (exact symmetry)
        put(B0 + ((int64_t)k * D + f1) * D + f2);
        if (f1 != f2) put(B0 + ((int64_t)k * D + f2) * D + f1);
    } else if (e < P * 256 + 16 * T) {
        const int f = e - P * 256;
        if (f < D) put(a0 + (int64_t)k * D + f);
    } else if (e == P * 256 + 16 * T) {
        put(ns0 + k);
    } else {
        put(h0 + k);
    }
}

// ---- rows grouped by their dominant component ----------------------------------------------------------------------
// Once the responsibilities are sparse, almost all work is per (row, dominant component) pair; with the rows of a
// component scattered over the matrix every list-driven access (x rows in the gather and in the list M-step, ln rho /
// lse entries) is a random 0.5 - 1 KB or 8-byte access.  At a bound pass the workspace therefore regroups its INTERNAL
// row order: internal row i holds the caller's row perm[i], rows with the same best component are contiguous (in
// ascending original order within a component: the grouping is a stable counting sort, deterministic).  Lists of
// ascending internal rows then are (almost) contiguous runs.  The caller never sees the internal order: read-outs and
// gmmvb_load_responsibilities translate through iperm.  Statistics are sums over rows: only their rounding changes.
//
// perm_new[start_k + j] = perm_old[lists[k][j]] (perm_old null: identity); lists[k] = ascending internal rows whose
// best component is k (select_mask_kernel<3> + scan_counts + fill_lists).  gridDim = (blocks over j, K).
static __global__ __launch_bounds__(256) void perm_compose_kernel(const int* __restrict__ lists, int64_t cap,
                                                           const int* __restrict__ counts, const int* __restrict__ perm_old,
                                                           int* __restrict__ perm_new) {
    const int k = blockIdx.y;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= counts[k]) return;
    int64_t start = 0;
    for (int q = 0; q < k; ++q) start += counts[q];
    const int src = lists[(int64_t)k * cap + j];
    perm_new[start + j] = perm_old ? perm_old[src] : src;
}

// Second key of the regrouping: how firmly a row sits in its best component.  bucket[n] in 0 .. kMarginBuckets - 1 from
// t = lse - ln rho_best = -ln r_best (1 - r_best ~ t: the runner-up lies about -ln t nats below): settled rows on top, then
// rows with r_best = 1.0 exactly, then decades of t.  Rows settle in this order as the components sharpen, so within a
// component's group the settled rows - which the selection kernels skip by whole waves and tiles - stay contiguous, and so
// do the rows the list-driven kernels still have to read.  (A NaN row lands in bucket 0.)
constexpr int kMarginBuckets = 8;
static __global__ __launch_bounds__(256) void margin_bucket_kernel(const double* __restrict__ lnrho, int64_t npad, int64_t n_rows,
                                                            const double* __restrict__ lse, const int* __restrict__ khat,
                                                            const unsigned char* __restrict__ lock /*null: no settled rows*/,
                                                            int* __restrict__ bucket) {
    const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (n >= n_rows) return;
    int b = 0;
    if (lock != nullptr && lock[n] == 1u) {
        b = 7;
    } else {
        const double t = lse[n] - lnrho[(int64_t)khat[n] * npad + n];
        if (t <= 0.0) b = 6;
        else if (t < 1e-20) b = 5;
        else if (t < 1e-14) b = 4;
        else if (t < 1e-9) b = 3;
        else if (t < 1e-5) b = 2;
        else if (t < 1e-2) b = 1;
    }
    bucket[n] = b;
}

static __global__ void perm_invert_kernel(const int* __restrict__ perm, int64_t n_rows, int* __restrict__ iperm) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_rows) iperm[perm[i]] = (int)i;
}

// The per-row state of the cache of single-component rows follows the rows into their new order: new internal row i holds
// the caller's row perm_new[i], which sat at internal row iperm_old[perm_new[i]] before (iperm_old null: the caller's order).
static __global__ void regroup_state_kernel(const int* __restrict__ perm_new, const int* __restrict__ iperm_old, int64_t n_rows,
                                     const unsigned char* __restrict__ lock, const unsigned char* __restrict__ lcomp,
                                     const float* __restrict__ dlock, unsigned char* __restrict__ lock_new,
                                     unsigned char* __restrict__ lcomp_new, float* __restrict__ dlock_new) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rows) return;
    const int c = perm_new[i];
    const int src = iperm_old ? iperm_old[c] : c;
    lock_new[i] = lock[src];
    lcomp_new[i] = lcomp[src];
    dlock_new[i] = dlock[src];
}

// xp[i][:] = x[perm[i]][:] (packed, ld = D)
template <typename XT>
__global__ void permute_rows_kernel(const XT* __restrict__ x, int64_t ldx, int64_t n_rows, int D,
                                    const int* __restrict__ perm, XT* __restrict__ xp) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_rows * D) return;
    const int64_t i = e / D;
    const int f = (int)(e - i * D);
    xp[e] = x[(int64_t)perm[i] * ldx + f];
}

// the same in 16-byte pieces (rows and row strides are multiples of 16 bytes, both bases 16-byte aligned): a quarter of the
// index arithmetic per byte moved
static __global__ void permute_rows16_kernel(const uint4* __restrict__ x, int64_t ld16, int64_t n_rows, int p16,
                                      const int* __restrict__ perm, uint4* __restrict__ xp) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_rows * p16) return;
    const int64_t i = e / p16;
    const int p = (int)(e - i * p16);
    xp[e] = x[(int64_t)perm[i] * ld16 + p];
}

}  // namespace gmmvb
