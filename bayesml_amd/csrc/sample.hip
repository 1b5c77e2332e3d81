// Device-side data generation behind the C ABI: GenModel.gen_sample of the mixture
// (bayesml/gaussianmixture/_gaussianmixture.py:241-264: one `choice` and one `multivariate_normal` per row in a Python loop)
// and of the HMM (bayesml/hiddenmarkovnormal/_hiddenmarkovnormal.py:344-358: the same along a Markov chain).
//
// The random stream is counter based and reproducible ON THE HOST: Philox4x64-10 exactly as numpy.random.Philox runs it
// (key = [seed, stream id], the block of four 64-bit outputs number L comes from the counter value L + 1), so
// `numpy.random.Philox(key=[seed, s]).random_raw(...)` IS the device's stream and a test can compare samples with a host
// restatement value by value (oracle/sampler_oracle.py) instead of by moments.
//   stream 0 - latent uniforms: u_t = (raw_t >> 11) 2^-53 in [0, 1), one per row / time step t;
//              the class is the number of entries of the inclusive cumulative distribution's first K - 1 that are <= u
//   stream 1 - normals: row r owns the ceil(D / 4) blocks from r ceil(D / 4) on; a block (r0 .. r3) gives four normals by
//              Box-Muller: sqrt(-2 ln(1 - (r0 >> 11) 2^-53)) (cos, sin)(2 pi (r1 >> 11) 2^-53), and the same from (r2, r3)
// Emissions: x = mu_z + eps A_z with A_z = L_z^-1 lower triangular, Lambda_z = L_z L_z^T (covariance A^T A = Lambda^-1).
// The Markov chain runs without a sequential pass over T: step t is the map i -> F_i^-1(u_t) on the K states and maps
// compose, so chunks of 256 steps carry every start state at once, groups of 256 chunk maps are composed the same way,
// one thread chains the group maps, and the start states flow back down (five launches, no host round trip).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "workspace.h"

using namespace gmmvb;

namespace {

struct Raw4 { uint64_t v[4]; };

__device__ __forceinline__ void mulhilo(uint64_t a, uint64_t b, uint64_t& hi, uint64_t& lo) {
    lo = a * b;
    hi = __umul64hi(a, b);
}

// block number `index` of the stream (seed, stream): numpy's Philox increments its 256-bit counter BEFORE generating
__device__ __forceinline__ Raw4 philox4x64_10(uint64_t index, uint64_t seed, uint64_t stream) {
    uint64_t c0 = index + 1, c1 = (c0 == 0) ? 1 : 0, c2 = 0, c3 = 0;      // (the carry: block 2^64 - 1)
    uint64_t k0 = seed, k1 = stream;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint64_t hi0, lo0, hi1, lo1;
        mulhilo(0xD2E7470EE14C6C93ull, c0, hi0, lo0);
        mulhilo(0xCA5A826395121157ull, c2, hi1, lo1);
        const uint64_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0, c1 = lo1, c2 = n2, c3 = lo0;
        k0 += 0x9E3779B97F4A7C15ull;
        k1 += 0xBB67AE8584CAA73Bull;
    }
    return Raw4{{c0, c1, c2, c3}};
}

__device__ __forceinline__ double unit_open_below(uint64_t raw) { return (double)(raw >> 11) * 0x1.0p-53; }        // [0, 1)

// number of entries of cdf[0 .. K-2] that are <= u (cdf nondecreasing): the inverse of the inclusive cumulative distribution
__device__ __forceinline__ int inverse_cdf(const double* __restrict__ cdf, int K, double u) {
    int lo = 0, hi = K - 1;                   // answer in [lo, hi]
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (u >= cdf[mid]) lo = mid + 1; else hi = mid;
    }
    return lo;
}

__global__ __launch_bounds__(256) void sample_latent_kernel(const double* __restrict__ cdf, int K, uint64_t seed, int64_t row0,
                                                            int64_t n_rows, int64_t* __restrict__ z) {
    extern __shared__ double s_cdf[];
    for (int i = threadIdx.x; i < K; i += 256) s_cdf[i] = cdf[i];
    __syncthreads();
    // a thread per block of four rows: rows [4 b, 4 b + 4) of the global numbering
    const int64_t first_blk = row0 >> 2, last_blk = (row0 + n_rows - 1) >> 2;
    for (int64_t b = first_blk + (int64_t)blockIdx.x * 256 + threadIdx.x; b <= last_blk; b += (int64_t)gridDim.x * 256) {
        const Raw4 r = philox4x64_10((uint64_t)b, seed, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t row = 4 * b + j - row0;
            if (row >= 0 && row < n_rows) z[row] = inverse_cdf(s_cdf, K, unit_open_below(r.v[j]));
        }
    }
}

// ---- Markov chain -------------------------------------------------------------------------------------------------

constexpr int kChunk = 256;       // time steps per chunk
constexpr int kGroup = 256;       // chunk maps per group

// maps[c][s] = state after chunk c when it is entered in state s (the sequence's very first step draws from pi whatever
// s is).  One workgroup per chunk; the chunk's uniforms go through LDS once.
__global__ __launch_bounds__(256) void chain_maps_kernel(const double* __restrict__ cdf_pi, const double* __restrict__ cdf_a, int K,
                                                         uint64_t seed, int64_t n_rows, int* __restrict__ maps) {
    __shared__ double s_u[kChunk];
    const int64_t c = blockIdx.x, t0 = c * kChunk;
    const int len = (int)min((int64_t)kChunk, n_rows - t0);
    if (threadIdx.x < kChunk / 4) {
        const Raw4 r = philox4x64_10((uint64_t)(t0 / 4 + threadIdx.x), seed, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) s_u[4 * threadIdx.x + j] = unit_open_below(r.v[j]);
    }
    __syncthreads();
    for (int s = threadIdx.x; s < K; s += 256) {
        int cur = s;
        for (int t = 0; t < len; ++t)
            cur = (t0 + t == 0) ? inverse_cdf(cdf_pi, K, s_u[0]) : inverse_cdf(cdf_a + (int64_t)cur * K, K, s_u[t]);
        maps[c * K + s] = cur;
    }
}

// gmaps[g][s] = composition of the group's chunk maps
__global__ __launch_bounds__(256) void chain_compose_kernel(const int* __restrict__ maps, int K, int64_t n_chunks, int* __restrict__ gmaps) {
    const int64_t g = blockIdx.x, c0 = g * kGroup, c1 = min(n_chunks, c0 + kGroup);
    for (int s = threadIdx.x; s < K; s += 256) {
        int cur = s;
        for (int64_t c = c0; c < c1; ++c) cur = maps[c * K + cur];
        gmaps[g * K + s] = cur;
    }
}

// one workgroup: thread 0 chains the groups, then a thread per group hands every chunk its entry state
__global__ __launch_bounds__(256) void chain_starts_kernel(const int* __restrict__ maps, const int* __restrict__ gmaps, int K,
                                                           int64_t n_chunks, int64_t n_groups, int* __restrict__ gstart,
                                                           int* __restrict__ cstart) {
    if (threadIdx.x == 0) {
        int cur = 0;                                    // (the first chunk's maps do not depend on it)
        for (int64_t g = 0; g < n_groups; ++g) {
            gstart[g] = cur;
            cur = gmaps[g * K + cur];
        }
    }
    __threadfence();
    __syncthreads();
    for (int64_t g = threadIdx.x; g < n_groups; g += 256) {
        int cur = gstart[g];
        const int64_t c0 = g * kGroup, c1 = min(n_chunks, c0 + kGroup);
        for (int64_t c = c0; c < c1; ++c) {
            cstart[c] = cur;
            cur = maps[c * K + cur];
        }
    }
}

// a thread per chunk replays it from its entry state
__global__ __launch_bounds__(64) void chain_replay_kernel(const double* __restrict__ cdf_pi, const double* __restrict__ cdf_a, int K,
                                                          uint64_t seed, int64_t n_rows, int64_t n_chunks, const int* __restrict__ cstart,
                                                          int64_t* __restrict__ z) {
    const int64_t c = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (c >= n_chunks) return;
    const int64_t t0 = c * kChunk;
    const int len = (int)min((int64_t)kChunk, n_rows - t0);
    int cur = cstart[c];
    for (int b = 0; 4 * b < len; ++b) {
        const Raw4 r = philox4x64_10((uint64_t)(t0 / 4 + b), seed, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int t = 4 * b + j;
            if (t < len) {
                const double u = unit_open_below(r.v[j]);
                cur = (t0 + t == 0) ? inverse_cdf(cdf_pi, K, u) : inverse_cdf(cdf_a + (int64_t)cur * K, K, u);
                z[t0 + t] = cur;
            }
        }
    }
}

// ---- emissions ----------------------------------------------------------------------------------------------------

constexpr int kEmitRows = 32;     // rows per pass of a workgroup

// x[row][j] = mu[z][j] + sum_{i >= j} eps[row][i] A[z][i][j]; the tile's normals are drawn into LDS by all threads,
// then a thread per (row, j) with j fastest so that a row's threads read A's rows as contiguous runs
// `order` (optional): the rows grouped by class, so that the workgroups running at one moment read the same few A_k
// from L2 instead of K D^2 doubles from everywhere (K = 64, D = 128: 8.4 MB against 4 MB of L2 per XCD - 794 -> see
// profiles/r6_sampler.json); a row's values depend on its global number only, so the grouping's order is free
template <typename XT>
__global__ __launch_bounds__(256) void sample_emissions_kernel(const int64_t* __restrict__ z, const double* __restrict__ mu,
                                                               const double* __restrict__ a, int K, int D, uint64_t seed,
                                                               int64_t row0, int64_t n_rows, XT* __restrict__ x, int64_t ldx,
                                                               const int* __restrict__ order) {
    extern __shared__ double s_eps[];                 // [kEmitRows][4 nb]
    const int nb = (D + 3) >> 2, dp = 4 * nb;
    const int64_t n_tiles = (n_rows + kEmitRows - 1) / kEmitRows;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t r0 = tile * kEmitRows;
        const int rows = (int)min((int64_t)kEmitRows, n_rows - r0);
        __syncthreads();
        for (int e = threadIdx.x; e < rows * nb; e += 256) {
            const int lr = e / nb, b = e - lr * nb;
            const int64_t row = order ? (int64_t)order[r0 + lr] : r0 + lr;
            const Raw4 r = philox4x64_10((uint64_t)(row0 + row) * (uint64_t)nb + (uint64_t)b, seed, 1);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const double u0 = 1.0 - unit_open_below(r.v[2 * h]);          // (0, 1]
                const double rad = sqrt(-2.0 * log(u0));
                double sn, cs;
                sincospi(2.0 * unit_open_below(r.v[2 * h + 1]), &sn, &cs);
                s_eps[lr * dp + 4 * b + 2 * h] = rad * cs;
                s_eps[lr * dp + 4 * b + 2 * h + 1] = rad * sn;
            }
        }
        __syncthreads();
        for (int e = threadIdx.x; e < rows * D; e += 256) {
            const int lr = e / D, j = e - lr * D;
            const int64_t row = order ? (int64_t)order[r0 + lr] : r0 + lr;
            const int k = (int)z[row];
            const double* ak = a + (int64_t)k * D * D;
            const double* ep = s_eps + lr * dp;
            double acc0 = mu[(int64_t)k * D + j], acc1 = 0.0;
            int i = j;
            for (; i + 1 < D; i += 2) {
                acc0 = fma(ep[i], ak[(int64_t)i * D + j], acc0);
                acc1 = fma(ep[i + 1], ak[(int64_t)(i + 1) * D + j], acc1);
            }
            if (i < D) acc0 = fma(ep[i], ak[(int64_t)i * D + j], acc0);
            x[row * ldx + j] = (XT)(acc0 + acc1);
        }
    }
}

// The grouped form: `order` lists the rows class by class, so a tile of 32 rows is one run of a class (two or three at
// class borders).  A thread owns column j and RPT = DJ / 8 of the tile's rows: every A[i][j] it loads (coalesced over j)
// feeds RPT accumulators, the normals come from LDS as broadcasts - 1 / RPT of the plain kernel's global loads, and four of
// them in flight (the plain kernel waits an L2 round trip per two multiply-adds: 700 ms at K 64, D 128, N 1e7).
template <typename XT, int DJ>
__global__ __launch_bounds__(256) void sample_emissions_grouped_kernel(const int64_t* __restrict__ z, const double* __restrict__ mu,
                                                                       const double* __restrict__ a, int K, int D, uint64_t seed,
                                                                       int64_t row0, int64_t n_rows, XT* __restrict__ x, int64_t ldx,
                                                                       const int* __restrict__ order) {
    constexpr int RPT = DJ / 8;
    static_assert((256 / DJ) * RPT == kEmitRows, "a workgroup covers the tile's rows once");
    extern __shared__ double s_eps[];                 // [kEmitRows][4 nb]
    __shared__ int s_row[kEmitRows], s_k[kEmitRows];
    const int nb = (D + 3) >> 2, dp = 4 * nb;
    const int64_t n_tiles = (n_rows + kEmitRows - 1) / kEmitRows;
    const int jl = threadIdx.x % DJ, q0 = (threadIdx.x / DJ) * RPT;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t r0 = tile * kEmitRows;
        const int rows = (int)min((int64_t)kEmitRows, n_rows - r0);
        __syncthreads();
        if (threadIdx.x < kEmitRows) {
            const int row = (int)threadIdx.x < rows ? order[r0 + threadIdx.x] : -1;
            s_row[threadIdx.x] = row;
            s_k[threadIdx.x] = row >= 0 ? (int)z[row] : -1;
        }
        __syncthreads();
        for (int e = threadIdx.x; e < kEmitRows * nb; e += 256) {
            const int lr = e / nb, b = e - lr * nb;
            if (lr < rows) {
                const Raw4 r = philox4x64_10((uint64_t)(row0 + s_row[lr]) * (uint64_t)nb + (uint64_t)b, seed, 1);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const double u0 = 1.0 - unit_open_below(r.v[2 * h]);
                    const double rad = sqrt(-2.0 * log(u0));
                    double sn, cs;
                    sincospi(2.0 * unit_open_below(r.v[2 * h + 1]), &sn, &cs);
                    s_eps[lr * dp + 4 * b + 2 * h] = rad * cs;
                    s_eps[lr * dp + 4 * b + 2 * h + 1] = rad * sn;
                }
            } else {
#pragma unroll
                for (int h = 0; h < 4; ++h) s_eps[lr * dp + 4 * b + h] = 0.0;
            }
        }
        __syncthreads();
        for (int lo = 0; lo < rows;) {
            const int k = s_k[lo];
            int hi = lo + 1;
            while (hi < rows && s_k[hi] == k) ++hi;
            if (q0 < hi && q0 + RPT > lo) {
                for (int j = jl; j < D; j += DJ) {
                    double acc[RPT];
                    const double m = mu[(int64_t)k * D + j];
#pragma unroll
                    for (int q = 0; q < RPT; ++q) acc[q] = m;
                    const double* ak = a + (int64_t)k * D * D + j;
                    const double* ep = s_eps + q0 * dp;
#pragma unroll 4
                    for (int i = j; i < D; ++i) {
                        const double av = ak[(int64_t)i * D];
#pragma unroll
                        for (int q = 0; q < RPT; ++q) acc[q] = fma(ep[q * dp + i], av, acc[q]);
                    }
#pragma unroll
                    for (int q = 0; q < RPT; ++q)
                        if (q0 + q >= lo && q0 + q < hi) x[(int64_t)s_row[q0 + q] * ldx + j] = (XT)acc[q];
                }
            }
            lo = hi;
        }
    }
}

template <typename XT>
void launch_grouped(int D, unsigned grid, size_t lds, hipStream_t st, const int64_t* z, const double* mu, const double* a, int K,
                    uint64_t seed, int64_t row0, int64_t n_rows, XT* x, int64_t ldx, const int* order) {
#define GMMVB_EMIT(DJ)                                                                                                           \
    do {                                                                                                                         \
        if (lds > 64 * 1024)                                                                                                     \
            (void)hipFuncSetAttribute((const void*)sample_emissions_grouped_kernel<XT, DJ>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                      (int)lds);                                                                                 \
        hipLaunchKernelGGL((sample_emissions_grouped_kernel<XT, DJ>), dim3(grid), dim3(256), lds, st, z, mu, a, K, D, seed, row0,   \
                           n_rows, x, ldx, order);                                                                               \
    } while (0)
    if (D <= 16) GMMVB_EMIT(16);
    else if (D <= 32) GMMVB_EMIT(32);
    else if (D <= 64) GMMVB_EMIT(64);
    else if (D <= 128) GMMVB_EMIT(128);
    else GMMVB_EMIT(256);
#undef GMMVB_EMIT
}

// rows grouped by class: histogram, scan over K, scatter (the order inside a class is whatever the atomics give - free, see
// sample_emissions_kernel)
__global__ __launch_bounds__(256) void class_count_kernel(const int64_t* __restrict__ z, int64_t n_rows, int K, int* __restrict__ counts) {
    extern __shared__ int s_cnt[];
    for (int i = threadIdx.x; i < K; i += 256) s_cnt[i] = 0;
    __syncthreads();
    for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < n_rows; r += (int64_t)gridDim.x * 256) atomicAdd(&s_cnt[(int)z[r]], 1);
    __syncthreads();
    for (int i = threadIdx.x; i < K; i += 256)
        if (s_cnt[i]) atomicAdd(&counts[i], s_cnt[i]);
}
__global__ __launch_bounds__(256) void class_scan_kernel(const int* __restrict__ counts, int K, int* __restrict__ cursor) {
    if (threadIdx.x == 0) {
        int run = 0;
        for (int k = 0; k < K; ++k) {
            cursor[k] = run;
            run += counts[k];
        }
    }
}
__global__ __launch_bounds__(256) void class_scatter_kernel(const int64_t* __restrict__ z, int64_t n_rows, int K, int* __restrict__ cursor,
                                                            int* __restrict__ order) {
    // a block claims a run per class for its rows at once (one global atomic per class and block), then fills it
    extern __shared__ int s_cnt[];                // [2 K]: counts, then bases
    int* s_base = s_cnt + K;
    const int64_t per = (n_rows + gridDim.x - 1) / gridDim.x, lo = blockIdx.x * per, hi = min(n_rows, lo + per);
    for (int i = threadIdx.x; i < K; i += 256) s_cnt[i] = 0;
    __syncthreads();
    for (int64_t r = lo + threadIdx.x; r < hi; r += 256) atomicAdd(&s_cnt[(int)z[r]], 1);
    __syncthreads();
    for (int i = threadIdx.x; i < K; i += 256) {
        s_base[i] = s_cnt[i] ? atomicAdd(&cursor[i], s_cnt[i]) : 0;
        s_cnt[i] = 0;
    }
    __syncthreads();
    for (int64_t r = lo + threadIdx.x; r < hi; r += 256) {
        const int k = (int)z[r];
        order[s_base[k] + atomicAdd(&s_cnt[k], 1)] = (int)r;
    }
}

int launch_error(const char* what) {
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? GMMVB_OK : fail(GMMVB_EHIP, what, e);
}

struct ChainPlan {
    int64_t n_chunks, n_groups;
    int64_t off_maps, off_gmaps, off_gstart, off_cstart, bytes;
};
ChainPlan chain_plan(int K, int64_t n_rows) {
    ChainPlan p;
    p.n_chunks = (n_rows + kChunk - 1) / kChunk;
    p.n_groups = (p.n_chunks + kGroup - 1) / kGroup;
    auto up = [](int64_t b) { return (b + 255) / 256 * 256; };
    p.off_maps = 0;
    p.off_gmaps = p.off_maps + up(p.n_chunks * K * 4);
    p.off_gstart = p.off_gmaps + up(p.n_groups * K * 4);
    p.off_cstart = p.off_gstart + up(p.n_groups * 4);
    p.bytes = p.off_cstart + up(p.n_chunks * 4);
    return p;
}

}  // namespace

extern "C" {

int gmmvb_sample_latent(int K, const double* cdf_dev, uint64_t seed, int64_t row0, int64_t n_rows, int64_t* z_dev, void* stream) {
    if (K < 1 || !cdf_dev || !z_dev || row0 < 0 || n_rows < 0) return fail(GMMVB_EINVAL, "gmmvb_sample_latent: bad argument");
    if ((size_t)K * 8 > 64 * 1024) return fail(GMMVB_EUNSUPPORTED, "gmmvb_sample_latent: K <= 8192");
    if (n_rows == 0) return GMMVB_OK;
    const int64_t blocks = ((row0 + n_rows - 1) >> 2) - (row0 >> 2) + 1;
    const unsigned grid = (unsigned)std::min<int64_t>((blocks + 255) / 256, 1 << 16);
    hipLaunchKernelGGL(sample_latent_kernel, dim3(grid), dim3(256), (size_t)K * 8, (hipStream_t)stream, cdf_dev, K, seed, row0, n_rows, z_dev);
    return launch_error("sample_latent_kernel");
}

int64_t gmmvb_sample_chain_work_bytes(int K, int64_t n_rows) {
    return (K < 1 || n_rows < 0) ? -1 : chain_plan(K, std::max<int64_t>(n_rows, 1)).bytes;
}

int gmmvb_sample_chain(int K, const double* cdf_pi_dev, const double* cdf_a_dev, uint64_t seed, int64_t n_rows, int64_t* z_dev,
                       void* work_dev, int64_t work_bytes, void* stream) {
    if (K < 1 || !cdf_pi_dev || !cdf_a_dev || !z_dev || n_rows < 0) return fail(GMMVB_EINVAL, "gmmvb_sample_chain: bad argument");
    if (n_rows == 0) return GMMVB_OK;
    const ChainPlan p = chain_plan(K, n_rows);
    if (!work_dev || work_bytes < p.bytes) return fail(GMMVB_EINVAL, "gmmvb_sample_chain: work buffer smaller than gmmvb_sample_chain_work_bytes");
    if (p.n_chunks > 0x7fffffff) return fail(GMMVB_EUNSUPPORTED, "gmmvb_sample_chain: more than 2^31 chunks");
    char* w = (char*)work_dev;
    int *maps = (int*)(w + p.off_maps), *gmaps = (int*)(w + p.off_gmaps), *gstart = (int*)(w + p.off_gstart), *cstart = (int*)(w + p.off_cstart);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(chain_maps_kernel, dim3((unsigned)p.n_chunks), dim3(256), 0, st, cdf_pi_dev, cdf_a_dev, K, seed, n_rows, maps);
    hipLaunchKernelGGL(chain_compose_kernel, dim3((unsigned)p.n_groups), dim3(256), 0, st, maps, K, p.n_chunks, gmaps);
    hipLaunchKernelGGL(chain_starts_kernel, dim3(1), dim3(256), 0, st, maps, gmaps, K, p.n_chunks, p.n_groups, gstart, cstart);
    hipLaunchKernelGGL(chain_replay_kernel, dim3((unsigned)((p.n_chunks + 63) / 64)), dim3(64), 0, st, cdf_pi_dev, cdf_a_dev, K, seed,
                       n_rows, p.n_chunks, cstart, z_dev);
    return launch_error("gmmvb_sample_chain");
}

int64_t gmmvb_sample_emissions_work_bytes(int K, int64_t n_rows) {
    if (K < 1 || n_rows < 0) return -1;
    if (K > 8192 || n_rows > 0x7fffffff) return 0;      // (ungrouped: still correct)
    return (2 * (int64_t)K + n_rows) * 4;
}

int gmmvb_sample_emissions(int K, int D, const int64_t* z_dev, const double* mu_dev, const double* a_dev, uint64_t seed,
                           int64_t row0, int64_t n_rows, int x_dtype, void* x_dev, int64_t ldx, void* work_dev, int64_t work_bytes,
                           void* stream) {
    if (K < 1 || D < 1 || !z_dev || !mu_dev || !a_dev || !x_dev || row0 < 0 || n_rows < 0 || ldx < D)
        return fail(GMMVB_EINVAL, "gmmvb_sample_emissions: bad argument");
    if (x_dtype != GMMVB_F32 && x_dtype != GMMVB_F64) return fail(GMMVB_EINVAL, "gmmvb_sample_emissions: x_dtype");
    const size_t lds = (size_t)kEmitRows * 4 * ((D + 3) / 4) * 8;
    if (lds > 150 * 1024) return fail(GMMVB_EUNSUPPORTED, "gmmvb_sample_emissions: D <= 600");
    if (n_rows == 0) return GMMVB_OK;
    const int64_t tiles = (n_rows + kEmitRows - 1) / kEmitRows;
    const unsigned grid = (unsigned)std::min<int64_t>(tiles, 256 * 16);
    hipStream_t st = (hipStream_t)stream;
    const int* order = nullptr;
    const int64_t want = gmmvb_sample_emissions_work_bytes(K, n_rows);
    if (work_dev && want > 0 && work_bytes >= want) {
        int *counts = (int*)work_dev, *cursor = counts + K, *ord = cursor + K;
        hipError_t e = hipMemsetAsync(counts, 0, (size_t)K * 4, st);
        if (e != hipSuccess) return fail(GMMVB_EHIP, "gmmvb_sample_emissions: memset", e);
        const unsigned g = (unsigned)std::min<int64_t>((n_rows + 4095) / 4096, 2048);
        hipLaunchKernelGGL(class_count_kernel, dim3(g), dim3(256), (size_t)K * 4, st, z_dev, n_rows, K, counts);
        hipLaunchKernelGGL(class_scan_kernel, dim3(1), dim3(256), 0, st, counts, K, cursor);
        hipLaunchKernelGGL(class_scatter_kernel, dim3(g), dim3(256), (size_t)K * 8, st, z_dev, n_rows, K, cursor, ord);
        order = ord;
    }
    if (order) {
        if (x_dtype == GMMVB_F32) launch_grouped<float>(D, grid, lds, st, z_dev, mu_dev, a_dev, K, seed, row0, n_rows, (float*)x_dev, ldx, order);
        else launch_grouped<double>(D, grid, lds, st, z_dev, mu_dev, a_dev, K, seed, row0, n_rows, (double*)x_dev, ldx, order);
        return launch_error("sample_emissions_grouped_kernel");
    }
    if (x_dtype == GMMVB_F32) {
        if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)sample_emissions_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(sample_emissions_kernel<float>, dim3(grid), dim3(256), lds, st, z_dev, mu_dev, a_dev, K, D, seed, row0, n_rows,
                           (float*)x_dev, ldx, nullptr);
    } else {
        if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)sample_emissions_kernel<double>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(sample_emissions_kernel<double>, dim3(grid), dim3(256), lds, st, z_dev, mu_dev, a_dev, K, D, seed, row0, n_rows,
                           (double*)x_dev, ldx, nullptr);
    }
    return launch_error("sample_emissions_kernel");
}

}  // extern "C"
