// Probe: a tile of 256 rows reads half of its K columns of a [K][npad] f32 array (1 KB per column, npad * 4 bytes apart)
// against the same bytes from a tile-blocked [tile][K][256] array.  hipcc --offload-arch=gfx950 -O3 -o probe tools/probe_ub_layout.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

template <bool BLOCKED, bool WRITE>
__global__ __launch_bounds__(256) void sweep_like(float* __restrict__ ub, int64_t npad, int K, int stride, float* __restrict__ out) {
    const int64_t tile = blockIdx.x;
    const int tid = threadIdx.x;
    const int64_t n = tile * 256 + tid;
    float acc = 0.0f;
    const int phase = (int)(tile % stride);
    for (int k0 = phase; k0 < K; k0 += 8 * stride) {
        float pre[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int k = k0 + q * stride;
            const int64_t idx = BLOCKED ? (tile * K + k) * 256 + tid : (int64_t)k * npad + n;
            pre[q] = k < K ? ub[idx] : 0.0f;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int k = k0 + q * stride;
            const float v = __builtin_amdgcn_sqrtf(pre[q] + pre[q]) * 0.99f - 0.01f;
            acc = fmaxf(acc, v);
            if (WRITE && k < K) {
                const int64_t idx = BLOCKED ? (tile * K + k) * 256 + tid : (int64_t)k * npad + n;
                ub[idx] = v * v * 0.5f;
            }
        }
    }
    if (acc == 12345.678f) out[n] = acc;
}

int main(int argc, char** argv) {
    const int K = argc > 1 ? atoi(argv[1]) : 256;
    const int64_t n = argc > 2 ? atoll(argv[2]) : 12500000;
    const int64_t tiles = (n + 255) / 256, npad = tiles * 256;
    float *ub, *out;
    hipMalloc(&ub, (size_t)K * npad * 4);
    hipMalloc(&out, (size_t)npad * 4);
    hipMemset(ub, 0x3c, (size_t)K * npad * 4);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int stride = 1; stride <= 4; stride *= 2)
        for (int v = 0; v < 4; ++v) {
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                hipEventRecord(a, 0);
                if (v == 0) hipLaunchKernelGGL((sweep_like<false, false>), dim3((unsigned)tiles), dim3(256), 0, 0, ub, npad, K, stride, out);
                if (v == 1) hipLaunchKernelGGL((sweep_like<true, false>), dim3((unsigned)tiles), dim3(256), 0, 0, ub, npad, K, stride, out);
                if (v == 2) hipLaunchKernelGGL((sweep_like<false, true>), dim3((unsigned)tiles), dim3(256), 0, 0, ub, npad, K, stride, out);
                if (v == 3) hipLaunchKernelGGL((sweep_like<true, true>), dim3((unsigned)tiles), dim3(256), 0, 0, ub, npad, K, stride, out);
                hipEventRecord(b, 0);
                hipEventSynchronize(b);
                float ms;
                hipEventElapsedTime(&ms, a, b);
                best = ms < best ? ms : best;
            }
            const double bytes = (double)K / stride * npad * 4 * (v >= 2 ? 2 : 1);
            printf("K=%d n=%lld open=1/%d %s %s: %.3f ms  %.2f TB/s\n", K, (long long)n, stride, (v & 1) ? "blocked" : "columns", v >= 2 ? "read+write" : "read", best, bytes / best / 1e9);
        }
    return 0;
}
