// Micro-benchmark: sustained v_mfma_f64_16x16x4_f64 and v_fma_f64 rates on this GPU, to pin the
// "peak" the roofline fractions in bench.py / DESIGN.md are quoted against (the programming guide
// lists no f64 MFMA row).  Build: hipcc --offload-arch=gfx950 -O3 tools/peak_probe.hip -o tools/peak_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(double* out, int iters, double a0, double b0) {
    d4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
    double a = a0 + threadIdx.x * 1e-9, b = b0 - threadIdx.x * 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void fma_loop(double* out, int iters, double a0, double b0) {
    double acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = i;
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = fma(acc[i], a, b);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
static float time_ms(F launch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    double* out;
    hipMalloc(&out, sizeof(double) * 256 * cus * 8);
    printf("device: %s, %d CUs, clock %d MHz\n", p.gcnArchName, cus, p.clockRate / 1000);
    const int iters = 20000;
    for (int wpc : {1, 2}) {              // workgroups (4 waves) per CU => waves per SIMD
        const int grid = cus * wpc;
        float ms = time_ms([&] { hipLaunchKernelGGL(mfma_loop<8>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0, 1.0); });
        double flops = 2.0 * 16 * 16 * 4 * 8.0 * iters * 4.0 * grid;
        printf("mfma_f64_16x16x4 x8 acc, %d wave/SIMD: %.2f ms  %.1f TFLOP/s  (%.1f cycles/MFMA/SIMD at %d MHz)\n", wpc, ms,
               flops / ms / 1e9, (double)ms * 1e-3 * p.clockRate * 1e3 / (8.0 * iters * wpc), p.clockRate / 1000);
    }
    {
        const int grid = cus;
        float ms = time_ms([&] { hipLaunchKernelGGL(mfma_loop<1>, dim3(grid), dim3(256), 0, 0, out, iters * 4, 1.0, 1.0); });
        printf("mfma_f64 dependent chain (1 acc): %.1f cycles/MFMA at nominal clock\n",
               (double)ms * 1e-3 * p.clockRate * 1e3 / (4.0 * iters));
    }
    for (int wpc : {1, 2, 4}) {
        const int grid = cus * wpc;
        float ms = time_ms([&] { hipLaunchKernelGGL(fma_loop, dim3(grid), dim3(256), 0, 0, out, iters, 1.0000001, 1e-9); });
        double flops = 2.0 * 16 * iters * 256.0 * grid;
        printf("v_fma_f64 x16 acc, %d wave/SIMD: %.2f ms  %.1f TFLOP/s\n", wpc, ms, flops / ms / 1e9);
    }
    hipFree(out);
    return 0;
}
