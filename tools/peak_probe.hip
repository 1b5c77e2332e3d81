// Micro-benchmark: sustained v_mfma_f64_16x16x4_f64 and v_fma_f64 rates on this GPU, to pin the
// "peak" the roofline fractions in bench.py / DESIGN.md are quoted against (the programming guide
// lists no f64 MFMA row).  Each configuration runs ~0.3 s warm + ~0.3 s timed on random-ish
// operands; the shader clock actually held is measured in-kernel (s_memtime / s_memrealtime).
// Build: hipcc --offload-arch=gfx950 -O3 tools/peak_probe.hip -o tools/peak_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef double d4 __attribute__((ext_vector_type(4)));

// NACC independent accumulators, each MFMA with its own A/B registers; NFMA extra independent
// v_fma_f64 per MFMA (to see whether the f64 vector ALU and the f64 MFMA share a pipe).
template <int NACC, int NFMA = 0>
__global__ __launch_bounds__(256) void mfma_loop(double* out, unsigned long long* clk, int iters, double a0, double b0) {
    d4 acc[NACC];
    double a[NACC], b[NACC], v[NFMA > 0 ? NFMA : 1];
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
        acc[i] = d4{0.1 * i, 0.2, 0.3, 0.4};
        a[i] = a0 + ((threadIdx.x + 3 * i) % 17) * 0.037;
        b[i] = b0 - ((threadIdx.x + 5 * i) % 13) * 0.011;
    }
#pragma unroll
    for (int j = 0; j < (NFMA > 0 ? NFMA : 1); ++j) v[j] = 1.0 + j;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters / 16; ++it) {
#pragma unroll
        for (int ui = 0; ui < 16 * NACC; ++ui) {
            const int i = ui % NACC;
            // inline asm pins the accumulator in AGPRs: the builtin form made hipcc shuttle every
            // accumulator VGPR<->AGPR on each loop iteration (128 v_accvgpr per 8 MFMAs)
            asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a[i]), "v"(b[i]));
#pragma unroll
            for (int j = 0; j < NFMA; ++j) v[j] = fma(v[j], 0.9999999, 1e-9);
        }
    }
    if (NFMA > 0) {
        double sv = 0;
#pragma unroll
        for (int j = 0; j < NFMA; ++j) sv += v[j];
        acc[0][0] += sv;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) {
        clk[2 * blockIdx.x] = t1 - t0;
        clk[2 * blockIdx.x + 1] = r1 - r0;
    }
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
static float time_ms(F launch, int reps) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int i = 0; i < reps; ++i) launch();     // warm: let the clock settle under this load
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) launch();
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

template <int NACC, int NFMA = 0>
static void run_mfma(int cus, int wpc, double* out, unsigned long long* clk) {
    const int grid = cus * wpc;
    const int iters = 200000 / NACC;
    float ms = time_ms([&] { hipLaunchKernelGGL((mfma_loop<NACC, NFMA>), dim3(grid), dim3(256), 0, 0, out, clk, iters, 1.01, 0.99); }, 6);
    std::vector<unsigned long long> h(2 * grid);
    (void)hipMemcpy(h.data(), clk, sizeof(unsigned long long) * 2 * grid, hipMemcpyDeviceToHost);
    std::vector<double> mhz;
    for (int i = 0; i < grid; ++i) mhz.push_back((double)h[2 * i] / (double)h[2 * i + 1] * 100.0);
    std::sort(mhz.begin(), mhz.end());
    const double clock_mhz = mhz[grid / 2];
    const double n_mfma = (double)NACC * iters * wpc;                  // per SIMD
    const double flops = 2.0 * 16 * 16 * 4 * n_mfma * 4.0 * cus;
    printf("mfma_f64_16x16x4  acc=%2d  +%d v_fma_f64/MFMA  waves/SIMD=%d : %7.2f ms  %6.1f TFLOP/s(mfma)  clock %4.0f MHz  %5.1f cycles/MFMA/SIMD\n",
           NACC, NFMA, wpc, ms, flops / ms / 1e9, clock_mhz, ms * 1e-3 * clock_mhz * 1e6 / n_mfma);
}

int main() {
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    double* out;
    unsigned long long* clk;
    (void)hipMalloc(&out, sizeof(double) * 256 * cus * 8);
    (void)hipMalloc(&clk, sizeof(unsigned long long) * 2 * cus * 8);
    printf("device: %s, %d CUs, nominal clock %d MHz\n", p.gcnArchName, cus, p.clockRate / 1000);
    run_mfma<1>(cus, 1, out, clk);
    run_mfma<2>(cus, 1, out, clk);
    run_mfma<4>(cus, 1, out, clk);
    run_mfma<8>(cus, 1, out, clk);
    run_mfma<16>(cus, 1, out, clk);
    run_mfma<8>(cus, 2, out, clk);
    run_mfma<4>(cus, 4, out, clk);
    run_mfma<8, 1>(cus, 1, out, clk);
    run_mfma<8, 2>(cus, 1, out, clk);
    run_mfma<8, 4>(cus, 1, out, clk);
    run_mfma<8, 8>(cus, 1, out, clk);
    run_mfma<8, 4>(cus, 2, out, clk);
    for (int wpc : {1, 2, 4}) {
        const int grid = cus * wpc;
        const int iters = 200000;
        float ms = time_ms([&] { hipLaunchKernelGGL(fma_loop, dim3(grid), dim3(256), 0, 0, out, iters, 1.0000001, 1e-9); }, 4);
        double flops = 2.0 * 16 * iters * 256.0 * grid;
        printf("v_fma_f64 x16 acc, %d wave/SIMD: %7.2f ms  %6.1f TFLOP/s\n", wpc, ms, flops / ms / 1e9);
    }
    (void)hipFree(out);
    (void)hipFree(clk);
    return 0;
}
