// Probe: operand/accumulator lane maps and issue rate of v_mfma_i32_32x32x32_i8 on gfx950.
// Hypothesis (by analogy with the bf16 32x32x16 map of the programming guide): lane l = (r = l & 31, h = l >> 5)
// holds A[r][16 h + j] and B[16 h + j][r] in byte j = 0..15 of its 4-VGPR operand; C/D: col = l & 31,
// row = (reg & 3) + 8 (reg >> 2) + 4 h.   Build: hipcc --offload-arch=gfx950 -O3 tools/i8_probe.hip -o tools/i8_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef int i4 __attribute__((ext_vector_type(4)));
typedef int i16 __attribute__((ext_vector_type(16)));

__global__ void one_tile(const signed char* A /*[32][32]*/, const signed char* B /*[32][32] (k, n)*/, int* C /*[32][32]*/) {
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    union { i4 v; signed char b[16]; } a, b;
    for (int j = 0; j < 16; ++j) {
        a.b[j] = A[r * 32 + 16 * h + j];
        b.b[j] = B[(16 * h + j) * 32 + r];
    }
    i16 acc = {0};
    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a.v, b.v, acc, 0, 0, 0);
    for (int g = 0; g < 16; ++g) C[((g & 3) + 8 * (g >> 2) + 4 * h) * 32 + r] = acc[g];
}

__device__ unsigned mix(unsigned v) {
    v ^= v >> 16; v *= 0x7feb352du; v ^= v >> 15; v *= 0x846ca68bu; v ^= v >> 16;
    return v;
}

// random = 1: operands are random 7-bit digits per lane (realistic toggling); 0: near-constant data
__global__ __launch_bounds__(256) void rate(int* out, int iters, int random) {
    i4 a = {(int)threadIdx.x * 0x01010101, 0x01020304, 0x05060708, 0x090a0b0c};
    i4 b = {0x11121314, (int)threadIdx.x, 0x0a0b0c0d, 0x01010101};
    if (random) {
        const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
        for (int i = 0; i < 4; ++i) {
            a[i] = (int)(mix(t * 8 + i) & 0x7f7f7f7fu) - 0x40404040;
            b[i] = (int)(mix(t * 8 + 4 + i) & 0x7f7f7f7fu) - 0x40404040;
        }
    }
    i16 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = i16{0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[i], 0, 0, 0);
    }
    int s = 0;
    for (int i = 0; i < 4; ++i)
        for (int g = 0; g < 16; ++g) s += acc[i][g];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    std::vector<signed char> A(1024), B(1024);
    srand(1);
    for (auto& v : A) v = (signed char)(rand() % 129 - 64);
    for (auto& v : B) v = (signed char)(rand() % 129 - 64);
    signed char *dA, *dB;
    int* dC;
    hipMalloc(&dA, 1024);
    hipMalloc(&dB, 1024);
    hipMalloc(&dC, 4096);
    hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(one_tile, dim3(1), dim3(64), 0, 0, dA, dB, dC);
    std::vector<int> C(1024);
    hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
            int ref = 0;
            for (int k = 0; k < 32; ++k) ref += (int)A[i * 32 + k] * (int)B[k * 32 + j];
            if (ref != C[i * 32 + j]) ++bad;
        }
    printf("layout check: %d mismatches of 1024\n", bad);
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    int* out;
    hipMalloc(&out, 4 * 256 * p.multiProcessorCount * 2);
    const int iters = 200000;
    for (int cfg : {1, 2, 6}) {
        const int wpc = cfg & 3, random = cfg >> 2;
        const int grid = p.multiProcessorCount * wpc;
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipLaunchKernelGGL(rate, dim3(grid), dim3(256), 0, 0, out, iters, random);
        hipDeviceSynchronize();
        for (int rep = 0; rep < 8; ++rep) hipLaunchKernelGGL(rate, dim3(grid), dim3(256), 0, 0, out, iters, random);   // let the clocks settle
        hipEventRecord(e0);
        hipLaunchKernelGGL(rate, dim3(grid), dim3(256), 0, 0, out, iters, random);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double n = 4.0 * iters * wpc;
        printf("i8 32x32x32, %s data, %d wave/SIMD: %.2f ms, %.1f cycles/MFMA/SIMD at %d MHz, %.0f Tops/s\n", random ? "random" : "constant", wpc, ms,
               ms * 1e-3 * p.clockRate * 1e3 / n, p.clockRate / 1000, 65536.0 * n * 4 * p.multiProcessorCount / ms / 1e9);
    }
    return 0;
}
