// Probe: do vector-ALU instructions issued between int8 MFMAs cost MFMA throughput on gfx950?
// Each loop iteration issues 4 independent v_mfma_i32_32x32x32_i8 (128 matrix-pipe cycles) and NV vector
// instructions of one kind (f32 fma, packed f32 fma, f64 fma, int lshl_add), 1 or 2 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 tools/overlap_probe.hip -o tools/overlap_probe
#include <hip/hip_runtime.h>
#include <cstdio>

typedef int i4 __attribute__((ext_vector_type(4)));
typedef int i16 __attribute__((ext_vector_type(16)));
typedef float f2 __attribute__((ext_vector_type(2)));

template <int KIND, int NV>
__global__ __launch_bounds__(256) void k(int* out, int iters) {
    i4 a = {(int)threadIdx.x * 0x01010101, 0x01020304, 0x05060708, 0x090a0b0c};
    i4 b = {0x11121314, (int)threadIdx.x, 0x0a0b0c0d, 0x01010101};
    i16 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = i16{0};
    float f[16];
    double d[16];
    f2 p[16];
    int n[16];
    for (int i = 0; i < 16; ++i) {
        f[i] = threadIdx.x * 0.001f + i;
        d[i] = threadIdx.x * 0.001 + i;
        p[i] = f2{f[i], f[i] + 1.0f};
        n[i] = threadIdx.x + i;
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            acc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[i], 0, 0, 0);
#pragma unroll
            for (int v = 0; v < NV / 4; ++v) {
                const int j = (i * (NV / 4) + v) & 15;
                if (KIND == 1) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f[j]));
                if (KIND == 2) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(p[j]));
                if (KIND == 3) asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(d[j]));
                if (KIND == 4) asm volatile("v_lshl_add_u32 %0, %0, 7, %0" : "+v"(n[j]));
                if (KIND == 5) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(n[j]));
            }
        }
    }
    int s = 0;
    for (int i = 0; i < 4; ++i)
        for (int g = 0; g < 16; ++g) s += acc[i][g];
    for (int i = 0; i < 16; ++i) s += (int)f[i] + (int)d[i] + (int)p[i][0] + n[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND, int NV>
void run(const char* what, int* out, int cus) {
    const int iters = 100000;
    for (int wpc : {1, 2}) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipLaunchKernelGGL((k<KIND, NV>), dim3(cus * wpc), dim3(256), 0, 0, out, iters);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<KIND, NV>), dim3(cus * wpc), dim3(256), 0, 0, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        // cycles per loop iteration per SIMD at the nominal clock (4 MFMAs = 128 matrix-pipe cycles)
        printf("%-28s NV=%2d  %d wave/SIMD: %7.2f ms = %6.1f cycles per 4 MFMAs per wave slot\n", what, NV, wpc, ms,
               ms * 1e-3 * 2.4e9 / iters / wpc);
    }
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    int* out;
    hipMalloc(&out, 4 * 256 * p.multiProcessorCount * 2);
    const int cus = p.multiProcessorCount;
    run<0, 0>("MFMA only", out, cus);
    run<1, 16>("+ v_fma_f32", out, cus);
    run<1, 32>("+ v_fma_f32", out, cus);
    run<2, 16>("+ v_pk_fma_f32", out, cus);
    run<2, 32>("+ v_pk_fma_f32", out, cus);
    run<3, 16>("+ v_fma_f64", out, cus);
    run<3, 32>("+ v_fma_f64", out, cus);
    run<4, 32>("+ v_lshl_add_u32", out, cus);
    run<5, 32>("+ v_cvt_f32_i32", out, cus);
    return 0;
}
