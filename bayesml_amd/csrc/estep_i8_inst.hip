// Instantiations of the int8-digit E-step for T32 = ceil(D/32) in 1..4, x in {f32, f64}: the 6-digit E-step and the
// 3-digit bound pass of the pruned E-step.
#include "estep_i8.h"
#include "launch.h"

namespace gmmvb {

int estep_i8_image_bytes(int D, int bound) { return i8_img_bytes(bound ? kBoundDigits : kDigits, i8_blocks(D)); }
int estep_i8_rows_per_wg() { return 8 * 32; }

hipError_t launch_pack_i8(const double* u, const double* m, const double* pivot, int K, int D, unsigned char* img,
                          int bound, hipStream_t st) {
    const int t32 = i8_blocks(D);
    if (bound)
        hipLaunchKernelGGL(pack_params_i8_kernel<kBoundDigits>, dim3(K), dim3(256), 0, st, u, m, pivot, K, D, t32,
                           i8_img_bytes(kBoundDigits, t32), img);
    else
        hipLaunchKernelGGL(pack_params_i8_kernel<kDigits>, dim3(K), dim3(256), 0, st, u, m, pivot, K, D, t32,
                           i8_img_bytes(kDigits, t32), img);
    return hipGetLastError();
}

template <int ND, bool BOUND, int T32, typename XT, bool VEC>
static hipError_t go(int grid, hipStream_t st, const EstepI8Args& a) {
    hipLaunchKernelGGL((estep_i8<ND, BOUND, T32, XT, VEC, 8>), dim3(grid), dim3(512), 0, st, static_cast<const XT*>(a.x),
                       a.ldx, a.n_rows, a.D, a.img, a.pivot, a.cvec, a.K, a.lnrho, a.npad);
    return hipGetLastError();
}

template <int ND, bool BOUND>
static hipError_t pick(int x_is_f64, bool vec, int grid, hipStream_t st, const EstepI8Args& a) {
#define CASE(TT)                                                                                                    \
    case TT:                                                                                                        \
        if (x_is_f64) return vec ? go<ND, BOUND, TT, double, true>(grid, st, a) : go<ND, BOUND, TT, double, false>(grid, st, a); \
        return vec ? go<ND, BOUND, TT, float, true>(grid, st, a) : go<ND, BOUND, TT, float, false>(grid, st, a);
    switch (i8_blocks(a.D)) {
        CASE(1) CASE(2) CASE(3) CASE(4)
    }
#undef CASE
    return hipErrorInvalidValue;
}

hipError_t launch_estep_i8(int x_is_f64, bool vec, int grid, hipStream_t st, const EstepI8Args& a, const char** name) {
    static const char* names[] = {"", "estep_i8<T32=1>", "estep_i8<T32=2>", "estep_i8<T32=3>", "estep_i8<T32=4>"};
    *name = names[i8_blocks(a.D) <= 4 ? i8_blocks(a.D) : 0];
    return pick<kDigits, false>(x_is_f64, vec, grid, st, a);
}

hipError_t launch_estep_i8_bound(int x_is_f64, bool vec, int grid, hipStream_t st, const EstepI8Args& a,
                                 const char** name) {
    static const char* names[] = {"", "estep_i8_bound<T32=1>", "estep_i8_bound<T32=2>", "estep_i8_bound<T32=3>",
                                  "estep_i8_bound<T32=4>"};
    *name = names[i8_blocks(a.D) <= 4 ? i8_blocks(a.D) : 0];
    return pick<kBoundDigits, true>(x_is_f64, vec, grid, st, a);
}

}  // namespace gmmvb
