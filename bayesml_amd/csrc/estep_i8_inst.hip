// Instantiations of the int8-digit E-step for T32 = ceil(D/32) in 1..4, x in {f32, f64}.
#include "estep_i8.h"
#include "launch.h"

namespace gmmvb {

int estep_i8_image_bytes(int D) { return i8_img_bytes(i8_blocks(D)); }
int estep_i8_rows_per_wg() { return 8 * 32; }

hipError_t launch_pack_i8(const double* u, const double* m, const double* pivot, int K, int D, unsigned char* img,
                          hipStream_t st) {
    const int t32 = i8_blocks(D);
    hipLaunchKernelGGL(pack_params_i8_kernel, dim3(K), dim3(256), 0, st, u, m, pivot, K, D, t32, i8_img_bytes(t32), img);
    return hipGetLastError();
}

template <int T32, typename XT, bool VEC>
static hipError_t go(int grid, hipStream_t st, const EstepI8Args& a) {
    hipLaunchKernelGGL((estep_i8<T32, XT, VEC, 8>), dim3(grid), dim3(512), 0, st, static_cast<const XT*>(a.x), a.ldx,
                       a.n_rows, a.D, a.img, a.pivot, a.cvec, a.K, a.lnrho, a.npad);
    return hipGetLastError();
}

#define CASE(TT)                                                                                     \
    case TT:                                                                                         \
        if (x_is_f64) {                                                                              \
            *name = vec ? "estep_i8<T32=" #TT ",x=f64,vec>" : "estep_i8<T32=" #TT ",x=f64,masked>";   \
            return vec ? go<TT, double, true>(grid, st, a) : go<TT, double, false>(grid, st, a);     \
        } else {                                                                                     \
            *name = vec ? "estep_i8<T32=" #TT ",x=f32,vec>" : "estep_i8<T32=" #TT ",x=f32,masked>";   \
            return vec ? go<TT, float, true>(grid, st, a) : go<TT, float, false>(grid, st, a);       \
        }

hipError_t launch_estep_i8(int x_is_f64, bool vec, int grid, hipStream_t st, const EstepI8Args& a, const char** name) {
    switch (i8_blocks(a.D)) {
        CASE(1) CASE(2) CASE(3) CASE(4)
    }
    return hipErrorInvalidValue;
}

}  // namespace gmmvb
