// Instantiations of the row-streamed dense E-step for 128 < D <= 256 (estep_rows.h): T = 10, 12, 14, 16 feature tiles,
// x in {f32, f64}; the sample tile is loaded with masked scalar loads (any D, any alignment: 16 T loads per lane against
// 4 T (T + 1) MFMAs per component).
#include "estep_rows.h"
#include "launch.h"

namespace gmmvb {

template <int T, typename XT>
static hipError_t go_rows(int grid, hipStream_t st, const EstepArgs& a) {
    hipLaunchKernelGGL((estep_rows_f64<T, XT, false>), dim3(grid), dim3(512), 0, st, static_cast<const XT*>(a.x), a.ldx, a.n_rows,
                       a.D, a.img, a.cvec, a.K, a.lnrho, a.npad);
    return hipGetLastError();
}

int estep_rows_rows_per_wg() { return 8 * 16; }

#define RCASE(TT)                                                                                         \
    case TT:                                                                                              \
        *name = x_is_f64 ? "estep_rows_f64<T=" #TT ",x=f64>" : "estep_rows_f64<T=" #TT ",x=f32>";         \
        return x_is_f64 ? go_rows<TT, double>(grid, st, a) : go_rows<TT, float>(grid, st, a);

hipError_t launch_estep_rows(int T, int x_is_f64, int grid, hipStream_t st, const EstepArgs& a, const char** name) {
    switch (T) {
        RCASE(10) RCASE(12) RCASE(14) RCASE(16)
    }
    return hipErrorInvalidValue;
}

}  // namespace gmmvb
