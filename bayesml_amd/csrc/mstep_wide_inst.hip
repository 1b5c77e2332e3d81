// Instantiations of the dense M-step for 128 < D <= 256 (T = 10, 12, 14, 16 feature tiles; an odd tile count is rounded
// up when the workspace is created): ceil(T / 2) waves per component, rows from the centred f64 copy (mstep.h).
#include "mstep.h"
#include "launch.h"

namespace gmmvb {

template <int T>
static hipError_t go_wide(int grid, hipStream_t st, const MstepArgs& a) {
    hipLaunchKernelGGL((mstep_wide_f64<T>), dim3(grid), dim3(32 * T), 0, st, static_cast<const double*>(a.x), a.n_rows, a.lnrho,
                       a.lse, a.aux, a.npad, a.K, a.KG, a.S, a.rows_per_split, a.direct_r, a.slabs);
    return hipGetLastError();
}

hipError_t launch_mstep_wide(int T, int grid, hipStream_t st, const MstepArgs& a, const char** name) {
    switch (T) {
        case 10: *name = "mstep_wide_f64<T=10,centred-f64,5 waves per component>"; return go_wide<10>(grid, st, a);
        case 12: *name = "mstep_wide_f64<T=12,centred-f64,6 waves per component>"; return go_wide<12>(grid, st, a);
        case 14: *name = "mstep_wide_f64<T=14,centred-f64,7 waves per component>"; return go_wide<14>(grid, st, a);
        case 16: *name = "mstep_wide_f64<T=16,centred-f64,8 waves per component>"; return go_wide<16>(grid, st, a);
    }
    return hipErrorInvalidValue;
}

}  // namespace gmmvb
