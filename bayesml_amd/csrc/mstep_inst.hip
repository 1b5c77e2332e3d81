// Instantiations of the M-step kernel.  The vector-load form needs T elements per lane to be a
// power-of-two vector (T in {2,4,8}); other T use the masked scalar-load form.
#include "mstep.h"
#include "launch.h"

namespace gmmvb {

template <int T, typename XT, bool VEC>
static hipError_t go(int grid, hipStream_t st, const MstepArgs& a) {
    hipLaunchKernelGGL((mstep_mfma_f64<T, XT, VEC>), dim3(grid), dim3(256), 0, st,
                       static_cast<const XT*>(a.x), a.ldx, a.n_rows, a.D, a.pivot, a.lnrho, a.lse, a.npad, a.K, a.KG,
                       a.S, a.rows_per_split, a.direct_r, a.slabs);
    return hipGetLastError();
}

int mstep_components_per_wg(int T) { return 4 / mstep_ws(T); }

#define CASE_V(TT)                                                                                \
    case TT:                                                                                      \
        if (x_is_f64) {                                                                           \
            *name = vec ? "mstep_mfma_f64<T=" #TT ",x=f64,vec>" : "mstep_mfma_f64<T=" #TT ",x=f64,masked>"; \
            return vec ? go<TT, double, true>(grid, st, a) : go<TT, double, false>(grid, st, a);  \
        } else {                                                                                  \
            *name = vec ? "mstep_mfma_f64<T=" #TT ",x=f32,vec>" : "mstep_mfma_f64<T=" #TT ",x=f32,masked>"; \
            return vec ? go<TT, float, true>(grid, st, a) : go<TT, float, false>(grid, st, a);    \
        }
#define CASE_S(TT)                                                                                \
    case TT:                                                                                      \
        if (x_is_f64) {                                                                           \
            *name = "mstep_mfma_f64<T=" #TT ",x=f64,masked>";                                     \
            return go<TT, double, false>(grid, st, a);                                            \
        } else {                                                                                  \
            *name = "mstep_mfma_f64<T=" #TT ",x=f32,masked>";                                     \
            return go<TT, float, false>(grid, st, a);                                             \
        }

hipError_t launch_mstep(int T, int x_is_f64, bool vec, int grid, hipStream_t st, const MstepArgs& a,
                        const char** name) {
    switch (T) {
        CASE_S(1) CASE_V(2) CASE_S(3) CASE_V(4) CASE_S(5) CASE_S(6) CASE_S(7) CASE_V(8)
    }
    return hipErrorInvalidValue;
}

}  // namespace gmmvb
