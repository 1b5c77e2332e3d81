// Instantiations of the M-step kernel.  Three load forms:
//   pre     rows come from the centred f64 copy in the workspace (default; no convert/subtract/mask in the loop)
//   vec     straight from x with T-element vector loads (T in {2,4,8}, D % 16 == 0, aligned)
//   masked  straight from x with scalar masked loads (any D)
#include "mstep.h"
#include "launch.h"

#include <cstdlib>

namespace gmmvb {

template <int T, typename XT, bool VEC, bool PRE>
static hipError_t go(int grid, hipStream_t st, const MstepArgs& a) {
    hipLaunchKernelGGL((mstep_mfma_f64<T, XT, VEC, PRE>), dim3(grid), dim3(64 * mstep_waves(T, PRE)), 0, st,
                       static_cast<const XT*>(a.x), a.ldx, a.n_rows, a.D, a.pivot, a.lnrho, a.lse, a.aux, a.npad, a.K, a.KG,
                       a.S, a.rows_per_split, a.direct_r, a.slabs);
    return hipGetLastError();
}

// one feature tile, rows from the centred copy: cw = 4 | 8 components per wave (mstep_small_f64; 5.4 against 5.1 ms at HMM
// config 5)
hipError_t launch_mstep_small(int grid, hipStream_t st, const MstepArgs& a, int KGW, int cw, const char** name) {
    if (cw == 8) {
        *name = "mstep_small_f64<T=1,8 components per wave,centred-f64>";
        hipLaunchKernelGGL((mstep_small_f64<8>), dim3(grid), dim3(256), 0, st, static_cast<const double*>(a.x), a.n_rows, a.lnrho,
                           a.lse, a.aux, a.npad, a.K, KGW, a.S, a.rows_per_split, a.direct_r, a.slabs);
    } else {
        *name = "mstep_small_f64<T=1,4 components per wave,centred-f64>";
        hipLaunchKernelGGL((mstep_small_f64<4>), dim3(grid), dim3(256), 0, st, static_cast<const double*>(a.x), a.n_rows, a.lnrho,
                           a.lse, a.aux, a.npad, a.K, KGW, a.S, a.rows_per_split, a.direct_r, a.slabs);
    }
    return hipGetLastError();
}

hipError_t launch_hmm_mstep_small(int grid, hipStream_t st, const MstepArgs& a, int KGW, const double* gamma_tm, int Kp,
                                  bool sparse, const char** name) {
    *name = sparse ? "hmm_mstep_small<T=1,8 components per wave,gamma time-major,steps above 2^-80>"
                   : "hmm_mstep_small<T=1,8 components per wave,gamma time-major>";
#define HMS(AUXV, SP)                                                                                                          \
    hipLaunchKernelGGL((hmm_mstep_small_kernel<8, AUXV, SP>), dim3(grid), dim3(256), 0, st, static_cast<const double*>(a.x), \
                       a.n_rows, gamma_tm, Kp, a.aux, a.npad, a.K, KGW, a.S, a.rows_per_split, a.slabs)
    if (a.aux) {
        if (sparse) HMS(true, true); else HMS(true, false);
    } else {
        if (sparse) HMS(false, true); else HMS(false, false);
    }
#undef HMS
    return hipGetLastError();
}

int mstep_components_per_wg(int T, bool pre) { return mstep_waves(T, pre) / mstep_ws(T); }
int mstep_threads(int T, bool pre) { return 64 * mstep_waves(T, pre); }

template <int T, bool HAS_VEC>
static hipError_t dispatch(int x_is_f64, bool vec, bool pre, int grid, hipStream_t st, const MstepArgs& a) {
    if (pre) return go<T, double, true, true>(grid, st, a);
    if constexpr (HAS_VEC) {
        if (vec) return x_is_f64 ? go<T, double, true, false>(grid, st, a) : go<T, float, true, false>(grid, st, a);
    }
    return x_is_f64 ? go<T, double, false, false>(grid, st, a) : go<T, float, false, false>(grid, st, a);
}

#define CASE(TT, HV)                                                                                       \
    case TT:                                                                                               \
        *name = pre ? "mstep_mfma_f64<T=" #TT ",centred-f64>"                                              \
                    : ((vec && HV) ? (x_is_f64 ? "mstep_mfma_f64<T=" #TT ",x=f64,vec>" : "mstep_mfma_f64<T=" #TT ",x=f32,vec>") \
                                   : (x_is_f64 ? "mstep_mfma_f64<T=" #TT ",x=f64,masked>" : "mstep_mfma_f64<T=" #TT ",x=f32,masked>")); \
        return dispatch<TT, HV>(x_is_f64, vec, pre, grid, st, a);

hipError_t launch_mstep(int T, int x_is_f64, bool vec, bool pre, int grid, hipStream_t st, const MstepArgs& a,
                        const char** name) {
    if (T > 8) return pre ? launch_mstep_wide(T, grid, st, a, name) : hipErrorInvalidValue;      // (mstep_wide_inst.hip)
    switch (T) {
        CASE(1, false) CASE(2, true) CASE(3, false) CASE(4, true) CASE(5, false) CASE(6, false) CASE(7, false) CASE(8, true)
    }
    return hipErrorInvalidValue;
}

template <int T>
static hipError_t go_list(int grid, hipStream_t st, const MstepListArgs& a) {
    hipLaunchKernelGGL(mstep_plan_kernel, dim3(1), dim3(64), 0, st, a.counts, a.K, a.cap_chunks, a.r_min, a.plan);
    if (a.x32)
        hipLaunchKernelGGL((mstep_list_x32_f64<T>), dim3(grid), dim3(64 * mstep_waves(T, true)), 0, st, a.x32, a.ldx, a.n_rows,
                           a.D, a.pivot, a.lnrho, a.lse, a.lists, a.cap, a.counts, a.plan, a.npad, a.K, a.slabs, a.direct_r);
    else
        hipLaunchKernelGGL((mstep_list_f64<T>), dim3(grid), dim3(64 * mstep_waves(T, true)), 0, st, a.xc, a.lnrho, a.lse,
                           a.lists, a.cap, a.counts, a.plan, a.npad, a.K, a.slabs, a.direct_r);
    return hipGetLastError();
}

#define LCASE(TT)                                        \
    case TT:                                             \
        *name = a.x32 ? "mstep_list_f64<T=" #TT ",x=f32>" : "mstep_list_f64<T=" #TT ",centred-f64>"; \
        return go_list<TT>(grid, st, a);

hipError_t launch_mstep_list(int T, int grid, hipStream_t st, const MstepListArgs& a, const char** name) {
    switch (T) {
        LCASE(1) LCASE(2) LCASE(3) LCASE(4) LCASE(5) LCASE(6) LCASE(7) LCASE(8)
    }
    return hipErrorInvalidValue;
}

}  // namespace gmmvb
