// Instantiations of the E-step kernels for T = ceil(D/16) in 1..8, x in {f32, f64}.
#include "estep.h"
#include "launch.h"

namespace gmmvb {

template <int T, typename XT, bool VEC>
static hipError_t go(int variant, int grid, hipStream_t st, const EstepArgs& a) {
    if (variant == kEstepDirect)
        hipLaunchKernelGGL((estep_mfma_f64<T, XT, VEC>), dim3(grid), dim3(256), 0, st, static_cast<const XT*>(a.x), a.ldx,
                           a.n_rows, a.D, a.img, a.cvec, a.K, a.lnrho, a.npad);
    else if (variant == kEstepLds8)
        hipLaunchKernelGGL((estep_lds_f64<T, XT, VEC, 8>), dim3(grid), dim3(512), 0, st, static_cast<const XT*>(a.x),
                           a.ldx, a.n_rows, a.D, a.img, a.cvec, a.K, a.lnrho, a.npad);
    else
        hipLaunchKernelGGL((estep_lds_f64<T, XT, VEC, 4>), dim3(grid), dim3(256), 0, st, static_cast<const XT*>(a.x),
                           a.ldx, a.n_rows, a.D, a.img, a.cvec, a.K, a.lnrho, a.npad);
    return hipGetLastError();
}

int estep_threads(int variant) { return variant == kEstepLds8 ? 512 : 256; }
int estep_rows_per_wg(int variant, int T, int x_is_f64) {
    const int nw = variant == kEstepLds8 ? 8 : 4;
    const int nb = variant == kEstepDirect ? (x_is_f64 ? estep_nb<double>(T) : estep_nb<float>(T))
                                           : (x_is_f64 ? estep_nb_w<double>(T, nw) : estep_nb_w<float>(T, nw));
    return nw * 16 * nb;
}
int estep_image_doubles(int T) { return img_doubles(T); }

#define NAME(V, TT, X, M)                                              \
    (V == kEstepDirect ? "estep_mfma_f64<T=" #TT ",x=" X "," M ">"     \
                       : (V == kEstepLds8 ? "estep_lds_f64<T=" #TT ",x=" X "," M ",8 waves>" : "estep_lds_f64<T=" #TT ",x=" X "," M ",4 waves>"))
#define CASE(TT)                                                                                         \
    case TT:                                                                                             \
        if (x_is_f64) {                                                                                  \
            *name = vec ? NAME(variant, TT, "f64", "vec") : NAME(variant, TT, "f64", "masked");          \
            return vec ? go<TT, double, true>(variant, grid, st, a) : go<TT, double, false>(variant, grid, st, a); \
        } else {                                                                                         \
            *name = vec ? NAME(variant, TT, "f32", "vec") : NAME(variant, TT, "f32", "masked");          \
            return vec ? go<TT, float, true>(variant, grid, st, a) : go<TT, float, false>(variant, grid, st, a);   \
        }

hipError_t launch_estep(int variant, int T, int x_is_f64, bool vec, int grid, hipStream_t st, const EstepArgs& a,
                        const char** name) {
    switch (T) {
        CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8)
    }
    return hipErrorInvalidValue;
}

// ---- one feature tile on the vector ALU (estep.h: estep_rows16_f64) -----------------------------------------------------
int estep_tri_image_doubles() { return kTriImg; }
int estep_rows16_rows_per_wg() { return 256; }
hipError_t launch_pack_tri16(const double* u, const double* m, int K, int D, double* tri, hipStream_t st) {
    hipLaunchKernelGGL(pack_tri16_kernel, dim3(K), dim3(64), 0, st, u, m, K, D, tri);
    return hipGetLastError();
}
hipError_t launch_estep_rows16(int x_is_f64, bool vec, int grid, hipStream_t st, const EstepArgs& a, const double* tri,
                               const char** name) {
#define R16(XT, V, NM)                                                                                                    \
    {                                                                                                                     \
        *name = NM;                                                                                                       \
        hipLaunchKernelGGL((estep_rows16_f64<XT, V>), dim3(grid), dim3(256), 0, st, static_cast<const XT*>(a.x), a.ldx,  \
                           a.n_rows, a.D, tri, a.cvec, a.K, a.lnrho, a.npad);                                             \
        return hipGetLastError();                                                                                         \
    }
    if (x_is_f64) {
        if (vec) R16(double, true, "estep_rows16_f64<x=f64,vec>")
        R16(double, false, "estep_rows16_f64<x=f64,masked>")
    }
    if (vec) R16(float, true, "estep_rows16_f64<x=f32,vec>")
    R16(float, false, "estep_rows16_f64<x=f32,masked>")
#undef R16
}

// ---- pruned E-step -------------------------------------------------------------------------------------------
// feature tiles from which the pruned E-step is built (D >= 49): the gather kernel is instantiated for T = 4 .. 8
int estep_bound_blocks(int T) { return T >= 6 ? 3 : (T >= 4 ? 2 : 0); }
int estep_gather_rows_per_wg(int T, int x_is_f64) {
    return 8 * 16 * (x_is_f64 ? estep_nb_w<double>(T, 8) : estep_nb_w<float>(T, 8)) * kGatherTiles;
}

template <int T, typename XT, bool VEC>
static hipError_t go_gather_dev(int grid, hipStream_t st, const EstepArgs& a, const int* lists, int64_t cap,
                                const int* counts, const int* plan, const float* thr, unsigned long long* exits, float margin) {
    if (thr)
        hipLaunchKernelGGL((estep_gather_dev_f64<T, XT, VEC, true>), dim3(grid), dim3(512), 0, st, static_cast<const XT*>(a.x),
                           a.ldx, a.D, a.img, a.cvec, a.K, lists, cap, counts, plan, a.lnrho, a.npad, thr, exits, margin);
    else
        hipLaunchKernelGGL((estep_gather_dev_f64<T, XT, VEC, false>), dim3(grid), dim3(512), 0, st, static_cast<const XT*>(a.x),
                           a.ldx, a.D, a.img, a.cvec, a.K, lists, cap, counts, plan, a.lnrho, a.npad, thr, exits, margin);
    return hipGetLastError();
}

#define GDCASE(TT)                                                                                                           \
    case TT:                                                                                                                 \
        if (x_is_f64)                                                                                                        \
            return vec ? go_gather_dev<TT, double, true>(grid, st, a, lists, cap, counts_dev, plan_dev, thr, exits, margin)                      \
                       : go_gather_dev<TT, double, false>(grid, st, a, lists, cap, counts_dev, plan_dev, thr, exits, margin);                    \
        return vec ? go_gather_dev<TT, float, true>(grid, st, a, lists, cap, counts_dev, plan_dev, thr, exits, margin)                           \
                   : go_gather_dev<TT, float, false>(grid, st, a, lists, cap, counts_dev, plan_dev, thr, exits, margin);

hipError_t launch_estep_gather_dev(int T, int x_is_f64, bool vec, int grid, hipStream_t st, const EstepArgs& a,
                                   const int* lists, int64_t cap, const int* counts_dev, const int* plan_dev,
                                   const float* thr, unsigned long long* exits, float margin) {
    if (a.K > 256) return hipErrorInvalidValue;
    switch (T) {
        GDCASE(4) GDCASE(5) GDCASE(6) GDCASE(7) GDCASE(8)
    }
    return hipErrorInvalidValue;
}

}  // namespace gmmvb
