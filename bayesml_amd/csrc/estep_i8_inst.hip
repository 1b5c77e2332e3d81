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

template <int ND, bool BOUND, int T32, int TB, typename XT, bool VEC>
static hipError_t go(int grid, hipStream_t st, const EstepI8Args& a) {
    hipLaunchKernelGGL((estep_i8<ND, BOUND, T32, TB, XT, VEC, 8>), dim3(grid), dim3(512), 0, st,
                       static_cast<const XT*>(a.x), a.ldx, a.n_rows, a.D, a.img, a.pivot, a.cvec, a.K, a.lnrho, a.npad, a.khat);
    return hipGetLastError();
}

template <int ND, bool BOUND, int T32, int TB>
static hipError_t pick_x(int x_is_f64, bool vec, int grid, hipStream_t st, const EstepI8Args& a) {
    if (x_is_f64) return vec ? go<ND, BOUND, T32, TB, double, true>(grid, st, a) : go<ND, BOUND, T32, TB, double, false>(grid, st, a);
    return vec ? go<ND, BOUND, T32, TB, float, true>(grid, st, a) : go<ND, BOUND, T32, TB, float, false>(grid, st, a);
}

hipError_t launch_estep_i8(int x_is_f64, bool vec, int grid, hipStream_t st, const EstepI8Args& a, const char** name) {
    static const char* names[] = {"", "estep_i8<T32=1>", "estep_i8<T32=2>", "estep_i8<T32=3>", "estep_i8<T32=4>"};
    const int t32 = i8_blocks(a.D);
    *name = names[t32 <= 4 ? t32 : 0];
    switch (t32) {
        case 1: return pick_x<kDigits, false, 1, 1>(x_is_f64, vec, grid, st, a);
        case 2: return pick_x<kDigits, false, 2, 2>(x_is_f64, vec, grid, st, a);
        case 3: return pick_x<kDigits, false, 3, 3>(x_is_f64, vec, grid, st, a);
        case 4: return pick_x<kDigits, false, 4, 4>(x_is_f64, vec, grid, st, a);
    }
    return hipErrorInvalidValue;
}

// tb = output blocks the bound pass evaluates, 1 .. ceil(D / 32)
hipError_t launch_estep_i8_bound(int x_is_f64, bool vec, int tb, int grid, hipStream_t st, const EstepI8Args& a,
                                 const char** name) {
    static const char* names[5][5] = {
        {"", "", "", "", ""},
        {"", "estep_i8_bound<T32=1,blocks=1>", "", "", ""},
        {"", "estep_i8_bound<T32=2,blocks=1>", "estep_i8_bound<T32=2,blocks=2>", "", ""},
        {"", "estep_i8_bound<T32=3,blocks=1>", "estep_i8_bound<T32=3,blocks=2>", "estep_i8_bound<T32=3,blocks=3>", ""},
        {"", "estep_i8_bound<T32=4,blocks=1>", "estep_i8_bound<T32=4,blocks=2>", "estep_i8_bound<T32=4,blocks=3>",
         "estep_i8_bound<T32=4,blocks=4>"}};
    const int t32 = i8_blocks(a.D);
    if (t32 < 1 || t32 > 4 || tb < 1 || tb > t32) return hipErrorInvalidValue;
    *name = names[t32][tb];
#define BC(T, B) \
    if (t32 == T && tb == B) return pick_x<kBoundDigits, true, T, B>(x_is_f64, vec, grid, st, a);
    BC(1, 1) BC(2, 1) BC(2, 2) BC(3, 1) BC(3, 2) BC(3, 3) BC(4, 1) BC(4, 2) BC(4, 3) BC(4, 4)
#undef BC
    return hipErrorInvalidValue;
}

// two-sided bounds of listed pairs (the settled rows' reference values): all ceil(D / 32) blocks, three digits
int estep_i8_pairs_per_chunk() { return i8_pairs_per_chunk(); }

template <int T32, typename XT, bool VEC>
static hipError_t go_pairs(int grid, hipStream_t st, const EstepI8Args& a, const int* lists, int64_t cap, const int* counts,
                           const int* plan, float* dist_up) {
    hipLaunchKernelGGL((estep_i8_pairs<T32, XT, VEC>), dim3(grid), dim3(512), 0, st, static_cast<const XT*>(a.x), a.ldx, a.D,
                       a.img, a.pivot, a.K, lists, cap, counts, plan, dist_up);
    return hipGetLastError();
}

hipError_t launch_estep_i8_pairs(int x_is_f64, bool vec, int grid, hipStream_t st, const EstepI8Args& a, const int* lists,
                                 int64_t cap, const int* counts, const int* plan, float* dist_up) {
    if (a.K > 256) return hipErrorInvalidValue;
#define PC(T)                                                                                                         \
    case T:                                                                                                           \
        if (x_is_f64)                                                                                                 \
            return vec ? go_pairs<T, double, true>(grid, st, a, lists, cap, counts, plan, dist_up)                    \
                       : go_pairs<T, double, false>(grid, st, a, lists, cap, counts, plan, dist_up);                  \
        return vec ? go_pairs<T, float, true>(grid, st, a, lists, cap, counts, plan, dist_up)                         \
                   : go_pairs<T, float, false>(grid, st, a, lists, cap, counts, plan, dist_up);
    switch (i8_blocks(a.D)) {
        PC(1) PC(2) PC(3) PC(4)
    }
#undef PC
    return hipErrorInvalidValue;
}

}  // namespace gmmvb
