// Instantiations of the int8-digit E-step for T32 = ceil(D/32) in 1..4, x in {f32, f64}: the 6-digit E-step and the
// 3-digit bound pass of the pruned E-step.
#include <cstdint>
#include <cstdlib>
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
                       static_cast<const XT*>(a.x), a.ldx, a.n_rows, a.D, a.img, a.pivot, a.cvec, a.K, a.lnrho, a.npad, a.khat, a.ub);
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

// sample digits in HBM; two-sided bounds of listed pairs over them (the proof round): all ceil(D / 32) blocks, three digits
int64_t estep_i8_digit_row_bytes(int D) { return i8_digit_row_bytes(i8_blocks(D)); }

hipError_t launch_x_digits(const void* x, int x_is_f64, int64_t ldx, int64_t n_rows, int D, const double* pivot,
                           unsigned char* xq, signed char* xqe, hipStream_t st) {
    const int t32 = i8_blocks(D);
    if (t32 < 1 || t32 > 4) return hipErrorInvalidValue;
    const unsigned grid = (unsigned)((n_rows + 31) / 32);
    if (x_is_f64)
        hipLaunchKernelGGL(x_digits_kernel<double>, dim3(grid), dim3(256), 0, st, static_cast<const double*>(x), ldx, n_rows, D,
                           t32, pivot, xq, xqe);
    else
        hipLaunchKernelGGL(x_digits_kernel<float>, dim3(grid), dim3(256), 0, st, static_cast<const float*>(x), ldx, n_rows, D,
                           t32, pivot, xq, xqe);
    return hipGetLastError();
}

int estep_i8_pairs_per_chunk() { return i8_pairs_per_chunk(); }

hipError_t launch_estep_i8_proof(int D, int grid, hipStream_t st, const unsigned char* xq, const signed char* xqe,
                                 const unsigned char* img, const double* cvec, int K, const int* lists, int64_t cap,
                                 const int* counts, const int* plan, float* ub, double* lb, int64_t npad) {
    if (K > 256) return hipErrorInvalidValue;
    grid *= 2;      // two workgroups per CU (108 registers: four waves per SIMD): 1.21 -> 1.10 ms per step at the benchmark shape;
                    // the kernel streams 385 B of digit planes per pair at 3.4 TB/s, three or four per CU change nothing
#define PC(T)                                                                                                              \
    case T:                                                                                                                \
        hipLaunchKernelGGL((estep_i8_proof<T>), dim3(grid), dim3(512), 0, st, xq, xqe, img, cvec, K, lists, cap, counts,   \
                           plan, ub, lb, npad);                                                                            \
        return hipGetLastError();
    switch (i8_blocks(D)) {
        PC(1) PC(2) PC(3) PC(4)
    }
#undef PC
    return hipErrorInvalidValue;
}


// The proof round over row superblocks (estep_i8.h, estep_i8_proof_blocked).  `work` holds the item table: per superblock
// totals and bases, per (superblock, component) item counts, then the items themselves - at most pairs / 256 + one per
// (superblock, component).  The caller hands over a buffer that is free during the E-step (the M-step's slabs).
int64_t estep_i8_proof_work_bytes(int K, int64_t n_rows) {
    const int64_t nblk = (n_rows + 255) / 256;
    const int64_t n_super = (nblk * 256 + kProofSuperRows - 1) / kProofSuperRows;
    const int64_t head = ((2 * n_super + kProofXcds + 3) / 4 * 4 + (n_super * (K + 1) + 3) / 4 * 4) * (int64_t)sizeof(int);
    const int64_t items = n_rows * (int64_t)K / kProofItem + n_super * K + 1;
    return head + items * (int64_t)sizeof(i4v);
}

hipError_t launch_estep_i8_proof_blocked(int D, int num_cu, hipStream_t st, const unsigned char* xq, const signed char* xqe,
                                         const unsigned char* img, const double* cvec, int K, const int* lists, int64_t cap,
                                         const int* counts, const int* blk_base, int nblk, void* work, float* ub, double* lb,
                                         int64_t npad) {
    if (K > 256 || nblk < 1 || !work || ((uintptr_t)work & 15)) return hipErrorInvalidValue;
    const int n_super = (int)(((int64_t)nblk * 256 + kProofSuperRows - 1) / kProofSuperRows);
    int* tot = static_cast<int*>(work);
    int* xb = tot + n_super;
    int* xtot = xb + n_super;
    int* cum = tot + (2 * (int64_t)n_super + kProofXcds + 3) / 4 * 4;
    i4v* items = reinterpret_cast<i4v*>(cum + ((int64_t)n_super * (K + 1) + 3) / 4 * 4);
    hipLaunchKernelGGL(proof_units_kernel, dim3(n_super), dim3(256), 0, st, blk_base, counts, K, nblk, cum, tot);
    hipLaunchKernelGGL(proof_order_kernel, dim3(kProofXcds), dim3(1024), 0, st, tot, n_super, xb, xtot);
    hipLaunchKernelGGL(proof_items_kernel, dim3(n_super), dim3(256), 0, st, blk_base, counts, K, nblk, cum, xb, xtot, items);
    // as many workgroups as are resident at once (182 - 222 registers past one feature block of 32: one per CU) - the items
    // are dealt out interleaved, and a workgroup that starts late would go through the superblocks a second time
    const int per_cu = i8_blocks(D) == 1 ? 2 : 1;
    const int grid = kProofXcds * ((per_cu * num_cu + kProofXcds - 1) / kProofXcds);
#define PB(T)                                                                                                              \
    case T:                                                                                                                \
        hipLaunchKernelGGL((estep_i8_proof_blocked<T>), dim3(grid), dim3(512), 0, st, xq, xqe, img, cvec, K, lists, cap,   \
                           items, xtot, ub, lb, npad);                                                                     \
        return hipGetLastError();
    switch (i8_blocks(D)) {
        PB(1) PB(2) PB(3) PB(4)
    }
#undef PB
    return hipErrorInvalidValue;
}

}  // namespace gmmvb
