// Launchers of the stateless sweep (project.h).
#include <algorithm>

#include "launch.h"
#include "project.h"

namespace gmmvb {

int64_t proj_image_len(int K, int D) { return (int64_t)K * proj_image_bytes(K, proj_blocks(D)); }
int64_t proj_const_len(int K) { return (int64_t)K * proj_kblocks(K) * 32; }

hipError_t launch_proj_table(const double* u, const double* m, const double* cvec, const double* pivot, int K, int D,
                             unsigned char* gimg, void* gconst, hipStream_t st) {
    const int t32 = proj_blocks(D);
    if (t32 < 1 || t32 > 4 || K < 1 || K > 256 || D > 128) return hipErrorInvalidValue;
    const size_t lds = (size_t)proj_tri_len(D) * sizeof(double);
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)proj_table_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
        attr_done = true;
    }
    // references dealt out over gridDim.y workgroups per component: a workgroup walks at most 16 of them one after the other
    // (a reference is two triangular matrix-vector products and two block reductions: ~7 us; with 85 per workgroup the
    // kernel was 0.67 ms at K = 256)
    const int js = std::max(1, (K + 15) / 16);
    hipLaunchKernelGGL(proj_table_kernel, dim3((unsigned)K, (unsigned)js), dim3(256), lds, st, u, m, cvec, pivot, K, D, t32, gimg,
                       static_cast<float4*>(gconst));
    return hipGetLastError();
}

hipError_t launch_proj_tile_ref(const int* counts, int K, int64_t n_tiles, int* tile_ref, hipStream_t st) {
    if (K > 1024) return hipErrorInvalidValue;
    hipLaunchKernelGGL(proj_tile_ref_kernel, dim3(1), dim3(256), 0, st, counts, K, n_tiles, tile_ref);
    return hipGetLastError();
}

hipError_t launch_rec_project(int grid, hipStream_t st, const ProjectArgs& a) {
    const int t32 = proj_blocks(a.D);
    if (t32 < 2 || t32 > 4 || a.K > 256) return hipErrorInvalidValue;
    const RecArrays rec{a.rec_k, a.rec_d, a.rec_B, a.rec_exact, a.rec_sel, a.rec_flags, a.npad};
#define GMMVB_PROJECT(T)                                                                                                        \
    do {                                                                                                                        \
        hipLaunchKernelGGL((rec_project_kernel<T>), dim3((unsigned)grid), dim3(kSelRows), 0, st, a.xq, a.xqe, a.gimg,            \
                           static_cast<const float4*>(a.gconst), a.tile_ref, a.lnrho, a.npad, a.n_rows, a.K, a.drift, a.cvec,  \
                           rec, a.masks, a.blk, a.epart, a.opart, a.lock, a.dlock, a.rthr, a.lcomp, a.pmask, a.pblk,           \
                           a.proof_all, a.own_fresh, a.cand_ctr);                                                              \
    } while (0)
    if (t32 == 2) GMMVB_PROJECT(2);
    else if (t32 == 3) GMMVB_PROJECT(3);
    else GMMVB_PROJECT(4);
#undef GMMVB_PROJECT
    return hipGetLastError();
}

hipError_t launch_proj_filter(int grid, hipStream_t st, const ProjectArgs& a) {
    const int t32 = proj_blocks(a.D);
    if (t32 < 2 || t32 > 4 || a.K > 256) return hipErrorInvalidValue;
    const RecArrays rec{a.rec_k, a.rec_d, a.rec_B, a.rec_exact, a.rec_sel, a.rec_flags, a.npad};
#define GMMVB_FILTER(T)                                                                                                         \
    hipLaunchKernelGGL((proj_filter_kernel<T>), dim3((unsigned)grid), dim3(kSelRows), 0, st, a.xq, a.xqe, a.gimg,               \
                       static_cast<const float4*>(a.gconst), a.tile_ref, a.npad, a.n_rows, a.K, a.rthr, rec, a.masks, a.pmask,  \
                       a.pblk, a.lcomp, a.cand_ctr)
    if (t32 == 2) GMMVB_FILTER(2);
    else if (t32 == 3) GMMVB_FILTER(3);
    else GMMVB_FILTER(4);
#undef GMMVB_FILTER
    return hipGetLastError();
}

}  // namespace gmmvb
