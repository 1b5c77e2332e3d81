// Host-side launch entry points implemented in the per-kernel translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct gmmvb_hmm_state;

namespace gmmvb {

struct EstepArgs {
    const void* x; int64_t ldx; int64_t n_rows; int D;
    const double* img; const double* cvec; int K;
    double* lnrho; int64_t npad;
};
struct MstepArgs {
    const void* x; int64_t ldx; int64_t n_rows; int D;
    const double* pivot; const double* lnrho; const double* lse; const double* aux; int64_t npad;
    int K; int KG; int S; int64_t rows_per_split; int direct_r; double* slabs;
};

// rows of x handled by one E-step workgroup for (variant, T, dtype); doubles per component parameter image
enum EstepVariant { kEstepLds = 0, kEstepDirect = 1, kEstepLds8 = 2, kEstepI8 = 3, kEstepValu16 = 4 };
int estep_rows_per_wg(int variant, int T, int x_is_f64);
int estep_threads(int variant);
int estep_image_doubles(int T);
// pruned E-step (estep.h): exact kernel over per-component sample lists; the bounds come from estep_i8.h
// D <= 16 on the vector ALU (estep.h: estep_rows16_f64): packed lower-triangular images [K][estep_tri_image_doubles()],
// estep_rows16_rows_per_wg() rows per workgroup (a lane per row, several rows per lane)
int estep_tri_image_doubles();
int estep_rows16_rows_per_wg();
hipError_t launch_pack_tri16(const double* u, const double* m, int K, int D, double* tri, hipStream_t st);
hipError_t launch_estep_rows16(int x_is_f64, bool vec, int grid, hipStream_t st, const EstepArgs& a, const double* tri,
                               const char** name);
// the emission of one feature tile written as rho' rows and row maxima into the HMM state (hmm.h: hmm_emission_mfma16_kernel;
// hmm_capi.hip)
hipError_t hmm_launch_emission16(::gmmvb_hmm_state* h, int x_is_f64, bool vec, hipStream_t st, const EstepArgs& a, const char** name);
int estep_bound_blocks(int T);                    // 0 = the pruned path is not built for this T (D < 49)
int estep_gather_rows_per_wg(int T, int x_is_f64);
// exact f64 evaluation of listed pairs (device lists [K][cap], device counts), chunk plan on the device
// (records.h: gather_plan_kernel), a fixed grid of persistent workgroups
hipError_t launch_estep_gather_dev(int T, int x_is_f64, bool vec, int grid, hipStream_t st, const EstepArgs& a,
                                   const int* lists, int64_t cap, const int* counts_dev, const int* plan_dev,
                                   const float* thr = nullptr /*[npad] relevance thresholds: early way out for irrelevant pairs*/,
                                   unsigned long long* exits = nullptr /*with thr: counter of the pairs that took it*/,
                                   float margin = 0.0f /*nats below thr a partial bound has to lie to take it*/);
// sample digits kept in HBM and the proof round over them (estep_i8.h: x_digits_kernel, estep_i8_proof)
int64_t estep_i8_digit_row_bytes(int D);
hipError_t launch_x_digits(const void* x, int x_is_f64, int64_t ldx, int64_t n_rows, int D, const double* pivot,
                           unsigned char* xq, signed char* xqe, hipStream_t st);
// the stateless sweep (project.h): table of the K x K reference frames from the parameters in force, the reference of every
// tile of 256 regrouped rows, and the sweep itself (outputs of rec_sweep_kernel<PREV>)
int64_t proj_image_len(int K, int D);         // bytes of gimg
int64_t proj_const_len(int K);                // float4 entries of gconst
hipError_t launch_proj_table(const double* u, const double* m, const double* cvec, const double* pivot, int K, int D,
                             unsigned char* gimg, void* gconst, hipStream_t st);
hipError_t launch_proj_tile_ref(const int* counts, int K, int64_t n_tiles, int* tile_ref, hipStream_t st);
struct ProjectArgs {
    const unsigned char* xq; const signed char* xqe;
    const unsigned char* gimg; const void* gconst; const int* tile_ref;
    const double* lnrho; int64_t npad; int64_t n_rows; int K; int D; const double* drift; const double* cvec;
    unsigned short* rec_k; float* rec_d; float* rec_B; unsigned char* rec_exact; unsigned char* rec_sel; unsigned char* rec_flags;
    unsigned long long* masks; int* blk; double* epart; double* opart;
    unsigned char* lock; float* dlock; float* rthr; const unsigned char* lcomp;
    unsigned long long* pmask; int* pblk; int proof_all; int own_fresh; unsigned long long* cand_ctr;
};
hipError_t launch_rec_project(int grid, hipStream_t st, const ProjectArgs& a);
// the table as a filter between a carried sweep and its proof round: uses xq .. tile_ref, npad, n_rows, K, D, rthr, rec_*, masks,
// pmask, pblk, lcomp of the same argument block; cand_ctr += proof pairs the table made unnecessary
hipError_t launch_proj_filter(int grid, hipStream_t st, const ProjectArgs& a);
int estep_i8_pairs_per_chunk();
// img = the 3-digit images; for every listed pair ub[k][row] <- upper bound (f32, rounded up), lb[k][row] <- lower bound of ln rho
hipError_t launch_estep_i8_proof(int D, int grid, hipStream_t st, const unsigned char* xq, const signed char* xqe,
                                 const unsigned char* img, const double* cvec, int K, const int* lists, int64_t cap,
                                 const int* counts, const int* plan, float* ub, double* lb, int64_t npad);
// the same with the pairs regrouped by row superblock so that a row's digit planes are fetched once per pass (estep_i8.h,
// estep_i8_proof_blocked): blk_base = the selection blocks' exclusive bases the lists were filled from ([nblk][K]), work =
// estep_i8_proof_work_bytes(K, n_rows) bytes of 16-byte aligned scratch
int64_t estep_i8_proof_work_bytes(int K, int64_t n_rows);
hipError_t launch_estep_i8_proof_blocked(int D, int num_cu, hipStream_t st, const unsigned char* xq, const signed char* xqe,
                                         const unsigned char* img, const double* cvec, int K, const int* lists, int64_t cap,
                                         const int* counts, const int* blk_base, int nblk, void* work, float* ub, double* lb,
                                         int64_t npad);
// int8-digit E-step (estep_i8.h): own parameter image (bytes per component), 256 rows per workgroup
struct EstepI8Args {
    const void* x; int64_t ldx; int64_t n_rows; int D;
    const unsigned char* img; const double* pivot; const double* cvec; int K;
    double* lnrho; int64_t npad;
    int* khat = nullptr;      // optional: first maximiser over k of what the kernel stores, per row
    float* ub = nullptr;      // bound pass: [K][npad] f32 upper bounds (rounded up) - written INSTEAD of lnrho
};
// bound = 1: the 3-digit image / kernel of the pruned E-step's bound pass
int estep_i8_image_bytes(int D, int bound);
int estep_i8_rows_per_wg();
hipError_t launch_pack_i8(const double* u, const double* m, const double* pivot, int K, int D, unsigned char* img,
                          int bound, hipStream_t st);
hipError_t launch_estep_i8(int x_is_f64, bool vec, int grid, hipStream_t st, const EstepI8Args& a, const char** name);
// tb = output blocks (32 rows of y each) the bound pass evaluates, 1 .. ceil(D / 32)
hipError_t launch_estep_i8_bound(int x_is_f64, bool vec, int tb, int grid, hipStream_t st, const EstepI8Args& a,
                                 const char** name);
// components handled by one M-step workgroup for T feature tiles (4 waves / waves-per-component)
int mstep_components_per_wg(int T, bool pre);
int mstep_threads(int T, bool pre);
// returns hipSuccess or the launch error; `name` receives a static description of the instantiation
hipError_t launch_estep(int variant, int T, int x_is_f64, bool vec, int grid, hipStream_t st, const EstepArgs& a,
                        const char** name);
// pre = true: a.x is the workspace's centred f64 copy [n_rows][16T] (see center_rows_kernel)
hipError_t launch_mstep(int T, int x_is_f64, bool vec, bool pre, int grid, hipStream_t st, const MstepArgs& a,
                        const char** name);

// sparse-responsibility M-step over the centred copy and per-component lists of active rows (mstep.h)
hipError_t launch_mstep_small(int grid, hipStream_t st, const MstepArgs& a, int KGW, int cw, const char** name);
// 128 < D <= 256 (T = 10, 12, 14, 16): dense M-step from the centred copy, dense E-step with the image streamed through LDS
hipError_t launch_mstep_wide(int T, int grid, hipStream_t st, const MstepArgs& a, const char** name);
hipError_t launch_estep_rows(int T, int x_is_f64, int grid, hipStream_t st, const EstepArgs& a, const char** name);
int estep_rows_rows_per_wg();
// ... for the HMM: gamma read time-major ([rows][Kp], lane order) through LDS instead of a component-major copy
// (sparse: only the MFMA steps with a gamma above the relevance line are walked)
hipError_t launch_hmm_mstep_small(int grid, hipStream_t st, const MstepArgs& a, int KGW, const double* gamma_tm, int Kp,
                                  bool sparse, const char** name);

struct MstepListArgs {
    const double* xc; const double* lnrho; const double* lse;
    const int* lists; int64_t cap; const int* counts; int* plan; int cap_chunks; int r_min;
    int64_t npad; int K; double* slabs;
    const float* x32 = nullptr; int64_t ldx = 0; int64_t n_rows = 0; int D = 0; const double* pivot = nullptr;   // f32 rows instead of xc
    int direct_r = 0;      // 3: delta lists of the settled-row cache (weights +-1 from the entries' sign bits)
};
hipError_t launch_mstep_list(int T, int grid, hipStream_t st, const MstepListArgs& a, const char** name);

}  // namespace gmmvb
