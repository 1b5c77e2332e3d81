// C ABI of the GMM-VB data-pass engine, the read-outs of a pass: responsibilities, ln rho, hard assignments (include/gmmvb.h; shared helpers in capi_internal.h).
#include "capi_internal.h"

extern "C" {

// Read-outs while rows are settled: evaluate their component's ln rho for the parameters of the last E-step
// (records.h, settled_mask_kernel).  Uses the list buffers: the next E-step rebuilds its first round from khat.
static int refresh_settled(gmmvb_workspace* ws, hipStream_t st) {
    if (!ws->params_used) return fail(GMMVB_ESTATE, "the parameters changed after the E-step whose settled rows are read");
    const int64_t n_rows = ws->e_rows;
    const int sel_grid = (int)((n_rows + kSelRows - 1) / kSelRows);
    if (!ws->rmask || !ws->rblk) return fail(GMMVB_ESTATE, "the list buffers of the pruned E-step are not allocated");
    const int is64 = ws->x_dtype == GMMVB_F64;
    bool vec = false;
    int rc = check_x(ws, ws->bounds_x, ws->bounds_ldx, n_rows, &vec);
    if (rc) return rc;
    EstepArgs a{ws->bounds_x, ws->bounds_ldx, n_rows, ws->D, ws->img, ws->cvec, ws->K, ws->lnrho, ws->npad};
    if (ws->sorted) {
        a.x = ws->xp;
        a.ldx = ws->D;
        vec = ws->D % 16 == 0;
    }
    hipLaunchKernelGGL(settled_mask_kernel, dim3(sel_grid), dim3(kSelRows), 0, st, ws->lock, ws->masks, ws->lcomp, ws->npad, n_rows,
                       ws->K, ws->rmask, ws->rblk);
    launch_scan_counts(st, ws->rblk, sel_grid, ws->K, ws->counts, ws->scan_parts);
    hipLaunchKernelGGL(fill_lists_kernel, dim3(sel_grid), dim3(kSelRows), 0, st, ws->rmask, ws->npad, n_rows, ws->K, ws->rblk,
                       ws->lists, ws->npad);
    hipLaunchKernelGGL(gather_plan_kernel, dim3(1), dim3(64), 0, st, ws->counts, ws->K, estep_gather_rows_per_wg(ws->T, is64),
                       ws->plan);
    hipError_t e = launch_estep_gather_dev(ws->T, is64, vec, 2 * ws->num_cu, st, a, ws->lists, ws->npad, ws->counts, ws->plan);
    if (e != hipSuccess) return fail(GMMVB_EHIP, "settled-row evaluation", e);
    hipLaunchKernelGGL(settled_lse_kernel, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, st, ws->lock, ws->masks, ws->lcomp,
                       ws->lnrho, ws->npad, n_rows, ws->K, ws->lse);
    e = hipGetLastError();
    if (e != hipSuccess) return fail(GMMVB_EHIP, "settled-row read-out", e);
    ws->active_lists = false;
    if (ws->mlists_done) ws->mlists_lost = true;
    ws->mlists_done = false;
    ws->settled_fresh = true;
    return GMMVB_OK;
}

static int readout(gmmvb_workspace* ws, int64_t row0, int64_t n_rows, double* out, void* stream, int mode) {
    if (!ws || !out) return fail(GMMVB_EINVAL, "null argument");
    if (ws->e_state == 0 || ws->e_state == 4) return fail(GMMVB_ESTATE, "no E-step output in the workspace");
    if (row0 < 0 || n_rows < 1 || row0 + n_rows > ws->e_rows) return fail(GMMVB_EINVAL, "row range outside the last E-step");
    if (mode == 0 && ws->e_state == 2) return fail(GMMVB_ESTATE, "ln rho is undefined after gmmvb_load_responsibilities");
    if (mode == 0 && ws->e_state == 3 && ws->hmm_no_lnrho)
        return fail(GMMVB_ESTATE, "the last gmmvb_estep formed no ln rho array (hmmvb_emission_target 1)");
    const int64_t total = n_rows * ws->K;
    const bool hmm_gamma = ws->e_state == 3 && mode == 1;      // responsibilities of an HMM pass = gamma
    if (ws->e_state == 1 && ws->rec_live) {                    // the pass lived on records: only listed pairs are exact
        if (ws->lock_live && ws->skip_used && !ws->settled_fresh) {
            const int rc = refresh_settled(ws, (hipStream_t)stream);
            if (rc) return rc;
        }
        const RecArrays rec{ws->rec_k, ws->rec_d, ws->rec_B, ws->rec_exact, ws->rec_sel, ws->rec_flags, ws->npad};
        hipLaunchKernelGGL(rec_readout_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rec,
                           ws->masks, ws->lnrho, ws->lse, ws->cvec, ws->npad, row0, n_rows, ws->K, mode, out,
                           ws->sorted ? ws->iperm : nullptr, ws->lock_live ? ws->lock : nullptr, ws->lcomp);
        hipError_t er = hipGetLastError();
        if (er != hipSuccess) return fail(GMMVB_EHIP, "rec_readout launch", er);
        return GMMVB_OK;
    }
    if (ws->lse_stale && mode == 1 && ws->e_state == 1) {       // (an HMM workspace skips the mixture's log-normaliser)
        hipLaunchKernelGGL(row_lse_kernel, dim3((unsigned)((ws->e_rows + kLseRows - 1) / kLseRows)), dim3(256), 0,
                           (hipStream_t)stream, ws->lnrho, ws->npad, ws->e_rows, ws->K, ws->lse, nullptr, nullptr, 1);
        ws->lse_stale = false;
    }
    if (hmm_gamma && hmm_ensure_gamma_cm(ws->hmm, (hipStream_t)stream) != hipSuccess) return fail(GMMVB_EHIP, "gamma transpose launch");
    hipLaunchKernelGGL(readout_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       hmm_gamma ? hmm_gamma_cm(ws->hmm) : ws->lnrho, ws->lse, ws->npad, row0, n_rows, ws->K, mode,
                       (ws->e_state == 2 || hmm_gamma) ? 1 : 0, out, ws->sorted ? ws->iperm : nullptr);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(GMMVB_EHIP, "readout launch", e);
    return GMMVB_OK;
}

int gmmvb_responsibilities(gmmvb_workspace* ws, int64_t row0, int64_t n_rows, double* r_dev, void* stream) {
    return readout(ws, row0, n_rows, r_dev, stream, 1);
}

int gmmvb_ln_rho(gmmvb_workspace* ws, int64_t row0, int64_t n_rows, double* out_dev, void* stream) {
    return readout(ws, row0, n_rows, out_dev, stream, 0);
}

int gmmvb_argmax(gmmvb_workspace* ws, int64_t row0, int64_t n_rows, int32_t* z_dev, void* stream) {
    if (!ws || !z_dev) return fail(GMMVB_EINVAL, "null argument");
    if (ws->e_state == 0 || ws->e_state == 4) return fail(GMMVB_ESTATE, "no E-step output in the workspace");
    if (row0 < 0 || n_rows < 1 || row0 + n_rows > ws->e_rows) return fail(GMMVB_EINVAL, "row range outside the last E-step");
    hipError_t e;
    const int* iperm = ws->sorted ? ws->iperm : nullptr;
    if (ws->e_state == 1 && ws->rec_live) {        // rec_finish_kernel left every row's first maximiser in khat
        hipLaunchKernelGGL(gather_int_kernel, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                           ws->khat, row0, n_rows, iperm, z_dev);
        e = hipGetLastError();
        if (e != hipSuccess) return fail(GMMVB_EHIP, "argmax read-out", e);
        return GMMVB_OK;
    }
    if (ws->e_state == 3 && hmm_ensure_gamma_cm(ws->hmm, (hipStream_t)stream) != hipSuccess) return fail(GMMVB_EHIP, "gamma transpose launch");
    hipLaunchKernelGGL(argmax_kernel, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       ws->e_state == 3 ? hmm_gamma_cm(ws->hmm) : ws->lnrho, ws->npad, row0, n_rows, ws->K, z_dev, iperm);
    e = hipGetLastError();
    if (e != hipSuccess) return fail(GMMVB_EHIP, "argmax launch", e);
    return GMMVB_OK;
}

}  // extern "C"
