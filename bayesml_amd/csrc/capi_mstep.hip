// C ABI of the GMM-VB data-pass engine, the M-step: gmmvb_load_responsibilities, gmmvb_mstep, gmmvb_estep_mstep (include/gmmvb.h; shared helpers in capi_internal.h).
#include "capi_internal.h"

extern "C" {

int gmmvb_load_responsibilities(gmmvb_workspace* ws, const double* r_dev, int64_t n_rows, void* stream) {
    if (!ws || !r_dev) return fail(GMMVB_EINVAL, "null argument");
    if (n_rows < 1 || n_rows > ws->max_rows) return fail(GMMVB_EINVAL, "n_rows must be in [1, max_rows]");
    claim_scratch(ws);
    const int tb = 256;
    hipLaunchKernelGGL(load_r_kernel, dim3((unsigned)((n_rows + tb - 1) / tb)), dim3(tb), 0, (hipStream_t)stream,
                       r_dev, n_rows, ws->K, ws->lnrho, ws->npad, ws->lse, ws->sorted ? ws->iperm : nullptr);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(GMMVB_EHIP, "load_r launch", e);
    ws->e_state = 2;
    ws->lost_estep = false;
    ws->e_rows = n_rows;
    ws->n_spans = 0;
    ws->bounds_rows = 0;               // the array holds responsibilities now, nothing a later E-step may carry over
    if (ws->lock_live) {               // the settled rows belonged to the previous state of affairs
        ws->lock_live = false;
        ws->lock_reset = true;
    }

    ws->rec_valid = false;
    ws->dense_valid = false;
    ws->rec_live = false;
    ws->act_rows = 0;
    return GMMVB_OK;
}

int gmmvb_mstep(gmmvb_workspace* ws, const void* x_dev, int64_t ldx, int64_t n_rows, double* stats_dev,
                void* stream) {
    bool vec = false;
    int rc = check_x(ws, x_dev, ldx, n_rows, &vec);
    if (rc) return rc;
    if (!stats_dev) return fail(GMMVB_EINVAL, "stats_dev is null");
    if (ws->e_state == 0 || ws->e_state == 4 || ws->e_rows != n_rows)
        return fail(GMMVB_ESTATE, "no responsibilities for these rows: call gmmvb_estep or gmmvb_load_responsibilities first");
    hipStream_t st = (hipStream_t)stream;
    if (ws->lse_stale && ws->e_state == 1) {       // a mixture M-step on an HMM workspace: the log-normaliser after all
        hipLaunchKernelGGL(row_lse_kernel, dim3((unsigned)((n_rows + kLseRows - 1) / kLseRows)), dim3(256), 0, st, ws->lnrho,
                           ws->npad, n_rows, ws->K, ws->lse, nullptr, nullptr, 1);
        ws->lse_stale = false;
    }
    if (ws->generic) {
        const int direct = ws->e_state == 2 ? 1 : (ws->e_state == 3 ? 2 : 0);
        if (ws->e_state == 3 && hmm_ensure_gamma_cm(ws->hmm, st) != hipSuccess) return fail(GMMVB_EHIP, "gamma transpose launch");
        const double* lr = ws->e_state == 3 ? hmm_gamma_cm(ws->hmm) : ws->lnrho;
        const double* aux = ws->e_state == 3 && !ws->hmm_skip_h ? ws->lnrho : nullptr;
        int S = ws->gen_S;
        const int64_t rps = round_up((n_rows + S - 1) / S, 64);
        S = (int)((n_rows + rps - 1) / rps);
        const int tiles = tri_pairs(ws->T);
        if (phase_events(ws)) note_hip(ws, hipEventRecord(ws->ev[2], st));
        span_begin(ws, kSpanMstepMain, st);
        if (ws->x_dtype == GMMVB_F64) {
            hipLaunchKernelGGL(mstep_generic_first_kernel<double>, dim3(ws->K, S), dim3(256), 0, st, (const double*)x_dev, ldx, n_rows,
                               ws->D, ws->pivot, lr, ws->lse, aux, ws->npad, ws->K, rps, direct, ws->gen_first);
            hipLaunchKernelGGL(mstep_generic_second_kernel<double>, dim3(tiles, ws->K, S), dim3(256), 0, st, (const double*)x_dev, ldx,
                               n_rows, ws->D, ws->pivot, lr, ws->lse, aux, ws->npad, ws->K, rps, direct, ws->T, ws->gen_second);
        } else {
            hipLaunchKernelGGL(mstep_generic_first_kernel<float>, dim3(ws->K, S), dim3(256), 0, st, (const float*)x_dev, ldx, n_rows,
                               ws->D, ws->pivot, lr, ws->lse, aux, ws->npad, ws->K, rps, direct, ws->gen_first);
            hipLaunchKernelGGL(mstep_generic_second_kernel<float>, dim3(tiles, ws->K, S), dim3(256), 0, st, (const float*)x_dev, ldx,
                               n_rows, ws->D, ws->pivot, lr, ws->lse, aux, ws->npad, ws->K, rps, direct, ws->T, ws->gen_second);
        }
        span_end(ws, st);
        if (phase_events(ws)) {
            note_hip(ws, hipEventRecord(ws->ev[3], st));
            ws->ev_m = true;
        }
        span_begin(ws, kSpanReduce, st);
        const int64_t elems = ws->D + 2 + (int64_t)tiles * 256;
        hipLaunchKernelGGL(reduce_generic_kernel, dim3((unsigned)((elems + 255) / 256), ws->K), dim3(256), 0, st, ws->gen_first,
                           ws->gen_second, S, ws->K, ws->D, ws->T, stats_dev);
        span_end(ws, st);
        hipError_t eg = hipGetLastError();
        if (eg != hipSuccess) return fail(GMMVB_EHIP, "mstep_generic launch", eg);
        ++ws->passes[5];
        const size_t used = std::strlen(ws->info);
        std::snprintf(ws->info + used, sizeof(ws->info) - used, " | mstep_generic_f64<D=%d> tiles=%d splits=%d", ws->D, tiles, S);
        return take_hip(ws, "event record inside the M-step");
    }
    // row splits: ~4 workgroups per CU in total, whole 64-row groups per split, S a multiple of 8 where possible
    int64_t S = ws->S_cap;
    const int64_t groups = (n_rows + 63) / 64;
    if (S > groups) S = groups;
    int64_t rows_per_split = round_up((n_rows + S - 1) / S, 64);
    if (ws->split_rows && rows_per_split > ws->split_rows) rows_per_split = ws->split_rows;
    S = (n_rows + rows_per_split - 1) / rows_per_split;
    bool pre = ws->xc && ws->xc_src == x_dev && ws->xc_rows == n_rows && ws->xc_ldx == ldx;
    if (ws->wide && !pre && ws->xc) {
        // past 8 feature tiles the M-step only exists over the centred copy: made here if the caller has not
        // (multivariate_normal.LearnModel's one-pass moments call gmmvb_mstep straight after gmmvb_load_responsibilities)
        rc = gmmvb_prepare_rows(ws, x_dev, ldx, n_rows, stream);
        if (rc) return rc;
        pre = true;
    }
    const int kpw = mstep_components_per_wg(ws->T, pre);
    const int KG = (ws->K + kpw - 1) / kpw;
    int64_t grid = 8 * ((S + 7) / 8) * KG;
    MstepArgs a{x_dev, ldx, n_rows, ws->D, ws->pivot, ws->lnrho, ws->lse, nullptr, ws->npad, ws->K, KG, (int)S,
                rows_per_split, ws->e_state == 2 ? 1 : 0, ws->slabs};
    const bool hmm_small = ws->e_state == 3 && ws->T == 1 && pre;      // reads gamma time-major (hmm_mstep_small_kernel)
    if (ws->e_state == 3) {          // HMM: responsibilities = gamma from the forward-backward pass, h = sum gamma ln rho
        if (!hmm_small && hmm_ensure_gamma_cm(ws->hmm, st) != hipSuccess) return fail(GMMVB_EHIP, "gamma transpose launch");
        a.lnrho = hmm_gamma_cm(ws->hmm);
        a.aux = (ws->hmm_no_lnrho || ws->hmm_skip_h) ? nullptr : ws->lnrho;     // (nullptr: h stays 0, see hmmvb_skip_h / hmmvb_emission_target)
        a.direct_r = 2;
    }
    if (pre) {
        a.x = ws->xc;
        a.ldx = 16 * ws->T;
        a.D = 16 * ws->T;
    } else if (ws->sorted) {
        return fail(GMMVB_ESTATE, "the workspace's rows are regrouped for another sample matrix: call gmmvb_prepare_rows first");
    }
    const char* name = "";
    hipError_t e;
    bool sparse = ws->sparse && ws->masks && pre && ws->e_state == 1 && ws->act_rows == n_rows;
    if (sparse) {      // the lists pay off while act kListMns < K kDenseMns (ws->pt.list_m_below())
        const double pairs = (double)n_rows * ws->K;
        if (ws->rec_live) {
            // a pruned E-step leaves exact values for the listed pairs only (the others are bounded in the f32 array, not in
            // ln rho): its M-step always runs over the lists, however many pairs are active
        } else {
            // after a dense E-step the host has been waiting for that kernel anyway: read this pass's own count
            rc = fetch_counters(ws);
            if (rc) return rc;
            sparse = ws->lag.valid && ws->lag.act <= ws->pt.list_m_below() * pairs;
        }
    }
    if (sparse && ws->K > 256) sparse = false;
    if (ws->rec_live && ws->e_state == 1 && !sparse)
        return fail(GMMVB_ESTATE, "a pruned E-step needs the list M-step over the matrix of the E-step (gmmvb_prepare_rows)");
    if (ws->lock_live && !sparse)
        return fail(GMMVB_ESTATE, "settled rows need the list M-step over the matrix of the E-step (gmmvb_prepare_rows)");
    if (sparse) {      // E-step output: only the samples that can change the f64 sums, through per-component lists
        rc = ensure_lists(ws);
        if (rc) return rc;
        const int nblk = (int)((n_rows + kSelRows - 1) / kSelRows);
        if (phase_events(ws)) note_hip(ws, hipEventRecord(ws->ev[2], st));      // the list building is part of the M-step's time
        const int cap_chunks0 = (int)std::min<int64_t>((int64_t)ws->S_cap * ws->K, 1 << 30);
#ifndef GMMVB_MLIST_RMIN
#define GMMVB_MLIST_RMIN 1024
#endif
        const int r_min0 = GMMVB_MLIST_RMIN;      // list entries per chunk (2048: +4 %, 4096: +19 % on the list M-step, round 2)
        MstepListArgs la0{ws->xc, ws->lnrho, ws->lse, ws->lists, ws->npad, ws->counts, ws->plan_m, cap_chunks0, r_min0,
                          ws->npad, ws->K, ws->slabs};
        // f32 rows with whole 16-feature tiles: read them instead of the twice as wide centred copy
        static const bool list_xc = dev_env("GMMVB_MLIST_XC") && dev_env("GMMVB_MLIST_XC")[0] == '1';      // developer switch: the centred f64 copy
        if (!list_xc && ws->x_dtype == GMMVB_F32 && ws->D == 16 * ws->T && (ws->T == 2 || ws->T == 4 || ws->T == 8)) {
            if (ws->sorted) {
                la0.x32 = (const float*)ws->xp;
                la0.ldx = ws->D;
            } else if (vec) {
                la0.x32 = (const float*)x_dev;
                la0.ldx = ldx;
            }
            la0.n_rows = n_rows;
            la0.D = ws->D;
            la0.pivot = ws->pivot;
        }
        if (!la0.x32 && ws->xc_stale) {        // this list kernel reads the centred copy: bring it to the internal row order
            e = recenter_rows(ws, n_rows, st);
            if (e != hipSuccess) return fail(GMMVB_EHIP, "center_rows launch", e);
        }
        int64_t lgrid = (cap_chunks0 + kpw - 1) / kpw;
        {
            const int64_t most = (n_rows * (int64_t)ws->K + r_min0 - 1) / r_min0 + ws->K;      // no more chunks than this can exist
            if ((most + kpw - 1) / kpw < lgrid) lgrid = (most + kpw - 1) / kpw;
        }
        const int elems0 = tri_pairs(ws->T) * 256 + 16 * ws->T + 2;
        if (ws->lock_live && ws->delta_pending) {
            // the rows that settled or came loose in this pass (rec_finish_kernel's delta masks) enter / leave the
            // cache of settled rows - before the pass's own lists are built in the same buffers
            span_begin(ws, kSpanLists, st);
            launch_scan_counts(st, ws->dblk, nblk, ws->K, ws->counts, ws->scan_parts);
            // (qpart: the delta and M-step pairs rec_finish_kernel counted per block - none of either: nothing to fill)
            hipLaunchKernelGGL(fill_lists_kernel, dim3(nblk), dim3(kSelRows), 0, st, ws->dmask, ws->npad, n_rows, ws->K,
                               ws->dblk, ws->lists, ws->npad, ws->lock, ws->lcomp, ws->e_state == 1 ? ws->qpart : nullptr);
            span_end(ws, st);
            MstepListArgs ld = la0;
            ld.direct_r = 3;
            // at most one entry per row: far fewer chunks than the lists of a pass can have
            int64_t dgrid = ((n_rows + r_min0 - 1) / r_min0 + ws->K + kpw - 1) / kpw;
            if (dgrid > lgrid) dgrid = lgrid;
            const char* dname = "";
            span_begin(ws, kSpanMstepMain, st);
            e = launch_mstep_list(ws->T, (int)dgrid, st, ld, &dname);
            span_end(ws, st);
            if (e != hipSuccess) return fail(GMMVB_EHIP, "settled-row delta launch", e);
            span_begin(ws, kSpanReduce, st);
            hipLaunchKernelGGL(reduce_chunks_kernel, dim3((elems0 + 255) / 256, ws->K), dim3(256), 0, st, ws->slabs, ws->plan_m,
                               ws->K, ws->D, ws->T, ws->cache, 1, nullptr);
            span_end(ws, st);
            ws->delta_pending = false;
            ws->active_lists = false;
            ws->mlists_done = false;
        }
        // masks and block counts of the active pairs were written by lse_mask_kernel / rec_finish_kernel at the end of the E-step
        if (ws->lock_live) {
            // the rows in the cache are left out: the M-step has its own masks (the E-step's next first round builds its
            // lists from the full ones)
            if (!ws->mlists_done) {
                if (ws->mlists_lost)
                    return fail(GMMVB_ESTATE, "the M-step's lists were used by a read-out of settled rows: call gmmvb_estep again");
                span_begin(ws, kSpanLists, st);
                launch_scan_counts(st, ws->mblk, nblk, ws->K, ws->counts, ws->scan_parts);
                hipLaunchKernelGGL(fill_lists_kernel, dim3(nblk), dim3(kSelRows), 0, st, ws->mmask, ws->npad, n_rows, ws->K,
                                   ws->mblk, ws->lists, ws->npad, nullptr, nullptr, ws->e_state == 1 ? ws->qpart : nullptr);
                e = hipGetLastError();
                span_end(ws, st);
                if (e != hipSuccess) return fail(GMMVB_EHIP, "active-sample lists", e);
                ws->mlists_done = true;
                ws->active_lists = false;
            }
        } else if (!ws->active_lists) {
            span_begin(ws, kSpanLists, st);
            launch_scan_counts(st, ws->blk, nblk, ws->K, ws->counts, ws->scan_parts);
            hipLaunchKernelGGL(fill_lists_kernel, dim3(nblk), dim3(kSelRows), 0, st, ws->masks, ws->npad, n_rows, ws->K,
                               ws->blk, ws->lists, ws->npad);
            e = hipGetLastError();
            span_end(ws, st);
            if (e != hipSuccess) return fail(GMMVB_EHIP, "active-sample lists", e);
            ws->active_lists = true;
            ws->blk_fresh = false;
        }
        // chunks of list entries (mstep.h): as many slabs as the workspace holds, at least 1024 entries per chunk
        grid = lgrid;
        S = 0;
        rows_per_split = r_min0;
        const MstepListArgs& la = la0;
        span_begin(ws, kSpanMstepMain, st);
        e = launch_mstep_list(ws->T, (int)grid, st, la, &name);
        span_end(ws, st);
        ++ws->passes[6];
    } else {
        ++ws->passes[5];
        if (phase_events(ws)) note_hip(ws, hipEventRecord(ws->ev[2], st));
        if (pre && ws->xc_stale) {             // the dense kernel reads the centred copy: bring it to the internal row order
            e = recenter_rows(ws, n_rows, st);
            if (e != hipSuccess) return fail(GMMVB_EHIP, "center_rows launch", e);
        }
        span_begin(ws, kSpanMstepMain, st);
        if (ws->T == 1 && pre) {
            // one feature tile: a wave walks the rows once for eight components (mstep.h, mstep_small_f64)
            constexpr int kSmallCw = 8;
            const int per_wg = 4 * kSmallCw;
            const int KGW = (ws->K + per_wg - 1) / per_wg;
            grid = 8 * ((S + 7) / 8) * KGW;
            if (hmm_small)
                e = launch_hmm_mstep_small((int)grid, st, a, KGW, hmm_gamma_tm(ws->hmm), hmm_padded_states(ws->hmm),
                                           !ws->opt_hmm_mstep_dense, &name);
            else
                e = launch_mstep_small((int)grid, st, a, KGW, kSmallCw, &name);
        } else {
            const bool cal = a.direct_r == 0 && cal_wanted(ws, 1, (double)n_rows * ws->K);
            if (cal) note_hip(ws, hipEventRecord(ws->cal_ev[2], st));
            e = launch_mstep(ws->T, ws->x_dtype == GMMVB_F64, vec, pre, (int)grid, st, a, &name);
            if (cal) cal_mark(ws, 1, (double)n_rows * ws->K, st);
        }
        span_end(ws, st);
    }
    if (e != hipSuccess) return fail(GMMVB_EHIP, "mstep launch", e);
    if (phase_events(ws)) {
        note_hip(ws, hipEventRecord(ws->ev[3], st));
        ws->ev_m = true;
    }
    const int elems = tri_pairs(ws->T) * 256 + 16 * ws->T + 2;
    span_begin(ws, kSpanReduce, st);
    if (sparse)
        hipLaunchKernelGGL(reduce_chunks_kernel, dim3((elems + 255) / 256, ws->K), dim3(256), 0, st, ws->slabs, ws->plan_m,
                           ws->K, ws->D, ws->T, stats_dev, 0, ws->lock_live ? ws->cache : nullptr);
    else
        hipLaunchKernelGGL(reduce_stats_kernel, dim3((elems + 255) / 256, ws->K), dim3(256), 0, st, ws->slabs, (int)S,
                           ws->K, ws->D, ws->T, stats_dev);
    span_end(ws, st);
    e = hipGetLastError();
    if (e != hipSuccess) return fail(GMMVB_EHIP, "reduce_stats launch", e);
    const size_t used = std::strlen(ws->info);
    std::snprintf(ws->info + used, sizeof(ws->info) - used, " | %s grid=%lldx%d splits=%lld rows/split=%lld", name,
                  (long long)grid, mstep_threads(ws->T, pre), (long long)S, (long long)rows_per_split);
    return take_hip(ws, "event record inside the M-step");
}

int gmmvb_estep_mstep(gmmvb_workspace* ws, const void* x_dev, int64_t ldx, int64_t n_rows, double* stats_dev,
                      void* stream) {
    int rc = gmmvb_estep(ws, x_dev, ldx, n_rows, stream);
    if (rc) return rc;
    return gmmvb_mstep(ws, x_dev, ldx, n_rows, stats_dev, stream);
}

}  // extern "C"
