// C ABI of the GMM-VB data-pass engine, the E-step: gmmvb_estep and the passes it is made of (include/gmmvb.h; shared helpers in capi_internal.h).
#include "capi_internal.h"

extern "C" {

// Regroup the internal row order by the best component of the last E-step (aux_kernels.h): new permutation, permuted copy
// of x, centred copy rebuilt from it.  Everything row-indexed in the workspace is stale afterwards: the caller (a bound
// pass) rebuilds it.
static hipError_t regroup_rows(gmmvb_workspace* ws, const void* x_dev, int64_t ldx, int64_t n_rows, hipStream_t st,
                               bool keep_state, bool margin_ok) {
    const int sel_grid = (int)((n_rows + kSelRows - 1) / kSelRows);
    const unsigned cgrid = (unsigned)((n_rows + 255) / 256);
    const int* key = ws->khat;                  // best components in the order the second pass sorts
    const int* perm_in = ws->sorted ? ws->perm : nullptr;
    // Two stable counting sorts, least significant key first: how firmly the rows sit in their component
    // (margin_bucket_kernel; needs the last pass's log-normalisers), then the component.
    const bool by_margin = ws->opt_regroup_margin && margin_ok && ws->K >= kMarginBuckets;
    if (by_margin) {
        int* bucket = ws->perm_tmp;             // (free until the first composition below writes it)
        hipLaunchKernelGGL(margin_bucket_kernel, dim3(cgrid), dim3(256), 0, st, ws->lnrho, ws->npad, n_rows, ws->lse, ws->khat,
                           keep_state ? ws->lock : nullptr, bucket);
        hipLaunchKernelGGL(select_mask_kernel<3>, dim3(sel_grid), dim3(kSelRows), 0, st, ws->lnrho, ws->npad, n_rows,
                           kMarginBuckets, bucket, ws->masks, ws->blk);
        launch_scan_counts(st, ws->blk, sel_grid, kMarginBuckets, ws->counts, ws->scan_parts);
        hipLaunchKernelGGL(fill_lists_kernel, dim3(sel_grid), dim3(kSelRows), 0, st, ws->masks, ws->npad, n_rows, kMarginBuckets,
                           ws->blk, ws->lists, ws->npad);
        // the best components and the caller's rows in the intermediate order (the keys in the records' slot array, which
        // the bound pass rewrites anyway)
        int* key1 = reinterpret_cast<int*>(ws->rec_d);
        hipLaunchKernelGGL(perm_compose_kernel, dim3(cgrid, kMarginBuckets), dim3(256), 0, st, ws->lists, ws->npad, ws->counts,
                           ws->khat, key1);
        hipLaunchKernelGGL(perm_compose_kernel, dim3(cgrid, kMarginBuckets), dim3(256), 0, st, ws->lists, ws->npad, ws->counts,
                           perm_in, ws->perm_tmp);
        key = key1;
        perm_in = ws->perm_tmp;
    }
    hipLaunchKernelGGL(select_mask_kernel<3>, dim3(sel_grid), dim3(kSelRows), 0, st, ws->lnrho, ws->npad, n_rows, ws->K,
                       const_cast<int*>(key), ws->masks, ws->blk);
    launch_scan_counts(st, ws->blk, sel_grid, ws->K, ws->counts, ws->scan_parts);
    hipLaunchKernelGGL(fill_lists_kernel, dim3(sel_grid), dim3(kSelRows), 0, st, ws->masks, ws->npad, n_rows, ws->K, ws->blk,
                       ws->lists, ws->npad);
    ws->tile_ref_valid = false;
    if (ws->tile_ref) {                // the groups' lengths are in counts now: every tile's reference component (project.h)
        if (launch_proj_tile_ref(ws->counts, ws->K, sel_grid, ws->tile_ref, st) == hipSuccess) ws->tile_ref_valid = true;
    }
    if (by_margin) {
        // (three row-sized index buffers in rotation: the new order goes where the best components were - the bound pass
        // that follows rewrites them)
        hipLaunchKernelGGL(perm_compose_kernel, dim3(cgrid, ws->K), dim3(256), 0, st, ws->lists, ws->npad, ws->counts, perm_in,
                           ws->khat);
        int* new_perm = ws->khat;
        ws->khat = ws->perm_tmp;
        ws->perm_tmp = ws->perm;
        ws->perm = new_perm;
    } else {
        hipLaunchKernelGGL(perm_compose_kernel, dim3(cgrid, ws->K), dim3(256), 0, st, ws->lists, ws->npad, ws->counts, perm_in,
                           ws->perm_tmp);
        std::swap(ws->perm, ws->perm_tmp);
    }
    if (keep_state) {
        // the cache of single-component rows is a sum over rows - it does not care about their order; what is kept per row
        // (in the cache or not, for which component, the settled rows' distance bound) moves with the rows.  The records'
        // byte arrays and the threshold array are free at this point of a bound pass (rec_build_kernel rewrites them).
        hipLaunchKernelGGL(regroup_state_kernel, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, st, ws->perm,
                           ws->sorted ? ws->iperm : nullptr, n_rows, ws->lock, ws->lcomp, ws->dlock, ws->rec_sel, ws->rec_flags,
                           ws->rthr);
        std::swap(ws->lock, ws->rec_sel);
        std::swap(ws->lcomp, ws->rec_flags);
        std::swap(ws->dlock, ws->rthr);
    }
    hipLaunchKernelGGL(perm_invert_kernel, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, st, ws->perm, n_rows, ws->iperm);
    const int64_t total = n_rows * ws->D;
    const unsigned pg = (unsigned)((total + 255) / 256);
    const int64_t esz = ws->x_dtype == GMMVB_F64 ? 8 : 4;
    if ((ws->D * esz) % 16 == 0 && (ldx * esz) % 16 == 0 && (uintptr_t)x_dev % 16 == 0) {
        const int p16 = (int)(ws->D * esz / 16);
        hipLaunchKernelGGL(permute_rows16_kernel, dim3((unsigned)((n_rows * p16 + 255) / 256)), dim3(256), 0, st,
                           (const uint4*)x_dev, ldx * esz / 16, n_rows, p16, ws->perm, (uint4*)ws->xp);
    } else if (ws->x_dtype == GMMVB_F64)
        hipLaunchKernelGGL(permute_rows_kernel<double>, dim3(pg), dim3(256), 0, st, (const double*)x_dev, ldx, n_rows, ws->D,
                           ws->perm, (double*)ws->xp);
    else
        hipLaunchKernelGGL(permute_rows_kernel<float>, dim3(pg), dim3(256), 0, st, (const float*)x_dev, ldx, n_rows, ws->D,
                           ws->perm, (float*)ws->xp);
    // the centred f64 copy follows the internal order too, but it is only read by the dense M-step (and by the list
    // M-step of f64 / ragged-D inputs): rebuilt there when needed (recenter_rows), not here - 4 ms and 10 GB at C3
    ws->xc_stale = ws->xc != nullptr;
    if (ws->xq && ws->xq_src == x_dev) {       // the digit planes follow the internal order (3 ms at C3, once or twice per fit)
        hipError_t eq = launch_x_digits(ws->xp, ws->x_dtype == GMMVB_F64, ws->D, n_rows, ws->D, ws->pivot, ws->xq, ws->xqe, st);
        if (eq != hipSuccess) return eq;
        ws->xq_gen = ws->pivot_gen;
    }
    ws->sorted = true;
    ++ws->sorts;
    return hipGetLastError();
}

// bound pass of the pruned E-step: an upper bound of ln rho for every pair (three int8 digits) and the best of them, khat
static hipError_t launch_bound_pass(gmmvb_workspace* ws, const EstepI8Args& a8, int is64, bool vec, hipStream_t st,
                                    const char** name, int* rpw_out, int64_t* grid_out) {
    const int rpw = estep_i8_rows_per_wg();
    int64_t grid = (a8.n_rows + rpw - 1) / rpw;
    if (grid > (1 << 20)) grid = 1 << 20;
    *rpw_out = rpw;
    *grid_out = grid;
    EstepI8Args ab = a8;
    ab.img = ws->img_i8b;
    ab.khat = ws->khat;
    ab.ub = ws->ub32;           // the bounds go straight into the f32 array the sweeps carry
    return launch_estep_i8_bound(is64, vec, ws->bound_tb, (int)grid, st, ab, name);
}

// masks -> per-component lists -> chunk plan -> exact f64 evaluation of the listed pairs (all sized on the device)
static hipError_t lists_and_gather(gmmvb_workspace* ws, const EstepArgs& a, int is64, bool vec, int sel_grid, hipStream_t st,
                                   const float* thr = nullptr, const double* block_total = nullptr /*listed pairs per block, if counted*/) {
    if (thr) {
        hipError_t em = hipMemsetAsync(ws->exit_ctr, 0, sizeof(unsigned long long), st);
        if (em != hipSuccess) return em;
    }
    span_begin(ws, kSpanSelect, st);
    launch_scan_counts(st, ws->blk, sel_grid, ws->K, ws->counts, ws->scan_parts);
    hipLaunchKernelGGL(fill_lists_kernel, dim3(sel_grid), dim3(kSelRows), 0, st, ws->masks, ws->npad, a.n_rows, ws->K, ws->blk,
                       ws->lists, ws->npad, nullptr, nullptr, block_total);
    hipLaunchKernelGGL(gather_plan_kernel, dim3(1), dim3(64), 0, st, ws->counts, ws->K,
                       estep_gather_rows_per_wg(ws->T, is64), ws->plan);
    hipError_t e = hipGetLastError();
    span_end(ws, st);
    if (e != hipSuccess) return e;
    span_begin(ws, kSpanGather, st);
    e = launch_estep_gather_dev(ws->T, is64, vec, 2 * ws->num_cu, st, a, ws->lists, ws->npad, ws->counts, ws->plan, thr,
                                thr ? ws->exit_ctr : nullptr, 0.0f);
    span_end(ws, st);
    ++ws->passes[7];
    return e;
}

// ---- pass policy: unit costs, thresholds and their calibration live in policy.h (ws->pt) --------------------------------

// Calibration of the policy table (policy.h) from the workspace's own passes: events around its first dense E-step, dense
// M-step and full bound pass of at least 2^23 pairs, taken over - like the pass counters - once they have completed.
bool cal_wanted(gmmvb_workspace* ws, int what, double pairs) {
    return ws->opt_calibrate && ws->cal_ev[0] != nullptr && !((ws->pt.measured >> what) & 1) && ws->cal_pairs[what] == 0.0 &&
           pairs >= (double)(int64_t(1) << 23) && ws->hmm == nullptr;
}
void cal_mark(gmmvb_workspace* ws, int what, double pairs, hipStream_t st) {
    note_hip(ws, hipEventRecord(ws->cal_ev[2 * what + 1], st));
    ws->cal_pairs[what] = pairs;
}
void cal_poll(gmmvb_workspace* ws) {
    for (int what = 0; what < 3; ++what) {
        if (ws->cal_pairs[what] <= 0.0 || hipEventQuery(ws->cal_ev[2 * what + 1]) != hipSuccess) continue;
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, ws->cal_ev[2 * what], ws->cal_ev[2 * what + 1]) == hipSuccess) {
            double ns = (double)ms * 1e6 / ws->cal_pairs[what];
            if (what == 2) ns += 0.010 * tri_pairs(ws->T) / 36.0;        // record building / selection around the bound kernel
            const bool took = ws->pt.take(what, ns);
            if (ws->opt_debug)
                std::fprintf(stderr, "[gmmvb] policy table: %s %.4f ns per pair %s (prune below %.3f, dense again above %.3f, list M below %.3f)\n",
                             what == 0 ? "dense E" : (what == 1 ? "dense M" : "bound pass"), ns, took ? "taken" : "out of range: literal kept",
                             ws->pt.prune_below(), ws->pt.dense_again_above(), ws->pt.list_m_below());
            if (!took) ws->pt.measured |= 8 << what;                     // (remembered as discarded: bits 3-5)
            // a discarded measurement (the process's first launch of a kernel pays its code upload; a small pass has a tail)
            // gets two more chances on later passes of the same kind
            ws->cal_pairs[what] = (took || ++ws->cal_tries[what] >= 3) ? -1.0 : 0.0;
        } else {
            ws->cal_pairs[what] = -1.0;
        }
    }
}

// The proof round over the lists just filled from the selection blocks' bases `blk_base`: by row superblocks when the item
// table fits the M-step's slabs (free during an E-step; estep_i8.h), else component after component.
static hipError_t proof_round(gmmvb_workspace* ws, hipStream_t st, const int* blk_base, int sel_grid, int64_t n_rows, float* ub) {
    if (ws->opt_proof_blocked && ws->slabs &&
        estep_i8_proof_work_bytes(ws->K, n_rows) <= ws->scratch->slabs_len * (int64_t)sizeof(double))
        return launch_estep_i8_proof_blocked(ws->D, ws->num_cu, st, ws->xq, ws->xqe, ws->img_i8b, ws->cvec, ws->K, ws->lists,
                                             ws->npad, ws->counts, blk_base, sel_grid, ws->slabs, ub, ws->lnrho, ws->npad);
    hipLaunchKernelGGL(gather_plan_kernel, dim3(1), dim3(64), 0, st, ws->counts, ws->K, estep_i8_pairs_per_chunk(), ws->plan);
    return launch_estep_i8_proof(ws->D, ws->num_cu, st, ws->xq, ws->xqe, ws->img_i8b, ws->cvec, ws->K, ws->lists, ws->npad,
                                 ws->counts, ws->plan, ub, ws->lnrho, ws->npad);
}

int gmmvb_estep(gmmvb_workspace* ws, const void* x_dev, int64_t ldx, int64_t n_rows, void* stream) {
    bool vec = false;
    int rc = check_x(ws, x_dev, ldx, n_rows, &vec);
    if (rc) return rc;
    if (!ws->have_params) return fail(GMMVB_ESTATE, "gmmvb_set_params has not been called");
    claim_scratch(ws);
    hipStream_t st = (hipStream_t)stream;
    const int is64 = ws->x_dtype == GMMVB_F64;
    if (ws->generic) {
        const int R = generic_rows(ws->D);
        const dim3 grid((unsigned)((n_rows + R - 1) / R), (unsigned)ws->K);
        const size_t lds = (size_t)ws->D * R * sizeof(double);
        if (phase_events(ws)) note_hip(ws, hipEventRecord(ws->ev[0], st));
        ws->n_spans = 0;
        span_begin(ws, kSpanEstepMain, st);
        if (is64)
            hipLaunchKernelGGL(estep_generic_kernel<double>, grid, dim3(64), lds, st, (const double*)x_dev, ldx, n_rows, ws->D,
                               ws->gen_u, ws->gen_m, ws->cvec, R, ws->lnrho, ws->npad);
        else
            hipLaunchKernelGGL(estep_generic_kernel<float>, grid, dim3(64), lds, st, (const float*)x_dev, ldx, n_rows, ws->D,
                               ws->gen_u, ws->gen_m, ws->cvec, R, ws->lnrho, ws->npad);
        span_end(ws, st);
        if (phase_events(ws)) {
            note_hip(ws, hipEventRecord(ws->ev[1], st));
            ws->ev_e = true;
        }
        span_begin(ws, kSpanLse, st);
        hipLaunchKernelGGL(row_lse_kernel, dim3((unsigned)((n_rows + kLseRows - 1) / kLseRows)), dim3(256), 0, st, ws->lnrho,
                           ws->npad, n_rows, ws->K, ws->lse, nullptr, nullptr, 1);
        span_end(ws, st);
        hipError_t eg = hipGetLastError();
        if (eg != hipSuccess) return fail(GMMVB_EHIP, "estep_generic launch", eg);
        ++ws->passes[0];
        ws->ctr_pending = false;
        ws->lag.valid = false;
        ws->act_rows = 0;
        ws->exp_counted = false;
        ws->rec_live = ws->rec_valid = false;
        ws->e_state = 1;
        ws->lost_estep = false;
        ws->e_rows = n_rows;
        ws->params_used = true;
        ws->have_drift = false;
        ws->prev_pass = 0;
        ws->evaluated = (double)n_rows * ws->K;
        std::snprintf(ws->info, sizeof(ws->info), "estep_generic_f64<D=%d> grid=%ux%ux64 rows/workgroup=%d", ws->D, grid.x, grid.y, R);
        return take_hip(ws, "event record inside the E-step");
    }
    const bool i8 = ws->estep_variant == kEstepI8;
    EstepArgs a{x_dev, ldx, n_rows, ws->D, ws->img, ws->cvec, ws->K, ws->lnrho, ws->npad};
    EstepI8Args a8{x_dev, ldx, n_rows, ws->D, ws->img_i8, ws->pivot_i8, ws->cvec, ws->K, ws->lnrho, ws->npad};
    if (ws->sorted && ws->xc_src != x_dev) ws->sorted = false;      // another matrix: the caller's order
    const char* name = "";
    hipError_t e = hipSuccess;
    const double pairs = (double)n_rows * ws->K;

    // ---- which kind of pass?  Decided from what the host knows WITHOUT waiting for the device: the counters of the
    // last E-step whose copy has arrived (they lag by one pass when the caller never synchronises; results do not
    // depend on the choice, only the time does).
    // A shard of a row-sharded job (gmmvb_set_shard) decides from the counters summed over all ranks and from the
    // job's size - nothing below differs between ranks, so neither do the decisions.
    if (ws->sharded) {
        rc = take_policy(ws);
        if (rc) return rc;
    }
    poll_counters(ws);
    cal_poll(ws);
    enum { kDense = 0, kBound = 1, kSweep = 3 };      // (2 was the pass on per-row records, gone in round 3)
    int mode = kDense;
    const gmmvb_pass_counters& L = ws->sharded ? ws->pol : ws->lag;
    // (a pruned E-step leaves exact ln rho for the listed pairs only, so its M-step has to run over the lists - which read
    // the rows through the workspace's prepared copy: without gmmvb_prepare_rows for this matrix the pass stays dense)
    const bool can_prune = ws->prune != 0 && ws->estep_variant == kEstepLds8 && ws->hmm == nullptr && ws->rec_k != nullptr &&
                           ws->xc != nullptr && ws->xc_src == x_dev && ws->xc_rows == n_rows && ws->xc_ldx == ldx;
    const int64_t size_rows = ws->sharded ? ws->shard_rows / ws->shard_ranks : n_rows;
    const bool big = ws->prune == 2 || size_rows * (int64_t)ws->K >= (int64_t(1) << 23);
    const bool same_rows = ws->bounds_rows == n_rows && ws->bounds_x == x_dev && ws->bounds_ldx == ldx;
    // counters of the previous pass, over rows_l rows (this rank's, or the job's)
    const bool known = L.valid && (ws->sharded || (L.rows == (double)n_rows && !ws->ctr_pending));
    // (this rank's own numbers of the previous pass: what its kernels did)
    const bool own_known = ws->lag.valid && ws->lag.rows == (double)n_rows && !ws->ctr_pending;
    const double rows_l = known ? L.rows : (double)n_rows;
    const double pairs_l = rows_l * ws->K;
    // the previous pass's M-step left its per-component lists of active rows (and their masks) in the workspace
    // (or the masks and block counts they are built from)
    // (an E-step whose output went to another tile of the group still left its masks, block counts and best components)
    const bool after_estep = ws->e_state == 1 || ws->lost_estep;
    const bool prev_lists = (ws->active_lists || ws->blk_fresh) && after_estep && ws->act_rows == n_rows && same_rows;
    // The stateless sweep (project.h) needs no carried per-pair bounds: the table gmmvb_set_params made for these parameters,
    // the digit planes of this matrix about the pivot in force, regrouped rows and the previous pass's lists.
    const bool can_project = ws->proj_table && ws->gimg != nullptr && ws->sorted && ws->tile_ref_valid && prev_lists &&
                             ws->xq != nullptr && ws->xq_src == x_dev && ws->xq_rows == n_rows &&
                             ws->xq_ldx == ldx && ws->xq_gen == ws->pivot_gen && ws->lock != nullptr;
    if (can_prune && big) {
        // sparse enough?  (never for an HMM workspace: forward-backward consumes every emission ln rho)
        bool sparse_ok = ws->prune == 2;
        if (!sparse_ok && known && !ws->forget) sparse_ok = L.act <= ws->pt.prune_below() * pairs_l;
        // a bound pass that left most pairs candidates (below) is not tried again until a quarter fewer pairs are active than
        // when it failed: at cluster spread 0.75 (31-40 of 64 active for twenty passes) every other pass was such an attempt
        if (sparse_ok && ws->prune != 2 && known && ws->bound_fail_act > 0.0 && L.act > 0.75 * ws->bound_fail_act * pairs_l)
            sparse_ok = false;
        if (sparse_ok) {
            mode = kBound;
            const bool hinted = same_rows && ws->have_drift && !ws->opt_carry_off;
            // Carrying the previous pass over the parameter update (gmmvb_set_drift): a sweep of the f32 per-pair bound
            // array, every entry with its own component's drift (1.5 ms at C3), after the previous pass's active pairs have
            // been evaluated under the new parameters.  (Round 2 also had a pass on 55-byte per-row records with ONE rest
            // bound per row; it eroded at the pace of the fastest-moving component and the default policy never chose it.)
            // typical_gamma is the caller's pessimistic summary min_k (gamma_k - delta_k / 30) (0.3, 0.6, 0.7, 0.8 in
            // the first iterations at C3, 0.94 by the 13th, 0.97 by the 20th, 0.99 by the 26th): below 0.5 the bounds are
            // made afresh.
            const double tg = ws->typical_gamma;
            bool sweep = hinted && (ws->dense_valid || (can_project && ws->opt_project == 2)) && !(tg > 0.0 && tg < ws->pt.gamma_no_carry);
            if (sweep && known && L.mode != kDense) {
                // spare candidates (listed but inactive) of the last pruned pass: carry on only while evaluating them
                // (they grow from pass to pass) costs less than a fresh bound pass, and while few rows overflow
                // (a pair of the proof round costs about a third of an exact evaluation)
                // (a bound pass's own proof stage works through the candidates its coarse bounds leave - not a sign of erosion)
                const double spare = (std::max(0.0, L.eval - (L.act - L.settled)) + (L.mode == kSweep ? ws->pt.proof_per_exact * L.proof : 0.0)) / pairs_l;
                ws->spare_last = spare;
                const int tb = ws->bound_tb > 0 ? ws->bound_tb : 3;
                const double bound_cost = ws->pt.i8_block_pair * tri_pairs(tb) + ws->pt.i8_row_of_y * 32 * tb, gpp = ws->pt.f64_tile_pair * tri_pairs(ws->T);
                if (gpp * spare * ws->pt.spare_growth >= bound_cost) sweep = false;
                // rows whose record had to be rebuilt in full cost K evaluations each and multiply from pass to pass
                // (x4 - x8 observed): stop carrying well before they dominate
                if (L.over > ws->pt.overflow_rows * rows_l || L.eval > ws->pt.carried_eval_above * pairs_l) sweep = false;
            }
            if (sweep && known && L.mode == kDense && L.act > ws->pt.sweep_after_dense_below * pairs_l) sweep = false;
            // straight from a dense pass the parameters usually still jump (second or third iteration of a restart): the
            // sweep's per-pair bounds are exact values then, but carried over such an update most of them end up
            // candidates (measured at C4: 118 of 256 per row, 171 ms) - a bound pass is the safe first pruned pass
            if (sweep && known && L.mode == kDense && tg > 0.0 && tg < ws->pt.gamma_no_carry_after_dense) sweep = false;
            if (sweep) mode = kSweep;
            // a bound pass that left most pairs candidates (the parameters jumped): back to the dense kernel
            if (mode == kBound && ws->prune != 2 && known && L.mode == kBound && L.eval > ws->pt.dense_again_above() * pairs_l) {
                mode = kDense;
                ++ws->passes[3];
                ws->bound_fail_act = L.act / pairs_l;
            }
        }
    }
    if (ws->forget) ws->bound_fail_act = -1.0;        // (a new restart: nothing is known about its bounds)
    ws->forget = false;
    // The cache of single-component rows (and the settled rows among them) survives every pruned pass over the same rows
    // whose M-step applied the delta lists - all of them end in rec_finish_kernel - including the one that regroups the
    // rows (regroup_rows moves the per-row state along).  A dense pass, new data or parameters unrelated to the last pass
    // drop it; rows that were settled then have no active pair on record, which only a pass that rebuilds everything
    // (bound or dense) can digest.
    // The rows are regrouped by dominant component at a bound pass (which rebuilds everything row-indexed anyway).  With
    // the proof round bound passes have become rare: the first time the responsibilities are sparse enough for the grouping
    // to pay (at most 2.5 active components per row) a carried pass therefore gives way to a bound pass, once - list-driven
    // kernels over ungrouped rows are 15-40 % slower for the rest of the fit.
    if (mode == kSweep && ws->sort_rows && ws->xp && !ws->sorted && ws->sorts == 0 && same_rows &&
        after_estep && known && L.act <= ws->pt.regroup_force_below * rows_l && ws->xc_src == x_dev && ws->xc_rows == n_rows &&
        ws->xc_ldx == ldx)
        mode = kBound;
    auto regroup_due = [&]() {
        return mode == kBound && ws->sort_rows && ws->xp && ws->hmm == nullptr && same_rows && after_estep && known &&
               L.act <= ws->pt.regroup_below * rows_l && ws->xc_src == x_dev && ws->xc_rows == n_rows && ws->xc_ldx == ldx &&
               (!ws->sorted || ws->moved_since_sort > ws->pt.regroup_moved * rows_l);      // (again once that share of the rows has moved on)
    };
    bool settle = false;
    if (ws->lock) {
        // (a regrouping of the rows takes the per-row state along: regroup_rows)
        const bool keep = mode != kDense && same_rows && !ws->lock_reset && !ws->delta_pending;
        if (ws->lock_reset || (ws->lock_live && !keep)) {
            if (mode == kSweep) mode = kBound;
            // (a failed reset would leave stale addends in the cache: the pass must not go on)
            e = hipMemsetAsync(ws->lock, 0, (size_t)ws->npad, st);
            if (e == hipSuccess) e = hipMemsetAsync(ws->cache, 0, (size_t)gmmvb_stats_len(ws->K, ws->D) * sizeof(double), st);
            if (e != hipSuccess) return fail(GMMVB_EHIP, "resetting the cache of single-component rows", e);
            ws->lock_live = false;
            ws->skip_used = false;
        }
        ws->lock_reset = false;
        ws->delta_pending = false;
        settle = mode != kDense && ws->cache_on && ws->sparse && ws->masks && ws->xc && ws->xc_src == x_dev &&
                 ws->xc_rows == n_rows && ws->xc_ldx == ldx;
    }
    // Rows with a single active component are settled (left out of the E-step as well as of the M-step) in every pruned
    // pass, provided the proof round is available - the int8 digit planes of this matrix are in the workspace, about the
    // pivot the component images were packed for: a settled row whose carried bounds no longer prove it then costs a few
    // int8 pairs.  (Without it such a row costs exact evaluations, and settling while the components still move by per
    // cents made rows come loose in masses - round 2 needed a gate with hysteresis on the drift, profiles/r2_experiments.md.)
    const bool proof_capable = settle && ws->opt_proof && ws->xq != nullptr && ws->img_i8b != nullptr && ws->xq_src == x_dev &&
                               ws->xq_rows == n_rows && ws->xq_ldx == ldx && ws->xq_gen == ws->img_gen;
    const double skip_margin = (proof_capable && ws->settle_margin >= 0.0) ? ws->settle_margin : -1.0;
    ws->settled_fresh = false;
    if (ws->opt_debug)
        std::fprintf(stderr, "[gmmvb] estep: mode=%d known=%d lag(mode=%d act=%.3g eval=%.3g over=%.3g settled=%.3g listed=%.3g) gamma=%.3f rec_valid=%d drift=%d settle=%d\n",
                     mode, (int)known, L.mode, L.act / rows_l, L.eval / rows_l, L.over / rows_l,
                     L.settled / rows_l, L.listed / rows_l, ws->typical_gamma, (int)ws->rec_valid,
                     (int)ws->have_drift, (int)settle);
    if (mode == kBound && ws->img_i8b) {
        // How many output blocks the bound pass evaluates.  Cost model per (sample, component) pair, in units of
        // 1e-11 s (namespace policy above): bound pass kI8BlockPair per block pair + kI8RowOfY per row of y; exact pass
        // kF64TilePair per f64 tile pair of every candidate.  Take the cheapest level among those observed in the last 32
        // bound passes; look one level down when the current one leaves hardly any spare candidates or one level up
        // when more than half of its candidates are spare, if that level is unknown.
        // When the bounds are going to be carried (sweeps follow for tens of passes), all blocks: every nat of slack a bound
        // starts with postpones the pass in which it erodes into a candidate - measured at the benchmark shape (round 3):
        // four blocks instead of the model's three cost 6 ms once and take the following twenty passes from 8.1 to 7.2 ms
        // each (proof pairs halved, a quarter instead of 43 % of the sweep's columns opened).
        const int t32 = (ws->D + 31) / 32;
        // (the caller hands over drift hints - a row-tiled pass, whose bounds do not survive the other tiles, does not)
        const bool carried_after = gmmvb_wants_drift(ws, n_rows) != 0 && ws->have_drift;
        if (ws->bound_tb == 0) ws->bound_tb = t32 > 3 ? 3 : t32;
        if (carried_after) {
            ws->bound_tb = t32;
        } else if (known && L.mode == kBound) {
            const int cur = ws->bound_tb;
            ws->tb_cand[cur] = L.eval / pairs_l;
            ws->tb_act[cur] = L.act / pairs_l;
            ws->tb_seen[cur] = 0;
            for (int l = 1; l <= t32; ++l)
                if (l != cur && (++ws->tb_seen[l] > 32 || ws->tb_act[l] > 1.5 * ws->tb_act[cur] ||
                                 ws->tb_act[l] < ws->tb_act[cur] / 1.5))
                    ws->tb_cand[l] = -1.0;
            const double gpp = ws->pt.f64_tile_pair * tri_pairs(ws->T);
            // carried passes follow a bound pass and inherit its spare candidates: a tighter bound pays for part of itself
            const double heirs = gmmvb_wants_drift(ws, n_rows) ? 3.0 : 0.0;
            auto cost = [&](int l) {
                const double spare_l = ws->tb_cand[l] > ws->tb_act[l] ? ws->tb_cand[l] - ws->tb_act[l] : 0.0;
                return ws->pt.i8_block_pair * tri_pairs(l) + ws->pt.i8_row_of_y * 32 * l + gpp * (ws->tb_cand[l] + heirs * spare_l);
            };
            int best = cur;
            for (int l = 1; l <= t32; ++l)
                if (ws->tb_cand[l] >= 0.0 && cost(l) < cost(best)) best = l;
            const double spare = ws->tb_cand[cur] - ws->tb_act[cur];
            if (best == cur) {
                if (cur > 1 && ws->tb_cand[cur - 1] < 0.0 && spare * ws->K < (heirs > 0.0 ? 0.02 : 0.25))
                    best = cur - 1;
                else if (cur < t32 && ws->tb_cand[cur + 1] < 0.0 &&
                         spare * gpp > ws->pt.i8_block_pair * (tri_pairs(cur + 1) - tri_pairs(cur)) + ws->pt.i8_row_of_y * 32)
                    best = cur + 1;
            }
            ws->bound_tb = best;
        }
    }
    if (mode == kDense)      // whatever was learnt about the bound levels belongs to another regime
        for (double& c : ws->tb_cand) c = -1.0;

    int rpw = 0;
    int64_t grid = 0;
    if (phase_events(ws)) note_hip(ws, hipEventRecord(ws->ev[0], st));
    ws->n_spans = 0;
    // a bound pass rebuilds everything row-indexed anyway: the moment to regroup the internal row order by the best
    // component of the previous pass (once at most 4 components per row are active: later passes are list-driven)
    bool sorted_now = false;
    if (regroup_due()) {
        span_begin(ws, kSpanSelect, st);
        e = regroup_rows(ws, x_dev, ldx, n_rows, st, ws->lock_live, ws->e_state == 1 && !ws->lse_stale);
        span_end(ws, st);
        if (e != hipSuccess) return fail(GMMVB_EHIP, "regrouping the rows", e);
        ws->moved_since_sort = 0.0;
        sorted_now = true;
    }
    if (ws->sorted) {           // the kernels read the workspace's permuted copy
        a.x = a8.x = ws->xp;
        a.ldx = a8.ldx = ws->D;
        vec = ws->D % 16 == 0;
    }
    const int sel_grid = (int)((n_rows + kSelRows - 1) / kSelRows);
    const RecArrays rec{ws->rec_k, ws->rec_d, ws->rec_B, ws->rec_exact, ws->rec_sel, ws->rec_flags, ws->npad};
    bool counted = false, proof_ran = false, tmeta_kept = false, projected = false, filtered = false;
    ws->lse_stale = false;
    const bool tmeta_was_valid = ws->tmeta_valid;
    ws->tmeta_valid = false;            // (only a lazy sweep that ran to its end leaves the tile state in step with the bounds)
    bool emission_to_hmm = false;
    if (mode == kDense) {
        const bool valu16 = ws->estep_variant == kEstepValu16 && ws->tri != nullptr;
        // an HMM pass that only the forward-backward recursions will read: rho' rows and row maxima straight into the HMM
        // state, no ln rho array (hmmvb_emission_target; hmm.h H0 + H1)
        emission_to_hmm = ws->T == 1 && !ws->wide && !i8 && !ws->sorted && hmm_fused_emission(ws->hmm);
        rpw = ws->wide ? estep_rows_rows_per_wg()
                       : (i8 ? estep_i8_rows_per_wg() : (valu16 ? estep_rows16_rows_per_wg() : estep_rows_per_wg(ws->estep_variant, ws->T, is64)));
        grid = (n_rows + rpw - 1) / rpw;
        if (grid > (1 << 20)) grid = 1 << 20;
        span_begin(ws, kSpanEstepMain, st);
        const bool cal_e = !ws->wide && !i8 && !emission_to_hmm && !valu16 && cal_wanted(ws, 0, pairs);
        if (cal_e) note_hip(ws, hipEventRecord(ws->cal_ev[0], st));
        e = ws->wide ? launch_estep_rows(ws->T, is64, (int)grid, st, a, &name)
                     : (i8 ? launch_estep_i8(is64, vec, (int)grid, st, a8, &name)
                           : (emission_to_hmm ? hmm_launch_emission16(ws->hmm, is64, vec, st, a, &name)
                              : (valu16 ? launch_estep_rows16(is64, vec, (int)grid, st, a, ws->tri, &name)
                                     : launch_estep(ws->estep_variant, ws->T, is64, vec, (int)grid, st, a, &name))));
        if (cal_e) cal_mark(ws, 0, pairs, st);
        span_end(ws, st);
        if (e != hipSuccess) return fail(GMMVB_EHIP, "estep launch", e);
        ++ws->passes[0];
        const int lse_blocks = (int)((n_rows + kLseRows - 1) / kLseRows);
        // small passes are launch-bound: no pair counting, no lists (the dense M-step takes microseconds there)
        const bool count_pairs = ws->sparse && ws->masks && ws->hmm == nullptr &&
                                 n_rows * (int64_t)ws->K >= (int64_t(1) << 18);
        span_begin(ws, kSpanLse, st);
        if (count_pairs) {
            // thresholds from a sample of the rows (every 16th block of 1024), then lse + active masks + counts in one pass
            const int stride = lse_blocks >= 64 ? 16 : 1;
            const int sampled = (lse_blocks + stride - 1) / stride;
            hipLaunchKernelGGL(row_lse_kernel, dim3((unsigned)sampled), dim3(256), 0, st, ws->lnrho, ws->npad, n_rows, ws->K,
                               ws->lse, ws->dpart, nullptr, stride);
            hipLaunchKernelGGL(thr_kernel, dim3((unsigned)ws->K), dim3(256), 0, st, ws->dpart, nullptr, sampled, ws->K, ws->thr,
                               ws->ctr);
            hipLaunchKernelGGL(lse_mask_kernel, dim3((unsigned)sel_grid), dim3(kSelRows), 0, st, ws->lnrho, ws->npad, n_rows,
                               ws->K, ws->thr, ws->lse, ws->masks, ws->blk, ws->apart, ws->khat);
            hipLaunchKernelGGL(sum_parts_kernel, dim3(1), dim3(1024), 0, st, ws->apart, nullptr, nullptr, nullptr, nullptr, nullptr,
                               nullptr, nullptr, sel_grid, ws->ctr);
            // records for the next pass (one more sweep of the array, ~1 % of the dense kernel's time)
            if (can_prune && big)
                hipLaunchKernelGGL(rec_build_kernel<false>, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, st, ws->lnrho,
                                   ws->npad, n_rows, ws->K, ws->cvec, nullptr, rec, ws->ub32);
            e = hipGetLastError();
            if (e != hipSuccess) return fail(GMMVB_EHIP, "row_lse / lse_mask launch", e);
            counted = true;
            ws->rec_valid = can_prune && big;
        } else if (ws->hmm != nullptr) {
            // the HMM pass normalises along the time axis (hmm_prep_kernel takes the row maxima): the mixture's
            // log-normaliser is only made if a read-out asks for mixture responsibilities before hmmvb_forward_backward
            ws->lse_stale = true;
            ws->rec_valid = false;
        } else {
            hipLaunchKernelGGL(row_lse_kernel, dim3((unsigned)lse_blocks), dim3(256), 0, st, ws->lnrho, ws->npad, n_rows, ws->K,
                               ws->lse, nullptr, nullptr, 1);
            e = hipGetLastError();
            if (e != hipSuccess) return fail(GMMVB_EHIP, "row_lse launch", e);
            ws->rec_valid = false;
        }
        span_end(ws, st);
        ws->rec_live = false;
        ws->evaluated = pairs;
    } else {
        rc = ensure_lists(ws);
        if (rc) return rc;
        if (mode == kBound) {
            span_begin(ws, kSpanEstepMain, st);
            // (only a pass over all output blocks measures what the table's bound_ns stands for)
            const bool cal_b = ws->bound_tb == (ws->D + 31) / 32 && cal_wanted(ws, 2, pairs);
            if (cal_b) note_hip(ws, hipEventRecord(ws->cal_ev[4], st));
            e = launch_bound_pass(ws, a8, is64, vec, st, &name, &rpw, &grid);
            if (cal_b) cal_mark(ws, 2, pairs, st);
            span_end(ws, st);
            if (e != hipSuccess) return fail(GMMVB_EHIP, "estep_bound launch", e);
            ++ws->passes[1];
            // the best component of every row, exactly
            span_begin(ws, kSpanSelect, st);
            hipLaunchKernelGGL(select_mask_kernel<3>, dim3(sel_grid), dim3(kSelRows), 0, st, ws->lnrho, ws->npad, n_rows,
                               ws->K, ws->khat, ws->masks, ws->blk);
            span_end(ws, st);
            e = lists_and_gather(ws, a, is64, vec, sel_grid, st);
            if (e != hipSuccess) return fail(GMMVB_EHIP, "E-step best-component evaluation", e);
            // records from the bounds (+ the one exact value), then every other candidate
            span_begin(ws, kSpanSelect, st);
            hipLaunchKernelGGL(rec_build_kernel<true>, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, st, ws->lnrho,
                               ws->npad, n_rows, ws->K, ws->cvec, ws->khat, rec, ws->ub32);
            hipLaunchKernelGGL(rec_select_kernel, dim3(sel_grid), dim3(kSelRows), 0, st, rec, n_rows, ws->K, ws->cvec, ws->masks,
                               ws->npad, ws->blk, ws->epart, ws->opart, ws->rthr);
            if (proof_capable) {
                // the candidates' bounds come from the bound pass's leading output blocks only: three int8 digits over ALL
                // blocks first (a third of an exact evaluation's cost), and only what still does not clear the threshold
                // goes to the exact gather
                launch_scan_counts(st, ws->blk, sel_grid, ws->K, ws->counts, ws->scan_parts);
                hipLaunchKernelGGL(fill_lists_kernel, dim3(sel_grid), dim3(kSelRows), 0, st, ws->masks, ws->npad, n_rows, ws->K,
                                   ws->blk, ws->lists, ws->npad, nullptr, nullptr, ws->epart);
                span_end(ws, st);
                span_begin(ws, kSpanProof, st);
                e = proof_round(ws, st, ws->blk, sel_grid, n_rows, ws->ub32);
                span_end(ws, st);
                if (e != hipSuccess) return fail(GMMVB_EHIP, "proof round (bound pass)", e);
                span_begin(ws, kSpanSelect, st);
                hipLaunchKernelGGL(rec_prune_kernel, dim3(sel_grid), dim3(kSelRows), 0, st, rec, ws->masks, ws->npad, n_rows, ws->K,
                                   ws->cvec, ws->ub32, ws->rthr, ws->blk, ws->epart, ws->ppart);
                proof_ran = true;
            }
            span_end(ws, st);
        } else if (mode == kSweep) {
            rpw = kSelRows;
            grid = sel_grid;
            name = "estep_sweep_bounds";
            ++ws->passes[4];
            ++ws->sweeps;
            // round 0: pairs to evaluate exactly under the new parameters before the sweep (its reference values).
            // If the previous pass's M-step ran over lists, those lists - every pair that was active - are still in the
            // workspace with their masks: evaluate them as they are (no list building); else the previous best
            // component of every row.
            if (prev_lists) {
                span_begin(ws, kSpanSelect, st);
                if (!ws->active_lists) {        // (the M-step's lists left out the rows in its cache)
                    launch_scan_counts(st, ws->blk, sel_grid, ws->K, ws->counts, ws->scan_parts);
                    // (gpart: the listed pairs rec_finish_kernel counted per block when it wrote these masks)
                    hipLaunchKernelGGL(fill_lists_kernel, dim3(sel_grid), dim3(kSelRows), 0, st, ws->masks, ws->npad, n_rows, ws->K,
                                       ws->blk, ws->lists, ws->npad, nullptr, nullptr, ws->prev_pass != 0 ? ws->gpart : nullptr);
                }
                hipLaunchKernelGGL(gather_plan_kernel, dim3(1), dim3(64), 0, st, ws->counts, ws->K,
                                   estep_gather_rows_per_wg(ws->T, is64), ws->plan);
                span_end(ws, st);
                span_begin(ws, kSpanGather, st);
                e = launch_estep_gather_dev(ws->T, is64, vec, 2 * ws->num_cu, st, a, ws->lists, ws->npad, ws->counts, ws->plan);
                span_end(ws, st);
                ++ws->passes[7];
                if (e != hipSuccess) return fail(GMMVB_EHIP, "E-step active-pair evaluation", e);
                const bool proof = proof_capable && (ws->skip_used || ws->opt_proof_all);       // (some rows may be settled)
                // Settled rows of components that moved noticeably: a fresh lower bound of their own pair first (three int8
                // digits), so that the sweep compares the other components' bounds with a tight reference instead of one
                // carried through Gamma and delta (records.h, own_first).  While the summary of the drift says that no
                // component moves that much the round is skipped altogether.
                const bool own_round = proof && ws->skip_used && !(ws->typical_gamma >= ws->pt.own_round_below);
                if (own_round) {
                    span_begin(ws, kSpanSelect, st);
                    hipLaunchKernelGGL(settled_mask_kernel, dim3(sel_grid), dim3(kSelRows), 0, st, ws->lock, ws->masks, ws->lcomp,
                                       ws->npad, n_rows, ws->K, ws->rmask, ws->rblk, ws->drift, ws->spart);
                    launch_scan_counts(st, ws->rblk, sel_grid, ws->K, ws->counts, ws->scan_parts);
                    hipLaunchKernelGGL(fill_lists_kernel, dim3(sel_grid), dim3(kSelRows), 0, st, ws->rmask, ws->npad, n_rows,
                                       ws->K, ws->rblk, ws->lists, ws->npad, nullptr, nullptr, ws->spart);
                    span_end(ws, st);
                    span_begin(ws, kSpanProof, st);
                    e = proof_round(ws, st, ws->rblk, sel_grid, n_rows, nullptr);
                    span_end(ws, st);
                    if (e != hipSuccess) return fail(GMMVB_EHIP, "proof round (settled rows' own pairs)", e);
                }
                span_begin(ws, kSpanSelect, st);
                if (can_project && ws->opt_project == 2) {
                    // bounds from the table of the parameters in force and the rows' digit planes: nothing carried, nothing
                    // written back (the per-pair array is void afterwards: ws->dense_valid below)
                    note_hip(ws, hipMemsetAsync(ws->exit_ctr + 2, 0, sizeof(unsigned long long), st));
                    ProjectArgs pa{ws->xq, ws->xqe, ws->gimg, ws->gconst, ws->tile_ref, ws->lnrho, ws->npad, n_rows, ws->K,
                                   ws->D, ws->drift, ws->cvec, ws->rec_k, ws->rec_d, ws->rec_B, ws->rec_exact, ws->rec_sel,
                                   ws->rec_flags, ws->masks, ws->blk, ws->epart, ws->opart, settle ? ws->lock : nullptr, ws->dlock,
                                   ws->rthr, ws->lcomp, proof ? ws->rmask : nullptr, ws->rblk, ws->opt_proof_all ? 1 : 0,
                                   own_round ? 1 : 0, ws->exit_ctr + 2};
                    e = launch_rec_project(sel_grid, st, pa);
                    if (e != hipSuccess) return fail(GMMVB_EHIP, "rec_project launch", e);
                    name = "estep_sweep_projected";
                    projected = true;
                } else if (ws->tmeta) {
                    note_hip(ws, hipMemsetAsync(ws->exit_ctr + 1, 0, sizeof(unsigned long long), st));
                    // (the tile state is void after any pass that rewrote the bounds wholesale: the first sweep after it
                    // opens every column and takes stock)
#define GMMVB_LAZY_SWEEP(WC)                                                                                                   \
    hipLaunchKernelGGL((rec_sweep_kernel<true, true, WC>), dim3(sel_grid), dim3(kSelRows), 0, st, ws->ub32, ws->lnrho,        \
                       ws->npad, n_rows, ws->K, ws->drift, ws->cvec, ws->khat, rec, ws->masks, ws->blk, ws->epart,            \
                       ws->opart, settle ? ws->lock : nullptr, ws->dlock, ws->rthr, ws->lcomp,                                 \
                       proof ? ws->rmask : nullptr, ws->rblk, ws->opt_proof_all ? 1 : 0, own_round ? 1 : 0, ws->tmeta,         \
                       tmeta_was_valid ? 0 : 1, ws->exit_ctr + 1, ws->ppart)
                    switch ((ws->K + 63) / 64) {        // (mask words as a compile-time constant)
                        case 1: GMMVB_LAZY_SWEEP(1); break;
                        case 2: GMMVB_LAZY_SWEEP(2); break;
                        case 3: GMMVB_LAZY_SWEEP(3); break;
                        default: GMMVB_LAZY_SWEEP(4); break;
                    }
#undef GMMVB_LAZY_SWEEP
                    tmeta_kept = true;
                } else {
                    hipLaunchKernelGGL(rec_sweep_kernel<true>, dim3(sel_grid), dim3(kSelRows), 0, st, ws->ub32, ws->lnrho, ws->npad,
                                       n_rows, ws->K, ws->drift, ws->cvec, ws->khat, rec, ws->masks, ws->blk, ws->epart, ws->opart,
                                       settle ? ws->lock : nullptr, ws->dlock, ws->rthr, ws->lcomp, proof ? ws->rmask : nullptr,
                                       ws->rblk, ws->opt_proof_all ? 1 : 0, own_round ? 1 : 0, nullptr, 0, nullptr, ws->ppart);
                }
                if (proof && can_project && !projected) {
                    // the table of the parameters in force first (project.h): a listed pair it clears needs no proof - most of
                    // them are far pairs whose carried bound has eroded to the relevance line
                    ProjectArgs pa{ws->xq, ws->xqe, ws->gimg, ws->gconst, ws->tile_ref, ws->lnrho, ws->npad, n_rows, ws->K,
                                   ws->D, ws->drift, ws->cvec, ws->rec_k, ws->rec_d, ws->rec_B, ws->rec_exact, ws->rec_sel,
                                   ws->rec_flags, ws->masks, ws->blk, ws->epart, ws->opart, ws->lock, ws->dlock, ws->rthr, ws->lcomp,
                                   ws->rmask, ws->rblk, 0, 0, ws->exit_ctr + 2};
                    note_hip(ws, hipMemsetAsync(ws->exit_ctr + 2, 0, sizeof(unsigned long long), st));
                    e = launch_proj_filter(sel_grid, st, pa);
                    if (e != hipSuccess) return fail(GMMVB_EHIP, "proj_filter launch", e);
                    filtered = true;
                }
                if (proof) {
                    // proof round: settled rows whose carried bounds left candidates - their component and the candidates
                    // get two-sided bounds from three int8 digits; rows that are proven stay settled, the others join
                    // the pass's lists (records.h)
                    launch_scan_counts(st, ws->rblk, sel_grid, ws->K, ws->counts, ws->scan_parts);
                    // (ppart: the proof pairs the sweep listed per block; rec_proof_decide_kernel overwrites it afterwards)
                    hipLaunchKernelGGL(fill_lists_kernel, dim3(sel_grid), dim3(kSelRows), 0, st, ws->rmask, ws->npad, n_rows,
                                       ws->K, ws->rblk, ws->lists, ws->npad, nullptr, nullptr, projected ? nullptr : ws->ppart);
                    span_end(ws, st);
                    span_begin(ws, kSpanProof, st);
                    e = proof_round(ws, st, ws->rblk, sel_grid, n_rows, ws->ub32);
                    span_end(ws, st);
                    if (e != hipSuccess) return fail(GMMVB_EHIP, "proof round", e);
                    span_begin(ws, kSpanSelect, st);
                    hipLaunchKernelGGL(rec_proof_decide_kernel, dim3(sel_grid), dim3(kSelRows), 0, st, rec, ws->rmask, ws->masks,
                                       ws->npad, n_rows, ws->K, ws->cvec, ws->ub32, ws->lnrho, ws->lcomp, ws->dlock, ws->rthr,
                                       ws->blk, ws->epart, ws->ppart, own_round ? ws->spart : nullptr);
                    proof_ran = true;
                }
                span_end(ws, st);
            } else {
                span_begin(ws, kSpanSelect, st);
                hipLaunchKernelGGL(select_mask_kernel<3>, dim3(sel_grid), dim3(kSelRows), 0, st, ws->lnrho, ws->npad, n_rows,
                                   ws->K, ws->khat, ws->masks, ws->blk);
                span_end(ws, st);
                e = lists_and_gather(ws, a, is64, vec, sel_grid, st);
                if (e != hipSuccess) return fail(GMMVB_EHIP, "E-step best-component evaluation", e);
                span_begin(ws, kSpanSelect, st);
                hipLaunchKernelGGL(rec_sweep_kernel<false>, dim3(sel_grid), dim3(kSelRows), 0, st, ws->ub32, ws->lnrho, ws->npad, n_rows,
                                   ws->K, ws->drift, ws->cvec, ws->khat, rec, ws->masks, ws->blk, ws->epart, ws->opart,
                                   settle ? ws->lock : nullptr, ws->dlock, ws->rthr, ws->lcomp, nullptr, nullptr, 0, 0, nullptr, 0);
                span_end(ws, st);
            }
            ws->sweep_prev = prev_lists;
        }
        // candidates: a pair whose first output blocks already put it below the row's threshold is not evaluated further
        e = lists_and_gather(ws, a, is64, vec, sel_grid, st, ws->gather_exit ? ws->rthr : nullptr, ws->epart);
        if (e != hipSuccess) return fail(GMMVB_EHIP, "E-step candidate evaluation", e);
        span_begin(ws, kSpanLse, st);
        hipLaunchKernelGGL(rec_finish_kernel, dim3(sel_grid), dim3(kSelRows), 0, st, rec, ws->lnrho, ws->npad, n_rows, ws->K,
                           ws->cvec, ws->lse, ws->khat, ws->masks, ws->blk, ws->apart, ws->mpart, ws->ub32,
                           settle ? ws->lock : nullptr, ws->dlock, skip_margin, settle ? ws->dmask : nullptr,
                           settle ? ws->dblk : nullptr, ws->mmask, ws->mblk, ws->spart, ws->gpart, ws->qpart, ws->rthr, ws->lcomp);
        if (!proof_ran) note_hip(ws, hipMemsetAsync(ws->ctr + 7, 0, sizeof(double), st));
        hipLaunchKernelGGL(sum_parts_kernel, dim3(proof_ran ? 8 : 7), dim3(1024), 0, st, ws->apart, ws->epart, ws->opart, ws->mpart,
                           ws->spart, ws->gpart, ws->qpart, ws->ppart, sel_grid, ws->ctr);
        e = hipGetLastError();
        span_end(ws, st);
        if (e != hipSuccess) return fail(GMMVB_EHIP, "rec_finish launch", e);
        counted = true;
        ws->rec_valid = true;
        ws->rec_live = true;
        ws->evaluated = -1.0;
        if (settle) {
            ws->lock_live = true;
            ws->delta_pending = true;
        }
    }
    ws->tmeta_valid = tmeta_kept;
    ws->pend_lazy = tmeta_kept;
    ws->pend_proj = projected || filtered;
    // the E phase of the profile ends behind the pass's LAST kernel (round 4; before, rec_finish / lse_mask - 0.2-0.4 ms of
    // E-step work at the benchmark shape - fell between the two phases and were booked as "outside the data pass")
    if (phase_events(ws)) {
        note_hip(ws, hipEventRecord(ws->ev[1], st));
        ws->ev_e = true;
    }
    // counters -> pinned host memory, behind an event (read by the next pass, or by gmmvb_last_sparsity)
    if (counted) {
        e = hipMemcpyAsync(ws->ctr_host, ws->ctr, 8 * sizeof(double), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess && ws->exit_ctr && mode != kDense)
            e = hipMemcpyAsync(ws->exit_host, ws->exit_ctr, 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipEventRecord(ws->ctr_ev, st);
        if (e != hipSuccess) return fail(GMMVB_EHIP, "E-step counters", e);
        ws->ctr_pending = true;
        ws->pend_mode = mode;
        ws->pend_rows = n_rows;
        // pairs evaluated before the counted selection: every row's best component, or (sweep over the previous
        // pass's lists) the previous pass's active pairs
        ws->pend_first_sorted = sorted_now;
        ws->pend_round0 = (mode == kSweep && ws->sweep_prev && own_known) ? ws->lag.listed : (double)n_rows;
        ws->act_rows = n_rows;
    } else {
        ws->ctr_pending = false;
        ws->lag.valid = false;
        ws->act_rows = 0;              // nothing counted: dense M-step, no pruning decision from this pass
    }
    ws->exp_counted = counted;
    ws->pol.valid = false;             // (a sharded job imports this pass's sums before the next E-step)
    ws->act_host = -1.0;
    ws->active_lists = false;
    ws->mlists_done = ws->mlists_lost = false;
    if (skip_margin >= 0.0) ws->skip_used = true;
    ws->blk_fresh = counted;
    ws->e_state = emission_to_hmm ? 4 : 1;
    ws->hmm_no_lnrho = emission_to_hmm;
    ws->lost_estep = false;
    ws->e_rows = n_rows;
    ws->params_used = true;
    ws->have_drift = false;
    ws->bounds_rows = n_rows;          // the records / the ln rho array now belong to the parameters in force, on these rows
    ws->bounds_x = x_dev;
    ws->bounds_ldx = ldx;
    ws->prev_pass = mode;
    // the f32 bound array holds a value or bound under the parameters in force for EVERY pair after a dense pass, a
    // bound pass or a sweep; a pass on records only refreshes the evaluated entries
    // (a projected sweep leaves the array alone: it is void until a dense or bound pass rewrites it)
    ws->dense_valid = !projected;
    if (mode == kDense || mode == kBound) ws->sweeps = 0;
    std::snprintf(ws->info, sizeof(ws->info), "%s grid=%lldx%d rows/workgroup=%d", name, (long long)grid,
                  (i8 || mode != kDense) ? 512 : estep_threads(ws->estep_variant), rpw);
    return take_hip(ws, "event record / counter reset inside the E-step");
}

}  // extern "C"
