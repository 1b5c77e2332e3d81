// C ABI of the GMM-VB data-pass engine (see include/gmmvb.h for the contract and the reference
// call sites each entry point replaces).
// This unit: workspace life cycle, parameters and drift hints, pass counters and the policy exchange of row shards, row
// preparation.  gmmvb_estep is capi_estep.hip, gmmvb_mstep capi_mstep.hip, the read-outs capi_readout.hip.
#include "capi_internal.h"

namespace {
thread_local std::string g_err;
}  // namespace

namespace gmmvb {
int fail(int code, const char* what, hipError_t e) {
    g_err = what;
    if (e != hipSuccess) {
        g_err += ": ";
        g_err += hipGetErrorString(e);
    }
    return code;
}
}  // namespace gmmvb

extern "C" {

int gmmvb_abi_version(void) { return GMMVB_ABI_VERSION; }
const char* gmmvb_last_error(void) { return g_err.c_str(); }

int64_t gmmvb_stats_len(int K, int D) {
    if (K < 1 || D < 1) return -1;
    return (int64_t)K * (2 + (int64_t)D + (int64_t)D * D);
}

static int create_workspace(int K, int D, int x_dtype, int64_t max_rows, gmmvb_workspace* first, gmmvb_workspace** out) {
    if (!out) return fail(GMMVB_EINVAL, "out is null");
    *out = nullptr;
    if (K < 1 || D < 1 || max_rows < 1) return fail(GMMVB_EINVAL, "K, D and max_rows must be positive");
    if (x_dtype != GMMVB_F32 && x_dtype != GMMVB_F64) return fail(GMMVB_EINVAL, "x_dtype must be GMMVB_F32 or GMMVB_F64");
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return fail(GMMVB_EHIP, "hipGetDevice", e);
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, dev);
    if (e != hipSuccess) return fail(GMMVB_EHIP, "hipGetDeviceProperties", e);

    gmmvb_workspace* ws = new (std::nothrow) gmmvb_workspace();
    if (!ws) return fail(GMMVB_ENOMEM, "host allocation failed");
    if (first) {
        ws->scratch = first->scratch;
        ++ws->scratch->refs;
    } else {
        ws->scratch = new (std::nothrow) gmmvb_scratch();
        if (!ws->scratch) {
            delete ws;
            return fail(GMMVB_ENOMEM, "host allocation failed");
        }
        ws->scratch->refs = 1;
    }
    ws->K = K;
    ws->D = D;
    ws->T = (D + 15) / 16;
    // 128 < D <= 256 (round 4): dense MFMA kernels of their own - the E-step streams U's block rows through LDS
    // (estep_rows.h), the M-step spreads a component's tile pairs over T / 2 waves (mstep.h) - instantiated for even tile
    // counts: an odd one is rounded up (zero padded images and centred rows).  No pruning, no lists, K-side through torch.
    ws->wide = D > 16 * kMaxTiles && D <= 32 * kMaxTiles;
    if (ws->wide) ws->T = 2 * ((ws->T + 1) / 2);
    ws->x_dtype = x_dtype;
    ws->max_rows = max_rows;
    ws->npad = round_up(max_rows, 64);
    ws->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (D > 32 * kMaxTiles) {
        // more than 16 feature tiles: the plain f64 kernels of generic.h (no parameter images, no pruning, no lists)
        ws->generic = true;
        ws->sparse = false;
        ws->prune = 0;
        ws->sort_rows = false;
        ws->cache_on = false;
        ws->gen_S = (int)std::min<int64_t>(64, std::max<int64_t>(1, max_rows / 16384));
        const int64_t tiles = tri_pairs(ws->T);
        struct { double** p; int64_t n; } gb[] = {
            {&ws->lnrho, (int64_t)K * ws->npad}, {&ws->lse, ws->npad}, {&ws->cvec, K}, {&ws->pivot, D},
            {&ws->gen_u, (int64_t)K * D * D}, {&ws->gen_m, (int64_t)K * D}, {&ws->gen_first, (int64_t)ws->gen_S * K * (D + 2)},
            {&ws->gen_second, (int64_t)ws->gen_S * K * tiles * 256}, {&ws->ctr, 8}};
        for (auto& b : gb) {
            e = hipMalloc((void**)b.p, (size_t)b.n * sizeof(double));
            if (e != hipSuccess) {
                gmmvb_workspace_destroy(ws);
                return fail(GMMVB_ENOMEM, "hipMalloc (workspace)", e);
            }
            ws->bytes += b.n * (int64_t)sizeof(double);
        }
        ws->scratch->lnrho = ws->lnrho;            // (freed with the scratch; the generic path has no tile groups)
        ws->scratch->npad = ws->npad;
        e = hipMemset(ws->pivot, 0, (size_t)D * sizeof(double));
        if (e == hipSuccess) e = hipMemset(ws->ctr, 0, 8 * sizeof(double));
        if (e == hipSuccess) e = hipHostMalloc((void**)&ws->ctr_host, 8 * sizeof(double), hipHostMallocDefault);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&ws->ctr_ev, hipEventDisableTiming);
        if (e == hipSuccess) {
            const size_t lds = (size_t)D * generic_rows(D) * sizeof(double);
            e = hipFuncSetAttribute((const void*)estep_generic_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e == hipSuccess)
                e = hipFuncSetAttribute((const void*)estep_generic_kernel<double>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        }
        if (e != hipSuccess) {
            gmmvb_workspace_destroy(ws);
            return fail(GMMVB_EHIP, "workspace initialisation", e);
        }
        *out = ws;
        return GMMVB_OK;
    }
    {
        const int kpw = mstep_components_per_wg(ws->T, false);     // the smaller of the two forms: sizes the slabs
        ws->KG = (K + kpw - 1) / kpw;
    }
    // row splits of the dense M-step: ~4 workgroups per CU; with a single feature tile a step is a row load and one
    // MFMA - latency, not arithmetic - so many more, shorter, splits (HMM config 5)
    ws->S_cap = (int)round_up(((int64_t)(ws->T == 1 ? GMMVB_T1_SPLITS : 4) * ws->num_cu + ws->KG - 1) / ws->KG, 8);      // (24: three of the HMM M-step's 50-KB workgroups per CU)
    if (ws->S_cap < 8) ws->S_cap = 8;
    {
        // Cap on the rows of one M-step split: 16 MB of centred rows (16384 rows at D = 128).  All component groups
        // of a split stream the same rows; short splits keep those workgroups within an L2's reach of each other
        // (measured at C3: fetch 96 GB -> 16-21 GB ~ the algorithmic 15.4 GB, kernel 175.6 -> 171.5 ms) at the
        // price of more slabs (+0.7 ms reduce).
        ws->split_rows = round_up(std::max<int64_t>(64, (16 << 20) / (16 * ws->T * 8)), 64);
        const int64_t need = round_up((max_rows + ws->split_rows - 1) / ws->split_rows, 8);
        if (need > ws->S_cap) ws->S_cap = (int)need;
    }
    ws->img_len = estep_image_doubles(ws->T);
    {
        const char* v = dev_env("GMMVB_ESTEP_VARIANT");
        ws->estep_variant = kEstepLds8;      // measured fastest (two waves per SIMD share one LDS image)
        if (ws->wide) v = nullptr;           // (one E-step kernel past 8 feature tiles)
        // ... except with a single feature tile (D <= 16): the 2.5-KB images stay in L1, staging them through LDS with a
        // barrier per group of components only costs (HMM config 5: emission 4.4 -> 3.1 ms).  No pruning at that size anyway.
        // (round 4) ... and with one tile the vector ALU, which can skip U's upper triangle, beats the matrix pipe (estep.h,
        // estep_rows16_f64: HMM emission 3.1 -> 2.7 ms); GMMVB_ESTEP_VARIANT=direct keeps the MFMA kernel
        if (ws->T == 1) ws->estep_variant = kEstepValu16;
        if (v && std::strcmp(v, "lds8") == 0) ws->estep_variant = kEstepLds8;
        if (v && std::strcmp(v, "direct") == 0) ws->estep_variant = kEstepDirect;
        if (v && std::strcmp(v, "lds4") == 0) ws->estep_variant = kEstepLds;
        if (v && std::strcmp(v, "i8") == 0) ws->estep_variant = kEstepI8;
    }
    if (ws->estep_variant == kEstepValu16) {
        e = hipMalloc((void**)&ws->tri, (size_t)K * estep_tri_image_doubles() * sizeof(double));
        if (e != hipSuccess) {
            gmmvb_workspace_destroy(ws);
            return fail(GMMVB_ENOMEM, "hipMalloc (packed triangular images)", e);
        }
        ws->bytes += (int64_t)K * estep_tri_image_doubles() * (int64_t)sizeof(double);
    }
    struct { double** p; int64_t n; } bufs[] = {
        {&ws->lnrho, (int64_t)K * ws->npad}, {&ws->lse, ws->npad},
        {&ws->img, (int64_t)K * ws->img_len},
        {&ws->cvec, K},                      {&ws->pivot, D},
        {&ws->slabs, (int64_t)ws->S_cap * K * slab_len(ws->T)},
        {&ws->xc, 0},
        {&ws->dpart, ((ws->npad + kLseRows - 1) / kLseRows) * K}, {&ws->thr, K},
        {&ws->apart, (ws->npad + kSelRows - 1) / kSelRows}, {&ws->ctr, 8}, {&ws->drift, 4 * (int64_t)K}};
    bufs[6].n = (ws->npad + 64) * 16 * (int64_t)ws->T;            // the centred copy
    {
        const char* v = std::getenv("GMMVB_MSTEP_SPARSE");         // "0" = always the dense M-step
        ws->sparse = !(v && std::strcmp(v, "0") == 0);
        if (max_rows > 2000000000) ws->sparse = false;             // the sample lists hold 32-bit row numbers
        if (ws->wide) ws->sparse = false;                          // (dense kernels only past 8 feature tiles)
        if (K == 1) {
            // one component: r = 1 for every row, nothing to prune, list or cache - and the one-pass moment computation of
            // multivariate_normal.LearnModel (K = 1, unit responsibilities) should not pay for a centred copy it never
            // builds: the M-step reads x directly
            ws->sparse = false;
            if (!ws->wide) bufs[6].n = 0;          // (past 8 feature tiles the M-step only exists over the centred copy)
        }
        v = dev_env("GMMVB_SORT_ROWS");
        ws->sort_rows = !(v && std::strcmp(v, "0") == 0) && (int64_t)max_rows <= 2000000000;
        v = std::getenv("GMMVB_ESTEP_PRUNE");
        ws->prune = (v && std::strcmp(v, "0") == 0) ? 0 : ((v && std::strcmp(v, "force") == 0) ? 2 : 1);
        if (!ws->sparse || estep_bound_blocks(ws->T) == 0 || K > 256) ws->prune = 0;
        v = dev_env("GMMVB_SETTLE_MARGIN");                    // nats; negative = never settle rows
        if (v) ws->settle_margin = std::atof(v);
        v = dev_env("GMMVB_PROOF");                            // "0": no int8 proof round (rows then never settle);
        ws->opt_proof = !(v && std::strcmp(v, "0") == 0);
        ws->opt_proof_all = !(v && std::strcmp(v, "settled") == 0);     // "settled": only the settled rows' pairs go through it
        v = dev_env("GMMVB_PROOF_BLOCKED");                    // "0": the proof round walks component after component
        ws->opt_proof_blocked = !(v && std::strcmp(v, "0") == 0);
        v = dev_env("GMMVB_SWEEP_LAZY");                       // "0": every sweep reads all K bounds of every row
        ws->opt_lazy = !(v && std::strcmp(v, "0") == 0);
        // the stateless table of project.h - off by default (measured, profiles/r6_experiments.md: on the benchmark's fits it
        // costs more than the proof pairs it saves): "filter" = it takes pairs off the carried sweep's proof lists, "only" =
        // it replaces the carried per-pair bounds
        v = dev_env("GMMVB_PROJECT");
        ws->opt_project = (v && std::strcmp(v, "filter") == 0) ? 1 : ((v && std::strcmp(v, "only") == 0) ? 2 : 0);
        v = dev_env("GMMVB_REGROUP_MARGIN");     // "0": the rows are regrouped by best component only
        ws->opt_regroup_margin = !(v && std::strcmp(v, "0") == 0);
        v = dev_env("GMMVB_GATHER_EXIT");                      // "0": candidates are always evaluated in full
        ws->gather_exit = !(v && std::strcmp(v, "0") == 0);
        v = dev_env("GMMVB_MSTEP_CACHE");
        ws->cache_on = !(v && std::strcmp(v, "0") == 0);
        ws->opt_carry_off = dev_env("GMMVB_ESTEP_CARRY_OFF") != nullptr;
        ws->opt_hmm_mstep_dense = dev_env("GMMVB_HMM_MSTEP_DENSE") != nullptr;
        {
            const char* dbg = std::getenv("GMMVB_DEBUG");
            ws->opt_debug = dbg && dbg[0] == '2';
        }
    }
    {
        const bool full = ws->estep_variant == kEstepI8, bound = ws->prune != 0;
        hipError_t e8 = hipSuccess;
        if (full) {
            ws->img_i8_len = estep_i8_image_bytes(D, 0);
            e8 = hipMalloc((void**)&ws->img_i8, (size_t)K * ws->img_i8_len);
            ws->bytes += (int64_t)K * ws->img_i8_len;
        }
        if (bound && e8 == hipSuccess) {
            ws->img_i8b_len = estep_i8_image_bytes(D, 1);
            e8 = hipMalloc((void**)&ws->img_i8b, (size_t)K * ws->img_i8b_len);
            ws->bytes += (int64_t)K * ws->img_i8b_len;
        }
        if ((full || bound) && e8 == hipSuccess) {
            e8 = hipMalloc((void**)&ws->pivot_i8, (size_t)D * sizeof(double));
            ws->bytes += D * (int64_t)sizeof(double);
        }
        if (e8 != hipSuccess) {
            gmmvb_workspace_destroy(ws);
            return fail(GMMVB_ENOMEM, "hipMalloc (int8 images)", e8);
        }
    }
    for (auto& b : bufs) {
        if (b.n == 0) continue;
        const bool shared = b.p == &ws->lnrho || b.p == &ws->xc || b.p == &ws->slabs;
        if (shared && first) continue;             // a further tile of a group: the first tile's buffers (below)
        e = hipMalloc((void**)b.p, (size_t)b.n * sizeof(double));
        if (e != hipSuccess) {
            if (shared) *b.p = nullptr;
            gmmvb_workspace_destroy(ws);
            return fail(GMMVB_ENOMEM, "hipMalloc (workspace)", e);
        }
        ws->bytes += b.n * (int64_t)sizeof(double);
    }
    {
        gmmvb_scratch* sc = ws->scratch;
        if (!first) {
            sc->lnrho = ws->lnrho;
            sc->xc = ws->xc;
            sc->slabs = ws->slabs;
            sc->npad = ws->npad;
            sc->xc_len = bufs[6].n;
            sc->slabs_len = bufs[5].n;
        } else {
            if (ws->npad > sc->npad || bufs[6].n > sc->xc_len || bufs[5].n > sc->slabs_len || (bufs[6].n > 0 && !sc->xc)) {
                gmmvb_workspace_destroy(ws);
                return fail(GMMVB_EINVAL, "a further tile must not be larger than the group's first workspace");
            }
            ws->lnrho = sc->lnrho;
            ws->xc = bufs[6].n > 0 ? sc->xc : nullptr;
            ws->slabs = sc->slabs;
        }
    }
    if (ws->sparse && K <= 256 && ensure_lists(ws) != GMMVB_OK) {      // not inside somebody's timed iteration
        gmmvb_workspace_destroy(ws);
        return GMMVB_ENOMEM;
    }
    e = hipMemset(ws->pivot, 0, (size_t)D * sizeof(double));
    if (e == hipSuccess) e = hipMemset(ws->ctr, 0, 8 * sizeof(double));
    if (e == hipSuccess) e = hipHostMalloc((void**)&ws->ctr_host, 8 * sizeof(double), hipHostMallocDefault);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ws->ctr_ev, hipEventDisableTiming);
    // the policy table at this shape, and the events its calibration records around the workspace's own first bulk passes
    ws->pt.init(ws->T, estep_bound_blocks(ws->T) ? (D + 31) / 32 : 0);
    {
        const char* v = dev_env("GMMVB_POLICY_CALIBRATE");
        ws->opt_calibrate = !(v && std::strcmp(v, "0") == 0);
    }
    for (int i = 0; i < 6 && e == hipSuccess && ws->prune != 0; ++i) e = hipEventCreate(&ws->cal_ev[i]);
    if (e != hipSuccess) {
        gmmvb_workspace_destroy(ws);
        return fail(GMMVB_EHIP, "workspace initialisation", e);
    }
    *out = ws;
    return GMMVB_OK;
}

int gmmvb_workspace_create(int K, int D, int x_dtype, int64_t max_rows, gmmvb_workspace** out) {
    return create_workspace(K, D, x_dtype, max_rows, nullptr, out);
}

int gmmvb_workspace_create_tile(gmmvb_workspace* first, int64_t max_rows, gmmvb_workspace** out) {
    if (!out) return fail(GMMVB_EINVAL, "out is null");
    *out = nullptr;
    if (!first || !first->scratch) return fail(GMMVB_EINVAL, "null argument");
    if (first->generic || first->hmm != nullptr)
        return fail(GMMVB_EUNSUPPORTED, "tile groups: mixture workspaces with c_degree <= 256 only (the HMM's time axis does not tile)");
    if (max_rows < 1 || max_rows > first->max_rows)
        return fail(GMMVB_EINVAL, "a further tile must not be larger than the group's first workspace");
    return create_workspace(first->K, first->D, first->x_dtype, max_rows, first, out);
}

int gmmvb_workspace_destroy(gmmvb_workspace* ws) {
    if (!ws) return GMMVB_OK;
    release_scratch(ws);
    double* bufs[] = {ws->lnrho, ws->lse, ws->img, ws->tri, ws->cvec, ws->pivot, ws->slabs, ws->xc, ws->dpart, ws->thr,
                      ws->apart, ws->ctr, ws->drift, ws->epart, ws->opart, ws->mpart, ws->gen_u, ws->gen_m, ws->gen_first,
                      ws->gen_second};
    int* ibufs[] = {ws->lists, ws->khat, ws->counts, ws->blk, ws->scan_parts, ws->plan, ws->plan_m, ws->perm, ws->iperm, ws->perm_tmp};
    if (ws->xp) (void)hipFree(ws->xp);
    void* rbufs[] = {ws->rec_k, ws->rec_d, ws->rec_B, ws->rec_exact, ws->rec_sel, ws->rec_flags, ws->ub32,
                     ws->lock, ws->lcomp, ws->dlock, ws->rthr, ws->exit_ctr, ws->dmask, ws->dblk, ws->mmask, ws->mblk, ws->cache, ws->spart, ws->gpart, ws->qpart,
                     ws->rmask, ws->rblk, ws->xq, ws->xqe, ws->ppart, ws->tmeta, ws->gimg, ws->gconst, ws->tile_ref};
    for (void* p : rbufs)
        if (p) (void)hipFree(p);
    if (ws->ctr_host) (void)hipHostFree(ws->ctr_host);
    if (ws->pol_host) (void)hipHostFree(ws->pol_host);
    if (ws->pol_ev) (void)hipEventDestroy(ws->pol_ev);
    if (ws->exit_host) (void)hipHostFree(ws->exit_host);
    if (ws->ctr_ev) (void)hipEventDestroy(ws->ctr_ev);
    for (int* p : ibufs)
        if (p) (void)hipFree(p);
    if (ws->masks) (void)hipFree(ws->masks);
    for (double* p : bufs)
        if (p) (void)hipFree(p);
    if (ws->img_i8) (void)hipFree(ws->img_i8);
    if (ws->img_i8b) (void)hipFree(ws->img_i8b);
    if (ws->pivot_i8) (void)hipFree(ws->pivot_i8);
    for (hipEvent_t e : ws->ev)
        if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : ws->span_ev)
        if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : ws->cal_ev)
        if (e) (void)hipEventDestroy(e);
    if (ws->hmm) hmm_state_destroy(ws->hmm);
    delete ws;
    return GMMVB_OK;
}

int64_t gmmvb_workspace_bytes(const gmmvb_workspace* ws) { return ws ? ws->bytes : -1; }

const char* gmmvb_last_launch_info(const gmmvb_workspace* ws) { return ws ? ws->info : ""; }

int gmmvb_pass_counts(const gmmvb_workspace* ws, int64_t* out /*[8]*/) {
    if (!ws || !out) return fail(GMMVB_EINVAL, "null argument");
    for (int i = 0; i < 8; ++i) out[i] = ws->passes[i];
    return GMMVB_OK;
}

int gmmvb_profile_enable(gmmvb_workspace* ws, int on) {
    if (!ws) return fail(GMMVB_EINVAL, "null argument");
    if (on && !ws->ev[0]) {
        for (auto& e : ws->ev) {
            hipError_t rc = hipEventCreate(&e);
            if (rc != hipSuccess) return fail(GMMVB_EHIP, "hipEventCreate", rc);
        }
        for (auto& e : ws->span_ev) {
            hipError_t rc = hipEventCreate(&e);
            if (rc != hipSuccess) return fail(GMMVB_EHIP, "hipEventCreate", rc);
        }
    }
    ws->prof = on != 0;
    ws->prof_light = on == 2;
    ws->span_open = false;
    ws->ev_e = ws->ev_m = false;
    ws->n_spans = 0;
    return GMMVB_OK;
}

const char* gmmvb_profile_span_name(int slot) { return (slot >= 0 && slot < kSpanSlots) ? kSpanNames[slot] : ""; }

int gmmvb_profile_spans(gmmvb_workspace* ws, float* ms /*[8]*/, int* launches /*[8]*/) {
    if (!ws || !ms || !launches) return fail(GMMVB_EINVAL, "null argument");
    for (int i = 0; i < kSpanSlots; ++i) {
        ms[i] = 0.0f;
        launches[i] = 0;
    }
    for (int i = 0; i < ws->n_spans; ++i) {
        float t = 0.0f;
        hipError_t rc = hipEventSynchronize(ws->span_ev[2 * i + 1]);
        if (rc == hipSuccess) rc = hipEventElapsedTime(&t, ws->span_ev[2 * i], ws->span_ev[2 * i + 1]);
        if (rc != hipSuccess) return fail(GMMVB_EHIP, "event timing (span)", rc);
        ms[ws->span_slot[i]] += t;
        ++launches[ws->span_slot[i]];
    }
    return GMMVB_OK;
}

int gmmvb_profile_last_ms(gmmvb_workspace* ws, float* estep_ms, float* mstep_ms) {
    if (!ws || !estep_ms || !mstep_ms) return fail(GMMVB_EINVAL, "null argument");
    *estep_ms = *mstep_ms = -1.0f;
    if (ws->ev_e) {
        hipError_t rc = hipEventSynchronize(ws->ev[1]);
        if (rc == hipSuccess) rc = hipEventElapsedTime(estep_ms, ws->ev[0], ws->ev[1]);
        if (rc != hipSuccess) return fail(GMMVB_EHIP, "event timing (estep)", rc);
    }
    if (ws->ev_m) {
        hipError_t rc = hipEventSynchronize(ws->ev[3]);
        if (rc == hipSuccess) rc = hipEventElapsedTime(mstep_ms, ws->ev[2], ws->ev[3]);
        if (rc != hipSuccess) return fail(GMMVB_EHIP, "event timing (mstep)", rc);
    }
    return GMMVB_OK;
}

int gmmvb_set_pivot(gmmvb_workspace* ws, const double* pivot_dev, void* stream) {
    if (!ws || !pivot_dev) return fail(GMMVB_EINVAL, "null argument");
    hipError_t e = hipMemcpyAsync(ws->pivot, pivot_dev, (size_t)ws->D * sizeof(double), hipMemcpyDeviceToDevice,
                                  (hipStream_t)stream);
    if (e != hipSuccess) return fail(GMMVB_EHIP, "hipMemcpyAsync(pivot)", e);
    ws->xc_src = nullptr;      // the centred copy (if any) is stale now
    ws->xq_src = nullptr;      // ... and so are the digit planes
    ++ws->pivot_gen;
    if (ws->lock_live) {               // the settled rows belonged to the previous state of affairs
        ws->lock_live = false;
        ws->lock_reset = true;
    }
    return GMMVB_OK;
}

int gmmvb_wants_drift(const gmmvb_workspace* ws, int64_t n_rows) {
    if (!ws || ws->prune == 0 || ws->estep_variant != kEstepLds8 || ws->hmm != nullptr || !ws->rec_k) return 0;
    if (ws->opt_carry_off) return 0;
    if (ws->sharded) n_rows = ws->shard_rows / ws->shard_ranks;        // the same answer on every rank
    return (ws->prune == 2 || n_rows * (int64_t)ws->K >= (int64_t(1) << 23)) ? 1 : 0;
}

namespace {
// drift[0..K) = gamma, [K..2K) = delta, [2K..3K) = c of the parameters the records belong to, [3K..4K) = Gamma - one launch
// instead of four device-to-device copies (each a dispatch of its own on the iteration's critical path)
__global__ void set_drift_kernel(const double* __restrict__ gamma, const double* __restrict__ delta, const double* __restrict__ big_gamma,
                                 const double* __restrict__ cvec, int K, double* __restrict__ drift) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    drift[k] = gamma[k];
    drift[K + k] = delta[k];
    drift[2 * K + k] = cvec[k];
    drift[3 * K + k] = big_gamma[k];
}
}  // namespace

int gmmvb_set_drift(gmmvb_workspace* ws, const double* gamma_dev, const double* delta_dev, const double* big_gamma_dev,
                    double typical_gamma, void* stream) {
    if (!ws || !gamma_dev || !delta_dev || !big_gamma_dev) return fail(GMMVB_EINVAL, "null argument");
    ws->typical_gamma = typical_gamma;
    hipStream_t st = (hipStream_t)stream;
    // (with the constants of the parameters the records belong to: the next gmmvb_set_params overwrites cvec)
    hipLaunchKernelGGL(set_drift_kernel, dim3((unsigned)((ws->K + 255) / 256)), dim3(256), 0, st, gamma_dev, delta_dev, big_gamma_dev,
                       ws->cvec, ws->K, ws->drift);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(GMMVB_EHIP, "set_drift_kernel", e);
    ws->have_drift = ws->have_params && ws->params_used;     // else: not the parameters the records belong to
    return GMMVB_OK;
}

// test / diagnostic read-out of one row's candidate record (blocking): out[0..7] slot components (-1 empty), out[8..15]
// slot distances, out[16] B, out[17] exact bits, out[18] selected bits, out[19] flags, out[20] khat, out[21] lse,
// out[22..25] the row's mask words
int gmmvb_debug_record(gmmvb_workspace* ws, int64_t row, double* out /*[26] host*/) {
    if (!ws || !out || !ws->rec_k || row < 0 || row >= ws->npad) return fail(GMMVB_EINVAL, "bad argument");
    hipError_t e = hipDeviceSynchronize();
    auto get = [&e](void* dst, const void* src, size_t n) {
        if (e == hipSuccess) e = hipMemcpy(dst, src, n, hipMemcpyDeviceToHost);
    };
    for (int j = 0; j < kRecSlots; ++j) {
        unsigned short k = 0;
        float d = 0.0f;
        get(&k, ws->rec_k + (int64_t)j * ws->npad + row, sizeof(k));
        get(&d, ws->rec_d + (int64_t)j * ws->npad + row, sizeof(d));
        out[j] = k == kRecEmpty ? -1.0 : (double)k;
        out[8 + j] = d;
    }
    float B = 0.0f;
    unsigned char ex = 0, sel = 0, fl = 0;
    int kh = 0;
    get(&B, ws->rec_B + row, sizeof(B));
    get(&ex, ws->rec_exact + row, 1);
    get(&sel, ws->rec_sel + row, 1);
    get(&fl, ws->rec_flags + row, 1);
    get(&kh, ws->khat + row, sizeof(kh));
    get(out + 21, ws->lse + row, sizeof(double));
    out[16] = B;
    out[17] = ex;
    out[18] = sel;
    out[19] = fl;
    out[20] = kh;
    for (int w = 0; w < 4; ++w) {
        unsigned long long m = 0;
        if (w < (ws->K + 63) / 64) get(&m, ws->masks + (int64_t)w * ws->npad + row, sizeof(m));
        out[22 + w] = (double)m;
    }
    if (e != hipSuccess) return fail(GMMVB_EHIP, "reading a record back", e);
    return GMMVB_OK;
}

int64_t gmmvb_regroup_count(const gmmvb_workspace* ws) { return ws ? ws->sorts : -1; }

// test / diagnostic: the proof round's kernel over (row, k) for EVERY row of the prepared matrix
namespace {
__global__ void debug_all_rows_kernel(int* __restrict__ list, int* __restrict__ counts, int K, int k, int64_t n_rows) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_rows) list[i] = (int)i;
    if (i < K) counts[i] = (int)(i == k ? n_rows : 0);
}
}  // namespace

int gmmvb_debug_proof(gmmvb_workspace* ws, int k, int64_t n_rows, float* ub_dev, double* lb_dev, void* stream) {
    if (!ws || !ub_dev || !lb_dev || k < 0 || k >= ws->K) return fail(GMMVB_EINVAL, "bad argument");
    if (!ws->xq || !ws->img_i8b || ws->xq_src == nullptr || ws->xq_rows != n_rows || ws->xq_gen != ws->img_gen || !ws->have_params)
        return fail(GMMVB_ESTATE, "no digit planes for these rows: gmmvb_set_pivot, gmmvb_prepare_rows, gmmvb_set_params first");
    claim_scratch(ws);
    hipStream_t st = (hipStream_t)stream;
    ws->tmeta_valid = false;                                       // (the bound array is written behind the sweeps' back)
    const unsigned grid = (unsigned)((std::max<int64_t>(n_rows, ws->K) + 255) / 256);
    hipLaunchKernelGGL(debug_all_rows_kernel, dim3(grid), dim3(256), 0, st, ws->lists + (int64_t)k * ws->npad, ws->counts, ws->K, k,
                       n_rows);
    hipLaunchKernelGGL(gather_plan_kernel, dim3(1), dim3(64), 0, st, ws->counts, ws->K, estep_i8_pairs_per_chunk(), ws->plan);
    hipError_t e = launch_estep_i8_proof(ws->D, ws->num_cu, st, ws->xq, ws->xqe, ws->img_i8b, ws->cvec, ws->K, ws->lists, ws->npad,
                                         ws->counts, ws->plan, ws->ub32, ws->lnrho, ws->npad);
    if (e == hipSuccess) e = hipMemcpyAsync(ub_dev, ws->ub32 + (int64_t)k * ws->npad, (size_t)n_rows * sizeof(float),
                                            hipMemcpyDeviceToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(lb_dev, ws->lnrho + (int64_t)k * ws->npad, (size_t)n_rows * sizeof(double),
                                            hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) return fail(GMMVB_EHIP, "proof kernel (diagnostic)", e);
    // whatever the workspace held of an E-step is gone
    ws->e_state = 0;
    ws->lost_estep = false;
    ws->bounds_rows = 0;
    ws->rec_valid = ws->dense_valid = ws->rec_live = false;
    ws->active_lists = ws->blk_fresh = false;
    ws->lag.valid = false;
    if (ws->lock_live) {
        ws->lock_live = false;
        ws->lock_reset = true;
    }
    return GMMVB_OK;
}

int gmmvb_forget(gmmvb_workspace* ws) {
    if (!ws) return fail(GMMVB_EINVAL, "null argument");
    ws->forget = true;
    ws->have_drift = false;
    if (ws->lock_live) {               // the settled rows belonged to the previous state of affairs
        ws->lock_live = false;
        ws->lock_reset = true;
    }
    return GMMVB_OK;
}

int gmmvb_set_params(gmmvb_workspace* ws, const double* c_dev, const double* m_dev, const double* u_dev,
                     void* stream) {
    if (!ws || !c_dev || !m_dev || !u_dev) return fail(GMMVB_EINVAL, "null argument");
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipSuccess;
    if (ws->generic) {
        e = hipMemcpyAsync(ws->cvec, c_dev, (size_t)ws->K * sizeof(double), hipMemcpyDeviceToDevice, st);
        if (e != hipSuccess) return fail(GMMVB_EHIP, "hipMemcpyAsync(c)", e);
        e = hipMemcpyAsync(ws->gen_m, m_dev, (size_t)ws->K * ws->D * sizeof(double), hipMemcpyDeviceToDevice, st);
        if (e == hipSuccess)
            e = hipMemcpyAsync(ws->gen_u, u_dev, (size_t)ws->K * ws->D * ws->D * sizeof(double), hipMemcpyDeviceToDevice, st);
        if (e != hipSuccess) return fail(GMMVB_EHIP, "hipMemcpyAsync(m, u)", e);
        ws->have_params = true;
        ws->params_used = false;
        return GMMVB_OK;
    }
    // (c and the int8 images' pivot ride along: the digits are taken about the pivot in force now; the int8 kernels read that
    // copy, not ws->pivot)
    hipLaunchKernelGGL(pack_params_kernel, dim3(ws->K), dim3(256), 0, st, u_dev, m_dev, ws->K, ws->D, ws->T,
                       ws->img_len, ws->img, c_dev, ws->cvec, ws->pivot_i8 ? ws->pivot : nullptr, ws->pivot_i8);
    e = hipGetLastError();
    if (e != hipSuccess) return fail(GMMVB_EHIP, "pack_params_kernel", e);
    if (ws->tri) {
        e = launch_pack_tri16(u_dev, m_dev, ws->K, ws->D, ws->tri, st);
        if (e != hipSuccess) return fail(GMMVB_EHIP, "pack_tri16_kernel", e);
    }
    if (ws->pivot_i8) {
        if (ws->img_i8) e = launch_pack_i8(u_dev, m_dev, ws->pivot_i8, ws->K, ws->D, ws->img_i8, 0, st);
        if (e == hipSuccess && ws->img_i8b) e = launch_pack_i8(u_dev, m_dev, ws->pivot_i8, ws->K, ws->D, ws->img_i8b, 1, st);
        if (e != hipSuccess) return fail(GMMVB_EHIP, "pack_params_i8_kernel", e);
        ws->img_gen = ws->pivot_gen;
    }
    // The stateless sweep's table for these parameters (project.h), when a sweep can follow: regrouped rows, their tiles'
    // references, digit planes about the pivot in force, and a drift hint (the settled rows' own bound is still carried).
    ws->proj_table = false;
    if (ws->gimg && ws->opt_project != 0 && ws->sorted && ws->tile_ref_valid && ws->have_drift && ws->xq_gen == ws->pivot_gen &&
        !ws->opt_carry_off) {
        e = launch_proj_table(u_dev, m_dev, c_dev, ws->pivot, ws->K, ws->D, ws->gimg, ws->gconst, st);
        if (e != hipSuccess) return fail(GMMVB_EHIP, "proj_table_kernel", e);
        ws->proj_table = true;
    }
    ws->have_params = true;
    ws->params_used = false;
    return GMMVB_OK;
}

// sample lists of the pruned E-step and the sparse M-step, candidate records, gather plan, per-block counters
int ensure_lists(gmmvb_workspace* ws) {
    if (ws->lists) return GMMVB_OK;
    const int64_t sel_blocks = (ws->npad + kSelRows - 1) / kSelRows, words = (ws->K + 63) / 64;
    const int64_t np = ws->npad;
    hipError_t e = hipSuccess;
    const bool own_lists = ws->scratch->lists == nullptr;
    if (own_lists) {                   // one set for the tile group, sized for its first (largest) workspace
        e = hipMalloc((void**)&ws->scratch->lists, (size_t)ws->K * ws->scratch->npad * sizeof(int));
        if (e != hipSuccess) ws->scratch->lists = nullptr;
    }
    if (e == hipSuccess) ws->lists = ws->scratch->lists;
    if (e == hipSuccess) e = hipMalloc((void**)&ws->khat, (size_t)np * sizeof(int));
    if (e == hipSuccess) e = hipMalloc((void**)&ws->counts, (size_t)ws->K * sizeof(int));
    if (e == hipSuccess) e = hipMalloc((void**)&ws->plan, (size_t)(ws->K + 1) * sizeof(int));
    if (e == hipSuccess) e = hipMalloc((void**)&ws->plan_m, (size_t)(ws->K + 2) * sizeof(int));
    if (e == hipSuccess) e = hipMalloc((void**)&ws->blk, (size_t)sel_blocks * ws->K * sizeof(int));
    if (e == hipSuccess) e = hipMalloc((void**)&ws->scan_parts, (size_t)ws->K * kScanParts * sizeof(int));
    if (e == hipSuccess) e = hipMalloc((void**)&ws->masks, (size_t)words * np * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMalloc((void**)&ws->lock, (size_t)np);
    if (e == hipSuccess) e = hipMemset(ws->lock, 0, (size_t)np);
    if (e == hipSuccess) e = hipMalloc((void**)&ws->lcomp, (size_t)np);
    if (e == hipSuccess) e = hipMalloc((void**)&ws->dlock, (size_t)np * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void**)&ws->rthr, (size_t)np * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void**)&ws->exit_ctr, 4 * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMemset(ws->exit_ctr, 0, 4 * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipHostMalloc((void**)&ws->exit_host, 4 * sizeof(unsigned long long), hipHostMallocDefault);
    if (e == hipSuccess) ws->exit_host[0] = ws->exit_host[1] = ws->exit_host[2] = ws->exit_host[3] = 0;
    if (e == hipSuccess) e = hipMalloc((void**)&ws->dmask, (size_t)words * np * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMalloc((void**)&ws->dblk, (size_t)sel_blocks * ws->K * sizeof(int));
    if (e == hipSuccess) e = hipMalloc((void**)&ws->mmask, (size_t)words * np * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMalloc((void**)&ws->rmask, (size_t)words * np * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMalloc((void**)&ws->rblk, (size_t)sel_blocks * ws->K * sizeof(int));
    if (e == hipSuccess) e = hipMalloc((void**)&ws->mblk, (size_t)sel_blocks * ws->K * sizeof(int));
    if (e == hipSuccess) e = hipMalloc((void**)&ws->cache, (size_t)gmmvb_stats_len(ws->K, ws->D) * sizeof(double));
    if (e == hipSuccess) e = hipMemset(ws->cache, 0, (size_t)gmmvb_stats_len(ws->K, ws->D) * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void**)&ws->spart, (size_t)sel_blocks * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void**)&ws->gpart, (size_t)sel_blocks * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void**)&ws->qpart, (size_t)sel_blocks * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void**)&ws->epart, (size_t)sel_blocks * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void**)&ws->opart, (size_t)sel_blocks * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void**)&ws->mpart, (size_t)sel_blocks * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void**)&ws->ppart, (size_t)sel_blocks * sizeof(double));
    if (e == hipSuccess) e = hipMemset(ws->ppart, 0, (size_t)sel_blocks * sizeof(double));
    if (ws->prune != 0 && ws->img_i8b && ws->opt_proof && ws->cache_on) {
        const int64_t rb = estep_i8_digit_row_bytes(ws->D);
        if (e == hipSuccess) e = hipMalloc((void**)&ws->xq, (size_t)(np * rb));
        if (e == hipSuccess) e = hipMalloc((void**)&ws->xqe, (size_t)np);
        if (e == hipSuccess) ws->bytes += np * (rb + 1);
        // the stateless sweep's table (project.h): needs the digit planes, regrouped rows and at least two feature blocks
        if (ws->opt_project != 0 && ws->sort_rows && ws->K <= kSelRows && ws->D > 32 && ws->D <= 128) {
            const int64_t ib = proj_image_len(ws->K, ws->D), cl = proj_const_len(ws->K);
            if (e == hipSuccess) e = hipMalloc((void**)&ws->gimg, (size_t)ib);
            if (e == hipSuccess) e = hipMemset(ws->gimg, 0, (size_t)ib);
            if (e == hipSuccess) e = hipMalloc(&ws->gconst, (size_t)cl * 16);
            if (e == hipSuccess) e = hipMalloc((void**)&ws->tile_ref, (size_t)sel_blocks * sizeof(int));
            if (e == hipSuccess) ws->bytes += ib + cl * 16 + sel_blocks * 4;
        }
    }
    if (e == hipSuccess) e = hipMalloc((void**)&ws->rec_k, (size_t)kRecSlots * np * sizeof(unsigned short));
    if (e == hipSuccess) e = hipMalloc((void**)&ws->rec_d, (size_t)kRecSlots * np * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void**)&ws->rec_B, (size_t)np * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void**)&ws->rec_exact, (size_t)np);
    if (e == hipSuccess) e = hipMalloc((void**)&ws->rec_sel, (size_t)np);
    if (e == hipSuccess) e = hipMalloc((void**)&ws->rec_flags, (size_t)np);
    if (e == hipSuccess) e = hipMalloc((void**)&ws->ub32, (size_t)ws->K * np * sizeof(float));
    if (e == hipSuccess) ws->bytes += (int64_t)ws->K * np * (int64_t)sizeof(float);
    if (ws->opt_lazy && ws->K <= kSelRows) {       // the lazy sweep's state per tile of kSelRows rows and component
        if (e == hipSuccess) e = hipMalloc((void**)&ws->tmeta, (size_t)sel_blocks * ws->K * sizeof(float4));
        if (e == hipSuccess) ws->bytes += sel_blocks * ws->K * (int64_t)sizeof(float4);
    }
    const size_t esz = ws->x_dtype == GMMVB_F64 ? 8 : 4;
    if (ws->sort_rows) {
        if (e == hipSuccess) e = hipMalloc(&ws->xp, (size_t)ws->max_rows * ws->D * esz);
        if (e == hipSuccess) e = hipMalloc((void**)&ws->perm, (size_t)np * sizeof(int));
        if (e == hipSuccess) e = hipMalloc((void**)&ws->iperm, (size_t)np * sizeof(int));
        if (e == hipSuccess) e = hipMalloc((void**)&ws->perm_tmp, (size_t)np * sizeof(int));
        if (e == hipSuccess) ws->bytes += (int64_t)ws->max_rows * ws->D * (int64_t)esz + 3 * np * (int64_t)sizeof(int);
    }
    if (e != hipSuccess) return fail(GMMVB_ENOMEM, "hipMalloc (sample lists / records)", e);
    // lists, khat, counts, plans, four sets of block counts and their scan parts; four sets of masks; lock / lcomp / dlock /
    // rthr; the cache; eight per-block counters; the records
    ws->bytes += ((own_lists ? (int64_t)ws->K * ws->scratch->npad : 0) + np + 4 * ws->K + 3 + 4 * sel_blocks * ws->K + (int64_t)ws->K * kScanParts) * (int64_t)sizeof(int) +
                 4 * words * np * 8 + np * (1 + 1 + 4 + 4) + gmmvb_stats_len(ws->K, ws->D) * 8 + 8 * sel_blocks * 8 +
                 np * (kRecSlots * 6 + 4 + 3);
    return GMMVB_OK;
}

// Counters of the last E-step, blocking: waits for the copy that the E-step enqueued (gmmvb_last_sparsity, and the
// M-step right after a dense E-step, where the host has been waiting for the dense kernel anyway).
int fetch_counters(gmmvb_workspace* ws) {
    if (ws->ctr_pending) {
        hipError_t e = hipEventSynchronize(ws->ctr_ev);
        if (e != hipSuccess) return fail(GMMVB_EHIP, "waiting for the E-step counters", e);
        ws->lag.act = ws->ctr_host[0];
        if (ws->pend_mode == 0) {              // dense pass: every pair evaluated, no records involved
            ws->lag.eval = (double)ws->pend_rows * ws->K;
            ws->lag.over = 0.0;
            ws->lag.settled = 0.0;
            ws->lag.listed = ws->lag.accum = ws->lag.act;
            ws->lag.exits = 0.0;
            ws->lag.cols = -1.0;
            ws->lag.left = -1.0;
            ws->lag.proof = 0.0;
            ws->lag.moved = 0.0;
        } else {
            ws->lag.proof = ws->ctr_host[7];
            ws->lag.exits = (ws->exit_host && ws->gather_exit) ? (double)ws->exit_host[0] : 0.0;
            ws->lag.cols = (ws->exit_host && ws->pend_lazy) ? (double)ws->exit_host[1] : -1.0;
            ws->lag.left = (ws->exit_host && ws->pend_proj) ? (double)ws->exit_host[2] : -1.0;
            ws->lag.settled = ws->ctr_host[4];
            ws->lag.listed = ws->ctr_host[5];
            ws->lag.accum = ws->ctr_host[6];                               // a bound pass / sweep also evaluated every row's (previous) best component
            ws->lag.eval = ws->ctr_host[1] + ((ws->pend_mode == 1 || ws->pend_mode == 3) ? ws->pend_round0 : 0.0);
            ws->lag.over = ws->ctr_host[2];
            // rows whose best component changed: after a regrouping they no longer sit with their component's rows.
            // (the first pass after a regrouping compares with the bound kernel's guess, not with a previous best)
            ws->lag.moved = ws->ctr_host[3];
            if (!ws->sharded && ws->sorted && !ws->pend_first_sorted) ws->moved_since_sort += ws->lag.moved;
        }
        ws->lag.rows = (double)ws->pend_rows;
        ws->lag.mode = ws->pend_mode;
        ws->lag.valid = true;
        ws->ctr_pending = false;
    }
    return GMMVB_OK;
}

// The same without waiting: takes the counters over if their copy has completed (the driver synchronises once per
// VB iteration, so by the next E-step it always has).
void poll_counters(gmmvb_workspace* ws) {
    if (ws->ctr_pending && hipEventQuery(ws->ctr_ev) == hipSuccess) (void)fetch_counters(ws);
}

// ---- row-sharded jobs: one policy for all ranks ------------------------------------------------------------------------
// The choice of pass (dense / bound / sweep / records, regrouping, bound level, settling) is driven by counters of the
// previous pass.  With the rows sharded over ranks each rank would see its own counters and the ranks would drift apart:
// a 40-ms bound pass on one rank while the others sweep in 8 ms stalls everybody at the iteration's all-reduce.  So the
// counters travel with the statistics block: gmmvb_policy_export writes them (GMMVB_POLICY_LEN doubles) where the caller's
// all-reduce picks them up, gmmvb_policy_import hands the sums back, and a sharded workspace (gmmvb_set_shard) decides
// from those sums only - every rank the same way.
namespace {
__global__ void policy_export_kernel(const double* __restrict__ ctr, const unsigned long long* __restrict__ exits, int mode,
                                     double rows, int K, double round0, int counted, int use_exits, double* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const bool dense = mode == 0;
    const double act = counted ? ctr[0] : 0.0;
    out[0] = act;
    out[1] = dense ? rows * K : ctr[1] + ((mode == 1 || mode == 3) ? round0 : 0.0);
    out[2] = dense ? 0.0 : ctr[2];
    out[3] = dense ? 0.0 : ctr[3];
    out[4] = dense ? 0.0 : ctr[4];
    out[5] = dense ? act : ctr[5];
    out[6] = dense ? act : ctr[6];
    out[7] = dense ? 0.0 : ctr[7];
    out[8] = (dense || !use_exits || !exits) ? 0.0 : (double)*exits;
    out[9] = rows;
    out[10] = 1.0;                      // ranks
    out[11] = counted ? 1.0 : 0.0;      // ranks whose pass counted its pairs
    out[12] = (double)mode;             // kind of the pass these counters describe (the same on every rank: sum / ranks)
    for (int i = 13; i < GMMVB_POLICY_LEN; ++i) out[i] = 0.0;
}
}  // namespace

int gmmvb_set_shard(gmmvb_workspace* ws, int64_t global_rows, int n_ranks) {
    if (!ws || global_rows < 1 || n_ranks < 1) return fail(GMMVB_EINVAL, "bad argument");
    ws->sharded = n_ranks > 1;
    ws->shard_rows = global_rows;
    ws->shard_ranks = n_ranks;
    ws->pol.valid = false;
    if (ws->sharded && !ws->pol_host) {
        hipError_t e = hipHostMalloc((void**)&ws->pol_host, GMMVB_POLICY_LEN * sizeof(double), hipHostMallocDefault);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&ws->pol_ev, hipEventDisableTiming);
        if (e != hipSuccess) return fail(GMMVB_EHIP, "policy buffers", e);
    }
    return GMMVB_OK;
}

int gmmvb_policy_export(gmmvb_workspace* ws, double* out_dev, void* stream) {
    if (!ws || !out_dev) return fail(GMMVB_EINVAL, "null argument");
    if (ws->e_state != 1) return fail(GMMVB_ESTATE, "no E-step output in the workspace");
    hipLaunchKernelGGL(policy_export_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, ws->ctr, ws->exit_ctr, ws->prev_pass,
                       (double)ws->e_rows, ws->K, ws->pend_round0, ws->exp_counted ? 1 : 0, ws->gather_exit ? 1 : 0, out_dev);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(GMMVB_EHIP, "policy export", e);
    return GMMVB_OK;
}

int gmmvb_policy_import(gmmvb_workspace* ws, const double* in_dev, void* stream) {
    if (!ws || !in_dev) return fail(GMMVB_EINVAL, "null argument");
    if (!ws->sharded) return GMMVB_OK;          // a single process decides from its own counters
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemcpyAsync(ws->pol_host, in_dev, GMMVB_POLICY_LEN * sizeof(double), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipEventRecord(ws->pol_ev, st);
    if (e != hipSuccess) return fail(GMMVB_EHIP, "policy import", e);
    ws->pol_pending = true;
    ws->pol_mode = ws->prev_pass;
    ws->pol_first_sorted = ws->pend_first_sorted;
    return GMMVB_OK;
}

// The imported counters, waited for: the copy was enqueued before the caller's per-iteration host sync, so this returns
// at once - and every rank must see them (a rank that decided without them would part ways with the others).
int take_policy(gmmvb_workspace* ws) {
    if (!ws->pol_pending) return GMMVB_OK;
    hipError_t e = hipEventSynchronize(ws->pol_ev);
    if (e != hipSuccess) return fail(GMMVB_EHIP, "waiting for the imported counters", e);
    const double* h = ws->pol_host;
    gmmvb_pass_counters& p = ws->pol;
    p.act = h[0];
    p.eval = h[1];
    p.over = h[2];
    p.moved = h[3];
    p.settled = h[4];
    p.listed = h[5];
    p.accum = h[6];
    p.proof = h[7];
    p.exits = h[8];
    p.rows = h[9];
    p.ranks = h[10];
    // the kind of the pass the counters were taken in travels with them: a row-tiled pass imports the same job-wide tail
    // before every tile, when this workspace's last pass is already the previous TILE of the current iteration
    p.mode = h[10] >= 1.0 ? (int)(h[12] / h[10] + 0.5) : ws->pol_mode;
    p.valid = h[10] >= 1.0 && h[11] == h[10];
    if (p.valid && ws->sorted && !ws->pol_first_sorted) ws->moved_since_sort += p.moved;
    ws->pol_pending = false;
    return GMMVB_OK;
}

int gmmvb_last_sparsity(gmmvb_workspace* ws, void* stream, double* active_pairs, double* evaluated_pairs) {
    (void)stream;
    if (!ws || !active_pairs || !evaluated_pairs) return fail(GMMVB_EINVAL, "null argument");
    if (ws->e_state != 1 && !ws->lost_estep) return fail(GMMVB_ESTATE, "no E-step output in the workspace");
    int rc = fetch_counters(ws);
    if (rc) return rc;
    *evaluated_pairs = ws->lag.mode == 0 ? (double)ws->e_rows * ws->K : ws->lag.eval;
    *active_pairs = (ws->sparse && ws->act_rows == ws->e_rows) ? ws->lag.act : -1.0;   // GMMVB_MSTEP_SPARSE=0: not counted
    return GMMVB_OK;
}

int gmmvb_last_work(gmmvb_workspace* ws, double* out) {
    if (!ws || !out) return fail(GMMVB_EINVAL, "null argument");
    if (ws->e_state != 1 && !ws->lost_estep) return fail(GMMVB_ESTATE, "no E-step output in the workspace");
    int rc = fetch_counters(ws);
    if (rc) return rc;
    const bool counted = ws->sparse && ws->act_rows == ws->e_rows;
    out[0] = counted ? ws->lag.act : -1.0;
    out[1] = ws->lag.mode == 0 ? (double)ws->e_rows * ws->K : ws->lag.eval;
    out[2] = counted ? ws->lag.accum : -1.0;
    out[3] = ws->lag.mode == 0 ? 0.0 : ws->lag.settled;
    out[4] = ws->lag.mode == 0 ? 0.0 : ws->lag.exits;
    out[5] = ws->lag.mode == 0 ? 0.0 : ws->lag.proof;
    out[6] = ws->lag.mode == 3 ? ws->lag.cols : -1.0;
    out[7] = ws->lag.mode == 3 ? ws->lag.left : -1.0;
    return GMMVB_OK;
}

int gmmvb_policy_calibrate(gmmvb_workspace* ws, int on) {
    if (!ws) return fail(GMMVB_EINVAL, "null argument");
    ws->opt_calibrate = on != 0;
    if (!on) {                        // back to the scaled literals
        ws->pt.init(ws->T, estep_bound_blocks(ws->T) ? (ws->D + 31) / 32 : 0);
        for (double& p : ws->cal_pairs) p = 0.0;
    }
    return GMMVB_OK;
}

int gmmvb_policy_table(gmmvb_workspace* ws, double* out /*[GMMVB_POLICY_TABLE_LEN], host*/) {
    if (!ws || !out) return fail(GMMVB_EINVAL, "null argument");
    cal_poll(ws);
    const gmmvb::PolicyTable& t = ws->pt;
    const double v[GMMVB_POLICY_TABLE_LEN] = {t.dense_e_ns, t.dense_m_ns, t.bound_ns, t.exact_ns, t.proof_ns, t.list_m_ns,
                                              t.lit_dense_e, t.lit_dense_m, t.lit_bound, t.prune_below(), t.dense_again_above(),
                                              t.list_m_below(), (double)(t.measured & 7), (double)((t.measured >> 3) & 7),
                                              ws->opt_calibrate ? 1.0 : 0.0, 0.0};
    for (int i = 0; i < GMMVB_POLICY_TABLE_LEN; ++i) out[i] = v[i];
    return GMMVB_OK;
}

int check_x(const gmmvb_workspace* ws, const void* x_dev, int64_t ldx, int64_t n_rows, bool* vec) {
    if (!ws || !x_dev) return fail(GMMVB_EINVAL, "null argument");
    if (n_rows < 1 || n_rows > ws->max_rows) return fail(GMMVB_EINVAL, "n_rows must be in [1, max_rows]");
    if (ldx < ws->D) return fail(GMMVB_EINVAL, "ldx must be >= D");
    const int64_t esz = ws->x_dtype == GMMVB_F64 ? 8 : 4;
    // vector loads: whole 16-feature blocks, 4-element (E) and T-element (M) vectors naturally aligned
    const int64_t valign = esz * (ws->T > 4 ? ws->T : 4);
    *vec = (ws->D % 16 == 0) && ((uintptr_t)x_dev % valign == 0) && ((ldx * esz) % valign == 0);
    return GMMVB_OK;
}

int gmmvb_prepare_rows(gmmvb_workspace* ws, const void* x_dev, int64_t ldx, int64_t n_rows, void* stream) {
    bool vec = false;
    int rc = check_x(ws, x_dev, ldx, n_rows, &vec);
    if (rc) return rc;
    claim_scratch(ws);                 // (the centred copy is written below)
    ws->bounds_rows = 0;               // (new) sample matrix: nothing of an earlier E-step may be carried over
    if (ws->lock_live) {               // the settled rows belonged to the previous state of affairs
        ws->lock_live = false;
        ws->lock_reset = true;
    }
    ws->sorted = false;                // ... and the internal row order is the caller's again
    ws->tile_ref_valid = false;
    ws->rec_valid = false;
    ws->dense_valid = false;
    ws->lag.valid = false;
    if (!ws->xc) return GMMVB_OK;      // disabled: the M-step reads x directly
    const int Dp = 16 * ws->T;
    const int64_t pad_rows = round_up(n_rows, 64) + 64;
    const int64_t total = pad_rows * Dp;
    const unsigned grid = (unsigned)((total + 255) / 256);
    hipStream_t st = (hipStream_t)stream;
    if (ws->x_dtype == GMMVB_F64)
        hipLaunchKernelGGL(center_rows_kernel<double>, dim3(grid), dim3(256), 0, st, (const double*)x_dev, ldx, n_rows,
                           pad_rows, ws->D, Dp, ws->pivot, ws->xc);
    else
        hipLaunchKernelGGL(center_rows_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)x_dev, ldx, n_rows,
                           pad_rows, ws->D, Dp, ws->pivot, ws->xc);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(GMMVB_EHIP, "center_rows launch", e);
    ws->xc_src = x_dev;
    ws->xc_rows = n_rows;
    ws->xc_ldx = ldx;
    ws->xc_stale = false;
    if (ws->xq) {              // the int8 digit planes of the proof round, about the same pivot
        e = launch_x_digits(x_dev, ws->x_dtype == GMMVB_F64, ldx, n_rows, ws->D, ws->pivot, ws->xq, ws->xqe, st);
        if (e != hipSuccess) return fail(GMMVB_EHIP, "x_digits launch", e);
        ws->xq_src = x_dev;
        ws->xq_rows = n_rows;
        ws->xq_ldx = ldx;
        ws->xq_gen = ws->pivot_gen;
    }
    return GMMVB_OK;
}

// The centred copy in the workspace's internal row order (after a regrouping; see regroup_rows)
hipError_t recenter_rows(gmmvb_workspace* ws, int64_t n_rows, hipStream_t st) {
    const int Dp = 16 * ws->T;
    const int64_t pad_rows = round_up(n_rows, 64) + 64;
    const unsigned cg = (unsigned)((pad_rows * Dp + 255) / 256);
    // regrouped rows: the workspace's permuted copy; else (the copy was lost to another tile of the group) the caller's matrix
    const void* src = ws->sorted ? ws->xp : ws->xc_src;
    const int64_t ld = ws->sorted ? (int64_t)ws->D : ws->xc_ldx;
    if (ws->x_dtype == GMMVB_F64)
        hipLaunchKernelGGL(center_rows_kernel<double>, dim3(cg), dim3(256), 0, st, (const double*)src, ld, n_rows,
                           pad_rows, ws->D, Dp, ws->pivot, ws->xc);
    else
        hipLaunchKernelGGL(center_rows_kernel<float>, dim3(cg), dim3(256), 0, st, (const float*)src, ld, n_rows,
                           pad_rows, ws->D, Dp, ws->pivot, ws->xc);
    ws->xc_stale = false;
    return hipGetLastError();
}


}  // extern "C"
