// C ABI of the GMM-VB data-pass engine (see include/gmmvb.h for the contract and the reference
// call sites each entry point replaces).
#include "workspace.h"

#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>

#include "aux_kernels.h"
#include "generic.h"
#include "launch.h"
#include "records.h"

using namespace gmmvb;

namespace {
thread_local std::string g_err;
}  // namespace

#ifndef GMMVB_T1_SPLITS
#define GMMVB_T1_SPLITS 24
#endif
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

static int ensure_lists(gmmvb_workspace* ws);
static void cal_poll(gmmvb_workspace* ws);

// profiling spans (gmmvb_profile_spans): HIP events on the launch stream around groups of kernels
enum { kSpanEstepMain = 0, kSpanSelect = 1, kSpanGather = 2, kSpanLse = 3, kSpanLists = 4, kSpanMstepMain = 5,
       kSpanReduce = 6, kSpanProof = 7, kSpanSlots = 8 };
static const char* const kSpanNames[kSpanSlots] = {"estep_main", "estep_select", "estep_gather", "estep_lse_mask",
                                                   "mstep_lists", "mstep_main", "mstep_reduce", "estep_proof"};
// A failed event record / counter reset inside a pass must not vanish: the first such error is kept in the workspace and
// gmmvb_estep / gmmvb_mstep return it (GMMVB_EHIP) before they hand anything to the caller.
static void note_hip(gmmvb_workspace* ws, hipError_t e) {
    if (e != hipSuccess && ws->hip_err == hipSuccess) ws->hip_err = e;
}
static int take_hip(gmmvb_workspace* ws, const char* what) {
    if (ws->hip_err == hipSuccess) return GMMVB_OK;
    const hipError_t e = ws->hip_err;
    ws->hip_err = hipSuccess;
    return fail(GMMVB_EHIP, what, e);
}
// An event record costs the stream about 10 us (the queue drains around the marker packet): at the benchmark shape a converged
// step has ~36 of them, 0.17 ms of a 3.7 ms step.  Profile level 2 keeps only the spans of the three groups that can dominate a
// step (the two E-step evaluation groups and the M-step's accumulation) and drops the phase events.
static bool span_kept(const gmmvb_workspace* ws, int slot) {
    return !ws->prof_light || slot == kSpanEstepMain || slot == kSpanGather || slot == kSpanMstepMain;
}
static void span_begin(gmmvb_workspace* ws, int slot, hipStream_t st) {
    ws->span_open = false;
    if (!ws->prof || ws->n_spans >= gmmvb_workspace::kMaxSpans || !span_kept(ws, slot)) return;
    ws->span_slot[ws->n_spans] = slot;
    ws->span_open = true;
    note_hip(ws, hipEventRecord(ws->span_ev[2 * ws->n_spans], st));
}
static void span_end(gmmvb_workspace* ws, hipStream_t st) {
    if (!ws->span_open) return;
    ws->span_open = false;
    note_hip(ws, hipEventRecord(ws->span_ev[2 * ws->n_spans + 1], st));
    ++ws->n_spans;
}
static bool phase_events(const gmmvb_workspace* ws) { return ws->prof && !ws->prof_light; }

// ---- the scratch of a tile group (workspace.h: gmmvb_scratch) -----------------------------------------------------------
// `w` loses the buffers to another workspace of its group: its E-step output, lists and centred copy are gone.  What it
// carries into its next E-step (bounds, records, settled rows, digit planes, row order, policy counters) is untouched; that
// E-step starts its first round from the rows' best components instead of the previous pass's lists.
static void yield_scratch(gmmvb_workspace* w) {
    // its counters, masks and block counts (per tile) still describe that pass: gmmvb_last_sparsity / gmmvb_last_work answer,
    // and the next sweep rebuilds its first round's lists from them (blk_fresh stays as it is)
    w->lost_estep = w->e_state == 1;
    w->e_state = 0;
    w->active_lists = false;
    if (w->mlists_done) w->mlists_lost = true;
    w->mlists_done = false;
    w->rec_live = false;
    w->settled_fresh = false;
    if (w->xc_src) w->xc_stale = true;         // rebuilt from the rows when a kernel needs it (restore_xc)
}
static void claim_scratch(gmmvb_workspace* ws) {
    gmmvb_scratch* s = ws->scratch;
    if (!s || s->owner == ws) return;
    if (s->owner) yield_scratch(s->owner);
    s->owner = ws;
}
static void release_scratch(gmmvb_workspace* ws) {
    gmmvb_scratch* s = ws->scratch;
    if (!s) return;
    if (s->owner == ws) s->owner = nullptr;
    if (--s->refs <= 0) {          // (a creation that failed half-way has not handed its buffers over yet)
        double* d[] = {s->lnrho ? s->lnrho : ws->lnrho, s->xc ? s->xc : ws->xc, s->slabs ? s->slabs : ws->slabs};
        for (double* p : d)
            if (p) (void)hipFree(p);
        if (s->lists) (void)hipFree(s->lists);
        delete s;
    }
    ws->scratch = nullptr;
    ws->lnrho = ws->xc = ws->slabs = nullptr;
    ws->lists = nullptr;
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
                     ws->rmask, ws->rblk, ws->xq, ws->xqe, ws->ppart, ws->tmeta, ws->gimg, ws->gconst, ws->hk, ws->tile_ref, ws->xqn};
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
        e = launch_proj_table(u_dev, m_dev, c_dev, ws->pivot, ws->K, ws->D, ws->hk, ws->gimg, ws->gconst, st);
        if (e != hipSuccess) return fail(GMMVB_EHIP, "proj_table_kernel", e);
        ws->proj_table = true;
    }
    ws->have_params = true;
    ws->params_used = false;
    return GMMVB_OK;
}

// sample lists of the pruned E-step and the sparse M-step, candidate records, gather plan, per-block counters
static int ensure_lists(gmmvb_workspace* ws) {
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
            if (e == hipSuccess) e = hipMalloc((void**)&ws->hk, (size_t)ws->K * sizeof(float));
            if (e == hipSuccess) e = hipMalloc((void**)&ws->tile_ref, (size_t)sel_blocks * sizeof(int));
            if (e == hipSuccess) e = hipMalloc((void**)&ws->xqn, (size_t)np * sizeof(float));
            if (e == hipSuccess) ws->bytes += ib + cl * 16 + ws->K * 4 + sel_blocks * 4 + np * 4;
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
static int fetch_counters(gmmvb_workspace* ws) {
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
static void poll_counters(gmmvb_workspace* ws) {
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
static int take_policy(gmmvb_workspace* ws) {
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

static int check_x(const gmmvb_workspace* ws, const void* x_dev, int64_t ldx, int64_t n_rows, bool* vec) {
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
        e = launch_x_digits(x_dev, ws->x_dtype == GMMVB_F64, ldx, n_rows, ws->D, ws->pivot, ws->xq, ws->xqe, st, ws->xqn);
        if (e != hipSuccess) return fail(GMMVB_EHIP, "x_digits launch", e);
        ws->xq_src = x_dev;
        ws->xq_rows = n_rows;
        ws->xq_ldx = ldx;
        ws->xq_gen = ws->pivot_gen;
    }
    return GMMVB_OK;
}

// The centred copy in the workspace's internal row order (after a regrouping; see regroup_rows)
static hipError_t recenter_rows(gmmvb_workspace* ws, int64_t n_rows, hipStream_t st) {
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
        hipError_t eq = launch_x_digits(ws->xp, ws->x_dtype == GMMVB_F64, ws->D, n_rows, ws->D, ws->pivot, ws->xq, ws->xqe, st, ws->xqn);
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
                                   const float* thr = nullptr) {
    if (thr) {
        hipError_t em = hipMemsetAsync(ws->exit_ctr, 0, sizeof(unsigned long long), st);
        if (em != hipSuccess) return em;
    }
    span_begin(ws, kSpanSelect, st);
    launch_scan_counts(st, ws->blk, sel_grid, ws->K, ws->counts, ws->scan_parts);
    hipLaunchKernelGGL(fill_lists_kernel, dim3(sel_grid), dim3(kSelRows), 0, st, ws->masks, ws->npad, a.n_rows, ws->K, ws->blk,
                       ws->lists, ws->npad);
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
static bool cal_wanted(gmmvb_workspace* ws, int what, double pairs) {
    return ws->opt_calibrate && ws->cal_ev[0] != nullptr && !((ws->pt.measured >> what) & 1) && ws->cal_pairs[what] == 0.0 &&
           pairs >= (double)(int64_t(1) << 23) && ws->hmm == nullptr;
}
static void cal_mark(gmmvb_workspace* ws, int what, double pairs, hipStream_t st) {
    note_hip(ws, hipEventRecord(ws->cal_ev[2 * what + 1], st));
    ws->cal_pairs[what] = pairs;
}
static void cal_poll(gmmvb_workspace* ws) {
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
                             ws->xq != nullptr && ws->xqn != nullptr && ws->xq_src == x_dev && ws->xq_rows == n_rows &&
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
                                   ws->blk, ws->lists, ws->npad);
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
                    hipLaunchKernelGGL(fill_lists_kernel, dim3(sel_grid), dim3(kSelRows), 0, st, ws->masks, ws->npad, n_rows, ws->K,
                                       ws->blk, ws->lists, ws->npad);
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
                                       ws->K, ws->rblk, ws->lists, ws->npad);
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
                    ProjectArgs pa{ws->xq, ws->xqe, ws->xqn, ws->gimg, ws->gconst, ws->tile_ref, ws->lnrho, ws->npad, n_rows, ws->K,
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
                       tmeta_was_valid ? 0 : 1, ws->exit_ctr + 1)
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
                                       ws->rblk, ws->opt_proof_all ? 1 : 0, own_round ? 1 : 0, nullptr, 0);
                }
                if (proof && can_project && !projected) {
                    // the table of the parameters in force first (project.h): a listed pair it clears needs no proof - most of
                    // them are far pairs whose carried bound has eroded to the relevance line
                    ProjectArgs pa{ws->xq, ws->xqe, ws->xqn, ws->gimg, ws->gconst, ws->tile_ref, ws->lnrho, ws->npad, n_rows, ws->K,
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
                    hipLaunchKernelGGL(fill_lists_kernel, dim3(sel_grid), dim3(kSelRows), 0, st, ws->rmask, ws->npad, n_rows,
                                       ws->K, ws->rblk, ws->lists, ws->npad);
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
        e = lists_and_gather(ws, a, is64, vec, sel_grid, st, ws->gather_exit ? ws->rthr : nullptr);
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
            hipLaunchKernelGGL(fill_lists_kernel, dim3(nblk), dim3(kSelRows), 0, st, ws->dmask, ws->npad, n_rows, ws->K,
                               ws->dblk, ws->lists, ws->npad, ws->lock, ws->lcomp);
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
                                   ws->mblk, ws->lists, ws->npad);
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
