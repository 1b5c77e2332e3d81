// What the translation units of the C ABI share (capi.hip: workspace life cycle, parameters, policy exchange, preparation;
// capi_estep.hip: gmmvb_estep; capi_mstep.hip: gmmvb_mstep and friends; capi_readout.hip: the read-outs): the small helpers
// around a pass (error latch, profiling spans, the tile group's shared scratch) and the functions one unit defines for
// the others.
#pragma once
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

#ifndef GMMVB_T1_SPLITS
#define GMMVB_T1_SPLITS 24
#endif

extern "C" {


// profiling spans (gmmvb_profile_spans): HIP events on the launch stream around groups of kernels
enum { kSpanEstepMain = 0, kSpanSelect = 1, kSpanGather = 2, kSpanLse = 3, kSpanLists = 4, kSpanMstepMain = 5,
       kSpanReduce = 6, kSpanProof = 7, kSpanSlots = 8 };
static const char* const kSpanNames[kSpanSlots] = {"estep_main", "estep_select", "estep_gather", "estep_lse_mask",
                                                   "mstep_lists", "mstep_main", "mstep_reduce", "estep_proof"};
// A failed event record / counter reset inside a pass must not vanish: the first such error is kept in the workspace and
// gmmvb_estep / gmmvb_mstep return it (GMMVB_EHIP) before they hand anything to the caller.
static inline void note_hip(gmmvb_workspace* ws, hipError_t e) {
    if (e != hipSuccess && ws->hip_err == hipSuccess) ws->hip_err = e;
}
static inline int take_hip(gmmvb_workspace* ws, const char* what) {
    if (ws->hip_err == hipSuccess) return GMMVB_OK;
    const hipError_t e = ws->hip_err;
    ws->hip_err = hipSuccess;
    return fail(GMMVB_EHIP, what, e);
}
// An event record costs the stream about 10 us (the queue drains around the marker packet): at the benchmark shape a converged
// step has ~36 of them, 0.17 ms of a 3.7 ms step.  Profile level 2 keeps only the spans of the three groups that can dominate a
// step (the two E-step evaluation groups and the M-step's accumulation) and drops the phase events.
static inline bool span_kept(const gmmvb_workspace* ws, int slot) {
    return !ws->prof_light || slot == kSpanEstepMain || slot == kSpanGather || slot == kSpanMstepMain;
}
static inline void span_begin(gmmvb_workspace* ws, int slot, hipStream_t st) {
    ws->span_open = false;
    if (!ws->prof || ws->n_spans >= gmmvb_workspace::kMaxSpans || !span_kept(ws, slot)) return;
    ws->span_slot[ws->n_spans] = slot;
    ws->span_open = true;
    note_hip(ws, hipEventRecord(ws->span_ev[2 * ws->n_spans], st));
}
static inline void span_end(gmmvb_workspace* ws, hipStream_t st) {
    if (!ws->span_open) return;
    ws->span_open = false;
    note_hip(ws, hipEventRecord(ws->span_ev[2 * ws->n_spans + 1], st));
    ++ws->n_spans;
}
static inline bool phase_events(const gmmvb_workspace* ws) { return ws->prof && !ws->prof_light; }

// ---- the scratch of a tile group (workspace.h: gmmvb_scratch) -----------------------------------------------------------
// `w` loses the buffers to another workspace of its group: its E-step output, lists and centred copy are gone.  What it
// carries into its next E-step (bounds, records, settled rows, digit planes, row order, policy counters) is untouched; that
// E-step starts its first round from the rows' best components instead of the previous pass's lists.
static inline void yield_scratch(gmmvb_workspace* w) {
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
static inline void claim_scratch(gmmvb_workspace* ws) {
    gmmvb_scratch* s = ws->scratch;
    if (!s || s->owner == ws) return;
    if (s->owner) yield_scratch(s->owner);
    s->owner = ws;
}
static inline void release_scratch(gmmvb_workspace* ws) {
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

// ---- defined in capi.hip
int ensure_lists(gmmvb_workspace* ws);
int fetch_counters(gmmvb_workspace* ws);
void poll_counters(gmmvb_workspace* ws);
int take_policy(gmmvb_workspace* ws);
int check_x(const gmmvb_workspace* ws, const void* x_dev, int64_t ldx, int64_t n_rows, bool* vec);
hipError_t recenter_rows(gmmvb_workspace* ws, int64_t n_rows, hipStream_t st);
// ---- defined in capi_estep.hip: calibration of the policy table from the workspace's own passes (policy.h)
bool cal_wanted(gmmvb_workspace* ws, int what, double pairs);
void cal_mark(gmmvb_workspace* ws, int what, double pairs, hipStream_t st);
void cal_poll(gmmvb_workspace* ws);

}  // extern "C"
