// Workspace object and error helper shared by the C-ABI translation units.
#pragma once
#include "../../include/gmmvb.h"

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdlib>

#include "policy.h"

struct gmmvb_hmm_state;      // HMM forward-backward buffers (hmm_capi.hip), allocated by hmmvb_enable
struct gmmvb_workspace;

// What an E-step leaves for the policy of the next one (and for gmmvb_last_work)
struct gmmvb_pass_counters {
    bool valid = false;
    double act = 0.0, eval = 0.0, over = 0.0, settled = 0.0, listed = 0.0, accum = 0.0, proof = 0.0, exits = 0.0, moved = 0.0;
    double cols = -1.0;          // (tile, component) columns of the bound array the last sweep went through; -1: not a lazy sweep
    double left = -1.0;          // pairs the stateless table (project.h) did not clear; -1: the pass was no projected sweep
    double rows = 0.0;           // rows the counters were taken over
    double ranks = 1.0;          // ranks they were summed over
    int mode = 0;                // kind of the pass: 0 dense, 1 bound pass, 2 carried records, 3 sweep
};

// Buffers that only live from an E-step to the M-step (or read-out) behind it: ln rho [K][npad] f64, the sample lists
// [K][npad] i32, the centred f64 copy of the rows and the M-step's slabs - 3.5 KB per row at K = 256, D = 64, two thirds of
// the workspace.  The workspaces of ONE row-tiled job (gmmvb_workspace_create_tile) share one set, sized for the largest
// tile: what a tile carries from iteration to iteration (f32 bounds, records, digit planes, settled rows, row order) stays
// its own.  `owner` is the workspace whose E-step output the buffers hold; another workspace that needs them takes them
// over (capi.hip: claim_scratch) and the previous owner is back to "no E-step output".  All on one stream.
struct gmmvb_scratch {
    double* lnrho = nullptr;
    double* xc = nullptr;
    double* slabs = nullptr;
    int* lists = nullptr;
    int64_t npad = 0;                  // rows (padded) the buffers were sized for
    int64_t xc_len = 0, slabs_len = 0; // doubles
    gmmvb_workspace* owner = nullptr;
    int refs = 0;
};

struct gmmvb_workspace {
    int K = 0, D = 0, T = 0, x_dtype = 0;
    gmmvb_scratch* scratch = nullptr;  // lnrho / xc / slabs / lists below point into it
    bool lost_estep = false;           // the last E-step's output went to another tile of the group (e_state 0); its counters,
                                       // masks and block counts are still that pass's
    int64_t max_rows = 0, npad = 0;
    int num_cu = 0, KG = 0, S_cap = 0;
    int64_t split_rows = 0;    // cap on rows per M-step split
    double* lnrho = nullptr;   // [K][npad]
    double* lse = nullptr;     // [npad]
    double* img = nullptr;     // [K][img_len] parameter images (layout: estep.h)
    int img_len = 0;
    int estep_variant = 0;     // kEstepLds8 (default); env GMMVB_ESTEP_VARIANT=direct|lds4|i8 selects the others
    double* tri = nullptr;             // [K][estep_tri_image_doubles()] packed lower-triangular images (D <= 16: estep_rows16_f64)
    unsigned char* img_i8 = nullptr;   // [K][img_i8_len] int8-digit parameter images (estep_i8.h), variant kEstepI8 only
    int img_i8_len = 0;
    double* pivot_i8 = nullptr;        // [D] the pivot those images (and the sample digits) are centred on
    unsigned char* img_i8b = nullptr;  // [K][img_i8b_len] 3-digit images of the pruned E-step's bound pass
    int img_i8b_len = 0;
    // output blocks the int8 bound pass evaluates (fewer blocks: cheaper pass, looser bound, more candidates for the
    // exact pass).  tb_cand[L] = candidates per pair the last pass at level L left, tb_seen[L] = pruned E-steps since
    // (levels not seen for 32 passes count as unknown); gmmvb_estep picks the level with the lowest modelled cost
    int bound_tb = 0;
    double tb_cand[5] = {-1.0, -1.0, -1.0, -1.0, -1.0};
    double tb_act[5] = {0.0, 0.0, 0.0, 0.0, 0.0};      // active pairs per pair when tb_cand[L] was observed
    int tb_seen[5] = {0, 0, 0, 0, 0};
    // carrying the E-step over a parameter update (gmmvb_set_drift, records.h): gamma / delta / Gamma of the pending
    // update, whether the records belong to the parameters of the last E-step on `bounds_rows` rows of `bounds_x`
    double* drift = nullptr;           // [4][K]: gamma, delta, c of the last E-step, Gamma
    bool have_drift = false;
    double typical_gamma = -1.0;       // mean gamma of the pending update if the caller knew it (<= 0: unknown)
    bool params_used = false;          // the parameters in force were the ones of the last E-step
    int64_t bounds_rows = 0;           // rows / matrix the records (and the ln rho array) belong to (0: nothing to carry)
    const void* bounds_x = nullptr;
    int64_t bounds_ldx = 0;
    int prev_pass = 0;                 // last E-step: 0 dense, 1 bound pass, 2 carried records, 3 dense sweep
    // per-row candidate records (records.h), allocated with the sample lists
    unsigned short* rec_k = nullptr;   // [8][npad]
    float* rec_d = nullptr;            // [8][npad]
    float* rec_B = nullptr;            // [npad]
    unsigned char* rec_exact = nullptr, *rec_sel = nullptr, *rec_flags = nullptr;   // [npad] each
    bool rec_valid = false;            // the records describe the last E-step's parameters on bounds_rows rows
    bool rec_live = false;             // the last E-step lived on records (read-outs go through them)
    float* ub32 = nullptr;             // [K][npad] upper bound of ln rho for every pair (rounded up), what the sweeps carry
    bool dense_valid = false;          // EVERY entry of ub32 is a value / bound under the last E-step's parameters
    int sweeps = 0;                    // dense sweeps since the last bound / dense pass (their bounds erode: at most 8)
    int* plan = nullptr;               // [K + 1] gather chunk plan (device)
    int* plan_m = nullptr;             // [K + 2] chunk plan of the list M-step
    double* epart = nullptr;           // [ceil(npad / 256)] listed pairs per selection block
    double* opart = nullptr;           // [ceil(npad / 256)] overflow rows per selection block
    double* mpart = nullptr;           // [ceil(npad / 256)] rows whose best component changed, per selection block
    double moved_since_sort = 0.0;     // their sum over the passes since the rows were last regrouped
    bool pend_first_sorted = false;    // the E-step whose counters are in flight regrouped the rows
    // counters of an E-step: [0] active pairs (r >= 2^-80), [1] pairs evaluated exactly, [2] overflow rows,
    // [3] rows whose best component changed.
    // Written on the device at the end of every E-step and copied to pinned host memory behind an event; the NEXT
    // E-step / M-step reads whatever has arrived (policy decisions lag one pass, results never depend on them).
    // [4] settled rows (see below), [5] pairs in the M-step's lists.
    // [6] pairs the M-step accumulates, [7] pairs of the proof round (int8 two-sided bounds instead of f64 values).
    double* ctr = nullptr;             // [8] device
    double* ctr_host = nullptr;        // [8] pinned
    hipEvent_t ctr_ev = nullptr;
    bool ctr_pending = false;          // a copy is in flight ...
    int pend_mode = 0;                 // ... of an E-step of this mode over pend_rows rows
    int64_t pend_rows = 0;
    double pend_round0 = 0.0;          // pairs that E-step evaluated before its counted selection round
    bool sweep_prev = false;           // the last sweep's first round used the previous pass's M-step lists
    // lag = counters of this rank's most recent E-step whose copy has arrived: what gmmvb_last_work reports, and the
    // policy's input in a single process.  pol = the same summed over the ranks of a row-sharded job
    // (gmmvb_policy_export / all-reduce / gmmvb_policy_import): once gmmvb_set_shard has been called the policy reads
    // nothing else that differs between ranks, so every rank takes the same decisions.
    gmmvb_pass_counters lag, pol;
    // unit costs and thresholds of the pass policy at this workspace's shape (policy.h); calibration from its own first
    // dense E-step / dense M-step / bound pass: HIP events around those launches, taken over when they have completed
    gmmvb::PolicyTable pt;
    bool opt_calibrate = true;         // gmmvb_policy_calibrate(ws, 0) / env GMMVB_POLICY_CALIBRATE=0: the scaled literals only
    hipEvent_t cal_ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};     // begin / end for dense E, dense M, bound pass
    double cal_pairs[3] = {0.0, 0.0, 0.0};       // pairs behind a pending measurement (0: none pending, -1: done)
    int cal_tries[3] = {0, 0, 0};                // measurements discarded so far
    bool sharded = false;              // gmmvb_set_shard: this workspace holds one shard of shard_rows rows over shard_ranks ranks
    int64_t shard_rows = 0;
    int shard_ranks = 1;
    double* pol_host = nullptr;        // [GMMVB_POLICY_LEN] pinned: the imported (summed) counters
    hipEvent_t pol_ev = nullptr;
    bool pol_pending = false;          // an import is in flight ...
    int pol_mode = 0;                  // ... of a pass of this mode
    bool pol_first_sorted = false;
    bool exp_counted = false;          // the last E-step counted its pairs (what gmmvb_policy_export may hand out)
    bool forget = false;               // gmmvb_forget: the next parameters are unrelated to the last E-step's
    double spare_last = -1.0;          // spare candidates per pair of the last pruned pass (diagnostics)
    double bound_fail_act = -1.0;      // active share of the pairs when a bound pass last left most of them candidates (< 0: never)
    double* cvec = nullptr;    // [K]
    double* pivot = nullptr;   // [D]
    double* dpart = nullptr;   // [ceil(npad / 1024)][K] block maxima of ln r (row_lse_kernel)
    double* thr = nullptr;     // [K] M-step skip thresholds: max_n ln r_nk - 80 ln 2 (valid while e_state == 1)
    bool sparse = true;        // env GMMVB_MSTEP_SPARSE=0: always run the dense M-step
    double* apart = nullptr;   // [ceil(npad / 256)] pairs with ln r >= -80 ln 2 per selection block
    int64_t act_rows = 0;      // rows of the E-step whose active pairs were counted (0 = not counted)
    double act_host = -1.0;    // host copy of ctr[0] for that E-step (-1 = not fetched yet)
    double evaluated = 0.0;    // pairs the last E-step evaluated exactly (-1: on the device, see gmmvb_last_sparsity)
    // pruned E-step (estep.h): env GMMVB_ESTEP_PRUNE = 0 never | force always | default: when the previous E-step
    // over the same rows left at most half of the pairs relevant and N K >= 2^23
    int prune = 1;
    int* lists = nullptr;      // [K][npad] sample lists (E-step candidates, then the M-step's active rows)
    bool active_lists = false; // the lists currently hold the active rows of the last E-step (scan + fill done)
    bool blk_fresh = false;    // blk still holds the block COUNTS of masks (not yet scanned into bases)
    int* khat = nullptr;       // [npad]
    int* counts = nullptr;     // [K]
    int* blk = nullptr;        // [blocks of 256 rows][K] candidates per selection block -> block bases (aux_kernels.h: blk_at)
    int* scan_parts = nullptr; // [K][kScanParts] partial sums of the scan over blk
    unsigned long long* masks = nullptr;   // [ceil(K / 64)][npad] candidate components of every sample
    double* slabs = nullptr;   // [S_cap][K][slab_len]
    // Settled rows (records.h): a row with ONE active component has r = 1.0 exactly, its addend to that component's
    // statistics does not depend on the parameters.  While the carried bounds prove that it stays so, the row is neither
    // evaluated (E) nor accumulated (M): its addend lives in `cache` (same layout as the statistics), which changes only
    // through the rows that settle or come loose in a pass (the M-step's delta lists).
    unsigned char* lock = nullptr;     // [npad] 0 free, 1 settled, 2 came loose in this pass, 3 settled in this pass
    unsigned char* lcomp = nullptr;    // [npad] cached rows: the component whose cache holds the row (K <= 256)
    float* dlock = nullptr;            // [npad] settled rows: upper bound of the whitened distance to their component
    float* rthr = nullptr;             // [npad] relevance threshold of the selection round (best exact value - 80 ln 2)
    unsigned long long* exit_ctr = nullptr;    // [4] device: candidate pairs of the pass that took the gather's early way out;
                                               //     (tile, component) columns the lazy sweep opened; pairs the table of
                                               //     project.h took off the proof lists (stateless sweep: pairs it left)
    unsigned long long* exit_host = nullptr;   // [4] pinned mirror (copied with the other counters)
    bool pend_lazy = false;                    // the pass behind the pending counters was a lazy sweep
    unsigned long long* dmask = nullptr;   // [ceil(K / 64)][npad] rows entering / leaving the cache in this pass
    int* dblk = nullptr;               // [blocks][K] their block counts
    unsigned long long* mmask = nullptr;   // [ceil(K / 64)][npad] the M-step's lists: active pairs of the rows not in the cache
    int* mblk = nullptr;               // [blocks][K] their block counts
    double* cache = nullptr;           // [stats_len] statistics of the settled rows
    double* spart = nullptr, *gpart = nullptr, *qpart = nullptr;   // [blocks] settled rows / listed pairs / M-step pairs per selection block
    unsigned long long* rmask = nullptr;   // read-outs of settled rows: their (row, component) pairs ...
    int* rblk = nullptr;                   // ... and block counts (allocated by the first such read-out)
    bool lock_live = false;            // some rows may be settled: the list M-step must add the cache
    bool delta_pending = false;        // an E-step marked rows 2 / 3 / 4 and no M-step has applied them yet
    bool mlists_done = false;          // the lists hold the M-step's own lists (mmask) of the last E-step
    bool mlists_lost = false;          // ... did, until a read-out of settled rows used the buffers
    bool skip_used = false;            // some pass since the cache was last emptied was allowed to settle rows
    bool lock_reset = false;           // the settled state belongs to something else now: drop it at the next E-step
    bool settled_fresh = false;        // read-outs: the settled rows' ln rho / lse were re-evaluated for the parameters in force
    double settle_margin = 1e300;      // nats of slack demanded before a row is settled (< 0: never; 1e300: every single-component row)
    bool cache_on = true;              // env GMMVB_MSTEP_CACHE=0: no cache of single-component rows
    bool gather_exit = true;           // env GMMVB_GATHER_EXIT=0: no early way out in the candidate gather
    // further switches, all read ONCE when the workspace is created (no getenv on the per-iteration path)
    bool opt_carry_off = false;        // GMMVB_ESTEP_CARRY_OFF: ignore gmmvb_set_drift
    bool opt_debug = false;            // GMMVB_DEBUG=2: one line per E-step on stderr
    // Proof round (estep_i8.h, records.h): the three int8 digit planes of every row, in the internal row order, made with
    // the centred copy (gmmvb_prepare_rows) and again when the rows are regrouped.  Valid for the matrix xq_src while the
    // pivot they are centred on is the one the component images were packed for (xq_gen == img_gen).
    unsigned char* xq = nullptr;       // [npad][3][32 ceil(D / 32)]
    signed char* xqe = nullptr;        // [npad] exponent of the row's largest |x - pivot| (127: no digits)
    const void* xq_src = nullptr;
    int64_t xq_rows = 0, xq_ldx = 0;
    int pivot_gen = 0, xq_gen = -1, img_gen = -2;
    bool opt_proof = true;             // env GMMVB_PROOF=0: settled rows with candidates go straight to the f64 gather
    bool opt_hmm_mstep_dense = false;  // env GMMVB_HMM_MSTEP_DENSE: the HMM's small M-step walks every (step, state), not only gamma >= 2^-80
    bool opt_proof_blocked = true;     // env GMMVB_PROOF_BLOCKED=0: estep_i8_proof instead of estep_i8_proof_blocked (row superblocks)
    bool opt_regroup_margin = true;    // env GMMVB_REGROUP_MARGIN=0: regroup by best component only (no second key: how firmly a row sits in it)
    bool opt_lazy = true;              // env GMMVB_SWEEP_LAZY=0: every sweep goes through all K bounds of every row
    bool opt_proof_all = true;         // the candidates of rows with an exact reference go through the proof round too (env
                                       // GMMVB_PROOF=settled: only the settled rows' pairs)
    // The stateless sweep (project.h): bounds from the parameters in force and the rows' int8 digit planes instead of the
    // carried per-pair array.  gimg / gconst are made by gmmvb_set_params when a sweep can follow (regrouped rows, digit
    // planes about the pivot in force, a drift hint for the settled rows' own bound), tile_ref by the regrouping.
    unsigned char* gimg = nullptr;     // [K][proj_image_bytes] int8 digit images of g_jk, per reference component j
    void* gconst = nullptr;            // [K][32 proj_kblocks] float4 constants per (reference, column)
    int* tile_ref = nullptr;           // [blocks] reference component of every tile of kSelRows rows (regrouped order)
    int opt_project = 0;               // env GMMVB_PROJECT: "filter" (1) the table filters the carried sweep's proof lists, "only" (2)
                                       // the sweep itself is stateless (rec_project_kernel); default 0: no table (it does not
                                       // pay on the benchmark's fits, profiles/r6_experiments.md)
    bool proj_table = false;           // gimg / gconst describe the parameters in force
    bool tile_ref_valid = false;       // tile_ref describes the row order in force
    bool pend_proj = false;            // the pass behind the pending counters used the table (exit_ctr[1]: pairs it left / removed)
    float4* tmeta = nullptr;           // [blocks][K] the lazy sweep's state per tile and component (records.h)
    bool tmeta_valid = false;          // it describes the bound array as it is (only sweeps have written to it since)
    double* ppart = nullptr;           // [blocks] pairs of the proof round per selection block
    // rows grouped by dominant component (aux_kernels.h): internal row i = the caller's row perm[i]
    void* xp = nullptr;        // [max_rows][D] x in internal row order (storage dtype), allocated with the lists
    int* perm = nullptr, *iperm = nullptr, *perm_tmp = nullptr;   // [npad] each
    bool sorted = false;       // xp / perm / iperm are in force for the matrix xc_src (else the internal order is the caller's)
    bool sort_rows = true;     // env GMMVB_SORT_ROWS=0: never regroup
    int64_t sorts = 0;         // regroupings since the workspace was created
    double* xc = nullptr;      // [npad][16T] centred f64 copy of the sample matrix (M-step operand), optional
    const void* xc_src = nullptr;   // the x it was made from (null = not prepared)
    int64_t xc_rows = 0, xc_ldx = 0;
    bool xc_stale = false;     // the rows were regrouped since xc was made: it is rebuilt from xp when a kernel needs it
    // c_degree > 128 (generic.h): plain f64 kernels on the raw parameters, no images, no lists
    bool wide = false;                 // 128 < D <= 256: dense MFMA kernels for 9 .. 16 feature tiles (T rounded up to even)
    bool generic = false;
    double* gen_u = nullptr;           // [K][D][D]
    double* gen_m = nullptr;           // [K][D]
    double* gen_first = nullptr;       // [gen_S][K][D + 2] partial ns, h, a per row split
    double* gen_second = nullptr;      // [gen_S][K][tri_pairs(T)][256] partial B tiles per row split
    int gen_S = 1;
    int64_t bytes = 0;
    bool have_params = false;
    int e_state = 0;           // 0 none, 1 E-step output, 2 responsibilities loaded directly, 3 HMM gamma,
                               // 4 emission as rho' / mx in the HMM state, no ln rho array (hmmvb_emission_target)
    int64_t e_rows = 0;
    char info[512] = {0};
    hipError_t hip_err = hipSuccess;   // first failed event record / counter reset of the pass in flight (capi.hip: note_hip)
    // launches since the workspace was created (gmmvb_pass_counts): E dense, E bound pass, E carried bounds,
    // pruned E-step that fell back to the dense kernel, E sweep of carried bounds, M dense, M lists,
    // candidate gathers
    int64_t passes[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    bool prof = false;
    bool prof_light = false;           // gmmvb_profile_enable(ws, 2): only the spans of the groups that can dominate a step
    bool span_open = false;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};   // E begin/end, M begin/end
    bool ev_e = false, ev_m = false;
    // per-kernel-group spans of the last E-step + M-step (gmmvb_profile_spans): slot, begin/end event
    static constexpr int kMaxSpans = 24;
    int n_spans = 0;
    int span_slot[kMaxSpans] = {0};
    hipEvent_t span_ev[2 * kMaxSpans] = {nullptr};
    gmmvb_hmm_state* hmm = nullptr;
    bool hmm_skip_h = false;           // hmmvb_skip_h: the M-step of an HMM pass leaves h = 0 (and does not read the ln rho array)
    bool hmm_no_lnrho = false;         // the last E-step handed its emission to the HMM state (e_state 4, then 3): no ln rho array
    bool lse_stale = false;            // the last dense E-step of an HMM workspace did not make lse (only a read-out wants it)
};

namespace gmmvb {
// Environment switches.  Two are part of the interface: GMMVB_ESTEP_PRUNE and GMMVB_MSTEP_SPARSE, read with
// std::getenv.  Every other GMMVB_* switch is a test / developer seam - it selects between kernels whose results agree, so
// that a parity test can run each of them - and is only looked at when GMMVB_DEBUG is set ("1"; "2" also prints one line
// per E-step on stderr).  All of them are read once, when a workspace / HMM state is created.
inline const char* dev_env(const char* name) {
    const char* on = std::getenv("GMMVB_DEBUG");
    return (on && on[0] != '\0' && on[0] != '0') ? std::getenv(name) : nullptr;
}
int fail(int code, const char* what, hipError_t e = hipSuccess);     // sets the thread-local message
void hmm_state_destroy(gmmvb_hmm_state* h);
const double* hmm_gamma_cm(const gmmvb_hmm_state* h);                 // [K][npad] responsibilities of the last pass (after hmm_ensure_gamma_cm)
const double* hmm_gamma_tm(const gmmvb_hmm_state* h);                 // [npad][Kp] the same, time-major in lane order (hmm.h)
int hmm_padded_states(const gmmvb_hmm_state* h);                      // Kp
bool hmm_fused_emission(const gmmvb_hmm_state* h);                    // gmmvb_estep hands rho' / mx straight to the HMM state (e_state 4)
hipError_t hmm_ensure_gamma_cm(gmmvb_hmm_state* h, hipStream_t st);   // transposes gamma once per pass, on demand
inline int64_t round_up(int64_t v, int64_t m) { return (v + m - 1) / m * m; }
}  // namespace gmmvb
