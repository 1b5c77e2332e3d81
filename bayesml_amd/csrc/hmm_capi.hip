// C ABI of the HMM forward-backward pass (include/gmmvb.h, hmmvb_* entry points).
#include "workspace.h"

#include <algorithm>
#include <cmath>
#include <new>

#include "hmm.h"
#include "launch.h"
#include "hmm_generic.h"
#include "hmm_wide.h"

#include <cstdlib>

using namespace gmmvb;

struct gmmvb_hmm_state {
    int K = 0, Kp = 0, KT = 0;
    int64_t npad = 0, max_chunks = 0, xi_waves = 0;
    double* rho_tm = nullptr;     // [npad][Kp] lane order
    double* alpha_tm = nullptr;   // [npad][Kp]
    double* gamma_tm = nullptr;   // [npad][Kp]
    double* w_tm = nullptr;       // [npad][Kp]
    double* gamma_cm = nullptr;   // [K][npad]  component-major copy of gamma, made on demand (hmm_ensure_gamma_cm): read-outs and
                                  //            the M-step kernels of D > 16; the D <= 16 M-step reads gamma_tm itself
    bool gamma_cm_valid = false;
    int64_t gamma_rows = 0;       // rows of the last forward-backward pass
    double* mx = nullptr;         // [npad]
    double* cprime = nullptr;     // [npad]
    double* prod = nullptr;       // [max_chunks][Kp][Kp]
    double* prod_t = nullptr;     // [max_chunks][Kp][Kp] their transposes (65 .. 128 states: the backward boundary pass reads them)
    double* qprod_t = nullptr;    // [max_chunks / kHmmSuper + 2][Kp][Kp] transposes of the super-chunk products (65 .. 128 states)
    double* fstart = nullptr;     // [max_chunks][Kp]
    double* bend = nullptr;       // [max_chunks][Kp]
    double* qprod = nullptr;      // [max_chunks / kHmmSuper + 2][Kp][Kp] super-chunk products (two-level boundary pass)
    double* fstart_s = nullptr, *bend_s = nullptr;   // [max_chunks / kHmmSuper + 2][Kp] their boundary vectors
    double* xi_slabs = nullptr;   // [xi_waves][Kp][Kp]
    double* lnc_partial = nullptr;   // [kLncBlocks]
    unsigned char* phi = nullptr; // [npad][Kp] Viterbi back-pointers (allocated on first use by hmmvb_enable)
    int* last_state = nullptr;
    // more than 64 states: the sequential kernels of hmm_generic.h; 65 .. 128 states and long sequences: the chunk-parallel
    // kernels of hmm_wide.h for the forward-backward pass (generic stays set: prep, xi-sum, Viterbi are the generic ones)
    bool generic = false;
    bool wide = false;
    double* a_t = nullptr;        // [K][K] transpose of A~
    unsigned short* phi16 = nullptr;   // [npad][K] back-pointers, natural order
    int64_t bytes = 0;
    int64_t vec_chunks = 0;       // rows of fstart / bend / fstart2 / bend2
    int64_t xi_slab_cap = 0;      // slabs xi_slabs has room for
    bool xi_separate = false;     // developer switch GMMVB_HMM_XI_SEPARATE: hmm_xi_sum_kernel as in round 3
    bool w_valid = false;         // the last pass wrote w_tm (false: the backward replay summed xi itself, hmm.h H5 XI)
    double* fstart2 = nullptr, *bend2 = nullptr;   // [max_chunks][Kp] the replays' own boundary vectors (forgetting pass)
    int* gate_dev = nullptr;      // 1: the forgetting pass's vectors do not stand, the products path runs
    int* gate_host = nullptr;     // pinned copy, read when the next call begins
    hipEvent_t gate_ev = nullptr;
    bool gate_pending = false, spec_on = true;
    bool vit_coalesced = false;   // the last hmmvb_viterbi call ran the coalescence pass (its gate: gate_dev[1])
    int64_t sweep_len = 64;       // steps next to a chunk boundary the forgetting pass's first stage walks (run<KT>; 32 failed one pass in seven at config 5, 64 none)
    int64_t chunk_floor = 0;      // run<KT>: a pass that did not stand on chunks shorter than kHmmLongChunk doubles them for the next ones
    int64_t gate_l = 0;           // chunk length of the pass whose gates are pending
    int short_hold = 0, short_hold_len = 4;      // calls that skip the short first stage after it did not stand (4, 8 ... 64 while it keeps failing)
    bool gate_two_stage = false;  // the pass whose gates are pending had a short first stage
    int spec_hold = 0, spec_hold_len = 8, last_gate = -1;   // last_gate: -1 no forgetting pass, 0 it stood, 1 products path behind it
    bool fuse_emission = false;   // hmmvb_emission_target: gmmvb_estep writes rho' / mx here (hmm.h H0 + H1) and no ln rho array
};

namespace gmmvb {
void hmm_state_destroy(gmmvb_hmm_state* h) {
    if (!h) return;
    double* bufs[] = {h->rho_tm, h->alpha_tm, h->gamma_tm, h->w_tm, h->gamma_cm, h->mx,
                      h->cprime, h->prod,     h->fstart,   h->bend, h->xi_slabs, h->lnc_partial,
                      h->qprod,  h->fstart_s, h->bend_s, h->a_t, h->prod_t, h->qprod_t};
    for (double* p : bufs)
        if (p) (void)hipFree(p);
    if (h->fstart2) (void)hipFree(h->fstart2);
    if (h->bend2) (void)hipFree(h->bend2);
    if (h->gate_dev) (void)hipFree(h->gate_dev);
    if (h->gate_host) (void)hipHostFree(h->gate_host);
    if (h->gate_ev) (void)hipEventDestroy(h->gate_ev);
    if (h->phi) (void)hipFree(h->phi);
    if (h->phi16) (void)hipFree(h->phi16);
    if (h->last_state) (void)hipFree(h->last_state);
    delete h;
}
const double* hmm_gamma_cm(const gmmvb_hmm_state* h) { return h ? h->gamma_cm : nullptr; }
const double* hmm_gamma_tm(const gmmvb_hmm_state* h) { return h ? h->gamma_tm : nullptr; }
int hmm_padded_states(const gmmvb_hmm_state* h) { return h ? h->Kp : 0; }
// the emission of one feature tile straight into rho' / mx (hmm_emission_mfma16_kernel): asked for, and a shape it covers
// (up to 32 states: beyond, a wave's K values per row no longer fit its registers beside the pipeline's operands)
bool hmm_fused_emission(const gmmvb_hmm_state* h) { return h && h->fuse_emission && !h->generic && h->KT <= 2; }
hipError_t hmm_launch_emission16(gmmvb_hmm_state* h, int x_is_f64, bool vec, hipStream_t st, const EstepArgs& a, const char** name) {
    const int64_t rows_per_wg = 4 * 16 * (h->KT == 1 ? 4 : 2);       // (hmm.h: NB row tiles per wave)
    const int64_t wgs = (a.n_rows + rows_per_wg - 1) / rows_per_wg;
    const unsigned grid = (unsigned)std::min<int64_t>(wgs, int64_t(1) << 20);
#define EM(XT, V, KTT)                                                                                                      \
    hipLaunchKernelGGL((hmm_emission_mfma16_kernel<XT, V, KTT>), dim3(grid), dim3(256), 0, st, static_cast<const XT*>(a.x), a.ldx, \
                       a.n_rows, a.D, a.img, a.cvec, a.K, h->rho_tm, h->mx)
#define EMK(XT, V)                                    \
    switch (h->KT) {                                  \
        case 1: EM(XT, V, 1); break;                  \
        default: EM(XT, V, 2); break;                 \
    }
    *name = "hmm_emission_mfma16_kernel";
    if (x_is_f64) {
        if (vec) EMK(double, true) else EMK(double, false)
    } else {
        if (vec) EMK(float, true) else EMK(float, false)
    }
#undef EMK
#undef EM
    return hipGetLastError();
}
// gamma component-major for whoever reads it that way: transposed once per forward-backward pass, and only if asked for
hipError_t hmm_ensure_gamma_cm(gmmvb_hmm_state* h, hipStream_t st) {
    if (!h || h->gamma_cm_valid || h->gamma_rows < 1) return hipSuccess;
    const int64_t T = h->gamma_rows;
    hipLaunchKernelGGL(hmm_gamma_to_cm_kernel, dim3((unsigned)((T + 63) / 64), (unsigned)((h->K + 63) / 64)), dim3(256), 0, st,
                       h->gamma_tm, T, h->K, h->Kp, h->npad, h->gamma_cm);
    h->gamma_cm_valid = true;
    return hipGetLastError();
}
}  // namespace gmmvb

namespace {

constexpr int kLncBlocks = 1024;
constexpr int kHmmCheckBlocks = 1024;      // grid of hmm_boundary_check_kernel (grid-stride over 40 MB of boundary vectors at config 5: 64 blocks took 0.08 ms)
// the forgetting pass stands if no entry of a (sum-1 normalised) boundary vector moves by more than this RELATIVE to itself
// when its chunk is started from the sweep's vector instead of the uniform one (hmm.h, hmm_boundary_check_kernel<true>: a test
// in the Hilbert metric, in which the recursion is non-expansive): a fully forgotten start leaves the rounding noise of sums
// of positive terms, a few 1e-16 relative per entry whatever its size
constexpr double kHmmForgetTol = 2e-13;
// the Viterbi pass's coalescence test (absolute, nats, on omega - max of a chunk's end vector): the two replays of a chunk
// round sums that reach ~1e4 in magnitude inside the chunk (ulp 2e-12) differently, so ~1e-11 is the noise; start-vector
// errors add up to at most chunks x tol over the sequence (max-plus maps are 1-Lipschitz): 8e-6 nats at 4e4 chunks, below
// the rounding of the sequential recursion's own sums at that length (scores ~5e8: ulp 6e-8 per step)
constexpr double kVitCoalesceTol = 2e-10;
constexpr int64_t kHmmGenericChunk = 256;      // more than 128 states: steps per workgroup in the forgetting pass

// chunk length: balances the sequential boundary scan (T/L steps of ~1.5 us) against the replay depth
// (L steps of ~4 us forward+backward): L ~ sqrt(T * 1.5 / 4), a power of two in [16, 4096]
// Long sequences (more than kHmmLongFrom steps): chunks of 256 steps and a two-level boundary pass (hmm.h, H3a / H3b).
// (round 4: from 2^15 steps instead of 2^18 - with the forgetting pass the long-sequence form costs two sweeps instead of the chunk
// products, and its short chunks shorten the replays' chains of dependent steps: T = 2e5 4.1 ms per iteration against 1.5)
constexpr int64_t kHmmLongFrom = int64_t(1) << 15;
constexpr int64_t kHmmShortForgetFrom = 4096;      // from here to kHmmLongFrom: the forgetting pass on chunks of kHmmShortChunk steps
constexpr int64_t kHmmShortChunk = 32;
int64_t chunk_len(int64_t T, bool one_level) {
    if (T > kHmmLongFrom && !one_level) return kHmmLongChunk;
    int64_t L = 16;
    while (L < 4096 && 8 * L * L < 3 * T) L *= 2;
    return L;
}

// The gate of the last forgetting pass, once its pinned copy has arrived: a pass that did not stand holds the next ones off -
// for 8 calls, then 16, ... 64 while it keeps failing (sticky chains with flat emissions pay for a sweep and a replay each time).
bool consume_gate(gmmvb_hmm_state* h, bool wait) {
    if (!h->gate_pending) return true;
    const hipError_t e = wait ? hipEventSynchronize(h->gate_ev) : hipEventQuery(h->gate_ev);
    if (e == hipErrorNotReady) return true;
    if (e != hipSuccess) return false;
    h->gate_pending = false;
    h->last_gate = h->gate_host[0];
    if (h->gate_two_stage) {
        // the first stage's sweeps walked only the steps next to the boundaries: when that was not enough the whole-chunk
        // stage behind its gate has done the work (a sweep and two replays more): go straight to whole chunks for a while
        if (h->gate_host[1] != 0) {
            h->short_hold = h->short_hold_len;
            h->short_hold_len = std::min(64, 2 * h->short_hold_len);
        } else {
            h->short_hold_len = 4;
        }
    }
    if (h->last_gate != 0 && h->gate_l > 0 && h->gate_l < kHmmLongChunk) {
        // the chunks were shorter than the recursions' memory: longer ones at once (no hold-off - that is for sequences
        // whose 256-step chunks do not stand either)
        h->chunk_floor = 2 * h->gate_l;
    } else if (h->last_gate != 0) {
        h->spec_hold = h->spec_hold_len;
        h->spec_hold_len = std::min(64, 2 * h->spec_hold_len);
    } else {
        h->spec_hold_len = 8;
    }
    h->gate_l = 0;
    return true;
}

template <int KT>
hipError_t run(gmmvb_workspace* ws, gmmvb_hmm_state* h, int64_t T, const double* pi_tilde, const double* a_tilde,
               double* out, hipStream_t st) {
    const int K = h->K, Kp = h->Kp;
    int64_t L = chunk_len(T, false);
    // long sequences whose 256-step chunks are few (64 to a replay workgroup, and the replays / sweeps are chains of dependent
    // steps whose length is the chunk's): chunks of 128 steps - twice the waves, half the chain
    const bool long_seq = T > kHmmLongFrom && L == kHmmLongChunk;
    // Short sequences (round 5): the one-level products path balances a sequential pass over T / L chunk products against
    // replay chains of L steps (T = 1e4: L = 64, 157 products one after the other - 0.25 ms - and two 64-step chains); with the
    // chunk start vectors from the forgetting pass the chunks can be short - 32 steps: a 32-step sweep and two 32-step replay
    // chains - and the products only run behind the gate.  While a failed gate holds the pass off, the formula's chunks.
    consume_gate(h, /*wait=*/false);
    const bool want_spec = h->spec_on && h->gate_dev != nullptr;
    const bool held = want_spec && h->spec_hold > 0;
    const bool short_spec = !long_seq && T >= kHmmShortForgetFrom && want_spec && !held;
    if (short_spec) L = std::min<int64_t>(kHmmLongChunk, std::max<int64_t>(kHmmShortChunk, h->chunk_floor));
    // (config 5 shape on one box: T = 4e5 2.64 -> 2.12 ms per iteration, 1e6 3.21 -> 2.61, 3e6 4.85 -> 4.56; at 8e6 the shorter
    // chunks lose, 9.79 -> 10.16: the limit is a replay workgroup per CU)
    // (round 5: down to 32 steps - with the products behind the gates a chunk only has to be long enough for the recursions to
    // forget their start, and the replays are chains of dependent steps: T = 1e5 ran 128-step chains on 13 workgroups)
    // ... and up again, for good, when a pass on short chunks did not stand (chunk_floor; consume_gate)
    if (long_seq)
        while (L > kHmmShortChunk && L > h->chunk_floor && (T - 1 + L - 1) / L < 64 * (int64_t)ws->num_cu) L /= 2;
    const int64_t n_chunks = T > 1 ? (T - 1 + L - 1) / L : 0;
    if (ws->e_state != 4)          // (4: the emission kernel has written rho' and mx itself)
        hipLaunchKernelGGL(hmm_prep_kernel, dim3((unsigned)((T + kPrepSteps - 1) / kPrepSteps)), dim3(256),
                           ((size_t)Kp * (kPrepSteps + 1) + kPrepSteps) * sizeof(double), st, ws->lnrho, ws->npad, T, K, Kp,
                           h->rho_tm, h->mx);
    const bool two_level = long_seq && n_chunks > 2 * kHmmSuper;
    const unsigned grid = (unsigned)((n_chunks + 4 * kReplayChunks - 1) / (4 * kReplayChunks));      // kReplayChunks chunks per wave, 4 waves per block
    // the xi sum inside the backward replay (one slab per replay wave) unless the slabs do not fit / developer switch
    // (up to 32 states: with three or four 16-state blocks the accumulators no longer fit beside the operator's registers)
    const bool xi_fused = KT <= 2 && n_chunks > 0 && kReplayChunks == 16 && (int64_t)grid * 4 <= h->xi_slab_cap && !h->xi_separate;
    auto replays = [&](const double* fs, const double* be, double* f_out, double* b_out, const int* gate) {
        hipLaunchKernelGGL((hmm_forward_replay_kernel<KT>), dim3(grid), dim3(256), 0, st, h->rho_tm, a_tilde, K, T, L,
                           n_chunks, fs, h->alpha_tm, h->cprime, 0, f_out, gate);
        if constexpr (KT <= 2 && kReplayChunks == 16) {
            if (xi_fused)
                hipLaunchKernelGGL((hmm_backward_replay_kernel<KT, true>), dim3(grid), dim3(256), 0, st, h->rho_tm, a_tilde, K, T, L,
                                   n_chunks, be, h->alpha_tm, h->cprime, h->gamma_tm, h->w_tm, h->xi_slabs, b_out, gate);
        }
        if (!xi_fused)
            hipLaunchKernelGGL((hmm_backward_replay_kernel<KT, false>), dim3(grid), dim3(256), 0, st, h->rho_tm, a_tilde, K, T, L,
                               n_chunks, be, h->alpha_tm, h->cprime, h->gamma_tm, h->w_tm, nullptr, b_out, gate);
    };
    // ---- the forgetting pass (round 4): chunk boundary vectors without the chunk products ------------------------------
    // The scaled recursions forget their start vector: started from the UNIFORM vector, a chunk of 256 steps of a sequence with
    // informative emissions ends in the same normalised alpha (beta~) as from the true one - to rounding.  So: a sweep of both
    // recursions over all chunks from uniform starts (K^2 per step, no stores) gives every chunk a start vector, the replays
    // run from those, and their own end vectors are compared with the sweeps': the difference IS (to first order) the error
    // of the start vectors used, and at <= 2e-13 RELATIVE per entry (kHmmForgetTol, a Hilbert-metric test) they stand.  Otherwise - slow mixing, flat emissions - the gate opens and
    // the products path below runs behind it (its kernels return at once while the gate is shut), replays included: the
    // result depends on the forgetting only through start vectors proven within chunks x 4e-13 of the exact ones in the Hilbert metric.  The products are T 2 K^3 flop (10 ms of f64 MFMA at config 5), the sweeps
    // two more passes of K^2 per step.  A call that needed the products holds the pass off for the next eight calls (the
    // gate is copied to pinned memory and looked at when the next call begins: no synchronisation).
    const int* gate = nullptr;
    // (a gate copy still in flight - a caller that does not synchronise between calls - only means the last outcome is not known yet)
    bool spec = (two_level || short_spec) && want_spec;
    if (held && (two_level || (!long_seq && T >= kHmmShortForgetFrom))) {
        --h->spec_hold;
        spec = false;
    }
    if (spec) {
        int* const gate_b = h->gate_dev;           // opens the chunk-product path
        int* const gate_a = h->gate_dev + 2;       // opens the whole-chunk stage ([1] is the Viterbi pass's)
        if (hipError_t eg = hipMemsetAsync(h->gate_dev, 0, sizeof(int), st); eg != hipSuccess) return eg;      // (the gates must be shut before the checks)
        if (hipError_t eg = hipMemsetAsync(gate_a, 0, sizeof(int), st); eg != hipSuccess) return eg;
        hipLaunchKernelGGL(hmm_alpha0_kernel, dim3(1), dim3(256), 0, st, h->rho_tm, pi_tilde, K, Kp, n_chunks, h->fstart, h->bend,
                           h->cprime);
        // first stage: the sweeps walk only the W steps next to every boundary (hmm.h, hmm_sweeps_kernel); if the replays' own
        // boundary vectors agree, done.  Otherwise gate_a opens the second stage - whole-chunk sweeps, replays, check -, and
        // only if that does not stand either gate_b opens the chunk products.
        int64_t W = std::min<int64_t>(L, h->sweep_len);
        if (W < L && h->short_hold > 0) {
            --h->short_hold;
            W = L;
        }
        h->gate_two_stage = W < L;
        h->gate_l = L;
        const int* stage_gate = nullptr;
        if (W < L) {
            hipLaunchKernelGGL((hmm_sweeps_kernel<KT>), dim3(grid, 2), dim3(256), 0, st, h->rho_tm, a_tilde, K, T, L, n_chunks,
                               h->fstart, h->bend, W, nullptr);
            replays(h->fstart, h->bend, h->fstart2, h->bend2, nullptr);
            hipLaunchKernelGGL(hmm_boundary_check_kernel<true>, dim3(kHmmCheckBlocks), dim3(256), 0, st, h->fstart, h->fstart2, h->bend, h->bend2,
                               (n_chunks - 1) * Kp, Kp, kHmmForgetTol, gate_a, nullptr);
            stage_gate = gate_a;
        }
        hipLaunchKernelGGL((hmm_sweeps_kernel<KT>), dim3(grid, 2), dim3(256), 0, st, h->rho_tm, a_tilde, K, T, L, n_chunks, h->fstart,
                           h->bend, L, stage_gate);
        replays(h->fstart, h->bend, h->fstart2, h->bend2, stage_gate);
        hipLaunchKernelGGL(hmm_boundary_check_kernel<true>, dim3(kHmmCheckBlocks), dim3(256), 0, st, h->fstart, h->fstart2, h->bend, h->bend2,
                           (n_chunks - 1) * Kp, Kp, kHmmForgetTol, gate_b, stage_gate);
        // (the pinned copies only steer the NEXT calls - hold the pass off after one that needed the products, the short stage
        // after one that needed whole chunks; if they cannot be made, the next calls simply try again)
        h->gate_pending = hipMemcpyAsync(h->gate_host, gate_b, sizeof(int), hipMemcpyDeviceToHost, st) == hipSuccess &&
                          hipMemcpyAsync(h->gate_host + 1, gate_a, sizeof(int), hipMemcpyDeviceToHost, st) == hipSuccess &&
                          hipEventRecord(h->gate_ev, st) == hipSuccess;
        gate = h->gate_dev;
    } else {
        h->last_gate = -1;
    }
    // ---- the products path (behind the gate, if the forgetting pass ran) -----------------------------------------------
    if (n_chunks > 0)
        hipLaunchKernelGGL((hmm_chunk_products_kernel<KT>), dim3((unsigned)((n_chunks + 3) / 4)), dim3(256), 0, st,
                           h->rho_tm, a_tilde, K, T, L, n_chunks, h->prod, gate);
    if (two_level) {
        const int64_t n_super = (n_chunks + kHmmSuper - 1) / kHmmSuper;
        hipLaunchKernelGGL((hmm_super_products_kernel<KT>), dim3((unsigned)n_super), dim3(256), 0, st, h->prod, n_chunks, h->qprod,
                           gate);
        hipLaunchKernelGGL((hmm_boundary_scan_kernel<KT>), dim3(1), dim3(128), 0, st, h->rho_tm, pi_tilde, h->qprod, K, n_super,
                           h->fstart_s, h->bend_s, h->cprime, h->alpha_tm, h->gamma_tm, h->w_tm, gate);
        hipLaunchKernelGGL((hmm_boundary_fill_kernel<KT>), dim3((unsigned)n_super), dim3(128), 0, st, h->prod, n_chunks,
                           h->fstart_s, h->bend_s, h->fstart, h->bend, gate);
    } else {
        hipLaunchKernelGGL((hmm_boundary_scan_kernel<KT>), dim3(1), dim3(128), 0, st, h->rho_tm, pi_tilde, h->prod, K,
                           n_chunks, h->fstart, h->bend, h->cprime, h->alpha_tm, h->gamma_tm, h->w_tm, gate);
    }
    if (n_chunks > 0) replays(h->fstart, h->bend, nullptr, nullptr, gate);
    h->w_valid = !xi_fused;
    // xi sum over t = 1 .. T-1
    int64_t n_waves = h->xi_waves;
    int64_t steps = T > 1 ? round_up((T - 1 + n_waves - 1) / n_waves, 4) : 4;
    n_waves = T > 1 && !xi_fused ? (T - 1 + steps - 1) / steps : 0;
    if (n_waves > 0)
        hipLaunchKernelGGL((hmm_xi_sum_kernel<KT>), dim3((unsigned)((n_waves + 3) / 4)), dim3(256), 0, st, h->alpha_tm,
                           h->w_tm, T, steps, h->xi_slabs);
    // waves of the last block beyond n_waves write slabs too (zeros): include them only if they exist
    const int64_t n_slabs = xi_fused ? (int64_t)grid * 4 : (n_waves > 0 ? ((n_waves + 3) / 4) * 4 : 0);
    const int n_part = (int)std::min<int64_t>(kLncBlocks, (T + 255) / 256);
    hipLaunchKernelGGL(hmm_lnc_partial_kernel, dim3(n_part), dim3(256), 0, st, h->cprime, h->mx, T, h->lnc_partial);
    hipLaunchKernelGGL(hmm_finish_kernel, dim3((unsigned)((K * K + 7) / 8)), dim3(256), 0, st, h->xi_slabs, n_slabs, a_tilde, K, Kp,
                       h->lnc_partial, n_part, T, h->gamma_tm, out);
    h->gamma_cm_valid = false;                 // (made on demand: hmm_ensure_gamma_cm)
    h->gamma_rows = T;
    return hipGetLastError();
}

// more than 64 states: one workgroup walks the sequence (hmm_generic.h); prep / finish / gamma transpose as above
template <typename Kern>
hipError_t seq_lds(Kern kern, size_t bytes) {
    return bytes > 48 * 1024 ? hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                   (int)bytes)
                             : hipSuccess;
}

// 65 .. 128 states, long sequences: chunk products / boundary pass / replays of hmm_wide.h between the generic prep and xi-sum
constexpr int64_t kHmmWideChunk = 256;
constexpr int64_t kHmmWideMinSteps = 2048;        // (shorter sequences: the sequential kernels - a handful of chunks fills nothing)

template <int KT>
hipError_t run_wide(gmmvb_workspace* ws, gmmvb_hmm_state* h, int64_t T, const double* pi_tilde, const double* a_tilde,
                    double* out, hipStream_t st) {
    const int K = h->K, Kp = h->Kp;
    // chunks of 256 steps, 64 to a replay workgroup (one per CU: the A fragments fill its LDS) - or of 128 steps while that leaves
    // CUs without a workgroup (T = 2e6: 123 workgroups of 256-step chunks on 256 CUs; the replays and sweeps are chains of
    // dependent steps, twice the workgroups halve them; chunks of 64 steps were too short for the forgetting pass at K = 128, D = 8)
    int64_t L = kHmmWideChunk;
    if ((T - 1 + L - 1) / L < 64 * (int64_t)ws->num_cu && (T - 1 + L / 2 - 1) / (L / 2) <= h->max_chunks) L /= 2;
    const int64_t n_chunks = (T - 1 + L - 1) / L;
    if (n_chunks > h->max_chunks) return hipErrorInvalidValue;
    const size_t fb = hmm_wide_frag_bytes<KT>();
    hipError_t e = seq_lds(hmm_chunk_products_wide_kernel<KT>, fb + 64);
    if (e == hipSuccess) e = seq_lds(hmm_forward_replay_wide_kernel<KT>, fb);
    if (e == hipSuccess) e = seq_lds(hmm_backward_replay_wide_kernel<KT>, fb);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(hmm_prep_generic_kernel, dim3((unsigned)((T + 63) / 64)), dim3(256), 0, st, ws->lnrho, ws->npad, T, K, Kp,
                       h->rho_tm, h->mx);
    const unsigned grid = (unsigned)((n_chunks + 63) / 64);          // 16 chunks per wave, 4 waves per workgroup
    auto replays = [&](const double* fs, const double* be, double* f_out, double* b_out, const int* gate) {
        hipLaunchKernelGGL((hmm_forward_replay_wide_kernel<KT>), dim3(grid), dim3(256), fb, st, h->rho_tm, a_tilde, K, T, L, n_chunks,
                           fs, h->alpha_tm, h->cprime, 0, f_out, gate);
        hipLaunchKernelGGL((hmm_backward_replay_wide_kernel<KT>), dim3(grid), dim3(256), fb, st, h->rho_tm, a_tilde, K, T, L, n_chunks,
                           be, h->alpha_tm, h->cprime, h->gamma_tm, h->w_tm, 0, b_out, gate);
    };
    // the forgetting pass (run<KT> above): here the products are 85 % of the iteration (T 2 Kp^3 flop)
    const bool two_level = n_chunks > 2 * kHmmSuper;
    const int* gate = nullptr;
    consume_gate(h, /*wait=*/false);
    bool spec = two_level && h->spec_on && h->gate_dev != nullptr;      // (a gate copy still in flight - a caller that does not synchronise between calls - only means the last outcome is not known yet)
    if (spec && h->spec_hold > 0) {
        --h->spec_hold;
        spec = false;
    }
    if (spec) {
        if (hipError_t eg = hipMemsetAsync(h->gate_dev, 0, sizeof(int), st); eg != hipSuccess) return eg;      // (the gate must be shut before the check)
        hipLaunchKernelGGL(hmm_alpha0_kernel, dim3(1), dim3(256), 0, st, h->rho_tm, pi_tilde, K, Kp, n_chunks, h->fstart, h->bend,
                           h->cprime);
        hipLaunchKernelGGL((hmm_forward_replay_wide_kernel<KT>), dim3(grid), dim3(256), fb, st, h->rho_tm, a_tilde, K, T, L, n_chunks,
                           h->fstart, h->alpha_tm, h->cprime, 1, h->fstart, nullptr);
        hipLaunchKernelGGL((hmm_backward_replay_wide_kernel<KT>), dim3(grid), dim3(256), fb, st, h->rho_tm, a_tilde, K, T, L, n_chunks,
                           h->bend, h->alpha_tm, h->cprime, h->gamma_tm, h->w_tm, 1, h->bend, nullptr);
        replays(h->fstart, h->bend, h->fstart2, h->bend2, nullptr);
        hipLaunchKernelGGL(hmm_boundary_check_kernel<true>, dim3(kHmmCheckBlocks), dim3(256), 0, st, h->fstart, h->fstart2, h->bend, h->bend2,
                           (n_chunks - 1) * Kp, Kp, kHmmForgetTol, h->gate_dev);
        // (the pinned copy only steers the NEXT calls - hold the pass off after one that needed the products; if it cannot be
        // made, they simply try the pass again)
        h->gate_pending = hipMemcpyAsync(h->gate_host, h->gate_dev, sizeof(int), hipMemcpyDeviceToHost, st) == hipSuccess &&
                          hipEventRecord(h->gate_ev, st) == hipSuccess;
        gate = h->gate_dev;
    } else {
        h->last_gate = -1;
    }
    hipLaunchKernelGGL((hmm_chunk_products_wide_kernel<KT>), dim3((unsigned)n_chunks), dim3(256), fb + 64, st, h->rho_tm, a_tilde, K,
                       T, L, n_chunks, h->prod, h->prod_t, gate);
    if (two_level) {
        // two levels: products of 64 chunk products, the sequential pass over those, every super-chunk fills in its own chunks
        const int64_t n_super = (n_chunks + kHmmSuper - 1) / kHmmSuper;
        e = seq_lds(hmm_super_products_wide_kernel<KT>, fb + 64);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((hmm_super_products_wide_kernel<KT>), dim3((unsigned)n_super), dim3(256), fb + 64, st, h->prod_t, n_chunks,
                           h->qprod, h->qprod_t, gate);
        hipLaunchKernelGGL((hmm_boundary_scan_wide_kernel<KT>), dim3(1, 2), dim3(kHmmWideScanThreads), 0, st, h->rho_tm, pi_tilde,
                           h->qprod, h->qprod_t, K, n_super, nullptr, nullptr, h->fstart_s, h->bend_s, h->cprime, h->alpha_tm,
                           h->gamma_tm, h->w_tm, gate);
        hipLaunchKernelGGL((hmm_boundary_scan_wide_kernel<KT>), dim3((unsigned)n_super, 2), dim3(kHmmWideScanThreads), 0, st, h->rho_tm,
                           pi_tilde, h->prod, h->prod_t, K, n_chunks, h->fstart_s, h->bend_s, h->fstart, h->bend, h->cprime,
                           h->alpha_tm, h->gamma_tm, h->w_tm, gate);
    } else {
        hipLaunchKernelGGL((hmm_boundary_scan_wide_kernel<KT>), dim3(1, 2), dim3(kHmmWideScanThreads), 0, st, h->rho_tm, pi_tilde,
                           h->prod, h->prod_t, K, n_chunks, nullptr, nullptr, h->fstart, h->bend, h->cprime, h->alpha_tm, h->gamma_tm,
                           h->w_tm);
    }
    replays(h->fstart, h->bend, nullptr, nullptr, gate);
    const int64_t steps = round_up((T - 1 + h->xi_waves - 1) / h->xi_waves, 4);
    const int64_t n_slabs = (T - 1 + steps - 1) / steps;
    hipLaunchKernelGGL((hmm_xi_sum_wide_kernel<KT>), dim3((unsigned)n_slabs), dim3(256), 0, st, h->alpha_tm, h->w_tm, T, steps,
                       h->xi_slabs);
    const int n_part = (int)std::min<int64_t>(kLncBlocks, (T + 255) / 256);
    hipLaunchKernelGGL(hmm_lnc_partial_kernel, dim3(n_part), dim3(256), 0, st, h->cprime, h->mx, T, h->lnc_partial);
    hipLaunchKernelGGL(hmm_finish_kernel, dim3((unsigned)((K * K + 7) / 8)), dim3(256), 0, st, h->xi_slabs, n_slabs, a_tilde, K, Kp,
                       h->lnc_partial, n_part, T, h->gamma_tm, out);
    h->w_valid = true;
    h->gamma_cm_valid = false;
    h->gamma_rows = T;
    return hipGetLastError();
}

hipError_t run_generic(gmmvb_workspace* ws, gmmvb_hmm_state* h, int64_t T, const double* pi_tilde, const double* a_tilde,
                       double* out, hipStream_t st) {
    const int K = h->K, Kp = h->Kp;
    const HmmSeqShape sh = hmm_seq_shape(K);
    hipError_t e = seq_lds(hmm_seq_forward_kernel, sh.lds_bytes);
    if (e == hipSuccess) e = seq_lds(hmm_seq_backward_kernel, sh.lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(hmm_prep_generic_kernel, dim3((unsigned)((T + 63) / 64)), dim3(256), 0, st, ws->lnrho, ws->npad, T, K, Kp,
                       h->rho_tm, h->mx);
    hipLaunchKernelGGL(hmm_transpose_kernel, dim3((unsigned)((K * K + 255) / 256)), dim3(256), 0, st, a_tilde, K, h->a_t);
    // The forgetting pass (run<KT>): the same two kernels in their chunk form, a workgroup per 256 steps - sweeps from the
    // uniform vector, replays from the sweeps' vectors, the test - and the walk of the whole sequence by ONE workgroup
    // (seconds per million steps) only behind the gate.  Sequences of at least 64 chunks.
    const int64_t L = kHmmGenericChunk;
    const int64_t n_chunks = T > 1 ? (T - 1 + L - 1) / L : 0;
    const int* gate = nullptr;
    consume_gate(h, /*wait=*/false);
    bool spec = h->spec_on && h->gate_dev != nullptr && n_chunks >= 64 && n_chunks <= h->vec_chunks;
    if (spec && h->spec_hold > 0) {
        --h->spec_hold;
        spec = false;
    }
    if (spec) {
        const unsigned g = (unsigned)n_chunks;
        if (hipError_t eg = hipMemsetAsync(h->gate_dev, 0, sizeof(int), st); eg != hipSuccess) return eg;      // (the gate must be shut before the check)
        hipLaunchKernelGGL(hmm_seq_forward_kernel, dim3(g), dim3(kHmmSeqThreads), sh.lds_bytes, st, h->rho_tm, pi_tilde, a_tilde, K,
                           Kp, T, sh.P, sh.J, sh.mat_in_lds, h->alpha_tm, h->cprime, h->gamma_tm, h->w_tm, L, h->fstart, 1, h->fstart,
                           nullptr);
        hipLaunchKernelGGL(hmm_seq_backward_kernel, dim3(g), dim3(kHmmSeqThreads), sh.lds_bytes, st, h->rho_tm, h->a_t, K, Kp, T,
                           sh.P, sh.J, sh.mat_in_lds, h->alpha_tm, h->cprime, h->gamma_tm, h->w_tm, L, h->bend, 1, h->bend, nullptr);
        hipLaunchKernelGGL(hmm_seq_forward_kernel, dim3(g), dim3(kHmmSeqThreads), sh.lds_bytes, st, h->rho_tm, pi_tilde, a_tilde, K,
                           Kp, T, sh.P, sh.J, sh.mat_in_lds, h->alpha_tm, h->cprime, h->gamma_tm, h->w_tm, L, h->fstart, 0, h->fstart2,
                           nullptr);
        hipLaunchKernelGGL(hmm_seq_backward_kernel, dim3(g), dim3(kHmmSeqThreads), sh.lds_bytes, st, h->rho_tm, h->a_t, K, Kp, T,
                           sh.P, sh.J, sh.mat_in_lds, h->alpha_tm, h->cprime, h->gamma_tm, h->w_tm, L, h->bend, 0, h->bend2, nullptr);
        hipLaunchKernelGGL(hmm_boundary_check_kernel<true>, dim3(kHmmCheckBlocks), dim3(256), 0, st, h->fstart, h->fstart2, h->bend, h->bend2,
                           (n_chunks - 1) * Kp, Kp, kHmmForgetTol, h->gate_dev);
        // (the pinned copy only steers the NEXT calls - hold the pass off after one that needed the products; if it cannot be
        // made, they simply try the pass again)
        h->gate_pending = hipMemcpyAsync(h->gate_host, h->gate_dev, sizeof(int), hipMemcpyDeviceToHost, st) == hipSuccess &&
                          hipEventRecord(h->gate_ev, st) == hipSuccess;
        gate = h->gate_dev;
    } else {
        h->last_gate = -1;
    }
    hipLaunchKernelGGL(hmm_seq_forward_kernel, dim3(1), dim3(kHmmSeqThreads), sh.lds_bytes, st, h->rho_tm, pi_tilde, a_tilde, K, Kp,
                       T, sh.P, sh.J, sh.mat_in_lds, h->alpha_tm, h->cprime, h->gamma_tm, h->w_tm, 0, nullptr, 0, nullptr, gate);
    hipLaunchKernelGGL(hmm_seq_backward_kernel, dim3(1), dim3(kHmmSeqThreads), sh.lds_bytes, st, h->rho_tm, h->a_t, K, Kp, T, sh.P,
                       sh.J, sh.mat_in_lds, h->alpha_tm, h->cprime, h->gamma_tm, h->w_tm, 0, nullptr, 0, nullptr, gate);
    int64_t n_slabs = 0;
    if (T > 1) {
        const int64_t steps = (T - 1 + h->xi_waves - 1) / h->xi_waves;
        n_slabs = (T - 1 + steps - 1) / steps;
        hipLaunchKernelGGL(hmm_xi_generic_kernel, dim3((unsigned)n_slabs, (unsigned)((Kp / 16) * (Kp / 16))), dim3(256), 0, st,
                           h->alpha_tm, h->w_tm, Kp, T, steps, h->xi_slabs);
    }
    const int n_part = (int)std::min<int64_t>(kLncBlocks, (T + 255) / 256);
    hipLaunchKernelGGL(hmm_lnc_partial_kernel, dim3(n_part), dim3(256), 0, st, h->cprime, h->mx, T, h->lnc_partial);
    hipLaunchKernelGGL(hmm_finish_kernel, dim3((unsigned)((K * K + 7) / 8)), dim3(256), 0, st, h->xi_slabs, n_slabs, a_tilde, K, Kp,
                       h->lnc_partial, n_part, T, h->gamma_tm, out);
    h->w_valid = true;
    h->gamma_cm_valid = false;
    h->gamma_rows = T;
    return hipGetLastError();
}

}  // namespace

extern "C" {

int64_t hmmvb_out_len(int K) { return K < 1 ? -1 : (int64_t)K * K + 2 * (int64_t)K + 1; }

int hmmvb_enable(gmmvb_workspace* ws) {
    if (!ws) return fail(GMMVB_EINVAL, "null argument");
    if (ws->hmm) return GMMVB_OK;
    if (ws->scratch && ws->scratch->refs > 1)
        return fail(GMMVB_EUNSUPPORTED, "HMM: the time axis does not tile (this workspace belongs to a tile group)");
    if (ws->K > 65535) return fail(GMMVB_EUNSUPPORTED, "HMM: at most 65535 states (16-bit back-pointers)");
    gmmvb_hmm_state* h = new (std::nothrow) gmmvb_hmm_state();
    if (!h) return fail(GMMVB_ENOMEM, "host allocation failed");
    h->K = ws->K;
    h->KT = (ws->K + 15) / 16;
    h->Kp = 16 * h->KT;
    h->npad = ws->npad;
    h->generic = ws->K > 64;                    // (the chunk-parallel kernels hold K x K products in registers)
    h->wide = ws->K > 64 && ws->K <= 128 && dev_env("GMMVB_HMM_WIDE_OFF") == nullptr;      // hmm_wide.h (developer switch: off)
    // (65 .. 128 states: chunks of 256 steps, or of 128 while those do not fill the CUs - run_wide)
    h->max_chunks = h->wide ? std::min<int64_t>(std::max<int64_t>(64 * (int64_t)ws->num_cu, ws->npad / kHmmWideChunk), ws->npad / 128) + 2
                            : (h->generic ? 1 : ws->npad / 16 + 2);          // chunk_len >= 16
    h->xi_waves = h->generic ? std::max<int64_t>(16, std::min<int64_t>(4 * (int64_t)ws->num_cu, (int64_t(1) << 27) / ((int64_t)h->Kp * h->Kp)))
                             : 16 * (int64_t)ws->num_cu;       // four xi-sum waves per SIMD: the kernel streams two [T][Kp] arrays and a wave
                                                               // has one load group in flight (round 4; one wave per SIMD: 1.9 ms, 2.6 TB/s)
    // room for one xi slab per replay wave (hmm.h H5 XI): sequences past 2^15 steps have chunks of kHmmLongChunk steps, 16 to a
    // wave; shorter ones at most ~850 chunks (chunk_len)
    h->xi_slab_cap = std::max<int64_t>(h->xi_waves + 4, h->generic ? 0 : h->npad / (16 * (kHmmLongChunk / 2)) + 72);
    h->xi_separate = dev_env("GMMVB_HMM_XI_SEPARATE") != nullptr;
    const int64_t tk = h->npad * h->Kp;
    // chunk boundary vectors: more than 128 states walk chunks of kHmmGenericChunk steps in the forgetting pass (run_generic)
    const int64_t vec_chunks = (h->generic && !h->wide) ? h->npad / kHmmGenericChunk + 2 : h->max_chunks;
    h->vec_chunks = vec_chunks;
    struct { double** p; int64_t n; } bufs[] = {
        {&h->rho_tm, tk}, {&h->alpha_tm, tk}, {&h->gamma_tm, tk}, {&h->w_tm, tk},
        {&h->gamma_cm, (int64_t)ws->K * h->npad}, {&h->mx, h->npad}, {&h->cprime, h->npad},
        {&h->prod, h->max_chunks * h->Kp * h->Kp}, {&h->fstart, vec_chunks * h->Kp},
        {&h->bend, vec_chunks * h->Kp}, {&h->xi_slabs, h->xi_slab_cap * h->Kp * h->Kp},
        {&h->lnc_partial, kLncBlocks},
        {&h->qprod, (h->max_chunks / kHmmSuper + 2) * h->Kp * h->Kp}, {&h->fstart_s, (h->max_chunks / kHmmSuper + 2) * h->Kp},
        {&h->bend_s, (h->max_chunks / kHmmSuper + 2) * h->Kp}, {&h->a_t, h->generic ? (int64_t)ws->K * ws->K : 0},
        {&h->prod_t, h->wide ? h->max_chunks * h->Kp * h->Kp : 0},
        {&h->qprod_t, h->wide ? (h->max_chunks / kHmmSuper + 2) * h->Kp * h->Kp : 0},
        {&h->fstart2, vec_chunks * h->Kp}, {&h->bend2, vec_chunks * h->Kp}};
    for (auto& b : bufs) {
        if (b.n == 0) continue;
        hipError_t e = hipMalloc((void**)b.p, (size_t)b.n * sizeof(double));
        if (e != hipSuccess) {
            hmm_state_destroy(h);
            return fail(GMMVB_ENOMEM, "hipMalloc (HMM buffers)", e);
        }
        h->bytes += b.n * (int64_t)sizeof(double);
    }
    hipError_t e2 = h->generic ? hipMalloc((void**)&h->phi16, (size_t)(h->npad * ws->K) * sizeof(unsigned short))
                               : hipMalloc((void**)&h->phi, (size_t)(h->npad * h->Kp));
    if (e2 == hipSuccess && h->wide) {          // (65 .. 128 states: byte back-pointers of the chunked Viterbi pass)
        e2 = hipMalloc((void**)&h->phi, (size_t)(h->npad * h->Kp));
        h->bytes += h->npad * h->Kp;
    }
    if (e2 == hipSuccess) e2 = hipMalloc((void**)&h->last_state, sizeof(int));
    if (e2 == hipSuccess && h->generic && !h->wide) {        // (their padding entries are never written and are compared)
        for (double* p : {h->fstart, h->bend, h->fstart2, h->bend2})
            if (e2 == hipSuccess) e2 = hipMemset(p, 0, (size_t)(vec_chunks * h->Kp) * sizeof(double));
    }
    if (e2 == hipSuccess) {      // the forgetting pass's gate (run<KT>, run_wide, run_generic): device flag, pinned copy, event
        h->spec_on = dev_env("GMMVB_HMM_FORGETTING_OFF") == nullptr;
        if (const char* v = dev_env("GMMVB_HMM_SWEEP_LEN")) h->sweep_len = std::max<int64_t>(1, std::atoll(v));      // developer switch
        e2 = hipMalloc((void**)&h->gate_dev, 4 * sizeof(int));      // [0] forward-backward (chunk products), [1] Viterbi, [2] forward-backward (whole-chunk stage)
        if (e2 == hipSuccess) e2 = hipHostMalloc((void**)&h->gate_host, 2 * sizeof(int));
        if (e2 == hipSuccess) e2 = hipEventCreateWithFlags(&h->gate_ev, hipEventDisableTiming);
        if (e2 == hipSuccess) h->gate_host[0] = h->gate_host[1] = 0;
    }
    if (e2 != hipSuccess) {
        hmm_state_destroy(h);
        return fail(GMMVB_ENOMEM, "hipMalloc (Viterbi buffers)", e2);
    }
    h->bytes += (h->generic ? 2 * h->npad * ws->K : h->npad * h->Kp) + 4;
    ws->hmm = h;
    ws->bytes += h->bytes;
    return GMMVB_OK;
}

int hmmvb_viterbi(gmmvb_workspace* ws, int64_t n_rows, const double* ln_pi_tilde_dev, const double* ln_a_tilde_dev,
                  int32_t* z_dev, void* stream) {
    if (!ws || !ln_pi_tilde_dev || !ln_a_tilde_dev || !z_dev) return fail(GMMVB_EINVAL, "null argument");
    if (!ws->hmm) return fail(GMMVB_ESTATE, "hmmvb_enable has not been called");
    if (ws->e_state == 4)
        return fail(GMMVB_ESTATE, "the last gmmvb_estep formed no ln rho array (hmmvb_emission_target 1): run it with target 0");
    if (ws->e_state != 1 || ws->e_rows != n_rows)
        return fail(GMMVB_ESTATE, "no emission ln rho for these rows: call gmmvb_estep first");
    gmmvb_hmm_state* h = ws->hmm;
    hipStream_t st = (hipStream_t)stream;
    h->vit_coalesced = false;
    // the pass takes the forward-backward pass's buffers as scratch (chunk products, boundary vectors, xi slabs, w_tm): whatever
    // that pass left is gone - said explicitly, not only through e_state (hmmvb_readout checks gamma_rows)
    h->w_valid = false;
    h->gamma_rows = 0;
    h->gamma_cm_valid = false;
    if (h->wide && h->phi && n_rows >= kHmmWideMinSteps) {
        // 65 .. 128 states: the chunked max-plus pass with two end states per lane and ln a~ in LDS (hmm_wide.h); scratch as below
        const int64_t L = kHmmWideChunk;
        const int64_t chunks = (n_rows - 1 + L - 1) / L;
        double* M = h->prod;
        double* wstart = h->fstart;
        unsigned char* map = reinterpret_cast<unsigned char*>(h->bend);
        int* endst = reinterpret_cast<int*>(h->fstart_s);
        if (chunks > h->max_chunks || chunks * (int64_t)sizeof(int) > (h->max_chunks / kHmmSuper + 2) * h->Kp * (int64_t)sizeof(double))
            return fail(GMMVB_ESTATE, "Viterbi scratch too small for this sequence");
        const size_t lds = (size_t)h->Kp * h->Kp * sizeof(double);
        hipError_t ew = hipSuccess;
        // (the coalescence pass of the narrow path below, same kernels' wide twins)
        const int* vgate = nullptr;
        const bool coalesce = chunks >= 64 && h->spec_on && h->gate_dev != nullptr && h->fstart2 != nullptr;
        h->vit_coalesced = coalesce;
        const unsigned rgrid = (unsigned)((chunks + kVitWideWaves - 1) / kVitWideWaves);
#define VITW(KTT)                                                                                                              \
    ew = seq_lds(hmm_vit_chunk_wide_kernel<KTT>, lds);                                                                         \
    if (ew == hipSuccess) ew = seq_lds(hmm_vit_replay_wide_kernel<KTT>, lds);                                                  \
    if (ew == hipSuccess && coalesce) {                                                                                        \
        vgate = h->gate_dev + 1;                                                                                               \
        ew = hipMemsetAsync(h->gate_dev + 1, 0, sizeof(int), st);                                                              \
        hipLaunchKernelGGL(hmm_vit_omega0_kernel, dim3(1), dim3(128), 0, st, ws->lnrho, ws->npad, ln_pi_tilde_dev, h->K, h->Kp, wstart); \
        hipLaunchKernelGGL((hmm_vit_replay_wide_kernel<KTT>), dim3(rgrid), dim3(64 * kVitWideWaves), lds, st, ws->lnrho,        \
                           ws->npad, ln_a_tilde_dev, wstart, h->K, n_rows, L, chunks, h->phi, h->last_state, 1, wstart, nullptr); \
        hipLaunchKernelGGL((hmm_vit_replay_wide_kernel<KTT>), dim3(rgrid), dim3(64 * kVitWideWaves), lds, st, ws->lnrho,        \
                           ws->npad, ln_a_tilde_dev, wstart, h->K, n_rows, L, chunks, h->phi, h->last_state, 0, h->fstart2, nullptr); \
        hipLaunchKernelGGL(hmm_boundary_check_kernel<false>, dim3(kHmmCheckBlocks), dim3(256), 0, st, wstart, h->fstart2, wstart, wstart,           \
                           (chunks - 1) * h->Kp, h->Kp, kVitCoalesceTol, h->gate_dev + 1);                                                \
    }                                                                                                                          \
    if (ew == hipSuccess) {                                                                                                    \
        hipLaunchKernelGGL((hmm_vit_chunk_wide_kernel<KTT>), dim3((unsigned)chunks, (unsigned)((h->K + 4 * kVitWideWaves - 1) / (4 * kVitWideWaves))), dim3(64 * kVitWideWaves), lds, st, \
                           ws->lnrho, ws->npad, ln_a_tilde_dev, h->K, n_rows, L, M, vgate);                                    \
        hipLaunchKernelGGL((hmm_vit_scan_wide_kernel<KTT>), dim3(1), dim3(kHmmWideScanThreads), 0, st, ws->lnrho, ws->npad,     \
                           ln_pi_tilde_dev, M, h->K, chunks, wstart, vgate);                                                   \
        hipLaunchKernelGGL((hmm_vit_replay_wide_kernel<KTT>), dim3(rgrid), dim3(64 * kVitWideWaves), lds, st, ws->lnrho,        \
                           ws->npad, ln_a_tilde_dev, wstart, h->K, n_rows, L, chunks, h->phi, h->last_state, 0, nullptr, vgate); \
    }
        switch (h->KT) {
            case 5: VITW(5) break;
            case 6: VITW(6) break;
            case 7: VITW(7) break;
            default: VITW(8) break;
        }
#undef VITW
        if (ew != hipSuccess) return fail(GMMVB_EHIP, "chunked viterbi (65 .. 128 states: LDS size)", ew);
        hipLaunchKernelGGL(hmm_vit_backmap_kernel, dim3((unsigned)chunks), dim3(128), 0, st, h->phi, h->Kp, n_rows, L, map);
        hipLaunchKernelGGL(hmm_vit_backscan_kernel, dim3(1), dim3(256), 0, st, map, h->Kp, chunks, h->last_state, endst);
        hipLaunchKernelGGL(hmm_vit_fill_kernel, dim3((unsigned)((chunks + 63) / 64)), dim3(64), 0, st, h->phi, h->Kp, n_rows, L,
                           chunks, endst, z_dev);
        ew = hipGetLastError();
        if (ew != hipSuccess) return fail(GMMVB_EHIP, "chunked viterbi launch (65 .. 128 states)", ew);
        return GMMVB_OK;
    }
    if (h->generic) {
        const HmmSeqShape sh = hmm_seq_shape(h->K);
        hipError_t eg = seq_lds(hmm_seq_viterbi_kernel, sh.lds_bytes);
        if (eg != hipSuccess) return fail(GMMVB_EHIP, "viterbi (LDS size)", eg);
        // chunk form through the coalescence pass (the narrow path below), the walk of the whole sequence by one workgroup behind
        // its gate; the path is traced back chunk-parallel either way
        const int64_t L = kHmmGenericChunk;
        const int64_t chunks = n_rows > 1 ? (n_rows - 1 + L - 1) / L : 0;
        const bool chunked = chunks >= 2 && chunks <= h->vec_chunks && h->fstart2 != nullptr;
        const bool coalesce = chunked && chunks >= 64 && h->spec_on && h->gate_dev != nullptr;
        h->vit_coalesced = coalesce;
        const int* vgate = nullptr;
        if (coalesce) {
            vgate = h->gate_dev + 1;
            if (hipMemsetAsync(h->gate_dev + 1, 0, sizeof(int), st) != hipSuccess) return fail(GMMVB_EHIP, "viterbi (gate reset)");
            hipLaunchKernelGGL(hmm_seq_viterbi_kernel, dim3((unsigned)chunks), dim3(kHmmSeqThreads), sh.lds_bytes, st, ws->lnrho, ws->npad,
                               ln_pi_tilde_dev, ln_a_tilde_dev, h->K, n_rows, sh.P, sh.J, sh.mat_in_lds, h->phi16, h->last_state, L, h->Kp,
                               h->fstart, 1, h->fstart, nullptr);
            hipLaunchKernelGGL(hmm_seq_viterbi_kernel, dim3((unsigned)chunks), dim3(kHmmSeqThreads), sh.lds_bytes, st, ws->lnrho, ws->npad,
                               ln_pi_tilde_dev, ln_a_tilde_dev, h->K, n_rows, sh.P, sh.J, sh.mat_in_lds, h->phi16, h->last_state, L, h->Kp,
                               h->fstart, 0, h->fstart2, nullptr);
            hipLaunchKernelGGL(hmm_boundary_check_kernel<false>, dim3(kHmmCheckBlocks), dim3(256), 0, st, h->fstart, h->fstart2, h->fstart, h->fstart,
                               (chunks - 1) * h->Kp, h->Kp, kVitCoalesceTol, h->gate_dev + 1);
        }
        hipLaunchKernelGGL(hmm_seq_viterbi_kernel, dim3(1), dim3(kHmmSeqThreads), sh.lds_bytes, st, ws->lnrho, ws->npad,
                           ln_pi_tilde_dev, ln_a_tilde_dev, h->K, n_rows, sh.P, sh.J, sh.mat_in_lds, h->phi16, h->last_state, 0, h->Kp,
                           nullptr, 0, nullptr, vgate);
        if (chunked) {
            // scratch: the forward-backward pass's w array ([npad][Kp] doubles; nothing reads it once gmmvb_estep has run again)
            unsigned short* map = reinterpret_cast<unsigned short*>(h->w_tm);        // [chunks][K]
            int* endst = reinterpret_cast<int*>(h->w_tm + (chunks * (int64_t)h->K + 3) / 4 + 1);      // [chunks]
            hipLaunchKernelGGL(hmm_seq_backmap_kernel, dim3((unsigned)chunks), dim3(256), 0, st, h->phi16, h->K, n_rows, L, map);
            hipLaunchKernelGGL(hmm_seq_backscan_kernel, dim3(1), dim3(64), 0, st, map, h->K, chunks, h->last_state, endst);
            hipLaunchKernelGGL(hmm_seq_fill_kernel, dim3((unsigned)((chunks + 63) / 64)), dim3(64), 0, st, h->phi16, h->K, n_rows, L,
                               chunks, endst, z_dev);
        } else {
            hipLaunchKernelGGL(hmm_seq_backtrack_kernel, dim3(1), dim3(64), 0, st, h->phi16, h->K, n_rows, h->last_state, z_dev);
        }
        eg = hipGetLastError();
        if (eg != hipSuccess) return fail(GMMVB_EHIP, "viterbi launch (generic)", eg);
        return GMMVB_OK;
    }
    // Long sequences: chunked max-plus scan (hmm.h, hmm_vit_*); its scratch is the forward-backward pass's (chunk products,
    // boundary vectors), which nothing reads once that pass has returned.  Short ones: the single sequential wave.
    const int64_t L = n_rows >= 65536 ? 256 : (n_rows >= 512 ? 32 : 0);
    if (L > 0) {
        const int64_t chunks = (n_rows - 1 + L - 1) / L;
        double* M = h->prod;                                             // [chunks][Kp][Kp]   (chunks <= max_chunks)
        double* wstart = h->fstart;                                      // [chunks][Kp]
        unsigned char* map = reinterpret_cast<unsigned char*>(h->bend);  // [chunks][Kp] bytes
        int* endst = reinterpret_cast<int*>(h->fstart_s);                // [chunks] ints
        if (chunks > h->max_chunks || chunks * (int64_t)sizeof(int) > (h->max_chunks / kHmmSuper + 2) * h->Kp * (int64_t)sizeof(double))
            return fail(GMMVB_ESTATE, "Viterbi scratch too small for this sequence");
        // chunk matrices: up to 32 states with a lane per start state (hmm_vit_chunk_lane_kernel), beyond with a wave per
        // (chunk, start state).  Chunk starts: with more than two super-chunks through super-chunk products (a sequential
        // pass over chunks / 64 products instead of over every chunk), else the single workgroup's pass.
        const int64_t supers = (chunks + kHmmSuper - 1) / kHmmSuper;
        const bool two_level = supers > 2 && h->qprod != nullptr && h->bend_s != nullptr;
        double* a_pad = h->xi_slabs;                                     // [Kp][Kp] (the forward-backward pass's slabs are free here)
        double* sstart = h->bend_s;                                      // [supers][Kp]
        // The coalescence pass (the max-plus twin of the forward-backward pass's forgetting, run<KT>): the best paths from all
        // start states of a chunk of 256 steps normally merge inside it, and then omega behind the chunk - minus its maximum -
        // does not depend on the chunk's start vector.  A sweep of the replay kernel from zero start vectors (no
        // back-pointers stored) gives every chunk a start vector, the replay runs from those and its own end vectors are
        // compared with the sweep's: equal to 2e-10 nats (kVitCoalesceTol), the back-pointers stand (only differences of omega enter them);
        // otherwise the gate opens and the chunk-matrix path below runs behind it, replay included.
        const int* vgate = nullptr;
        const bool coalesce = L == 256 && chunks >= 64 && h->spec_on && h->gate_dev != nullptr && h->fstart2 != nullptr;
        h->vit_coalesced = coalesce;
#define VITC(KTT)                                                                                                           \
    if (coalesce) {                                                                                                         \
        vgate = h->gate_dev + 1;                                                                                            \
        if (hipMemsetAsync(h->gate_dev + 1, 0, sizeof(int), st) != hipSuccess) return fail(GMMVB_EHIP, "viterbi (gate reset)"); \
        hipLaunchKernelGGL(hmm_vit_omega0_kernel, dim3(1), dim3(64), 0, st, ws->lnrho, ws->npad, ln_pi_tilde_dev, h->K, h->Kp, wstart); \
        hipLaunchKernelGGL((hmm_vit_replay_kernel<KTT>), dim3((unsigned)chunks), dim3(64), 0, st, ws->lnrho, ws->npad,       \
                           ln_a_tilde_dev, wstart, h->K, n_rows, L, chunks, h->phi, h->last_state, 1, wstart, nullptr);     \
        hipLaunchKernelGGL((hmm_vit_replay_kernel<KTT>), dim3((unsigned)chunks), dim3(64), 0, st, ws->lnrho, ws->npad,       \
                           ln_a_tilde_dev, wstart, h->K, n_rows, L, chunks, h->phi, h->last_state, 0, h->fstart2, nullptr);  \
        hipLaunchKernelGGL(hmm_boundary_check_kernel<false>, dim3(kHmmCheckBlocks), dim3(256), 0, st, wstart, h->fstart2, wstart, wstart,        \
                           (chunks - 1) * h->Kp, h->Kp, kVitCoalesceTol, h->gate_dev + 1);                                             \
    }                                                                                                                       \
    if (KTT <= 2) {                                                                                                         \
        constexpr int KPL = KTT <= 1 ? 16 : 32;                                                                             \
        hipLaunchKernelGGL(hmm_vit_pad_kernel, dim3((KPL * KPL + 255) / 256), dim3(256), 0, st, ln_a_tilde_dev, h->K, KPL, a_pad); \
        hipLaunchKernelGGL((hmm_vit_chunk_lane_kernel<KPL>), dim3((unsigned)((chunks + 64 / KPL - 1) / (64 / KPL))), dim3(64), 0, \
                           st, ws->lnrho, ws->npad, a_pad, h->K, n_rows, L, chunks, M, vgate);                              \
    } else {                                                                                                                \
        hipLaunchKernelGGL((hmm_vit_chunk_kernel<KTT>), dim3((unsigned)chunks, (unsigned)((h->K + 3) / 4)), dim3(256), 0, st, \
                           ws->lnrho, ws->npad, ln_a_tilde_dev, h->K, n_rows, L, M, vgate);                                 \
    }                                                                                                                       \
    if (two_level) {                                                                                                        \
        hipLaunchKernelGGL((hmm_vit_super_kernel<16 * KTT>), dim3((unsigned)supers), dim3(256), 0, st, M, h->K, chunks, h->qprod, vgate); \
        hipLaunchKernelGGL((hmm_vit_scan2_kernel<16 * KTT>), dim3(1), dim3(64), 0, st, ws->lnrho, ws->npad, ln_pi_tilde_dev, \
                           h->qprod, h->K, supers, sstart, vgate);                                                          \
        hipLaunchKernelGGL((hmm_vit_fill2_kernel<16 * KTT>), dim3((unsigned)supers), dim3(64), 0, st, M, h->K, chunks, sstart, \
                           wstart, vgate);                                                                                  \
    } else {                                                                                                                \
        hipLaunchKernelGGL((hmm_vit_scan_kernel<KTT>), dim3(1), dim3(256), 0, st, ws->lnrho, ws->npad, ln_pi_tilde_dev, M, h->K, \
                           chunks, wstart, vgate);                                                                          \
    }                                                                                                                       \
    hipLaunchKernelGGL((hmm_vit_replay_kernel<KTT>), dim3((unsigned)chunks), dim3(64), 0, st, ws->lnrho, ws->npad,           \
                       ln_a_tilde_dev, wstart, h->K, n_rows, L, chunks, h->phi, h->last_state, 0, nullptr, vgate)
        switch (h->KT) {
            case 1: VITC(1); break;
            case 2: VITC(2); break;
            case 3: VITC(3); break;
            default: VITC(4); break;
        }
#undef VITC
        hipLaunchKernelGGL(hmm_vit_backmap_kernel, dim3((unsigned)chunks), dim3(64), 0, st, h->phi, h->Kp, n_rows, L, map);
        hipLaunchKernelGGL(hmm_vit_backscan_kernel, dim3(1), dim3(256), 0, st, map, h->Kp, chunks, h->last_state, endst);
        hipLaunchKernelGGL(hmm_vit_fill_kernel, dim3((unsigned)((chunks + 63) / 64)), dim3(64), 0, st, h->phi, h->Kp, n_rows, L,
                           chunks, endst, z_dev);
        hipError_t ec = hipGetLastError();
        if (ec != hipSuccess) return fail(GMMVB_EHIP, "chunked viterbi launch", ec);
        return GMMVB_OK;
    }
#define VIT(KTT)                                                                                                     \
    hipLaunchKernelGGL((hmm_viterbi_forward_kernel<KTT>), dim3(1), dim3(64), 0, st, ws->lnrho, ws->npad, ln_pi_tilde_dev, \
                       ln_a_tilde_dev, h->K, n_rows, h->phi, h->last_state)
    switch (h->KT) {
        case 1: VIT(1); break;
        case 2: VIT(2); break;
        case 3: VIT(3); break;
        default: VIT(4); break;
    }
#undef VIT
    hipLaunchKernelGGL(hmm_viterbi_backtrack_kernel, dim3(1), dim3(256), 0, st, h->phi, h->Kp, n_rows, h->last_state,
                       z_dev);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(GMMVB_EHIP, "viterbi launch", e);
    return GMMVB_OK;
}

int hmmvb_last_viterbi_pass(gmmvb_workspace* ws) {
    if (!ws || !ws->hmm) return -2;
    if (!ws->hmm->vit_coalesced) return -1;
    int g = 0;
    if (hipMemcpy(&g, ws->hmm->gate_dev + 1, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return -2;      // (synchronises)
    return g;
}

int hmmvb_last_boundary_pass(gmmvb_workspace* ws) {
    if (!ws || !ws->hmm) return -2;
    gmmvb_hmm_state* h = ws->hmm;
    if (!consume_gate(h, /*wait=*/true)) return -2;
    return h->last_gate;
}

int hmmvb_skip_h(gmmvb_workspace* ws, int skip) {
    if (!ws) return fail(GMMVB_EINVAL, "null argument");
    if (!ws->hmm) return fail(GMMVB_ESTATE, "hmmvb_enable has not been called");
    ws->hmm_skip_h = skip != 0;
    return GMMVB_OK;
}

int hmmvb_emission_target(gmmvb_workspace* ws, int fused, int* in_effect) {
    if (!ws) return fail(GMMVB_EINVAL, "null argument");
    if (!ws->hmm) return fail(GMMVB_ESTATE, "hmmvb_enable has not been called");
    ws->hmm->fuse_emission = fused != 0 && ws->T == 1 && !ws->wide && dev_env("GMMVB_HMM_FUSED_EMISSION_OFF") == nullptr;
    if (in_effect) *in_effect = hmm_fused_emission(ws->hmm) ? 1 : 0;
    return GMMVB_OK;
}

int hmmvb_forward_backward(gmmvb_workspace* ws, int64_t n_rows, const double* pi_tilde_dev, const double* a_tilde_dev,
                           double* out_dev, void* stream) {
    if (!ws || !pi_tilde_dev || !a_tilde_dev || !out_dev) return fail(GMMVB_EINVAL, "null argument");
    if (!ws->hmm) return fail(GMMVB_ESTATE, "hmmvb_enable has not been called");
    if ((ws->e_state != 1 && ws->e_state != 4) || ws->e_rows != n_rows)
        return fail(GMMVB_ESTATE, "no emission ln rho for these rows: call gmmvb_estep first");
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipSuccess;
    if (ws->hmm->wide && n_rows >= kHmmWideMinSteps) {
        switch (ws->hmm->KT) {
            case 5: e = run_wide<5>(ws, ws->hmm, n_rows, pi_tilde_dev, a_tilde_dev, out_dev, st); break;
            case 6: e = run_wide<6>(ws, ws->hmm, n_rows, pi_tilde_dev, a_tilde_dev, out_dev, st); break;
            case 7: e = run_wide<7>(ws, ws->hmm, n_rows, pi_tilde_dev, a_tilde_dev, out_dev, st); break;
            default: e = run_wide<8>(ws, ws->hmm, n_rows, pi_tilde_dev, a_tilde_dev, out_dev, st); break;
        }
        if (e != hipSuccess) return fail(GMMVB_EHIP, "HMM forward-backward launch (65 .. 128 states)", e);
        ws->e_state = 3;
        return GMMVB_OK;
    }
    if (ws->hmm->generic) {
        e = run_generic(ws, ws->hmm, n_rows, pi_tilde_dev, a_tilde_dev, out_dev, st);
        if (e != hipSuccess) return fail(GMMVB_EHIP, "HMM forward-backward launch (generic)", e);
        ws->e_state = 3;
        return GMMVB_OK;
    }
    switch (ws->hmm->KT) {
        case 1: e = run<1>(ws, ws->hmm, n_rows, pi_tilde_dev, a_tilde_dev, out_dev, st); break;
        case 2: e = run<2>(ws, ws->hmm, n_rows, pi_tilde_dev, a_tilde_dev, out_dev, st); break;
        case 3: e = run<3>(ws, ws->hmm, n_rows, pi_tilde_dev, a_tilde_dev, out_dev, st); break;
        case 4: e = run<4>(ws, ws->hmm, n_rows, pi_tilde_dev, a_tilde_dev, out_dev, st); break;
        default: return fail(GMMVB_EUNSUPPORTED, "K > 64");
    }
    if (e != hipSuccess) return fail(GMMVB_EHIP, "HMM forward-backward launch", e);
    ws->e_state = 3;
    return GMMVB_OK;
}

int hmmvb_readout(gmmvb_workspace* ws, int what, int64_t row0, int64_t n_rows, const double* a_tilde_dev, double* out_dev,
                  void* stream) {
    if (!ws || !ws->hmm || !out_dev) return fail(GMMVB_EINVAL, "null argument");
    if (what != 0 && what != 1 && what != 3) return fail(GMMVB_EINVAL, "what must be 0 (alpha), 1 (beta) or 3 (xi)");
    if (what == 3 && !a_tilde_dev) return fail(GMMVB_EINVAL, "a_tilde_dev is needed for xi");
    if (ws->e_state != 3 || row0 < 0 || n_rows < 1 || row0 + n_rows > ws->e_rows || row0 + n_rows > ws->hmm->gamma_rows)
        return fail(GMMVB_ESTATE, "row range outside the last hmmvb_forward_backward (or hmmvb_viterbi has reused its buffers)");
    gmmvb_hmm_state* h = ws->hmm;
    const int64_t total = n_rows * h->K * (what == 3 ? h->K : 1);
    hipLaunchKernelGGL(hmm_readout_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, h->alpha_tm,
                       h->gamma_tm, h->w_valid ? h->w_tm : nullptr, a_tilde_dev, h->K, h->Kp, what, row0, n_rows, out_dev, h->rho_tm,
                       h->cprime);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(GMMVB_EHIP, "hmm_readout launch", e);
    return GMMVB_OK;
}

/* Read-outs of the last forward-backward pass for rows [row0, row0 + n_rows): alpha / beta~ are not kept in
 * natural order; this returns alpha (mode 0) or c' (mode 1, [n_rows]) for tests. */
int hmmvb_debug_readout(gmmvb_workspace* ws, int what, int64_t row0, int64_t n_rows, double* out_dev, void* stream) {
    if (!ws || !ws->hmm || !out_dev) return fail(GMMVB_EINVAL, "null argument");
    if (ws->e_state != 3 || row0 < 0 || n_rows < 1 || row0 + n_rows > ws->e_rows) return fail(GMMVB_EINVAL, "bad range");
    gmmvb_hmm_state* h = ws->hmm;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e;
    if (what == 1) {
        e = hipMemcpyAsync(out_dev, h->cprime + row0, (size_t)n_rows * sizeof(double), hipMemcpyDeviceToDevice, st);
    } else if (what == 2) {
        e = hipMemcpyAsync(out_dev, h->mx + row0, (size_t)n_rows * sizeof(double), hipMemcpyDeviceToDevice, st);
    } else {
        // alpha in lane order, [n_rows][Kp]; the caller un-permutes (positions: hmm_pos)
        e = hipMemcpyAsync(out_dev, h->alpha_tm + row0 * h->Kp, (size_t)n_rows * h->Kp * sizeof(double),
                           hipMemcpyDeviceToDevice, st);
    }
    if (e != hipSuccess) return fail(GMMVB_EHIP, "debug readout", e);
    return GMMVB_OK;
}

}  // extern "C"
