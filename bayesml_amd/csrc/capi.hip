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
#include "launch.h"

using namespace gmmvb;

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

static int ensure_lists(gmmvb_workspace* ws);

// profiling spans (gmmvb_profile_spans): HIP events on the launch stream around groups of kernels
enum { kSpanEstepMain = 0, kSpanSelect = 1, kSpanGather = 2, kSpanLse = 3, kSpanLists = 4, kSpanMstepMain = 5,
       kSpanReduce = 6, kSpanSlots = 8 };
static const char* const kSpanNames[kSpanSlots] = {"estep_main", "estep_select", "estep_gather", "estep_lse_mask",
                                                   "mstep_lists", "mstep_main", "mstep_reduce", ""};
static void span_begin(gmmvb_workspace* ws, int slot, hipStream_t st) {
    if (!ws->prof || ws->n_spans >= gmmvb_workspace::kMaxSpans) return;
    ws->span_slot[ws->n_spans] = slot;
    (void)hipEventRecord(ws->span_ev[2 * ws->n_spans], st);
}
static void span_end(gmmvb_workspace* ws, hipStream_t st) {
    if (!ws->prof || ws->n_spans >= gmmvb_workspace::kMaxSpans) return;
    (void)hipEventRecord(ws->span_ev[2 * ws->n_spans + 1], st);
    ++ws->n_spans;
}

int gmmvb_workspace_create(int K, int D, int x_dtype, int64_t max_rows, gmmvb_workspace** out) {
    if (!out) return fail(GMMVB_EINVAL, "out is null");
    *out = nullptr;
    if (K < 1 || D < 1 || max_rows < 1) return fail(GMMVB_EINVAL, "K, D and max_rows must be positive");
    if (x_dtype != GMMVB_F32 && x_dtype != GMMVB_F64) return fail(GMMVB_EINVAL, "x_dtype must be GMMVB_F32 or GMMVB_F64");
    if (D > 16 * kMaxTiles) return fail(GMMVB_EUNSUPPORTED, "D > 128 is not supported by this version");
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return fail(GMMVB_EHIP, "hipGetDevice", e);
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, dev);
    if (e != hipSuccess) return fail(GMMVB_EHIP, "hipGetDeviceProperties", e);

    gmmvb_workspace* ws = new (std::nothrow) gmmvb_workspace();
    if (!ws) return fail(GMMVB_ENOMEM, "host allocation failed");
    ws->K = K;
    ws->D = D;
    ws->T = (D + 15) / 16;
    ws->x_dtype = x_dtype;
    ws->max_rows = max_rows;
    ws->npad = round_up(max_rows, 64);
    ws->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    {
        const int kpw = mstep_components_per_wg(ws->T, false);     // the smaller of the two forms: sizes the slabs
        ws->KG = (K + kpw - 1) / kpw;
    }
    ws->S_cap = (int)round_up(((int64_t)4 * ws->num_cu + ws->KG - 1) / ws->KG, 8);
    if (ws->S_cap < 8) ws->S_cap = 8;
    {
        // Cap on the rows of one M-step split: 16 MB of centred rows (16384 rows at D = 128).  All component groups
        // of a split stream the same rows; short splits keep those workgroups within an L2's reach of each other
        // (measured at C3: fetch 96 GB -> 16-21 GB ~ the algorithmic 15.4 GB, kernel 175.6 -> 171.5 ms) at the
        // price of more slabs (+0.7 ms reduce).  env GMMVB_MSTEP_SPLIT_ROWS overrides; 0 disables the cap.
        const char* v = std::getenv("GMMVB_MSTEP_SPLIT_ROWS");
        ws->split_rows = round_up(std::max<int64_t>(64, (16 << 20) / (16 * ws->T * 8)), 64);
        if (v) ws->split_rows = std::atoll(v) > 0 ? round_up(std::max<int64_t>(64, std::atoll(v)), 64) : 0;
        if (ws->split_rows) {
            const int64_t need = round_up((max_rows + ws->split_rows - 1) / ws->split_rows, 8);
            if (need > ws->S_cap) ws->S_cap = (int)need;
        }
    }
    ws->img_len = estep_image_doubles(ws->T);
    {
        const char* v = std::getenv("GMMVB_ESTEP_VARIANT");
        ws->estep_variant = kEstepLds8;      // measured fastest (two waves per SIMD share one LDS image)
        if (v && std::strcmp(v, "direct") == 0) ws->estep_variant = kEstepDirect;
        if (v && std::strcmp(v, "lds4") == 0) ws->estep_variant = kEstepLds;
        if (v && std::strcmp(v, "i8") == 0) ws->estep_variant = kEstepI8;
    }
    {
        const char* v = std::getenv("GMMVB_ESTEP_BOUND");
        ws->bound_i8 = !(v && std::strcmp(v, "f64") == 0);
    }
    struct { double** p; int64_t n; } bufs[] = {
        {&ws->lnrho, (int64_t)K * ws->npad}, {&ws->lse, ws->npad},
        {&ws->img, (int64_t)K * ws->img_len},
        {&ws->cvec, K},                      {&ws->pivot, D},
        {&ws->slabs, (int64_t)ws->S_cap * K * slab_len(ws->T)},
        {&ws->xc, 0},
        {&ws->dpart, ((ws->npad + kLseRows - 1) / kLseRows) * K}, {&ws->thr, K},
        {&ws->apart, (ws->npad + kSelRows - 1) / kSelRows}, {&ws->act_total, 1}, {&ws->drift, 3 * (int64_t)K}};
    {
        const char* v = std::getenv("GMMVB_MSTEP_PRECENTER");      // "0" = never make the centred copy
        if (!(v && std::strcmp(v, "0") == 0)) bufs[6].n = (ws->npad + 64) * 16 * (int64_t)ws->T;
    }
    {
        const char* v = std::getenv("GMMVB_MSTEP_SPARSE");         // "0" = always the dense M-step
        ws->sparse = !(v && std::strcmp(v, "0") == 0);
        if (max_rows > 2000000000) ws->sparse = false;             // the sample lists hold 32-bit row numbers
        v = std::getenv("GMMVB_ESTEP_PRUNE");
        ws->prune = (v && std::strcmp(v, "0") == 0) ? 0 : ((v && std::strcmp(v, "force") == 0) ? 2 : 1);
        if (!ws->sparse || estep_bound_blocks(ws->T) == 0 || K > 256) ws->prune = 0;
    }
    {
        const bool full = ws->estep_variant == kEstepI8, bound = ws->prune != 0 && ws->bound_i8;
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
        e = hipMalloc((void**)b.p, (size_t)b.n * sizeof(double));
        if (e != hipSuccess) {
            gmmvb_workspace_destroy(ws);
            return fail(GMMVB_ENOMEM, "hipMalloc (workspace)", e);
        }
        ws->bytes += b.n * (int64_t)sizeof(double);
    }
    if (ws->sparse && K <= 256 && ensure_lists(ws) != GMMVB_OK) {      // not inside somebody's timed iteration
        gmmvb_workspace_destroy(ws);
        return GMMVB_ENOMEM;
    }
    e = hipMemset(ws->pivot, 0, (size_t)D * sizeof(double));
    if (e != hipSuccess) {
        gmmvb_workspace_destroy(ws);
        return fail(GMMVB_EHIP, "hipMemset(pivot)", e);
    }
    *out = ws;
    return GMMVB_OK;
}

int gmmvb_workspace_destroy(gmmvb_workspace* ws) {
    if (!ws) return GMMVB_OK;
    double* bufs[] = {ws->lnrho, ws->lse, ws->img, ws->cvec, ws->pivot, ws->slabs, ws->xc, ws->dpart, ws->thr,
                      ws->apart, ws->act_total, ws->drift};
    int* ibufs[] = {ws->lists, ws->khat, ws->counts, ws->blk};
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
    return GMMVB_OK;
}

int gmmvb_wants_drift(const gmmvb_workspace* ws, int64_t n_rows) {
    if (!ws || ws->prune == 0 || ws->estep_variant != kEstepLds8 || ws->hmm != nullptr || !ws->masks) return 0;
    if (std::getenv("GMMVB_ESTEP_CARRY_OFF") != nullptr) return 0;
    return (ws->prune == 2 || n_rows * (int64_t)ws->K >= (int64_t(1) << 23)) ? 1 : 0;
}

int gmmvb_set_drift(gmmvb_workspace* ws, const double* gamma_dev, const double* delta_dev, double typical_gamma,
                    void* stream) {
    if (!ws || !gamma_dev || !delta_dev) return fail(GMMVB_EINVAL, "null argument");
    ws->typical_gamma = typical_gamma;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemcpyAsync(ws->drift, gamma_dev, (size_t)ws->K * sizeof(double), hipMemcpyDeviceToDevice, st);
    if (e == hipSuccess)
        e = hipMemcpyAsync(ws->drift + ws->K, delta_dev, (size_t)ws->K * sizeof(double), hipMemcpyDeviceToDevice, st);
    // the constants of the parameters the ln rho array belongs to (the next gmmvb_set_params overwrites cvec)
    if (e == hipSuccess)
        e = hipMemcpyAsync(ws->drift + 2 * ws->K, ws->cvec, (size_t)ws->K * sizeof(double), hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) return fail(GMMVB_EHIP, "hipMemcpyAsync(drift)", e);
    ws->have_drift = ws->have_params && ws->params_used;     // else: not the parameters the ln rho array belongs to
    return GMMVB_OK;
}

int gmmvb_set_params(gmmvb_workspace* ws, const double* c_dev, const double* m_dev, const double* u_dev,
                     void* stream) {
    if (!ws || !c_dev || !m_dev || !u_dev) return fail(GMMVB_EINVAL, "null argument");
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemcpyAsync(ws->cvec, c_dev, (size_t)ws->K * sizeof(double), hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) return fail(GMMVB_EHIP, "hipMemcpyAsync(c)", e);
    hipLaunchKernelGGL(pack_params_kernel, dim3(ws->K), dim3(256), 0, st, u_dev, m_dev, ws->K, ws->D, ws->T,
                       ws->img_len, ws->img);
    e = hipGetLastError();
    if (e != hipSuccess) return fail(GMMVB_EHIP, "pack_params_kernel", e);
    if (ws->pivot_i8) {
        // the digits are taken about the pivot in force now; the int8 kernels read this copy, not ws->pivot
        e = hipMemcpyAsync(ws->pivot_i8, ws->pivot, (size_t)ws->D * sizeof(double), hipMemcpyDeviceToDevice, st);
        if (e == hipSuccess && ws->img_i8) e = launch_pack_i8(u_dev, m_dev, ws->pivot_i8, ws->K, ws->D, ws->img_i8, 0, st);
        if (e == hipSuccess && ws->img_i8b) e = launch_pack_i8(u_dev, m_dev, ws->pivot_i8, ws->K, ws->D, ws->img_i8b, 1, st);
        if (e != hipSuccess) return fail(GMMVB_EHIP, "pack_params_i8_kernel", e);
    }
    ws->have_params = true;
    ws->params_used = false;
    return GMMVB_OK;
}

// sample lists of the pruned E-step and the sparse M-step (allocated at first use)
static int ensure_lists(gmmvb_workspace* ws) {
    if (ws->lists) return GMMVB_OK;
    const int64_t sel_blocks = (ws->npad + kSelRows - 1) / kSelRows, words = (ws->K + 63) / 64;
    hipError_t e = hipMalloc((void**)&ws->lists, (size_t)ws->K * ws->npad * sizeof(int));
    if (e == hipSuccess) e = hipMalloc((void**)&ws->khat, (size_t)ws->npad * sizeof(int));
    if (e == hipSuccess) e = hipMalloc((void**)&ws->counts, (size_t)ws->K * sizeof(int));
    if (e == hipSuccess) e = hipMalloc((void**)&ws->blk, (size_t)sel_blocks * ws->K * sizeof(int));
    if (e == hipSuccess) e = hipMalloc((void**)&ws->masks, (size_t)words * ws->npad * sizeof(unsigned long long));
    if (e != hipSuccess) return fail(GMMVB_ENOMEM, "hipMalloc (sample lists)", e);
    ws->bytes += ((int64_t)ws->K * ws->npad + ws->npad + ws->K + sel_blocks * ws->K) * (int64_t)sizeof(int) +
                 words * ws->npad * 8;
    return GMMVB_OK;
}

// active-pair count of the last E-step (one 8-byte read behind a stream sync, cached until the next E-step)
static int fetch_active(gmmvb_workspace* ws, hipStream_t st, double* out) {
    if (ws->act_host < 0.0) {
        double act = 0.0;
        hipError_t e = hipMemcpyAsync(&act, ws->act_total, sizeof(double), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) return fail(GMMVB_EHIP, "reading the active-pair count", e);
        ws->act_host = act;
    }
    *out = ws->act_host;
    return GMMVB_OK;
}

int gmmvb_last_sparsity(gmmvb_workspace* ws, void* stream, double* active_pairs, double* evaluated_pairs) {
    if (!ws || !active_pairs || !evaluated_pairs) return fail(GMMVB_EINVAL, "null argument");
    if (ws->e_state != 1) return fail(GMMVB_ESTATE, "no E-step output in the workspace");
    *evaluated_pairs = ws->evaluated;
    if (!ws->sparse || ws->act_rows != ws->e_rows) {      // GMMVB_MSTEP_SPARSE=0: the pairs are not counted
        *active_pairs = -1.0;
        return GMMVB_OK;
    }
    return fetch_active(ws, (hipStream_t)stream, active_pairs);
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
    ws->bounds_rows = 0;               // (new) sample matrix: nothing of an earlier E-step may be carried over
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
    return GMMVB_OK;
}

int gmmvb_estep(gmmvb_workspace* ws, const void* x_dev, int64_t ldx, int64_t n_rows, void* stream) {
    bool vec = false;
    int rc = check_x(ws, x_dev, ldx, n_rows, &vec);
    if (rc) return rc;
    if (!ws->have_params) return fail(GMMVB_ESTATE, "gmmvb_set_params has not been called");
    hipStream_t st = (hipStream_t)stream;
    const int is64 = ws->x_dtype == GMMVB_F64;
    const bool i8 = ws->estep_variant == kEstepI8;
    EstepArgs a{x_dev, ldx, n_rows, ws->D, ws->img, ws->cvec, ws->K, ws->lnrho, ws->npad};
    EstepI8Args a8{x_dev, ldx, n_rows, ws->D, ws->img_i8, ws->pivot_i8, ws->cvec, ws->K, ws->lnrho, ws->npad};
    const char* name = "";
    hipError_t e = hipSuccess;
    // Prune?  Needs the default f64 kernel family, and (unless forced) evidence from the previous E-step over
    // these rows that at most half of the (sample, component) pairs matter.
    // (never for an HMM workspace: forward-backward consumes every emission ln rho, bounds will not do)
    bool prune = ws->prune != 0 && ws->estep_variant == kEstepLds8 && ws->hmm == nullptr;
    if (prune && ws->prune == 1) {
        prune = false;
        if (n_rows * (int64_t)ws->K >= (int64_t(1) << 23) && ws->act_rows == n_rows) {
            double act = 0.0;
            rc = fetch_active(ws, st, &act);
            if (rc) return rc;
            prune = act <= 0.5 * (double)n_rows * ws->K;
        }
    }
    if (prune && ws->img_i8b) {
        // How many output blocks the bound pass evaluates.  Cost model per (sample, component) pair, in units of
        // 1e-11 s measured at C3 (profiles/r1_v6_*): bound pass 0.12 per block pair + 0.039 per row of y; exact pass
        // 0.81 per f64 tile pair of every candidate.  Take the cheapest level among those observed in the last 32
        // pruned passes; look one level down when the current one leaves hardly any spare candidates (two levels
        // exist below) or one level up when more than half of its candidates are spare, if that level is unknown.
        const int t32 = (ws->D + 31) / 32;
        const char* pin = std::getenv("GMMVB_ESTEP_BOUND_BLOCKS");      // pins the level (1 .. ceil(D/32))
        if (ws->bound_tb == 0) ws->bound_tb = t32 > 3 ? 3 : t32;
        if (pin && std::atoi(pin) >= 1 && std::atoi(pin) <= t32) {
            ws->bound_tb = std::atoi(pin);
        } else if (ws->evaluated_prev >= 0.0 && ws->act_rows == n_rows && ws->prev_pass == 1) {
            double act = 0.0;
            rc = fetch_active(ws, st, &act);
            if (rc) return rc;
            const double pairs = (double)n_rows * ws->K;
            const int cur = ws->bound_tb;
            ws->tb_cand[cur] = ws->evaluated_prev / pairs;
            ws->tb_act[cur] = act / pairs;
            ws->tb_seen[cur] = 0;
            // an observation is forgotten after 32 passes, or once the sparsity is no longer what it was made at
            for (int l = 1; l <= t32; ++l)
                if (l != cur && (++ws->tb_seen[l] > 32 || ws->tb_act[l] > 1.5 * ws->tb_act[cur] ||
                                 ws->tb_act[l] < ws->tb_act[cur] / 1.5))
                    ws->tb_cand[l] = -1.0;
            const double gpp = 0.81 * tri_pairs(ws->T);
            // with drift hints in use a few carried passes follow a bound pass and inherit its spare candidates: a
            // tighter bound pays for part of itself there
            const double heirs = ws->have_drift ? 3.0 : 0.0;
            auto cost = [&](int l) {
                const double spare_l = ws->tb_cand[l] > ws->tb_act[l] ? ws->tb_cand[l] - ws->tb_act[l] : 0.0;
                return 0.12 * tri_pairs(l) + 0.039 * 32 * l + gpp * (ws->tb_cand[l] + heirs * spare_l);
            };
            int best = cur;
            for (int l = 1; l <= t32; ++l)
                if (ws->tb_cand[l] >= 0.0 && cost(l) < cost(best)) best = l;
            const double spare = ws->tb_cand[cur] - act / pairs;         // candidates that turned out irrelevant
            if (best == cur) {
                if (cur > 1 && ws->tb_cand[cur - 1] < 0.0 && spare * ws->K < (ws->have_drift ? 0.02 : 0.25))
                    best = cur - 1;        // under a quarter of a spare candidate per sample: the bound is tight,
                                           // try the cheaper one (candidates grow steeply once rows are dropped)
                else if (cur < t32 && ws->tb_cand[cur + 1] < 0.0 &&
                         spare * gpp > 0.12 * (tri_pairs(cur + 1) - tri_pairs(cur)) + 0.039 * 32)
                    best = cur + 1;        // the spare candidates cost more than a tighter bound would
            }
            ws->bound_tb = best;
        }
    }
    // Carry the previous pass's values / bounds over the parameter update instead of bounding every pair again
    // (gmmvb_set_drift)?  Up to 8 passes in a row, and not once the spare candidates predicted for this pass would
    // cost the exact pass more than a real bound pass costs: then the bounds are refreshed (see below).
    bool carry = prune && ws->have_drift && ws->bounds_rows == n_rows && ws->bounds_x == x_dev &&
                 ws->bounds_ldx == ldx && ws->masks && ws->carried < 8 &&
                 std::getenv("GMMVB_ESTEP_CARRY_OFF") == nullptr;
    // early in a fit the components still move by tens of per cent per iteration (mean gamma 0.3, 0.7, 0.87, 0.91,
    // 0.94 ... at C3): carried bounds shrink by gamma^2 and leave a dozen candidates per sample; bound afresh then
    if (carry && ws->typical_gamma > 0.0 && ws->typical_gamma < 0.9) carry = false;
    if (prune && ws->prev_pass != 0 && ws->evaluated_prev >= 0.0 && ws->act_rows == n_rows) {
        // spare candidates (listed but inactive) per pair of the last pruned pass, and of the one before
        double act = 0.0;
        rc = fetch_active(ws, st, &act);
        if (rc) return rc;
        const double pairs = (double)n_rows * ws->K;
        ws->spare_before = ws->spare_last;
        ws->spare_last = (ws->evaluated_prev - act) / pairs;
        if (ws->spare_last < 0.0) ws->spare_last = 0.0;
        if (carry && ws->prev_pass == 2) {
            // the spare candidates grow from pass to pass (by a factor of two to three at the benchmark's scale):
            // carry on only while the exact pass over the predicted spare ones costs less than a bound pass
            const int tb = ws->bound_tb > 0 ? ws->bound_tb : 3;
            const double bound_cost = 0.12 * tri_pairs(tb) + 0.039 * 32 * tb, gpp = 0.81 * tri_pairs(ws->T);
            carry = gpp * ws->spare_last * 2.5 < bound_cost;
        }
        if (std::getenv("GMMVB_DEBUG"))
            std::fprintf(stderr, "[gmmvb] estep: prev_pass=%d evaluated_prev=%.3g act=%.3g spare %.4g <- %.4g carry=%d carried=%d\n",
                         ws->prev_pass, ws->evaluated_prev / n_rows, act / n_rows, ws->spare_last, ws->spare_before,
                         (int)carry, ws->carried);
    } else if (ws->prev_pass == 0) {
        ws->spare_last = ws->spare_before = -1.0;
        if (carry && ws->act_rows == n_rows) {
            // out of a dense pass the parameters are still moving fast (the second or third iteration of a fit):
            // carried values would leave most pairs candidates unless the responsibilities are sparse already
            double act = 0.0;
            rc = fetch_active(ws, st, &act);
            if (rc) return rc;
            carry = act <= 0.1 * (double)n_rows * ws->K;
        }
    }
    ws->evaluated_prev = -1.0;
    if (prune) {
        rc = ensure_lists(ws);
        if (rc) return rc;
    }
    int rpw = 0;
    int64_t grid = 0;
    bool pruned_fell_back = false;
    if (ws->prof) (void)hipEventRecord(ws->ev[0], st);
    ws->n_spans = 0;
    ws->evaluated = prune ? 0.0 : (double)n_rows * ws->K;
    if (!prune)      // a dense pass: whatever was learnt about the bound levels belongs to another regime
        for (double& c : ws->tb_cand) c = -1.0;
    if (prune) {
        if (carry) {
            // nothing to launch here: the previous pass's best components are evaluated first (round 0), and the second
            // selection carries every other value over the update as it reads it (select_mask_kernel<4>)
            rpw = kSelRows;
            grid = (n_rows + kSelRows - 1) / kSelRows;
            name = "estep_carried_bounds";
            ++ws->carried;
            ++ws->passes[2];
        } else {
            rpw = ws->img_i8b ? estep_i8_rows_per_wg() : estep_bound_rows_per_wg(ws->T, is64);
            grid = (n_rows + rpw - 1) / rpw;
            if (grid > (1 << 20)) grid = 1 << 20;
            span_begin(ws, kSpanEstepMain, st);
            if (ws->img_i8b) {
                EstepI8Args ab = a8;
                ab.img = ws->img_i8b;
                ab.khat = ws->khat;             // the bound kernel also finds every row's best component
                e = launch_estep_i8_bound(is64, vec, ws->bound_tb, (int)grid, st, ab, &name);
            } else {
                e = launch_estep_bound(ws->T, is64, vec, (int)grid, st, a, &name);
            }
            span_end(ws, st);
            ws->carried = 0;
            ++ws->passes[1];
        }
        if (e != hipSuccess) return fail(GMMVB_EHIP, "estep_bound launch", e);
        const bool i8_bound = ws->img_i8b != nullptr;
        bool& fell_back = pruned_fell_back;
        int counts_host[256];
        const int sel_grid = (int)((n_rows + kSelRows - 1) / kSelRows);
        for (int round = 0; round < 2; ++round) {
            span_begin(ws, kSpanSelect, st);
            if (round == 0 && (carry || i8_bound))
                hipLaunchKernelGGL(select_mask_kernel<3>, dim3(sel_grid), dim3(kSelRows), 0, st, ws->lnrho, ws->npad,
                                   n_rows, ws->K, ws->khat, ws->masks, ws->blk);
            else if (round == 0)
                hipLaunchKernelGGL(select_mask_kernel<0>, dim3(sel_grid), dim3(kSelRows), 0, st, ws->lnrho, ws->npad,
                                   n_rows, ws->K, ws->khat, ws->masks, ws->blk);
            else if (carry)
                hipLaunchKernelGGL(select_mask_kernel<4>, dim3(sel_grid), dim3(kSelRows), 0, st, ws->lnrho, ws->npad,
                                   n_rows, ws->K, ws->khat, ws->masks, ws->blk, ws->drift, ws->cvec);
            else
                hipLaunchKernelGGL(select_mask_kernel<1>, dim3(sel_grid), dim3(kSelRows), 0, st, ws->lnrho, ws->npad,
                                   n_rows, ws->K, ws->khat, ws->masks, ws->blk);
            hipLaunchKernelGGL(scan_counts_kernel, dim3(ws->K), dim3(256), 0, st, ws->blk, sel_grid, ws->K, ws->counts);
            hipLaunchKernelGGL(fill_lists_kernel, dim3(sel_grid), dim3(kSelRows), 0, st, ws->masks, ws->npad, n_rows, ws->K,
                               ws->blk, ws->lists, ws->npad);
            e = hipGetLastError();
            span_end(ws, st);
            if (e == hipSuccess)
                e = hipMemcpyAsync(counts_host, ws->counts, (size_t)ws->K * sizeof(int), hipMemcpyDeviceToHost, st);
            if (e == hipSuccess) e = hipStreamSynchronize(st);
            if (e != hipSuccess) return fail(GMMVB_EHIP, "E-step candidate selection", e);
            double listed = 0.0;
            for (int k = 0; k < ws->K; ++k) listed += counts_host[k];
            if (round == 1 && carry && ws->evaluated + listed > 0.35 * (double)n_rows * ws->K) {
                // the carried bounds have become too loose: bound afresh (the exact values written so far stay valid
                // upper bounds until the bound kernel overwrites them) and select again
                carry = false;
                ws->carried = 0;
                rpw = ws->img_i8b ? estep_i8_rows_per_wg() : estep_bound_rows_per_wg(ws->T, is64);
                grid = (n_rows + rpw - 1) / rpw;
                if (grid > (1 << 20)) grid = 1 << 20;
                span_begin(ws, kSpanEstepMain, st);
                if (ws->img_i8b) {
                    EstepI8Args ab = a8;
                    ab.img = ws->img_i8b;
                    ab.khat = ws->khat;
                    e = launch_estep_i8_bound(is64, vec, ws->bound_tb, (int)grid, st, ab, &name);
                } else {
                    e = launch_estep_bound(ws->T, is64, vec, (int)grid, st, a, &name);
                }
                span_end(ws, st);
                if (e != hipSuccess) return fail(GMMVB_EHIP, "estep_bound launch", e);
                ++ws->passes[1];
                ++ws->passes[4];
                ws->evaluated = 0.0;
                round = -1;                 // start the selection over
                continue;
            }
            if (round == 1 && ws->prune != 2 && ws->evaluated + listed > 0.6 * (double)n_rows * ws->K) {
                // the parameters moved a long way since the last E-step (a new restart): most pairs are candidates
                // again, so evaluate everything with the dense kernel instead of gathering almost everything
                const int rpd = estep_rows_per_wg(ws->estep_variant, ws->T, is64);
                int64_t gd = (n_rows + rpd - 1) / rpd;
                if (gd > (1 << 20)) gd = 1 << 20;
                span_begin(ws, kSpanEstepMain, st);
                e = launch_estep(ws->estep_variant, ws->T, is64, vec, (int)gd, st, a, &name);
                span_end(ws, st);
                if (e != hipSuccess) return fail(GMMVB_EHIP, "estep launch", e);
                ws->evaluated = (double)n_rows * ws->K;
                ws->evaluated_prev = -1.0;
                fell_back = true;
                ++ws->passes[3];
                if (ws->img_i8b && !carry) {   // this level left everything a candidate: remember, and go back up
                    ws->tb_cand[ws->bound_tb] = 1.0;
                    ws->tb_seen[ws->bound_tb] = 0;
                    if (ws->bound_tb < (ws->D + 31) / 32) ++ws->bound_tb;
                }
                break;
            }
            ws->evaluated += listed;
            ws->evaluated_prev = ws->evaluated;
            span_begin(ws, kSpanGather, st);
            e = launch_estep_gather(ws->T, is64, vec, st, a, ws->lists, ws->npad, ws->counts, counts_host);
            span_end(ws, st);
            if (e != hipSuccess) return fail(GMMVB_EHIP, "estep_gather launch", e);
            ++ws->passes[7];
        }
    } else {
        rpw = i8 ? estep_i8_rows_per_wg() : estep_rows_per_wg(ws->estep_variant, ws->T, is64);
        grid = (n_rows + rpw - 1) / rpw;
        if (grid > (1 << 20)) grid = 1 << 20;
        span_begin(ws, kSpanEstepMain, st);
        e = i8 ? launch_estep_i8(is64, vec, (int)grid, st, a8, &name)
               : launch_estep(ws->estep_variant, ws->T, is64, vec, (int)grid, st, a, &name);
        span_end(ws, st);
        if (e != hipSuccess) return fail(GMMVB_EHIP, "estep launch", e);
        ++ws->passes[0];
    }
    if (ws->prof) {
        (void)hipEventRecord(ws->ev[1], st);
        ws->ev_e = true;
    }
    const int lse_blocks = (int)((n_rows + kLseRows - 1) / kLseRows);
    // small passes are launch-bound: no pair counting, no lists (the dense M-step takes microseconds there)
    const bool count_pairs = ws->sparse && ws->masks && ws->hmm == nullptr &&
                             (prune || n_rows * (int64_t)ws->K >= (int64_t(1) << 18));
    span_begin(ws, kSpanLse, st);
    if (count_pairs) {
        // thresholds from a sample of the rows (every 16th block of 1024), then lse + active masks + counts in one pass
        const int stride = lse_blocks >= 64 ? 16 : 1;
        const int sampled = (lse_blocks + stride - 1) / stride;
        const int nblk = (int)((n_rows + kSelRows - 1) / kSelRows);
        hipLaunchKernelGGL(row_lse_kernel, dim3((unsigned)sampled), dim3(256), 0, st, ws->lnrho, ws->npad, n_rows, ws->K,
                           ws->lse, ws->dpart, nullptr, stride);
        hipLaunchKernelGGL(thr_kernel, dim3((unsigned)ws->K), dim3(256), 0, st, ws->dpart, nullptr, sampled, ws->K, ws->thr,
                           ws->act_total);
        hipLaunchKernelGGL(lse_mask_kernel, dim3((unsigned)nblk), dim3(kSelRows), 0, st, ws->lnrho, ws->npad, n_rows, ws->K,
                           ws->thr, ws->lse, ws->masks, ws->blk, ws->apart, ws->khat);
        hipLaunchKernelGGL(thr_kernel, dim3((unsigned)(ws->K + 1)), dim3(256), 0, st, nullptr, ws->apart, nblk, ws->K,
                           ws->thr, ws->act_total);
        e = hipGetLastError();
        if (e != hipSuccess) return fail(GMMVB_EHIP, "row_lse / lse_mask launch", e);
        ws->act_rows = n_rows;
        ws->act_host = -1.0;
        ws->active_lists = false;
    } else {
        hipLaunchKernelGGL(row_lse_kernel, dim3((unsigned)lse_blocks), dim3(256), 0, st, ws->lnrho, ws->npad, n_rows, ws->K,
                           ws->lse, nullptr, nullptr, 1);
        e = hipGetLastError();
        if (e != hipSuccess) return fail(GMMVB_EHIP, "row_lse launch", e);
        ws->act_rows = 0;              // nothing counted: dense M-step, no pruning decision from this pass
    }
    span_end(ws, st);
    ws->e_state = 1;
    ws->e_rows = n_rows;
    ws->params_used = true;
    ws->have_drift = false;
    ws->bounds_rows = n_rows;          // the ln rho array now belongs to the parameters in force, on these rows
    ws->bounds_x = x_dev;
    ws->bounds_ldx = ldx;
    if (!prune || pruned_fell_back) ws->carried = 0;
    ws->prev_pass = (!prune || pruned_fell_back) ? 0 : (ws->carried > 0 ? 2 : 1);
    std::snprintf(ws->info, sizeof(ws->info), "%s grid=%lldx%d rows/workgroup=%d", name, (long long)grid,
                  (i8 || prune) ? 512 : estep_threads(ws->estep_variant), rpw);
    return GMMVB_OK;
}

int gmmvb_load_responsibilities(gmmvb_workspace* ws, const double* r_dev, int64_t n_rows, void* stream) {
    if (!ws || !r_dev) return fail(GMMVB_EINVAL, "null argument");
    if (n_rows < 1 || n_rows > ws->max_rows) return fail(GMMVB_EINVAL, "n_rows must be in [1, max_rows]");
    const int tb = 256;
    hipLaunchKernelGGL(load_r_kernel, dim3((unsigned)((n_rows + tb - 1) / tb)), dim3(tb), 0, (hipStream_t)stream,
                       r_dev, n_rows, ws->K, ws->lnrho, ws->npad, ws->lse);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(GMMVB_EHIP, "load_r launch", e);
    ws->e_state = 2;
    ws->e_rows = n_rows;
    ws->n_spans = 0;
    ws->bounds_rows = 0;               // the array holds responsibilities now, nothing a later E-step may carry over
    return GMMVB_OK;
}

int gmmvb_mstep(gmmvb_workspace* ws, const void* x_dev, int64_t ldx, int64_t n_rows, double* stats_dev,
                void* stream) {
    bool vec = false;
    int rc = check_x(ws, x_dev, ldx, n_rows, &vec);
    if (rc) return rc;
    if (!stats_dev) return fail(GMMVB_EINVAL, "stats_dev is null");
    if (ws->e_state == 0 || ws->e_rows != n_rows)
        return fail(GMMVB_ESTATE, "no responsibilities for these rows: call gmmvb_estep or gmmvb_load_responsibilities first");
    hipStream_t st = (hipStream_t)stream;
    // row splits: ~4 workgroups per CU in total, whole 64-row groups per split, S a multiple of 8 where possible
    int64_t S = ws->S_cap;
    const int64_t groups = (n_rows + 63) / 64;
    if (S > groups) S = groups;
    int64_t rows_per_split = round_up((n_rows + S - 1) / S, 64);
    if (ws->split_rows && rows_per_split > ws->split_rows) rows_per_split = ws->split_rows;
    S = (n_rows + rows_per_split - 1) / rows_per_split;
    const bool pre = ws->xc && ws->xc_src == x_dev && ws->xc_rows == n_rows && ws->xc_ldx == ldx;
    const int kpw = mstep_components_per_wg(ws->T, pre);
    const int KG = (ws->K + kpw - 1) / kpw;
    int64_t grid = 8 * ((S + 7) / 8) * KG;
    MstepArgs a{x_dev, ldx, n_rows, ws->D, ws->pivot, ws->lnrho, ws->lse, nullptr, ws->npad, ws->K, KG, (int)S,
                rows_per_split, ws->e_state == 2 ? 1 : 0, ws->slabs};
    if (ws->e_state == 3) {          // HMM: responsibilities = gamma from the forward-backward pass, h = sum gamma ln rho
        a.lnrho = hmm_gamma_cm(ws->hmm);
        a.aux = ws->lnrho;
        a.direct_r = 2;
    }
    if (pre) {
        a.x = ws->xc;
        a.ldx = 16 * ws->T;
        a.D = 16 * ws->T;
    }
    const char* name = "";
    hipError_t e;
    bool sparse = ws->sparse && ws->masks && pre && ws->e_state == 1 && ws->act_rows == n_rows;
    if (sparse) {      // the lists pay off when most pairs are negligible
        double act = 0.0;
        rc = fetch_active(ws, st, &act);
        if (rc) return rc;
        sparse = act <= 0.35 * (double)n_rows * ws->K;
    }
    if (sparse && ws->K > 256) sparse = false;
    if (sparse) {      // E-step output: only the samples that can change the f64 sums, through per-component lists
        rc = ensure_lists(ws);
        if (rc) return rc;
        // a split = a whole number of 256-row selection blocks
        const int nblk = (int)((n_rows + kSelRows - 1) / kSelRows);
        int bps = (int)((rows_per_split + kSelRows - 1) / kSelRows);
        if (bps < 1) bps = 1;
        rows_per_split = (int64_t)bps * kSelRows;
        S = (n_rows + rows_per_split - 1) / rows_per_split;
        if (ws->prof) (void)hipEventRecord(ws->ev[2], st);      // the list building is part of the M-step's time
        // masks and block counts of the active pairs were written by lse_mask_kernel at the end of the E-step
        if (!ws->active_lists) {
            span_begin(ws, kSpanLists, st);
            hipLaunchKernelGGL(scan_counts_kernel, dim3(ws->K), dim3(256), 0, st, ws->blk, nblk, ws->K, ws->counts);
            hipLaunchKernelGGL(fill_lists_kernel, dim3(nblk), dim3(kSelRows), 0, st, ws->masks, ws->npad, n_rows, ws->K,
                               ws->blk, ws->lists, ws->npad);
            e = hipGetLastError();
            span_end(ws, st);
            if (e != hipSuccess) return fail(GMMVB_EHIP, "active-sample lists", e);
            ws->active_lists = true;
        }
        grid = 8 * ((S + 7) / 8) * KG;
        MstepListArgs la{ws->xc, ws->lnrho, ws->lse, ws->lists, ws->npad, ws->blk, ws->counts, nblk, bps,
                         ws->npad, ws->K, KG, (int)S, ws->slabs};
        span_begin(ws, kSpanMstepMain, st);
        e = launch_mstep_list(ws->T, (int)grid, st, la, &name);
        span_end(ws, st);
        ++ws->passes[6];
    } else {
        ++ws->passes[5];
        if (ws->prof) (void)hipEventRecord(ws->ev[2], st);
        span_begin(ws, kSpanMstepMain, st);
        e = launch_mstep(ws->T, ws->x_dtype == GMMVB_F64, vec, pre, (int)grid, st, a, &name);
        span_end(ws, st);
    }
    if (e != hipSuccess) return fail(GMMVB_EHIP, "mstep launch", e);
    if (ws->prof) {
        (void)hipEventRecord(ws->ev[3], st);
        ws->ev_m = true;
    }
    const int elems = tri_pairs(ws->T) * 256 + 16 * ws->T + 2;
    span_begin(ws, kSpanReduce, st);
    hipLaunchKernelGGL(reduce_stats_kernel, dim3((elems + 255) / 256, ws->K), dim3(256), 0, st, ws->slabs, (int)S,
                       ws->K, ws->D, ws->T, stats_dev);
    span_end(ws, st);
    e = hipGetLastError();
    if (e != hipSuccess) return fail(GMMVB_EHIP, "reduce_stats launch", e);
    const size_t used = std::strlen(ws->info);
    std::snprintf(ws->info + used, sizeof(ws->info) - used, " | %s grid=%lldx%d splits=%lld rows/split=%lld", name,
                  (long long)grid, mstep_threads(ws->T, pre), (long long)S, (long long)rows_per_split);
    return GMMVB_OK;
}

int gmmvb_estep_mstep(gmmvb_workspace* ws, const void* x_dev, int64_t ldx, int64_t n_rows, double* stats_dev,
                      void* stream) {
    int rc = gmmvb_estep(ws, x_dev, ldx, n_rows, stream);
    if (rc) return rc;
    return gmmvb_mstep(ws, x_dev, ldx, n_rows, stats_dev, stream);
}

static int readout(gmmvb_workspace* ws, int64_t row0, int64_t n_rows, double* out, void* stream, int mode) {
    if (!ws || !out) return fail(GMMVB_EINVAL, "null argument");
    if (ws->e_state == 0) return fail(GMMVB_ESTATE, "no E-step output in the workspace");
    if (row0 < 0 || n_rows < 1 || row0 + n_rows > ws->e_rows) return fail(GMMVB_EINVAL, "row range outside the last E-step");
    if (mode == 0 && ws->e_state == 2) return fail(GMMVB_ESTATE, "ln rho is undefined after gmmvb_load_responsibilities");
    const int64_t total = n_rows * ws->K;
    const bool hmm_gamma = ws->e_state == 3 && mode == 1;      // responsibilities of an HMM pass = gamma
    hipLaunchKernelGGL(readout_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       hmm_gamma ? hmm_gamma_cm(ws->hmm) : ws->lnrho, ws->lse, ws->npad, row0, n_rows, ws->K, mode,
                       (ws->e_state == 2 || hmm_gamma) ? 1 : 0, out);
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
    if (ws->e_state == 0) return fail(GMMVB_ESTATE, "no E-step output in the workspace");
    if (row0 < 0 || n_rows < 1 || row0 + n_rows > ws->e_rows) return fail(GMMVB_EINVAL, "row range outside the last E-step");
    hipLaunchKernelGGL(argmax_kernel, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       ws->e_state == 3 ? hmm_gamma_cm(ws->hmm) : ws->lnrho, ws->npad, row0, n_rows, ws->K, z_dev);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(GMMVB_EHIP, "argmax launch", e);
    return GMMVB_OK;
}

}  // extern "C"
