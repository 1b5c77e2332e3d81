/*
 * gmmvb.h — C ABI of the MI355X (gfx950) GMM variational-Bayes data-pass engine.
 *
 * The reference (bayesml/BayesML v0.3.1) is pure Python and has no FFI; its boundary for this
 * path is the Python class contract of gaussianmixture.LearnModel.  These entry points are what
 * a binding for that class calls in place of the two NumPy methods that touch the N-sized data:
 *
 *   _update_q_z(x)        bayesml/gaussianmixture/_gaussianmixture.py:772-784   (E-step)
 *   _calc_n_x_bar_s(x)    bayesml/gaussianmixture/_gaussianmixture.py:725-732   (M-step statistics)
 *   -sum xlogy(r, r)      bayesml/gaussianmixture/_gaussianmixture.py:704       (N-sized VL term)
 *   estimate_latent_vars  bayesml/gaussianmixture/_gaussianmixture.py:1186-1193 (r / argmax read-out)
 *
 * Conventions
 *   - every pointer argument named *_dev is a DEVICE pointer owned by the caller; `stream` is a
 *     hipStream_t passed as void* (NULL = the null stream).  Calls only enqueue work: they do not allocate (only
 *     gmmvb_workspace_create/destroy and hmmvb_enable touch the allocator), do not throw and - with the exceptions
 *     listed at gmmvb_last_sparsity - do not synchronise.  Policy decisions inside gmmvb_estep / gmmvb_mstep (dense
 *     kernel, bound pass or carried records; dense or list M-step) use counters of EARLIER passes that have already
 *     arrived in pinned host memory; results never depend on them.
 *   - return value: GMMVB_OK or an error code; gmmvb_last_error() gives a thread-local message.
 *   - all K-sized quantities and all outputs are IEEE binary64.  The sample matrix x stays in its
 *     storage dtype (f32 or f64) in HBM and is widened on load; all arithmetic is f64
 *     (v_mfma_f64_16x16x4_f64), because f32 arithmetic misses the 1e-5 parity target.
 *   - shapes: K >= 1, D >= 1, n_rows <= max_rows.  Up to D = 128 (8 feature tiles) the data pass runs on the f64 MFMA
 *     kernels with pruning; for 128 < D <= 256 on dense f64 MFMA kernels of their own (the parameter image streamed through
 *     LDS by block rows; gmmvb_mstep then works from the workspace's centred copy and makes it if gmmvb_prepare_rows has not
 *     been called); beyond, on plain f64 vector kernels (csrc/generic.h): same results, far slower.  The K-sized
 *     entry points (gmmvb_kside_*) keep their matrices in LDS and stop at D = 128.
 */
#ifndef GMMVB_H
#define GMMVB_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GMMVB_ABI_VERSION 8

enum gmmvb_status {
    GMMVB_OK = 0,
    GMMVB_EINVAL = 1,       /* bad argument (null pointer, bad shape, n_rows > max_rows, misaligned x) */
    GMMVB_EUNSUPPORTED = 2, /* shape outside this version's range (gmmvb_kside_*: D > 128)                 */
    GMMVB_EHIP = 3,         /* a HIP runtime call failed (see gmmvb_last_error)                       */
    GMMVB_ENOMEM = 4,       /* device allocation failed                                               */
    GMMVB_ESTATE = 5        /* call order violated (e.g. mstep before estep / load_responsibilities)  */
};

enum gmmvb_dtype { GMMVB_F32 = 0, GMMVB_F64 = 1 };

typedef struct gmmvb_workspace gmmvb_workspace;

int gmmvb_abi_version(void);
const char* gmmvb_last_error(void);

/* Number of doubles in the statistics block written by gmmvb_mstep:
 *   [ ns[K] | h[K] | a[K][D] | B[K][D][D] ]   (len = K*(2 + D + D*D))
 * ns[k] = sum_n r_nk; h[k] = sum_n r_nk ln r_nk; a[k] = sum_n r_nk (x_n - pivot);
 * B[k] = sum_n r_nk (x_n - pivot)(x_n - pivot)^T (exactly symmetric).
 * The sums are linear in the rows, so row shards are combined by adding their blocks
 * (one all-reduce per VB iteration).  The caller turns them into the reference's
 * ns / x_bar_vecs / s_mats: x_bar = pivot + a/ns, S = B/ns - (a/ns)(a/ns)^T. */
int64_t gmmvb_stats_len(int K, int D);

/* Workspace on the CURRENT HIP device for up to max_rows rows of a [*, D] matrix of dtype x_dtype. */
int gmmvb_workspace_create(int K, int D, int x_dtype, int64_t max_rows, gmmvb_workspace** out);
int gmmvb_workspace_destroy(gmmvb_workspace* ws);
int64_t gmmvb_workspace_bytes(const gmmvb_workspace* ws);

/* A further workspace of a ROW-TILED job: same K, D, x_dtype as `first`, for up to max_rows <= first's rows, sharing
 * first's pass-local buffers (ln rho [K][rows] f64, the sample lists [K][rows] i32, the centred copy of the rows, the
 * M-step's slabs: two thirds of a workspace's bytes) and keeping everything it carries from one VB iteration to the next
 * (f32 bounds, records, digit planes, settled rows, row order, policy counters) of its own.  This is how a matrix whose
 * [N, K] arrays do not fit one GPU (reference `_gaussianmixture.py:835-836` keeps them in host RAM) runs with carried
 * bounds: N = 1e8, K = 256 in eight tiles is 219 GB instead of 550.  Use: per tile gmmvb_prepare_rows once, then per VB
 * iteration gmmvb_set_params (and gmmvb_set_drift) on every tile and gmmvb_estep_mstep tile after tile, adding up the
 * statistics blocks.  The shared buffers hold ONE tile's E-step output at a time: a call that writes them on another
 * tile (gmmvb_estep, gmmvb_load_responsibilities, gmmvb_prepare_rows) takes them over, after which the previous tile's
 * gmmvb_mstep and read-outs return GMMVB_ESTATE until its next E-step.  All workspaces of a group must be driven on one
 * stream; they may be destroyed in any order.  Not for HMM workspaces (hmmvb_enable refuses a group member) nor D > 256. */
int gmmvb_workspace_create_tile(gmmvb_workspace* first, int64_t max_rows, gmmvb_workspace** out);

/* Expansion point for the second moments (default: zeros).  Any fixed vector near the data keeps
 * B/ns - (a/ns)(a/ns)^T free of cancellation; results do not depend on it beyond rounding. */
int gmmvb_set_pivot(gmmvb_workspace* ws, const double* pivot_dev /*[D]*/, void* stream);

/* Optional, once per sample matrix (after gmmvb_set_pivot): build the workspace's centred f64 copy
 * xc = (double)x - pivot that gmmvb_mstep then streams instead of x whenever it is called with the same
 * (x_dev, ldx, n_rows).  Results are identical; the M-step loop loses its convert/subtract work, which on
 * gfx950 competes with the f64 MFMA pipe.  Costs 8 * 16*ceil(D/16) bytes per row of workspace; disabled
 * (a no-op) when the workspace was created under GMMVB_MSTEP_PRECENTER=0. */
int gmmvb_prepare_rows(gmmvb_workspace* ws, const void* x_dev, int64_t ldx, int64_t n_rows, void* stream);

/* Posterior expectations consumed by the E-step, replacing the reads of _e_ln_pi_vec,
 * _e_ln_lambda_dets, hn_kappas, hn_m_vecs and _e_lambda_mats at _gaussianmixture.py:773-781:
 *   c[k]  = E[ln pi_k] + (E[ln det Lambda_k] - D ln 2pi - D/kappa_k)/2
 *   m[k]  = hn_m_vecs[k]
 *   u[k]  = lower-triangular D x D (row-major) with u^T u = E[Lambda_k] = nu_k W_k, so that
 *           (x-m)^T E[Lambda_k] (x-m) = || u (x-m) ||^2.
 * Packs them into the kernel layout (MFMA tiles + bias -u m). */
int gmmvb_set_params(gmmvb_workspace* ws, const double* c_dev /*[K]*/, const double* m_dev /*[K][D]*/,
                     const double* u_dev /*[K][D][D]*/, void* stream);

/* E-step over rows [0, n_rows) of x (row stride ldx elements): ln rho_nk and the row log-normaliser
 * ln sum_k exp(ln rho_nk) are left in the workspace (replaces _gaussianmixture.py:773-783). */
int gmmvb_estep(gmmvb_workspace* ws, const void* x_dev, int64_t ldx, int64_t n_rows, void* stream);

/* Initialise from given responsibilities instead of an E-step (r row-major [n_rows][K]);
 * used by the 'random_responsibility' restart (_gaussianmixture.py:734-736). */
int gmmvb_load_responsibilities(gmmvb_workspace* ws, const double* r_dev, int64_t n_rows, void* stream);

/* M-step statistics for the responsibilities currently in the workspace (replaces
 * _gaussianmixture.py:725-732 and the N-sized term of :704).  stats_dev: gmmvb_stats_len doubles.
 * After a pruned E-step the sums run over per-component lists, and rows with a single active component (r = 1.0 exactly:
 * their addend does not depend on the parameters) are kept in a cache inside the workspace that only changes through the
 * rows entering or leaving it; the result is the same sum.  The cache follows the E-step / M-step alternation of one
 * sample matrix: any other call order is legal (a second gmmvb_mstep returns the same block, an E-step without an M-step
 * drops the cache) but a pass's first gmmvb_mstep must see the (x_dev, ldx, n_rows) of its gmmvb_estep. */
int gmmvb_mstep(gmmvb_workspace* ws, const void* x_dev, int64_t ldx, int64_t n_rows, double* stats_dev,
                void* stream);

/* gmmvb_estep followed by gmmvb_mstep (one VB data pass). */
int gmmvb_estep_mstep(gmmvb_workspace* ws, const void* x_dev, int64_t ldx, int64_t n_rows,
                      double* stats_dev, void* stream);

/* Read-outs for rows [row0, row0 + n_rows) of the last E-step, row-major [n_rows][K].
 * When the E-step pruned (large N K, sparse responsibilities; see gmmvb_last_sparsity), gmmvb_ln_rho returns, for the
 * pairs it did not evaluate, an upper bound of ln rho that lies at least 80 ln 2 below the row's log-normaliser, and
 * gmmvb_responsibilities returns exactly 0 for them (r < 2^-80): hard assignments and statistics are unaffected.  Pruning is never
 * used once hmmvb_enable has been called, or with GMMVB_ESTEP_PRUNE=0 in the environment. */
int gmmvb_responsibilities(gmmvb_workspace* ws, int64_t row0, int64_t n_rows, double* r_dev, void* stream);
int gmmvb_ln_rho(gmmvb_workspace* ws, int64_t row0, int64_t n_rows, double* out_dev, void* stream);
int gmmvb_argmax(gmmvb_workspace* ws, int64_t row0, int64_t n_rows, int32_t* z_dev, void* stream);

/* ---- row sharding over one process per GPU (the reference has no distributed code) -----------
 * The statistics block is linear in the rows: rank g runs gmmvb_estep / gmmvb_mstep on its own rows and ONE in-place
 * all-reduce(sum, f64) of the block per VB iteration makes every rank hold the statistics of the whole matrix (the
 * K-sized update is then computed identically on every rank).  RCCL over xGMI; librccl.so.1 is resolved at run time.
 *   rank 0:      gmmvb_comm_unique_id(id)  -> hand the 128 bytes to every rank (MPI / TCP / a file: the caller's choice)
 *   every rank:  gmmvb_comm_create(id, n_ranks, rank, &comm)      (collective; the current HIP device is the rank's GPU)
 *   per pass:    gmmvb_allreduce_stats(comm, stats_dev, gmmvb_stats_len(K, D), stream)     (enqueued on `stream`) */
typedef struct gmmvb_comm gmmvb_comm;
int gmmvb_comm_unique_id(unsigned char* id_out /*[128]*/);
int gmmvb_comm_create(const unsigned char* id /*[128]*/, int n_ranks, int rank, gmmvb_comm** out);
int gmmvb_comm_destroy(gmmvb_comm* comm);
int gmmvb_allreduce_stats(gmmvb_comm* comm, double* stats_dev, int64_t len, void* stream);

/* The block on the wire (ABI v7).  B_k = sum_n r_nk (x_n - p)(x_n - p)^T is symmetric and gmmvb_mstep mirrors it exactly,
 * so the lower triangle of every D x D block is redundant in the exchange: gmmvb_stats_pack writes
 *   [ns K | h K | a K D | upper triangles of B, row by row, K D (D + 1) / 2]   = gmmvb_stats_packed_len(K, D) doubles
 * (4.3 MB instead of 8.5 MB at K = 64, D = 128 and at K = 256, D = 64), gmmvb_stats_unpack restores the full block after
 * the all-reduce, mirroring the summed upper triangle - which also makes the reduced B exactly symmetric whatever order a
 * ring reduces its segments in.  Device pointers; the two buffers must not overlap.  The policy tail of a sharded job
 * (below) rides behind the packed block the same way it rode behind the full one.
 *   per pass:  gmmvb_estep_mstep(..., stats); gmmvb_stats_pack(K, D, stats, wire, stream);
 *              gmmvb_allreduce_stats(comm, wire, gmmvb_stats_packed_len(K, D) [+ GMMVB_POLICY_LEN], stream);
 *              gmmvb_stats_unpack(K, D, wire, stats, stream) */
int64_t gmmvb_stats_packed_len(int K, int D);
int gmmvb_stats_pack(int K, int D, const double* stats_dev, double* packed_dev, void* stream);
int gmmvb_stats_unpack(int K, int D, const double* packed_dev, double* stats_dev, void* stream);

/* One pass policy for all ranks.  Inside gmmvb_estep the library chooses between its dense kernel, a fresh bound pass
 * and carrying the previous pass's bounds from counters of the previous pass (how many pairs were active, evaluated,
 * ...).  Ranks deciding from their own shards' counters would part ways, and with one all-reduce per iteration the
 * slowest choice sets the step time for everybody.  A sharded workspace therefore decides from JOB-WIDE numbers only:
 *   once:       gmmvb_set_shard(ws, rows of the whole job, n_ranks)
 *   per pass:   gmmvb_estep_mstep(...); gmmvb_policy_export(ws, tail_dev, stream);
 *               all-reduce(sum) of [statistics block | tail] - the tail is GMMVB_POLICY_LEN doubles, e.g. kept right
 *               behind the statistics block so that the iteration still has ONE collective;
 *               gmmvb_policy_import(ws, tail_dev, stream)     (enqueues a copy to pinned host memory; the next
 *               gmmvb_estep waits for it, which costs nothing after the caller's per-iteration synchronisation).
 * Results never depend on any of this - only which kernels run.  gmmvb_last_work keeps reporting the rank's own numbers.
 * (The M-step's choice between its dense and its list form right after a DENSE E-step still uses the rank's own count.) */
#define GMMVB_POLICY_LEN 16
int gmmvb_set_shard(gmmvb_workspace* ws, int64_t global_rows, int n_ranks);
int gmmvb_policy_export(gmmvb_workspace* ws, double* out_dev /*[GMMVB_POLICY_LEN]*/, void* stream);
int gmmvb_policy_import(gmmvb_workspace* ws, const double* summed_dev /*[GMMVB_POLICY_LEN]*/, void* stream);

/* ---- Small problems: the whole restart x iteration loop of update_posterior in ONE launch (csrc/small.hip) ----------
 * Replaces, for shapes gmmvb_small_supported accepts (D <= 8, K <= 32, K (1 + D + D (D + 1) / 2) <= 256, n_rows <= 16384:
 * the sizes of BayesML's tutorials, where the general path is bound by its launches), the reference's driver loops
 *   for i in range(num_init): ... for t in range(max_itr): _update_q_mu_lambda / _update_q_pi / _update_q_z / _calc_vl,
 *   convergence test abs((vl - vl_before) / vl_before) < tolerance        (_gaussianmixture.py:846-872)
 * Workgroup r runs restart r; the caller draws the restarts' initial states in the reference's order:
 *   prior_dev = [alpha K | m KD | kappa K | nu K | w_inv KDD | ln B(W0, nu0) K | ln C(alpha0) 1]   (h0_* and :661-669)
 *   init_type 0 ("subsampling", :786-796):           init_dev = per restart [m KD | w_inv KDD] of the K sub-samples
 *   init_type 1 ("random_responsibility", :734-736): init_dev = per restart r [n_rows][K]
 * out_dev, per restart (gmmvb_small_out_len doubles):
 *   [ number of lower bounds L | converged 0/1 | p_x p_z p_pi p_mu_lambda q_z q_pi q_mu_lambda vl of the last pass |
 *     trace[max_itr + 1] (L valid) | alpha K | m KD | kappa K | nu K | w_inv KDD | w KDD |
 *     E[ln pi] K | E[ln det Lambda] K | ln B(W, nu) K | ns K | x_bar KD | s KDD ]
 * - the posterior that produced the restart's last data pass and that pass's moments (what ref :895 recomputes for the
 * winner).  r_dev (optional): [n_restarts][n_rows][K] responsibilities of every restart's last pass.  The winner rule
 * (:873) and the progress lines stay with the caller. */
int gmmvb_small_supported(int K, int D, int64_t n_rows);
int64_t gmmvb_small_out_len(int K, int D, int max_itr);
int gmmvb_small_fit(int K, int D, int x_dtype, const void* x_dev, int64_t ldx, int64_t n_rows, const double* pivot_dev /*[D]*/,
                    const double* prior_dev, int n_restarts, int init_type, const double* init_dev, int max_itr,
                    double tolerance, double* out_dev, double* r_dev, void* stream);

/* ---- Gaussian-emission HMM (bayesml/hiddenmarkovnormal/_hiddenmarkovnormal.py) -------------------------
 * The emission term is the GMM E-step without E[ln pi]: call gmmvb_set_params with
 *   c[k] = (E[ln det Lambda_k] - D ln 2pi - D/kappa_k)/2     (_calc_rho, :988-996)
 * and gmmvb_estep; then hmmvb_forward_backward replaces _forward :999-1006, _backward :1008-1011,
 * _update_gamma :1013-1014, _update_xi :1016-1018 and the ms / gamma part of _calc_n_m_x_bar_s :837-845:
 *   pi_tilde[k]   = exp(ln pi~_k - max)          (:862-863)
 *   a_tilde[i][j] = exp(ln a~_ij - max over all) (:866-867)
 *   out = [ ms[K][K] = sum_{t>=1} xi_t | gamma_0[K] | gamma_{T-1}[K] | sum_t ln c_t ]   (hmmvb_out_len doubles)
 * where c_t are the reference's scaling constants (the engine shifts ln rho per row and adds the shift back).
 * Afterwards the workspace holds gamma as the responsibilities: gmmvb_mstep returns ns / a / B for gamma
 * and h[k] = sum_t gamma_tk ln rho_tk (the first term of _vl_q_z, :905), gmmvb_responsibilities returns
 * gamma and gmmvb_argmax its row-wise argmax.  hmmvb_enable allocates the forward-backward buffers
 * (4 [T][16 ceil(K/16)] + 1 [K][T] doubles); K <= 64. */
int64_t hmmvb_out_len(int K);
int hmmvb_enable(gmmvb_workspace* ws);
int hmmvb_forward_backward(gmmvb_workspace* ws, int64_t n_rows, const double* pi_tilde_dev, const double* a_tilde_dev,
                           double* out_dev, void* stream);
/* How the last hmmvb_forward_backward call got its chunk boundary vectors (diagnostics / tests; waits for that call):
 *   -1  from the chunk transfer products (short sequences, or the forgetting pass held off),
 *    0  from the forgetting pass - both recursions swept from uniform start vectors, the replays' own boundary vectors agreed
 *       with the sweeps' to 2e-14 -, 1  the forgetting pass ran, its vectors did not stand and the products path ran behind it.
 *   -2  no HMM state / error.  The result of hmmvb_forward_backward is the same in all three cases (to rounding). */
int hmmvb_last_boundary_pass(gmmvb_workspace* ws);
/* The same for the last hmmvb_viterbi call: -1 chunk start vectors from the max-plus chunk matrices (or the sequential
 * kernels), 0 from the coalescence pass (a sweep of the recursion from zero start vectors; the replay's own end vectors agreed
 * with the sweep's to 1e-9 nats), 1 the pass ran, did not stand, and the chunk-matrix path ran behind it.  Synchronises. */
int hmmvb_last_viterbi_pass(gmmvb_workspace* ws);

/* skip = 1: the gmmvb_mstep calls that follow hmmvb_forward_backward leave the h block of the statistics at 0 and do not read
 * the ln rho array (a third of that kernel's traffic): sum_t gamma_tk ln rho_tk, the only use of h on the HMM path
 * (_hiddenmarkovnormal.py:905), follows from the moments of the same block in closed form,
 * ns_k (c_k - ((s_k o nu_k W_k).sum() + (x_bar_k - m_k)^T nu_k W_k (x_bar_k - m_k)) / 2), the expression the reference
 * itself uses for E[ln p(x|z)] (:871-877).  Default 0. */
int hmmvb_skip_h(gmmvb_workspace* ws, int skip);

/* Where the next gmmvb_estep calls of an HMM workspace put the emission.  fused = 0 (default): the ln rho array, as for a
 * mixture - hmmvb_forward_backward, hmmvb_viterbi and the ln rho read-out all work from it.  fused = 1: for shapes the
 * library covers (*in_effect = 1: one feature tile, D <= 16, and at most 32 states) the emission kernel writes
 * rho' = exp(ln rho - row maximum) and the maxima straight into the forward-backward buffers and forms NO ln rho array
 * (its 16 N K bytes of traffic and a launch per pass are saved).  Then only hmmvb_forward_backward may follow;
 * hmmvb_viterbi and gmmvb_ln_rho return GMMVB_ESTATE until a pass with target 0, and the statistics block of the following
 * gmmvb_mstep carries h = 0 as with hmmvb_skip_h.  Other shapes: *in_effect = 0 and nothing changes.  in_effect may be null. */
int hmmvb_emission_target(gmmvb_workspace* ws, int fused, int* in_effect);

/* Viterbi path (estimate_latent_vars(loss="0-1", viterbi=True), :1465-1481) from the emission ln rho of the
 * last gmmvb_estep: z_dev[t] = state index of the most probable path (first maximiser on ties). */
int hmmvb_viterbi(gmmvb_workspace* ws, int64_t n_rows, const double* ln_pi_tilde_dev, const double* ln_a_tilde_dev,
                  int32_t* z_dev, void* stream);

/* Row-range read-outs of the last hmmvb_forward_backward, natural state order, rows [row0, row0 + n_rows) - the
 * reference's alpha_vecs / beta_vecs [T][K] and xi_mats [T][K][K] attributes (_hiddenmarkovnormal.py:1063-1069), formed on
 * demand instead of being materialised (xi_mats is 82 GB at config 5):
 *   what 0: alpha_t [n_rows][K];  1: beta_t = gamma_t / alpha_t [n_rows][K] (gamma = alpha o beta as in the reference);
 *   what 3: xi_t [n_rows][K][K] = (alpha_{t-1}^T w_t) o a_tilde with xi_0 = 0; a_tilde_dev = the matrix given to the
 *           forward-backward call (ignored for what 0 / 1). */
int hmmvb_readout(gmmvb_workspace* ws, int what, int64_t row0, int64_t n_rows, const double* a_tilde_dev, double* out_dev,
                  void* stream);

/* test/diagnostic read-out of the last pass: what = 0 alpha ([n_rows][16 ceil(K/16)], lane order: state
 * 16b + (g + 4r) at position 16b + 4g + r), 1 c' ([n_rows]), 2 row shift max_k ln rho ([n_rows]). */
int hmmvb_debug_readout(gmmvb_workspace* ws, int what, int64_t row0, int64_t n_rows, double* out_dev, void* stream);

/* Optional in-library timing with HIP events recorded on the launch stream: gmmvb_profile_last_ms gives the
 * E-step phase (first to last kernel of gmmvb_estep: the dense kernel estep_lds_f64 and the log-normaliser pass, or bound
 * pass / sweep + selections + proof round + gathers + rec_finish; v4: up to v3 the phase ended before the log-normaliser /
 * rec_finish kernels) and the M-step phase (list building + mstep_mfma_f64 / mstep_list_f64) of the
 * last passes (bench.py's roofline leg); it waits for those events.  on = 2 ("dominant groups only"): an event record
 * costs the stream ~10 us, a converged step at the benchmark shape has ~36 of them; level 2 records only the spans
 * estep_main, estep_gather and mstep_main (gmmvb_profile_spans) and no phase events (gmmvb_profile_last_ms gives -1). */
int gmmvb_profile_enable(gmmvb_workspace* ws, int on);
int gmmvb_profile_last_ms(gmmvb_workspace* ws, float* estep_ms, float* mstep_ms);

/* Finer view of the same profile: HIP-event time (ms, summed over the launches of a group) and number of launch
 * groups per slot for the last gmmvb_estep + gmmvb_mstep; gmmvb_profile_span_name(slot) names the slots
 * ("estep_main" = dense / bound kernel, "estep_select", "estep_gather", "estep_lse_mask", "mstep_lists",
 * "mstep_main" = dense / list kernel, "mstep_reduce", "estep_proof" = the int8 proof round).  Waits for the last recorded event. */
int gmmvb_profile_spans(gmmvb_workspace* ws, float* ms /*[8]*/, int* launches /*[8]*/);
const char* gmmvb_profile_span_name(int slot);

/* Kernel names and launch geometry of the last estep/mstep (for profiling reports); returns a
 * static string such as "estep_mfma_f64<8,2,f32,vec> grid=1024x256". */
const char* gmmvb_last_launch_info(const gmmvb_workspace* ws);

/* K-sized linear algebra of the posterior update (replaces np.linalg.inv + slogdet of _gaussianmixture.py:746-756,
 * 769): for each of K symmetric positive definite D x D matrices W^-1 (row-major, lower triangle read) the Cholesky
 * factor G (W^-1 = G G^T), its inverse G^-1 (both lower triangular, upper part zero) and ln det W^-1.  The caller forms
 * u = sqrt(nu) G^-1, W = G^-T G^-1.  D <= 128; one workgroup per matrix; no workspace, no synchronisation (graph-capturable). */
int gmmvb_kside_factor(int K, int D, const double* w_inv_dev /*[K][D][D]*/, double* g_dev /*[K][D][D]*/,
                       double* g_inv_dev /*[K][D][D]*/, double* logdet_dev /*[K]*/, void* stream);

/* The drift hint of gmmvb_set_drift for the update (m_old, u_old) -> (m_new, u_new) of K components, in one launch:
 * gamma = 1 / ub(|| u_old u_new^-1 ||_2), big_gamma = ub(|| u_new u_old^-1 ||_2), delta = || u_new (m_new - m_old) ||, where
 * ub is the rigorous upper bound || (A^T A)^(2^s) ||_F^(1/2^(s+1)) (s = squarings resp. squarings_big repeated
 * squarings; at most D^(1/2^(s+1)) above the true norm), and enorm = ub(|| u_new u_old^-1 - I ||_2): the caller may
 * sharpen the hint to gamma = max(gamma, 1 - enorm), big_gamma = min(big_gamma, 1 + enorm), which is far tighter once
 * the components hardly move.  u, u^-1: [K][D][D] row-major (lower triangular); D <= 128. */
int gmmvb_kside_drift(int K, int D, const double* u_old_dev, const double* uinv_old_dev, const double* m_old_dev,
                      const double* u_new_dev, const double* uinv_new_dev, const double* m_new_dev, int squarings,
                      int squarings_big, double* gamma_dev /*[K]*/, double* delta_dev /*[K]*/,
                      double* big_gamma_dev /*[K]*/, double* enorm_dev /*[K]*/, void* stream);

/* The whole K-sized part of one VB iteration (everything update_posterior does between two data passes,
 * _gaussianmixture.py:671-770 minus the N-sized sums) in one call - three launches, no synchronisation:
 *   moments    x_bar = pivot + a/ns, S = B/ns - (a/ns)(a/ns)^T from the statistics block (ns > 0; else 0 / the previous S)
 *   scal[0..7] the lower bound's terms p_x, p_z, p_pi, p_mu_lambda, q_z, q_pi, q_mu_lambda and their sum under q
 *   q_next     the closed-form update from the prior and the moments, with its derived expectations, the whitening factor
 *              u (u^T u = E[Lambda]), u^-1 and the E-step constant c - ready for gmmvb_set_params
 *   drift      (want_drift) gamma / delta / big_gamma of q -> q_next for gmmvb_set_drift; scal[8] = min_k (gamma_k - delta_k/30)
 * All pointers are device pointers; q and q_next must not alias.  s_prev [K][D][D] is read (components with ns = 0 keep
 * their previous S, like the reference's stale s_mats) and overwritten with S.  scratch: 13 K doubles.  D <= 128. */
typedef struct gmmvb_prior_view {
    const double *alpha, *m, *kappa, *nu, *w_inv, *ln_b_w_nu;      /* [K], [K][D], [K], [K], [K][D][D], [K] */
    double ln_c_alpha;                                             /* ln C(alpha_0), _gaussianmixture.py:662 */
} gmmvb_prior_view;
typedef struct gmmvb_post_view {
    double *alpha, *m, *kappa, *nu, *w_inv, *w, *u, *u_inv;       /* [K], [K][D], [K], [K], then four [K][D][D] */
    double *e_ln_pi, *e_ln_lambda_det, *ln_b_w_nu, *c;            /* [K] each */
} gmmvb_post_view;
int gmmvb_kside_step(int K, int D, const gmmvb_prior_view* prior, const gmmvb_post_view* q, const gmmvb_post_view* q_next,
                     const double* stats_dev, const double* pivot_dev, double* s_prev_dev, double* ns_dev /*[K]*/,
                     double* x_bar_dev /*[K][D]*/, double* s_dev /*[K][D][D]*/, int want_drift, double* gamma_dev,
                     double* delta_dev, double* big_gamma_dev, double* scal_dev /*[9]*/, double* scratch_dev /*[13 K]*/,
                     void* stream);

/* The Dirichlet half of hiddenmarkovnormal.LearnModel's K-side in one launch (ABI v7; csrc/kside.hip).  With gmmvb_kside_step on
 * views of the HMM posterior's Normal-Wishart buffers (the mixture's Dirichlet fields of those views are dummies) it replaces
 * _hiddenmarkovnormal.py:883-943 (_calc_vl), :980-986 (_update_q_pi, _update_q_a) and :861-868 (their features):
 *   scal[10] <- p_x p_z p_pi p_a p_mu_lambda q_z q_pi q_a q_mu_lambda vl under the CURRENT posterior (eta, zeta, ln pi~, ln a~,
 *               ln C(zeta) summed over rows), the forward-backward summary fb = [ms K x K | gamma_0 K | gamma_last K | sum ln c]
 *               (hmmvb_forward_backward's output) and scal_nw = gmmvb_kside_step's nine doubles ([0], [3], [6] are used);
 *               h_scale[0] = 0 when the pass had no emission (_init_random_responsibility), else 1;
 *   eta', zeta' = prior + ns / ms, ln pi~', pi~', ln a~', a~' (one global maximum), ln C(zeta') of the next posterior.
 * Device pointers, float64; K <= 256. */
int hmmvb_kside_dirichlet(int K, const double* eta0_dev, const double* zeta0_dev, double ln_c_eta0, double ln_c_zeta0,
                          const double* eta_dev, const double* zeta_dev, const double* ln_pi_dev, const double* ln_a_dev,
                          const double* ln_c_zeta_dev, const double* fb_dev, const double* ns_dev, const double* scal_nw_dev,
                          const double* h_scale_dev, double* eta_next_dev, double* zeta_next_dev, double* ln_pi_next_dev,
                          double* pi_next_dev, double* ln_a_next_dev, double* a_next_dev, double* ln_c_zeta_next_dev,
                          double* scal_dev, void* stream);

/* Which kernels have run in this workspace since it was created (for tests and profiling reports):
 *   out[0] dense E-steps, out[1] bound passes of the pruned E-step, out[2] E-steps on carried records, out[3] E-steps
 *   sent back to the dense kernel after a bound pass that pruned nothing, out[4] E-steps on a sweep of carried bounds
 *   over the dense ln rho array, out[5] dense M-steps,
 *   out[6] M-steps over active-row lists, out[7] candidate gathers (exact f64 evaluation of listed pairs). */
int gmmvb_pass_counts(const gmmvb_workspace* ws, int64_t* out /*[8]*/);

/* Optional hint for the pruned E-step, to be given BEFORE the gmmvb_set_params of new parameters: for every component k
 *   gamma[k] <= sigma_min(u_new u_old^-1),  big_gamma[k] >= sigma_max(u_new u_old^-1),  delta[k] >= || u_new (m_new - m_old) ||_2
 * where (m_old, u_old) are the parameters of the last gmmvb_estep.  Then for every x
 *   gamma || u_old (x - m_old) || - delta  <=  || u_new (x - m_new) ||  <=  big_gamma || u_old (x - m_old) || + delta,
 * which lets the next gmmvb_estep carry its per-row candidate records (csrc/records.h: up to 8 (component, distance)
 * slots and one bound for all other components, 55 bytes per row) over to the new parameters instead of bounding every
 * pair afresh: pairs whose carried bound proves r < 2^-80 are skipped, the others are evaluated exactly, and a row
 * whose record has become too loose has all K pairs evaluated.  Loose values only cost candidates, wrong ones (gamma too
 * large, big_gamma or delta too small) break the bounds.  typical_gamma: a pessimistic summary of the drift if the
 * caller has one on the host - min_k (gamma_k - delta_k / 30) is what bayesml_amd passes - or a value <= 0 if not: below
 * 0.985 the library carries by a sweep of its dense ln rho array (every pair with its own component's drift) instead of the
 * records, below 0.5 it bounds afresh.  The hint is consumed by the next gmmvb_estep. */
int gmmvb_set_drift(gmmvb_workspace* ws, const double* gamma_dev /*[K]*/, const double* delta_dev /*[K]*/,
                    const double* big_gamma_dev /*[K]*/, double typical_gamma, void* stream);
/* 1 if an E-step over n_rows rows of this workspace can make use of gmmvb_set_drift (pruning is possible at all),
 * else 0: lets the caller skip computing the hint. */
int gmmvb_wants_drift(const gmmvb_workspace* ws, int64_t n_rows);
/* The parameters given next are unrelated to those of the last E-step (a new restart): the next gmmvb_estep does not
 * assume that the responsibilities are still as sparse as they were and runs the dense kernel. */
int gmmvb_forget(gmmvb_workspace* ws);

/* How often the workspace has regrouped its internal row order by dominant component since it was created (DESIGN.md
 * section 5c; the caller never sees that order: every read-out is in the caller's row order). */
int64_t gmmvb_regroup_count(const gmmvb_workspace* ws);

/* test / diagnostic read-out (blocking) of one row's candidate record, 26 doubles to HOST memory: slot components
 * (-1 = empty) [0..7], slot distances [8..15], rest bound B [16], exact / selected bits [17] [18], flags [19], best
 * component [20], lse [21], the row's mask words [22..25]. */
int gmmvb_debug_record(gmmvb_workspace* ws, int64_t row, double* out /*[26], host*/);

/* test / diagnostic: the proof round's kernel (csrc/estep_i8.h, estep_i8_proof: three int8 digits per operand, exact int32
 * accumulation, rigorous error term) for component k and EVERY row of the matrix last given to gmmvb_prepare_rows, in the
 * caller's row order if the rows have not been regrouped: ub_dev[n] >= ln rho_nk >= lb_dev[n] under the parameters of
 * the last gmmvb_set_params.  Discards the workspace's E-step state. */
int gmmvb_debug_proof(gmmvb_workspace* ws, int k, int64_t n_rows, float* ub_dev /*[n_rows]*/, double* lb_dev /*[n_rows]*/,
                      void* stream);

/* Sparsity of the last gmmvb_estep: *active_pairs = number of (row, component) pairs whose responsibility is at
 * least 2^-80 of the row's total, *evaluated_pairs = pairs whose ln rho was evaluated exactly (n_rows * K unless
 * the E-step pruned; pruned pairs hold an upper bound that proves r < 2^-80).  *active_pairs = -1 when the
 * library did not count (GMMVB_MSTEP_SPARSE=0, tiny passes).  Waits for the E-step's counters (the one entry point
 * besides gmmvb_profile_* that blocks; gmmvb_mstep blocks the same way right after a DENSE gmmvb_estep of N K >= 2^18,
 * to choose between its dense and its list form). */
int gmmvb_last_sparsity(gmmvb_workspace* ws, void* stream, double* active_pairs, double* evaluated_pairs);
/* The same (blocking) with the work the pass left for the M-step.  out[0] active pairs (r >= 2^-80; -1: not counted),
 * out[1] pairs the E-step evaluated exactly, out[2] pairs the list M-step accumulates (rows whose single component has
 * r = 1.0 exactly keep their addend in a cache and are only touched when that changes; -1: not counted), out[3] rows the
 * E-step did not evaluate at all (settled: their carried bounds prove that nothing changed), out[4] of the pairs in
 * out[1], those whose evaluation stopped after half of the output blocks because the partial sum already put them below
 * the row's relevance threshold (they cost 10 of the 36 tile pairs at D = 128), out[5] pairs of the proof round: settled
 * rows whose carried bounds left candidates get two-sided bounds of their component and of the candidates from three int8
 * digits (about a fifth of an exact evaluation's cost) - rows that are proven to keep a single active component are not
 * evaluated at all (their responsibility is 1.0 to the last bit whatever the value); out[6] after a sweep: the (tile of
 * 256 rows, component) columns of the per-pair bound array it had to go through, of ceil(n_rows / 256) * K (-1: the pass
 * was no sweep, or GMMVB_SWEEP_LAZY=0); out[7] after a sweep that used the table of csrc/project.h (bounds from the parameters
 * in force and the rows' int8 digit planes): the pairs it took off the proof round's lists - with GMMVB_PROJECT=only, where the
 * table replaces the carried bounds, the pairs it did not clear (-1: the pass used no table). */
int gmmvb_last_work(gmmvb_workspace* ws, double* out /*[8], host*/);

/* The pass policy's unit costs (ABI v8; csrc/policy.h).  gmmvb_estep / gmmvb_mstep choose between kernels whose results agree;
 * the thresholds of that choice follow from per-pair costs.  A workspace starts from literals measured on MI355X at the
 * benchmark shape, scaled to its own tile counts, and - unless gmmvb_policy_calibrate(ws, 0) - replaces the three bulk costs
 * by what its OWN first dense E-step, dense M-step and full bound pass (of at least 2^23 pairs) take on its device: HIP
 * events around those launches, taken over without a synchronisation once they have completed; a measurement outside
 * [1/2, 2] x the scaled literal is discarded.  Results never depend on the table.
 * gmmvb_policy_table: out[0..5] = ns per (row, component) pair in force: dense E, dense M, bound pass, exact pair, proof pair,
 * list M-step; out[6..8] the scaled literals of the first three; out[9..11] the thresholds that follow (pruned E-step while
 * active / K is below [9], dense again above [10] evaluated / K, list M-step below [11]); out[12] bit mask of the costs
 * measured (1 dense E, 2 dense M, 4 bound pass), out[13] of measurements discarded, out[14] calibration on, out[15] 0. */
#define GMMVB_POLICY_TABLE_LEN 16
int gmmvb_policy_calibrate(gmmvb_workspace* ws, int on);
int gmmvb_policy_table(gmmvb_workspace* ws, double* out /*[GMMVB_POLICY_TABLE_LEN], host*/);

/* ---- device-side data generation (ABI v8) -------------------------------------------------------------
 * GenModel.gen_sample of the mixture (bayesml/gaussianmixture/_gaussianmixture.py:241-264: a `choice` and a
 * `multivariate_normal` per row in a Python loop) and of the HMM (bayesml/hiddenmarkovnormal/_hiddenmarkovnormal.py:344-358,
 * the same along a Markov chain) as kernels over a COUNTER-BASED stream that the host can reproduce value by value:
 * Philox4x64-10 exactly as numpy.random.Philox(key=[seed, stream]) runs it (block L of four 64-bit outputs comes from the
 * counter value L + 1).  Stream 0 carries one uniform per row / time step t, u_t = (raw_t >> 11) 2^-53; the class drawn
 * from an inclusive cumulative distribution cdf[K] is the number of its first K - 1 entries that are <= u.  Stream 1 carries
 * the normals: row r owns the ceil(D / 4) blocks from r ceil(D / 4) on, a block (r0..r3) gives four normals by Box-Muller
 * (sqrt(-2 ln(1 - (r0 >> 11) 2^-53)) (cos, sin)(2 pi (r1 >> 11) 2^-53), the same from (r2, r3)).  Rows are numbered
 * globally: `row0` is the number of z_dev[0] / x_dev's first row, so any window of a sample can be drawn on its own (row
 * shards, tiles).  Not the reference's own stream (PCG64 through Generator.choice / multivariate_normal): same
 * distributions, reproducible per seed; the host path of gen_sample keeps the reference's stream. */
/* z[t] ~ Categorical(pi), t in [row0, row0 + n_rows): cdf_dev = inclusive cumulative sums of pi [K] */
int gmmvb_sample_latent(int K, const double* cdf_dev /*[K]*/, uint64_t seed, int64_t row0, int64_t n_rows,
                        int64_t* z_dev /*[n_rows]*/, void* stream);
/* z[0] ~ pi, z[t] ~ a_mat[z[t-1]] for a whole sequence [0, n_rows) without a sequential pass over it: chunks of 256 steps
 * carry all K start states at once (a step is a map on the states and maps compose), groups of 256 chunk maps are composed
 * the same way, one thread chains the groups, and the start states flow back down - five launches, no host round trip.
 * work_dev: gmmvb_sample_chain_work_bytes(K, n_rows) bytes of 256-byte aligned scratch. */
int64_t gmmvb_sample_chain_work_bytes(int K, int64_t n_rows);
int gmmvb_sample_chain(int K, const double* cdf_pi_dev /*[K]*/, const double* cdf_a_dev /*[K][K] row-wise cumulative*/,
                       uint64_t seed, int64_t n_rows, int64_t* z_dev /*[n_rows]*/, void* work_dev, int64_t work_bytes,
                       void* stream);
/* x[r] = mu[z[r]] + eps_r A[z[r]] in f64, stored as x_dtype; A[k] lower triangular with A[k]^T A[k] = Lambda_k^-1
 * (A = L^-1 for Lambda = L L^T); eps_r = the D normals of global row row0 + r.  D <= 600.  work_dev (optional, NULL = none):
 * gmmvb_sample_emissions_work_bytes(K, n_rows) bytes of scratch in which the rows are grouped by class first, so that the
 * factors are read from L2 - same values, about ten times faster once K D^2 doubles exceed an L2. */
int64_t gmmvb_sample_emissions_work_bytes(int K, int64_t n_rows);
int gmmvb_sample_emissions(int K, int D, const int64_t* z_dev /*[n_rows]*/, const double* mu_dev /*[K][D]*/,
                           const double* a_dev /*[K][D][D]*/, uint64_t seed, int64_t row0, int64_t n_rows, int x_dtype,
                           void* x_dev, int64_t ldx, void* work_dev, int64_t work_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GMMVB_H */
