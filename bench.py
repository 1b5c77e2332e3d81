#!/usr/bin/env python3
"""GMM-VB E+M samples/sec at K=64, D=128 (BASELINE.json metric), one process per GPU.

A "step" is one full VB iteration of ``gaussianmixture.LearnModel.update_posterior``'s inner loop
(reference ``_gaussianmixture.py:864-867``): K-side posterior update -> parameter packing -> E-step ->
row log-normaliser -> M-step statistics -> slab reduction -> (all-reduce over row shards) -> moments ->
variational lower bound (one host sync).  x is resident in HBM before the timed region.
``value`` = rows processed by all ranks per second.

    python bench.py --gpus N --steps K --warmup W [--config c3|c4|c2] [--scaling weak|strong] [--overlap] [--dense]

``--gpus N`` with N > 1 and no WORLD_SIZE in the environment starts the N rank processes itself (before this
process touches a GPU); under ``python -m torch.distributed.run`` the ranks come from the environment.  Every rank
joins an RCCL ("nccl") process group and the JSON line reports ``ranks`` and ``backend``.  A SIGTERM / SIGINT to the parent
stops every rank (they run in their own sessions).

Configurations (BASELINE.json ``configs``): c3 = K64 D128 N1e7 f32 (default, the one the metric is quoted on),
c4 = K256 D64 1.25e7 rows per GPU (N = 1e8 over 8 GPUs), c2 = K16 D32 N1e6 f64.  ``--scaling weak`` (default):
every rank holds the configuration's rows; ``--scaling strong``: ``--total-rows`` are split over the ranks.

Output: stdout carries ONE compact JSON line (< 6 KB: the contract's keys, `roofline`, `cpu_baseline`, both parity legs,
`window`, `full_fit_seconds`, `dense_floor_samples_per_s`, the sub-lines `hmm_c5`, `c2`, `c4_shard`, `c4_strong`); the full
record with every leg, per-step lists and kernel groups goes to ``--detail`` (default gpurun_out/bench_detail.json).
Default legs: dense, full, hmm, c2, c4, c4strong (about a minute); hard, spread, offpath, small on request (``--legs all``).
With ``--gpus N`` > 1 the line carries the C3 weak-scaling window and, as `c4_strong`, config 4's N = 1e8 split over the N
ranks (ms_per_step, per_rank_ms_per_step, allreduce_ms from its own HIP events, estep_kinds_identical_across_ranks); the
one-GPU record carries the same job as resident row tiles - the strong-scaling curve is the ratio of those sub-lines.

Legs of the single-GPU run (full record):
  value/roofline  the default policy (sparse where the responsibilities are), timed
  dense           the same data pass with the dense f64 MFMA kernels (every pair evaluated): the floor the policy
                  falls back to, its executed TFLOP/s against the f64 MFMA peak, and the difference of the
                  statistics block between the two paths on identical parameters
  hard_workload   heavily overlapping clusters (means 0.3 * randn): what the policy does when nothing can be pruned
  spread_sweep    the middle of the separation spectrum: means spread * randn for spread in {0.5, 0.75, 1.0, 1.5} at N = 2e6,
                  default policy against dense kernels only over iterations 3-22, each with an oracle parity run
  full_fit        update_posterior(max_itr=25, num_init=2, tolerance=0) through the public API on the resident matrix
  offpath         off the headline's recipe at N = 2e6: K_model = 2 K_data, mixing weights ~ Dirichlet(0.3), anisotropic clusters -
                  default policy against dense kernels only and an oracle parity run under both policies (opt-in: --legs)
  hmm_c5          BASELINE.json configs[4]: hiddenmarkovnormal.LearnModel K=32, D=16, T=1e7 (tools/bench_hmm.py's measurement)
  c2, c4_shard    BASELINE.json configs[1] (K16 D32 N1e6 f64) and one GPU's shard of configs[3] (K256 D64, 1.25e7 rows):
                  VB iterations 6-25 of one restart, dominant kernel, frac_executed and frac_on_F
  c4_strong       configs[3] itself: N = 1e8 over all ranks (one GPU: eight resident row tiles)
  small_c1        BASELINE.json configs[0] (K=3, D=2, N=1000) with the reference's defaults: the one-launch path and the
                  general engine
  cpu_baseline    the oracle (NumPy port of the reference's formulation) on this host's cores, 10 VB iterations over
                  the first N_ref rows
  parity          GPU driver vs oracle on those rows after 10 iterations (north_star tolerance 1e-5), twice: with the
                  default policy (dense kernels at that size) and with the sparse path forced (int8 bound pass,
                  carried bounds, candidate gathers, list M-step - the kernels of the timed steps)

roofline: `frac` = `frac_executed` = f64 MFMA flops the dominant kernel issued / its HIP-event time /
78.6 TFLOP/s (a utilisation); `frac_on_F` = SURVEY 8d's dense flop count F of that phase x rows / the same time / the same
peak (>> 1 on the pruned path: an algorithmic saving, not a utilisation); `step_frac_on_F` the same for the whole step.
Every kernel group carries both fractions - ``hbm_frac`` = algorithmic bytes (rows it
has to read x D x s) / its HIP-event time / 8 TB/s, ``f64_mfma_frac`` = executed f64 MFMA flops / 78.6 TFLOP/s -
and ``bound`` names the larger; the line's bound / achieved / peak / frac are the dominant group's.  ``step_hbm_frac`` = the
whole step's N D s bytes / step time / 8 TB/s (= value / HBM-roofline samples/s).  ``per_rank``: every rank's own step times.
"""
import argparse
import gc
import json
import os
import socket
import subprocess
import sys
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SEED = 20250711
PEAK_HBM_GBPS = 8000.0           # MI355X_MICROARCH.md: HBM3E ~8 TB/s
PEAK_F64_MFMA_TFLOPS = 78.6      # MI355X datasheet FP64 matrix (the guide lists no f64 row; tools/peak_probe.hip: 72.7-74.2)
CONFIGS = {"c2": dict(classes=16, degree=32, rows=1_000_000, dtype="f64", total=1_000_000),
           "c3": dict(classes=64, degree=128, rows=10_000_000, dtype="f32", total=10_000_000),
           "c4": dict(classes=256, degree=64, rows=12_500_000, dtype="f32", total=100_000_000)}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="c3", choices=sorted(CONFIGS))
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--total-rows", type=int, default=None, help="strong scaling: rows of the whole job")
    ap.add_argument("--rows", type=int, default=None, help="rows per GPU (overrides the configuration)")
    ap.add_argument("--classes", type=int, default=None)
    ap.add_argument("--degree", type=int, default=None)
    ap.add_argument("--dtype", default=None, choices=["f32", "f64"], help="storage dtype of x in HBM")
    ap.add_argument("--ref-rows", type=int, default=20_000)
    ap.add_argument("--overlap", action="store_true", help="hard workload: cluster means 0.3 * randn instead of 2 * randn")
    ap.add_argument("--spread", type=float, default=None, help="cluster means spread * randn (default 2.0; --overlap = 0.3)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline / parity legs")
    ap.add_argument("--no-legs", action="store_true",
                    help="skip the dense, hard-workload, spread-sweep, full-fit and HMM legs")
    ap.add_argument("--legs", default="dense,full,hmm,c2,c4,c4strong",
                    help="comma-separated legs (dense, full, hmm and the sub-lines c2, c4 (one shard of config 4), c4strong "
                         "(config 4's N = 1e8 over all ranks; on one GPU as resident row tiles) by default: about a minute "
                         "together; hard, spread, offpath, small on request; 'all' = every leg).  With --gpus N > 1 only "
                         "c4strong runs")
    ap.add_argument("--strong-total-rows", type=int, default=CONFIGS["c4"]["total"],
                    help="rows of the whole job in the c4strong sub-line (default config 4's 1e8)")
    ap.add_argument("--detail", default=os.path.join("gpurun_out", "bench_detail.json"),
                    help="file that receives the full record (every leg, per-step lists, kernel groups); stdout carries "
                         "only the compact line.  '' = do not write it, '-' = print it on stdout BEFORE the compact line")
    ap.add_argument("--spans", choices=("dominant", "full"), default="dominant",
                    help="HIP-event spans recorded inside the timed steps: 'dominant' = only the kernel groups that can dominate "
                         "a step (estep_main, estep_gather, mstep_main: what the roofline block needs); 'full' = every group and "
                         "the E/M phase events (~36 records a step, ~10 us of stream time each)")
    ap.add_argument("--dense", action="store_true", help="no pruning, no sparse M-step: every pair in f64")
    ap.add_argument("--force-dist", action="store_true",
                    help="join an RCCL process group even with one rank (exercises the N > 1 code path on a 1-GPU box)")
    ap.add_argument("--init-timeout", type=float, default=600.0,
                    help="seconds a rank may spend joining the process group and finishing its first collective before its "
                         "watchdog dumps every thread's stack and exits non-zero (a stalled RCCL init must not hang the job)")
    ap.add_argument("--native-allreduce", action="store_true",
                    help="per-iteration all-reduce through the library's own RCCL communicator (gmmvb_allreduce_stats)")
    return ap.parse_args()


def visible_gpus():
    """GPUs this process could use, WITHOUT touching the HIP runtime (the parent of the rank processes must never
    initialise a GPU, and on ROCm even torch.cuda.device_count() may): the *_VISIBLE_DEVICES lists if set, else the KFD
    topology's nodes that have SIMDs.  None if neither is available."""
    for key in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(key)
        if v is not None and v.strip():
            return len([t for t in v.split(",") if t.strip()])
    root = "/sys/class/kfd/kfd/topology/nodes"
    try:
        n = 0
        for node in os.listdir(root):
            with open(os.path.join(root, node, "properties")) as f:
                props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
            n += int(props.get("simd_count", "0")) > 0
        return n
    except OSError:
        return None


def launch_ranks(n):
    """Start n rank processes of this script (rank r on GPU r).  The parent never initialises a GPU."""
    have = visible_gpus()
    if os.environ.get("BENCH_SHARE_GPU") == "1":      # developer switch: every rank on GPU 0, gloo instead of RCCL
        have = None
    if have is not None and have < n:
        print(f"bench.py: --gpus {n} but only {have} GPU(s) visible", file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    import signal

    def stop_ranks(grace=5.0):
        """SIGTERM to every live rank's process group (each rank is the leader of its own session), SIGKILL after `grace`."""
        live = [p for p in procs if p.poll() is None]
        for p in live:
            try:
                os.killpg(p.pid, signal.SIGTERM)          # the exact process groups started below
            except OSError:
                pass
        end = time.time() + grace
        while time.time() < end and any(p.poll() is None for p in live):
            time.sleep(0.1)
        for p in live:
            if p.poll() is None:
                try:
                    os.killpg(p.pid, signal.SIGKILL)
                except OSError:
                    pass

    def on_signal(signum, _frame):
        # `timeout ... python bench.py --gpus N` or Ctrl-C reaches only this parent: the ranks live in their own sessions
        # and would run on as orphans inside a collective, holding the GPUs
        print(f"bench.py: signal {signum}; stopping the rank processes", file=sys.stderr)
        stop_ranks()
        os._exit(128 + signum)

    old_handlers = {sig: signal.signal(sig, on_signal) for sig in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP)}
    rc = 0
    try:
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(port))
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            env.setdefault("NCCL_DEBUG", "WARN")
            env.setdefault("GLOO_SOCKET_IFNAME", "lo")          # (one node: the container's hostname may not resolve)
            # host BLAS threads per rank (the K-sized NumPy linear algebra of the model's constructor): n ranks with one pool of
            # os.cpu_count() spinning threads each stall one another - measured: eight ranks on a 256-core host sat in
            # numpy.linalg.inv for minutes (profiles/r6_experiments.md); torch.distributed.run sets OMP_NUM_THREADS=1 itself
            per_rank = str(max(1, (os.cpu_count() or n) // n))
            env.setdefault("OMP_NUM_THREADS", per_rank)
            env.setdefault("OPENBLAS_NUM_THREADS", per_rank)
            # fresh child processes, each in its own session (never a re-exec of a process that has touched a GPU)
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                          start_new_session=True))
        # a rank that dies (its own watchdog, an RCCL error) must take the others down instead of leaving them in a collective
        alive = list(procs)
        while alive:
            time.sleep(0.2)
            for p in list(alive):
                code = p.poll()
                if code is None:
                    continue
                alive.remove(p)
                if code != 0 and rc == 0:           # (the ranks stopped below exit with -SIGTERM: the first failure is the job's)
                    rc = abs(code) or 1
                    print(f"bench.py: rank process {procs.index(p)} exited with {code}; stopping the other ranks", file=sys.stderr)
                    stop_ranks()
    except BaseException:
        rc = rc or 1
        raise
    finally:
        stop_ranks()               # (no-op when every rank has exited)
        for sig, h in old_handlers.items():
            signal.signal(sig, h)
    return rc


def recipe_means(K, D, spread):
    """First draw of the synthetic recipe: mu = spread * standard_normal((K, D))."""
    return spread * np.random.default_rng(SEED).standard_normal((K, D))


def recipe_rows_host(K, D, n, dtype, spread):
    """First n rows of the recipe, drawn on the host exactly like oracle.synth_gmm (chunk boundary 2**20 > n)."""
    rng = np.random.default_rng(SEED)
    mu = spread * rng.standard_normal((K, D))
    z = rng.integers(0, K, n)
    return (mu[z] + rng.standard_normal((n, D))).astype(dtype)


def device_rows(K, D, n, dtype, dev, seed, spread, head=None):
    """Same mixture drawn with the device generator, in chunks; the first len(head) rows are `head`."""
    import torch
    mu = torch.from_numpy(recipe_means(K, D, spread)).to(dev)
    x = torch.empty((n, D), dtype=dtype, device=dev)
    gen = torch.Generator(device=dev).manual_seed(seed)
    step = 1 << 20
    for lo in range(0, n, step):
        hi = min(n, lo + step)
        z = torch.randint(0, K, (hi - lo,), device=dev, generator=gen)
        x[lo:hi] = (mu[z] + torch.randn(hi - lo, D, dtype=torch.float64, device=dev, generator=gen)).to(dtype)
    if head is not None:
        x[: head.shape[0]] = torch.from_numpy(head).to(dev)
    return x


class env_vars:
    """Library switches are read when a workspace is created (GMMVB_ESTEP_CARRY_OFF: at every E-step)."""
    KEYS = ("GMMVB_ESTEP_PRUNE", "GMMVB_MSTEP_SPARSE")

    def __init__(self, **kv):
        self.kv = kv

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in set(self.KEYS) | set(self.kv)}
        for k, v in self.kv.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v

    def __exit__(self, *exc):
        for k, v in self.old.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v


def cpu_baseline_and_parity(K, D, x_ref, dev, iters=10, sweep_threads=False):
    """Oracle (test infrastructure) on the host cores vs the GPU driver on the same rows: with the default policy and
    with the sparse path forced (the kernels of the timed steps: int8 bound pass, carried bounds, gathers, lists)."""
    from bayesml_amd import gaussianmixture as gm
    from oracle import gmm_vb_oracle as orc
    import copy
    x64 = x_ref.astype(np.float64)
    p = orc.Prior.default(K, D)
    q = orc.Posterior.from_prior(p)
    orc.init_subsampling(x64, q, np.random.default_rng(0))
    st = orc.data_pass(x64, q)

    def iterate(q_, st_, n):
        t0 = time.perf_counter()
        for _ in range(n):
            orc.update_q_mu_lambda(p, q_, st_)
            orc.update_q_pi(p, q_, st_)
            st_ = orc.data_pass(x64, q_, st_.s)
            orc.lower_bound(p, q_, st_)
        return time.perf_counter() - t0, st_

    # The host's best: one iteration per BLAS thread count on a copy of the state, then the timed run at the fastest
    # (128 OpenBLAS threads on [20000, 128] operands is oversubscription - round 5 reported 6.2e3 samples/s that way against
    # the 1.1e4 BASELINE.md section 2 measured on 8 cores)
    sweep, limits = {}, None
    try:
        from threadpoolctl import threadpool_info, threadpool_limits
        default_threads = max([i.get("num_threads", 1) for i in threadpool_info()] or [1])
        if sweep_threads:
            for t in sorted({t for t in (8, 16, 32, 64, 128) if t <= default_threads} | {default_threads}):
                with threadpool_limits(limits=t):
                    s_t, _ = iterate(copy.deepcopy(q), copy.deepcopy(st), 1)
                sweep[t] = x_ref.shape[0] / s_t
            limits = max(sweep, key=sweep.get)
        threads = limits or default_threads
    except Exception:      # noqa: BLE001
        threads = os.cpu_count() or 1
    if limits:
        with threadpool_limits(limits=limits):
            cpu_s, st = iterate(q, st, iters)
    else:
        cpu_s, st = iterate(q, st, iters)
    base = dict(value=x_ref.shape[0] * iters / cpu_s, unit="samples/s", cores=int(threads), kind="port",
                sample=f"{iters} VB iterations (K-side + E + M + lower bound, fp64 NumPy/OpenBLAS) over the first "
                       f"{x_ref.shape[0]} rows, {threads} BLAS threads of {os.cpu_count()} cores"
                       + (f" = best of {sorted(sweep)}" if sweep else ""),
                threads_swept={str(t): round(v, 1) for t, v in sweep.items()}, seconds=cpu_s)

    def rel(a, b):
        return float(np.max(np.abs(a - b)) / np.max(np.abs(b)))

    def gpu_run(label, **switches):
        with env_vars(**switches):
            m = gm.LearnModel(K, D, seed=0, device=dev, verbose=False)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                m.update_posterior(x_ref, max_itr=iters, num_init=1, tolerance=0.0)
        errs = dict(hn_alpha_vec=rel(m.hn_alpha_vec, q.alpha), hn_m_vecs=rel(m.hn_m_vecs, q.m),
                    hn_kappas=rel(m.hn_kappas, q.kappa), hn_nus=rel(m.hn_nus, q.nu),
                    hn_w_mats=rel(m.hn_w_mats, q.w), hn_w_mats_inv=rel(m.hn_w_mats_inv, q.w_inv))
        counts = m._engine.pass_counts()
        m._engine.close()
        return dict(max_rel_err=max(errs.values()), tolerance=1e-5, passed=max(errs.values()) < 1e-5, per_array=errs,
                    rows=int(x_ref.shape[0]), iterations=iters, kernel_launches=counts,
                    path=label + ": " + ", ".join(f"{k} x{v}" for k, v in counts.items() if v))

    return base, gpu_run("default policy"), gpu_run("GMMVB_ESTEP_PRUNE=force", GMMVB_ESTEP_PRUNE="force")


class Workload:
    """One model + sample matrix driven through the same internals update_posterior uses."""

    def __init__(self, K, D, x, dev, comm, spans="full"):
        from bayesml_amd import _kside
        from bayesml_amd import gaussianmixture as gm
        self.K, self.D, self.n = K, D, x.shape[0]
        self.m = gm.LearnModel(K, D, seed=0, device=dev, comm=comm, verbose=False)
        self.eng, self.xd = self.m._open(x)
        # HIP events in the library (include/gmmvb.h: gmmvb_profile_enable): every record costs the stream ~10 us, so the
        # timed steps of the headline keep only the spans of the groups that can dominate ("dominant", level 2)
        self.eng.profile(2 if spans == "dominant" else 1)
        prior = self.m._prior_tensors(dev)
        self.ks = self.m._stepper(self.eng, prior, self.xd)
        q = self.m._init_subsampling(self.eng, self.xd, _kside.post_from_prior(prior), self.m._comm.global_rows)
        self.ks.load(q)
        self.m._give_params(self.eng, q)
        self.m._data_pass(self.eng, self.xd, self.ks)
        self.ks.step()
        self.terms, self.gmean = self.ks.read()

    @property
    def q(self):
        return self.ks.q

    def step(self):
        # update_posterior's loop body: parameter hand-over (+ drift hint) -> data pass (+ all-reduce) -> K-side step
        # (one hipGraph replay) -> ONE device-to-host copy (lower bound terms, mean drift)
        m, ks = self.m, self.ks
        m._give_params(self.eng, ks.q_next, ks.hint(self.gmean))
        ks.advance()
        m._data_pass(self.eng, self.xd, ks)
        ks.step()
        self.terms, self.gmean = ks.read()
        return self.terms["vl"]

    def snapshot(self):
        a, e = self.eng.sparsity()
        em, mm = self.eng.last_kernel_ms()
        return dict(estep_ms=round(em, 2) if em >= 0 else None, mstep_ms=round(mm, 2) if mm >= 0 else None,
                    kernels=[p.strip().split(" ")[0] for p in self.eng.launch_info.split("|")],
                    active_components_per_sample=round(a / self.n, 2) if a >= 0 else None,
                    evaluated_components_per_sample=round(e / self.n, 2),
                    accumulated_components_per_sample=round(self.eng.work()["accumulated"] / self.n, 2))

    def close(self):
        self.eng.close()
        self.m._engine = None


def kernel_name(info_part):
    return info_part.strip().split("<")[0].split(" ")[0]


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus))

    hide_stdout()
    import torch
    import torch.distributed as dist
    from bayesml_amd import RowShard

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if args.gpus != 1:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
        args.gpus = world
    if args.dense:
        os.environ["GMMVB_ESTEP_PRUNE"] = "0"
        os.environ["GMMVB_MSTEP_SPARSE"] = "0"
    # BENCH_SHARE_GPU=1 (developer switch): all ranks on GPU 0 with the gloo backend - the N > 1 code path of this script
    # on a one-GPU box (the collective then goes through host memory; the numbers mean nothing)
    share_gpu = os.environ.get("BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    comm = None
    n_ranks, backend = 1, None
    use_dist = world > 1 or args.force_dist
    if use_dist:
        # a rank that the parent stops (SIGTERM: another rank died, or the job's own timeout) says where it was
        import faulthandler
        import signal
        faulthandler.register(signal.SIGTERM, all_threads=True, chain=True, file=sys.stderr)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("NCCL_DEBUG", "WARN")          # (RCCL's own account of a failed init goes to stderr)
        # per-rank watchdog: if joining the group or the first collective stalls, dump every thread's stack and exit 1
        print(f"bench.py: rank {rank}/{world} joining the process group (watchdog {args.init_timeout:.0f} s)", file=sys.stderr, flush=True)
        faulthandler.dump_traceback_later(args.init_timeout, exit=True, file=sys.stderr)
        if share_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        assert dist.get_world_size() == args.gpus
        comm = RowShard(native=args.native_allreduce, always=args.force_dist)
        # every rank contributes 1: proves the RCCL group really spans `world` processes
        one = torch.ones(1, dtype=torch.float64, device=dev)
        dist.all_reduce(one)
        n_ranks = int(one.item())
        backend = dist.get_backend()          # "nccl" = RCCL on ROCm; "gloo" only under BENCH_SHARE_GPU=1
        assert n_ranks == world
        faulthandler.cancel_dump_traceback_later()

    cfg = CONFIGS[args.config]
    K = args.classes or cfg["classes"]
    D = args.degree or cfg["degree"]
    dt = args.dtype or cfg["dtype"]
    if args.scaling == "strong":
        total = args.total_rows or cfg["total"]
        n_local = total // world + (1 if rank < total % world else 0)
    else:
        n_local = args.rows or cfg["rows"]
    spread = args.spread if args.spread is not None else (0.3 if args.overlap else 2.0)
    args.overlap = args.overlap or spread != 2.0          # (the legs that assume the headline recipe are skipped)
    tdtype = torch.float32 if dt == "f32" else torch.float64
    ndtype = np.float32 if dt == "f32" else np.float64
    esz = 4 if dt == "f32" else 8

    # ---- workload, resident in HBM before anything is timed
    x_ref = recipe_rows_host(K, D, min(args.ref_rows, n_local), ndtype, spread)
    x = device_rows(K, D, n_local, tdtype, dev, SEED + 1 + rank, spread, head=x_ref if rank == 0 else None)

    cpu_base = parity = parity_sparse = None
    if rank == 0 and world == 1 and not args.no_cpu:
        cpu_base, parity, parity_sparse = cpu_baseline_and_parity(K, D, x_ref, dev, sweep_threads=True)
    do_cpu = rank == 0 and world == 1 and not args.no_cpu

    w = Workload(K, D, x, dev, comm, spans=args.spans)
    eng = w.eng
    torch.cuda.synchronize()
    warm = [dict(w.snapshot(), what="pass after the subsampling initialisation")]
    for _ in range(args.warmup):
        w.step()
        warm.append(dict(w.snapshot(), what="warm-up iteration"))

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    ker, launches, spars, spans, works = [], [], [], [], []
    counts0 = eng.pass_counts()
    # A full collection of Python's garbage collector inside the timed steps is a 40-70 ms host stall once
    # torch.distributed has been imported and initialised (measured: always the same step, profiles/r2_experiments.md):
    # collect now and move what is alive to the permanent generation.
    gc.collect()
    gc.freeze()
    fence()
    t0 = time.perf_counter()
    walls = []
    for _ in range(args.steps):
        ts = time.perf_counter()
        vl = w.step()
        walls.append((time.perf_counter() - ts) * 1e3)
        ker.append(eng.last_kernel_ms())        # events already complete: step() ended with a host sync
        spans.append(eng.kernel_spans())
        launches.append(eng.launch_info)
        spars.append(eng.sparsity())
        works.append(eng.work())
    fence()
    elapsed = time.perf_counter() - t0
    counts1 = eng.pass_counts()
    policy_same = None
    if use_dist:
        # every rank's sequence of E-step kinds over the timed steps: one policy for all ranks (gmmvb_set_shard) means
        # they are identical
        seqs = [None] * world
        dist.all_gather_object(seqs, [kernel_name(l) for l in launches])
        policy_same = all(q == seqs[0] for q in seqs)
        # every rank's own clock and step times, so that a straggler shows (value uses the max over ranks)
        ranks_info = [None] * world
        dist.all_gather_object(ranks_info, dict(rank=rank, elapsed_s=elapsed, ms_per_step=elapsed / args.steps * 1e3,
                                                wall_ms=[round(v, 2) for v in walls], rows=int(n_local)))
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        tot = torch.tensor([float(n_local)], dtype=torch.float64, device=dev)
        dist.all_reduce(tot)
        n_total = int(tot.item())
    else:
        n_total = n_local
        ranks_info = [dict(rank=0, elapsed_s=elapsed, ms_per_step=elapsed / args.steps * 1e3,
                           wall_ms=[round(v, 2) for v in walls], rows=int(n_local))]

    # config 4 strong-scaled over all ranks (north_star: ">= 6x at 8 GPUs on N = 1e8"): every --gpus N record carries the
    # same job, so the curve falls out of the driver's N = 1, 2, 4, 8 records without extra flags
    leg_names = set(args.legs.split(","))
    if "all" in leg_names:
        leg_names = {"dense", "hard", "spread", "full", "hmm", "small", "offpath", "c2", "c4", "c4strong"}
    strong_leg = None
    if world > 1 and "c4strong" in leg_names and not args.no_legs and not args.dense and args.config == "c3":
        tot4 = args.strong_total_rows
        n4 = tot4 // world + (1 if rank < tot4 % world else 0)
        strong_leg = config_subline("c4", dev, 5, 20, comm=comm, use_dist=True, rows=n4, seed=100 + rank, world=world)

    # RCCL writes its version banner through C stdio, which buffers when stdout is a pipe or a file: left alone it comes out
    # at process exit - AFTER the JSON line, which must be the last line of stdout.  Every rank empties its C buffers now,
    # and rank 0 prints only after all have.
    flush_c_stdio()
    if use_dist:
        dist.barrier()
    if rank == 0:
        steps = args.steps
        step_ms = elapsed / steps * 1e3
        roof, groups, fl_pair, timed_counts = build_roofline(eng, K, D, dt, n_local, steps, args.warmup, step_ms, ker, spans,
                                                             launches, spars, works, counts0, counts1, args.spans)
        last_launch = eng.launch_info
        dense_leg = hard = spread_leg = full_leg = hmm_leg = small_leg = offpath = c2_leg = c4_leg = None
        if not args.dense and world == 1 and not args.no_legs:
            legs = leg_names
            if "dense" in legs:
                dense_leg = dense_leg_run(w, K, D, n_local, fl_pair)
            w.close()
            del w
            torch.cuda.empty_cache()
            if "full" in legs and not args.overlap:
                full_leg = full_fit_leg(K, D, x, dev)
            del x
            torch.cuda.empty_cache()
            if "hard" in legs and not args.overlap:
                hard = hard_workload_leg(K, D, n_local, tdtype, ndtype, dev, parity=do_cpu)
            if "spread" in legs and not args.overlap:
                spread_leg = spread_sweep_leg(K, D, min(n_local, 2_000_000), tdtype, ndtype, dev, parity=do_cpu)
            if "offpath" in legs and not args.overlap:
                offpath = offpath_leg(K, D, min(n_local, 2_000_000), tdtype, ndtype, dev, parity=do_cpu)
            if "hmm" in legs and args.config == "c3":
                hmm_leg = hmm_c5_leg(dev, cpu=do_cpu)
            if "small" in legs:
                small_leg = small_c1_leg(dev)
            if args.config == "c3" and not args.overlap:
                if "c2" in legs:
                    c2_leg = config_subline("c2", dev, 5, 20)
                if "c4" in legs:
                    c4_leg = config_subline("c4", dev, 5, 20, what="GMM-VB K=256 D=64, one shard of config 4 (1.25e7 of N=1e8 rows), "
                                                                      "x stored f32, VB iterations 6-25 of one restart")
                if "c4strong" in legs:
                    strong_leg = config_subline("c4", dev, 5, 20, rows=args.strong_total_rows, seed=100)
        out = {
            "metric": "GMM-VB E+M samples/sec at K=64,D=128,N=1e7; 1/2/4/8-GPU scaling",
            "value": n_total * steps / elapsed, "unit": "samples/s", "n_gpus": world, "ranks": n_ranks, "backend": backend,
            "estep_kinds_identical_across_ranks": policy_same,
            "allreduce": (("gmmvb_allreduce_stats (C ABI, RCCL)" if args.native_allreduce else
                           ("torch.distributed gloo (BENCH_SHARE_GPU=1: all ranks on one GPU)" if share_gpu else "torch.distributed nccl (RCCL)"))
                          if use_dist else None),
            "steps": steps, "warmup": args.warmup, "ms_per_step": step_ms, "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"GMM-VB K={K} D={D} N={n_local} rows/GPU x {world} GPU, x stored {dt}, "
                                   f"one VB iteration per step ({args.config} of BASELINE.json configs"
                                   f"{', overlapping clusters' if args.overlap else ''})",
                       "classes": K, "degree": D, "rows_per_gpu": n_local, "rows_total": n_total, "x_storage": dt,
                       "cluster_spread": spread, "parallelism": f"rows{world}",
                       "workspace_GB": round(eng.workspace_bytes / 1e9, 2),
                       "row_tiles": getattr(eng, "n_tiles", 1)},
            "roofline": roof, "dense": dense_leg, "hard_workload": hard, "spread_sweep": spread_leg,
            "full_fit": full_leg, "hmm_c5": hmm_leg, "small_c1": small_leg, "offpath": offpath, "per_rank": ranks_info,
            "c2": c2_leg, "c4_shard": c4_leg, "c4_strong": strong_leg,
            "cpu_baseline": cpu_base, "parity": parity, "parity_sparse_path": parity_sparse, "final_vl": vl,
            "launch": last_launch, "warmup_steps": warm,
            "per_step": {"wall_ms": [round(v, 2) for v in walls],
                         "estep_ms": [round(k[0], 2) if k[0] >= 0 else None for k in ker],
                         "mstep_ms": [round(k[1], 2) if k[1] >= 0 else None for k in ker],
                         "estep_kernel": [kernel_name(l) for l in launches],
                         "active_components_per_sample": [round(a / n_local, 2) if a >= 0 else None for a, _ in spars],
                         "evaluated_components_per_sample": [round(e / n_local, 2) for _, e in spars],
                         "accumulated_components_per_sample": [round(wk["accumulated"] / n_local, 2) for wk in works],
                         "settled_rows_per_sample": [round(wk["settled_rows"] / n_local, 3) for wk in works],
                         "proof_pairs_per_sample": [round(wk.get("proof_pairs", 0.0) / n_local, 3) for wk in works],
                         "sweep_share_of_bound_array": [round(wk.get("sweep_share", -1.0), 3) for wk in works]},
        }
        out["window"] = (f"VB iterations {args.warmup + 1}-{args.warmup + steps} of one restart (subsampling initialisation + its "
                         f"data pass + {args.warmup} warm-up iterations are outside the timed region)")
        emit(out, args)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


class TimedComm:
    """Proxy of a RowShard that brackets the per-iteration collective with its own HIP events on the launch stream (the
    collective itself may run on RCCL's stream: the second event sits behind the wait for it)."""

    def __init__(self, inner):
        self._inner, self.events = inner, []

    def __getattr__(self, name):
        return getattr(self._inner, name)

    def all_reduce_(self, t):
        import torch
        if not t.is_cuda or not (self._inner.world > 1 or getattr(self._inner, "always", False)):
            return self._inner.all_reduce_(t)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        out = self._inner.all_reduce_(t)
        b.record()
        self.events.append((a, b))
        return out

    def take_ms(self):
        """Event times of the collectives since the last call (the caller has synchronised)."""
        ms = [a.elapsed_time(b) for a, b in self.events]
        self.events = []
        return ms


def timed_window(K, D, dt, x, dev, comm, warmup, steps, spans="dominant", use_dist=False):
    """Workload on x -> `warmup` iterations -> `steps` timed ones (barrier + synchronize on both sides, MAX over ranks).
    Returns a dict with the window's step time, its roofline block and the per-rank clocks."""
    import torch
    import torch.distributed as dist
    tc = TimedComm(comm) if comm is not None else None
    w = Workload(K, D, x, dev, tc, spans=spans)
    eng, n_local = w.eng, x.shape[0]
    for _ in range(warmup):
        w.step()

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    if tc is not None:
        torch.cuda.synchronize()
        tc.take_ms()
    ker, launches, spars, spans_rec, works, walls = [], [], [], [], [], []
    counts0 = eng.pass_counts()
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        ts = time.perf_counter()
        w.step()
        walls.append((time.perf_counter() - ts) * 1e3)
        ker.append(eng.last_kernel_ms())
        spans_rec.append(eng.kernel_spans())
        launches.append(eng.launch_info)
        spars.append(eng.sparsity())
        works.append(eng.work())
    fence()
    elapsed = own = time.perf_counter() - t0
    counts1 = eng.pass_counts()
    ar_ms = tc.take_ms() if tc is not None else []
    world, rank = (dist.get_world_size(), dist.get_rank()) if use_dist else (1, 0)
    policy_same, per_rank, n_total = None, [own / steps * 1e3], n_local
    if use_dist:
        seqs = [None] * world
        dist.all_gather_object(seqs, dict(kinds=[kernel_name(l) for l in launches], ms=own / steps * 1e3, rows=int(n_local),
                                          allreduce_ms=float(np.mean(ar_ms)) if ar_ms else None))
        policy_same = all(q["kinds"] == seqs[0]["kinds"] for q in seqs)
        per_rank = [q["ms"] for q in seqs]
        elapsed = max(per_rank) * steps / 1e3
        n_total = sum(q["rows"] for q in seqs)
        ar_all = [q["allreduce_ms"] for q in seqs if q["allreduce_ms"] is not None]
    else:
        ar_all = [float(np.mean(ar_ms))] if ar_ms else []
    step_ms = elapsed / steps * 1e3
    roof, groups, _fl, timed_counts = build_roofline(eng, K, D, dt, n_local, steps, warmup, step_ms, ker, spans_rec, launches,
                                                     spars, works, counts0, counts1, spans)
    out = dict(step_ms=step_ms, rows_total=n_total, rows_local=n_local, per_rank_ms_per_step=per_rank, roofline=roof,
               estep_kinds_identical_across_ranks=policy_same, allreduce_ms=(max(ar_all) if ar_all else None),
               allreduce_ms_per_rank=ar_all, wall_ms=[round(v, 2) for v in walls], timed_counts=timed_counts,
               row_tiles=getattr(eng, "n_tiles", 1), launch=eng.launch_info)
    w.close()
    return out


def subline(name, what, win, steps, warmup, world=1):
    """A compact sub-line of the JSON record for another configuration's window (like `hmm_c5`)."""
    r = win["roofline"]
    out = {"workload": what, "ms_per_step": round(win["step_ms"], 4), "steps": steps, "warmup": warmup,
           "samples_per_s": win["rows_total"] / (win["step_ms"] * 1e-3), "kernel": r["kernel"], "bound": r["bound"],
           "frac_executed": r.get("frac_executed"), "frac_on_F": r.get("frac_on_F"), "step_frac_on_F": r.get("step_frac_on_F"),
           "step_hbm_frac": r["step_hbm_frac"], "pruned": r.get("pruned"), "row_tiles": win["row_tiles"],
           "kernel_groups_ms": {g: round(v["ms"], 3) for g, v in r["kernel_groups"].items() if v["ms"] > 0.005}}
    if world > 1 or win["allreduce_ms"] is not None:
        out.update(n_gpus=world, per_rank_ms_per_step=[round(v, 3) for v in win["per_rank_ms_per_step"]],
                   allreduce_ms=win["allreduce_ms"], estep_kinds_identical_across_ranks=win["estep_kinds_identical_across_ranks"])
    return out


def config_subline(name, dev, warmup, steps, comm=None, use_dist=False, rows=None, seed=0, world=1, what=None):
    """Window of another BASELINE.json configuration with its own synthetic rows (same recipe), as a sub-line."""
    import torch
    cfg = CONFIGS[name]
    K, D, dt = cfg["classes"], cfg["degree"], cfg["dtype"]
    n = int(rows or cfg["rows"])
    x = device_rows(K, D, n, torch.float32 if dt == "f32" else torch.float64, dev, SEED + 1 + seed, 2.0)
    win = timed_window(K, D, dt, x, dev, comm, warmup, steps, use_dist=use_dist)
    del x
    torch.cuda.empty_cache()
    tot = win["rows_total"]
    return subline(name, what or f"GMM-VB K={K} D={D} N={tot}{'' if world == 1 else f' over {world} GPUs'}, x stored {dt}, "
                                 f"VB iterations {warmup + 1}-{warmup + steps} of one restart", win, steps, warmup, world)


def build_roofline(eng, K, D, dt, n_local, steps, warmup, step_ms, ker, spans, launches, spars, works, counts0, counts1, spans_mode):
    """The `roofline` block of a timed window from what the window's steps recorded: per kernel group the
    HIP-event time, algorithmic bytes and executed f64 MFMA flops; the line's fields are the dominant group's.  Two fractions
    with one meaning each: `frac_executed` = flops the kernel really issued on the f64 matrix pipe / its time / peak (a
    utilisation, <= 1), `frac_on_F` = the flops the REFERENCE's dense formulation spends on that phase (E
    K(2D^2+3D)+6K, M K(2D^2+2D)+2KD per row) / the same time / the same peak - far above 1 once rows and pairs are pruned,
    where it measures the algorithmic saving, not the hardware."""
    esz = 4 if dt == "f32" else 8
    # (phase events only with --spans full)
    e_ms = float(np.mean([k[0] for k in ker])) if all(k[0] >= 0 for k in ker) else None
    m_ms = float(np.mean([k[1] for k in ker])) if all(k[1] >= 0 for k in ker) else None
    names = [kernel_name(p) for p in eng.launch_info.split("|")]      # kernels of the last step
    tiles = (D + 15) // 16
    fl_pair = 512 * tiles * (tiles + 1) // 2        # executed f64 MFMA flops per exactly evaluated (sample, component)
    ev = float(np.mean([e for _, e in spars]))      # pairs evaluated exactly per E-step
    ac = float(np.mean([a for a, _ in spars]))      # active pairs (r >= 2^-80)
    # pairs the list M-step accumulates: active pairs minus the rows whose single component has r = 1.0 exactly and
    # did not change (their addends sit in the workspace's cache), plus the rows entering / leaving that cache
    acc = float(np.mean([wk["accumulated"] if wk["accumulated"] >= 0 else n_local * K for wk in works]))
    settled = float(np.mean([wk["settled_rows"] for wk in works]))
    exits = float(np.mean([wk["early_exits"] for wk in works]))      # candidates that stopped after T/2 output blocks
    timed_counts = {k: counts1[k] - counts0[k] for k in counts1}
    m_sparse = timed_counts["mstep_list"] > 0
    # ---- per kernel group: mean HIP-event ms per step, algorithmic bytes per step (rows the group must read x D x s)
    groups = {}
    for g in sorted({g for s in spans for g in s}):
        groups[g] = dict(ms=float(np.mean([s.get(g, (0.0, 0))[0] for s in spans])),
                         launch_groups_per_step=float(np.mean([s.get(g, (0.0, 0))[1] for s in spans])))
    row_bytes = D * esz
    sparse_e = timed_counts["estep_bound"] + timed_counts["estep_carried"] + timed_counts["estep_sweep"] > 0
    alg = {"estep_main": n_local * row_bytes * (timed_counts["estep_dense"] + timed_counts["estep_bound"]
                                                + timed_counts["estep_fell_back_dense"]) / steps,
           "estep_gather": ev * row_bytes if sparse_e else 0.0,
           "mstep_main": (acc if m_sparse else n_local) * row_bytes}
    for g, b in alg.items():
        if g in groups and groups[g]["ms"] > 0:
            groups[g]["algorithmic_bytes"] = b
            groups[g]["algorithmic_GBps"] = b / (groups[g]["ms"] * 1e-3) / 1e9
    if groups.get("estep_gather", {}).get("ms", 0) > 0:
        half = (tiles // 2) * (tiles // 2 + 1) // 2            # tile pairs of the first T/2 output blocks
        done = ev - exits * (1.0 - half / (tiles * (tiles + 1) // 2)) if tiles >= 2 else ev
        groups["estep_gather"]["executed_f64_tflops"] = fl_pair * done / groups["estep_gather"]["ms"] / 1e9
    if "mstep_main" in groups:
        groups["mstep_main"]["executed_f64_tflops"] = fl_pair * (acc if m_sparse else n_local * K) / groups["mstep_main"]["ms"] / 1e9
    if groups.get("estep_main", {}).get("ms", 0) > 0 and timed_counts["estep_dense"] + timed_counts["estep_fell_back_dense"] > 0 \
            and timed_counts["estep_bound"] == 0:
        groups["estep_main"]["executed_f64_tflops"] = fl_pair * n_local * K * (timed_counts["estep_dense"]
                                                                             + timed_counts["estep_fell_back_dense"]) / steps / groups["estep_main"]["ms"] / 1e9
    # both yardsticks per kernel group: HBM (algorithmic bytes) and the f64 matrix pipe (executed flops); a group is bound
    # by whichever fraction is larger
    for gname, gv in groups.items():
        if "algorithmic_GBps" in gv:
            gv["hbm_frac"] = gv["algorithmic_GBps"] / PEAK_HBM_GBPS
        if "executed_f64_tflops" in gv:
            gv["f64_mfma_frac"] = gv["executed_f64_tflops"] / PEAK_F64_MFMA_TFLOPS
        if "hbm_frac" in gv or "f64_mfma_frac" in gv:
            gv["bound"] = "mfma" if gv.get("f64_mfma_frac", 0.0) > gv.get("hbm_frac", 0.0) else "hbm"
    cand = [g for g in ("estep_main", "estep_gather", "mstep_main") if g in groups and "algorithmic_GBps" in groups[g]]
    if cand:
        dom = max(cand, key=lambda g: groups[g]["ms"])
        dom_kernel = {"estep_main": names[0], "estep_gather": "estep_gather_f64", "mstep_main": names[-1]}[dom]
        ach = groups[dom]["algorithmic_GBps"]
    else:       # a row-tiled run (the workspace does not fit: _engine.TiledDataPass) keeps no per-group events
        dom, dom_kernel = None, "tiled data pass (whole step)"
        ach = n_local * row_bytes / (step_ms * 1e-3) / 1e9
    # HBM-side bytes per launch of that kernel: PMC counters cannot be read from inside this process, so they come
    # from the committed rocprofv3 --pmc passes of this same command (tools/summarize_pmc.py), and only when that
    # file was made for the kernel that actually ran here
    traffic = traffic_src = None
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            pm = json.load(f).get({"estep_i8_bound": "estep_i8", "estep_gather_f64": "estep_gather_dev_f64",
                                   "mstep_list_f64": "mstep_list_x32_f64" if row_bytes == 4 * D else "mstep_list_f64"}
                                  .get(dom_kernel, dom_kernel))
        ran = (dom_kernel == "estep_gather_f64" and timed_counts["estep_gather"] > 0) or any(dom_kernel in l for l in launches)
        if pm and pm.get("config") == f"K{K} D{D} N{n_local} {dt}" and ran:
            # per step like `achieved`: the counters' average per launch x this run's launches of the group per step
            per_step = groups[dom]["launch_groups_per_step"] if dom else 1.0
            n_last = int(round(per_step * steps))
            fl, wl = pm.get("fetch_bytes_raw_launches"), pm.get("write_bytes_launches")
            if pm.get("window") == f"w{warmup}s{steps}" and fl and wl and 0 < n_last <= min(len(fl), len(wl)):
                # the passes were made with this very command: the kernel's last n launches are the timed steps'
                traffic = (2.0 * sum(fl[-n_last:]) + sum(wl[-n_last:])) / steps
                traffic_src = (f"profiles/pmc_traffic.json ({pm['kernel']}, its last {n_last} launches = the {steps} timed "
                               "steps of this command, per step): " + pm["note"])
            else:
                traffic = (pm["fetch_bytes"] + pm["write_bytes"]) * per_step
                traffic_src = (f"profiles/pmc_traffic.json ({pm['kernel']}, average per launch x {per_step:.2f} launches "
                               "per step): " + pm["note"])
    except (OSError, ValueError, KeyError):
        pass
    step_bytes = n_local * row_bytes
    roof = {"bound": "hbm", "kernel": dom_kernel, "achieved": ach, "peak": PEAK_HBM_GBPS, "unit": "GB/s",
            "frac": ach / PEAK_HBM_GBPS, "traffic": traffic, "traffic_source": traffic_src,
            "step_algorithmic_bytes": step_bytes,
            "step_hbm_GBps": step_bytes / (step_ms * 1e-3) / 1e9,
            "step_hbm_frac": step_bytes / (step_ms * 1e-3) / 1e9 / PEAK_HBM_GBPS,
            "hbm_roofline_samples_per_s": PEAK_HBM_GBPS * 1e9 / row_bytes,
            "f64_mfma_ceiling_samples_per_s": (PEAK_F64_MFMA_TFLOPS * 1e12 / (fl_pair * (ev + acc) / n_local)
                                               if (ev + acc) > 0 else None),
            "pairs_per_sample": {"active": ac / n_local, "evaluated_exactly": ev / n_local,
                                 "accumulated_by_mstep": acc / n_local, "settled_rows": settled / n_local,
                                 "early_exits": exits / n_local,
                                 "proof_round_int8": float(np.mean([wk.get("proof_pairs", 0.0) for wk in works])) / n_local},
            "spans": spans_mode,
            "kernel_groups": groups,
            "events_ms_per_step": sum(g["ms"] for g in groups.values()),
            "outside_events_ms_per_step": step_ms - sum(g["ms"] for g in groups.values()),
            "phase_ms": {"estep": e_ms, "mstep": m_ms},
            "timed_kernel_launches": timed_counts,
            "note": "every kernel group carries hbm_frac (algorithmic bytes: rows it must read x D x s, SURVEY 8d, / its "
                    "HIP-event time / 8 TB/s) and f64_mfma_frac (executed f64 MFMA flops / 78.6 TFLOP/s); `bound`, `achieved`, "
                    "`peak`, `frac` are the dominant group's LARGER fraction; frac <= 1 by construction.  step_hbm_frac = value / "
                    "hbm_roofline_samples_per_s.  f64_mfma_ceiling = the rate at which the f64 matrix pipe alone could "
                    "evaluate the (sample, component) pairs this step evaluates exactly (E) and accumulates (M).  "
                    "executed_f64_tflops of estep_gather charges the pairs that take the gather's early way out "
                    "(DESIGN.md 4b; pairs_per_sample.early_exits) with the tile pairs they really do"}
    if dom:
        roof["hbm_frac"] = groups[dom].get("hbm_frac")
        roof["f64_mfma_frac"] = groups[dom].get("f64_mfma_frac")
    if dom and groups[dom].get("bound") == "mfma":
        # the dominant kernel's larger fraction is of the f64 matrix pipe: executed flops against its peak
        roof.update(bound="mfma", unit="TFLOP/s", peak=PEAK_F64_MFMA_TFLOPS, achieved=groups[dom]["executed_f64_tflops"],
                    frac=groups[dom]["f64_mfma_frac"], hbm_achieved_GBps=ach)
    assert roof["frac"] <= 1.0 + 1e-9, roof

    # ---- SURVEY 8d's convention beside the executed one
    f_e, f_m = K * (2 * D * D + 3 * D) + 6 * K, K * (2 * D * D + 2 * D) + 2 * K * D
    pruned = bool(sparse_e or m_sparse)
    for gname, phase_f in (("estep_main", f_e), ("estep_gather", f_e), ("mstep_main", f_m)):
        gv = groups.get(gname)
        if gv and gv["ms"] > 0:
            gv["frac_executed"] = gv.get("f64_mfma_frac")
            gv["frac_on_F"] = phase_f * n_local / (gv["ms"] * 1e-3) / 1e12 / PEAK_F64_MFMA_TFLOPS
    roof["frac_executed"] = groups[dom].get("f64_mfma_frac") if dom else None
    roof["frac_on_F"] = groups[dom].get("frac_on_F") if dom else None
    roof["step_frac_on_F"] = (f_e + f_m) * n_local / (step_ms * 1e-3) / 1e12 / PEAK_F64_MFMA_TFLOPS
    roof["F_flop_per_sample"] = f_e + f_m
    roof["pruned"] = pruned
    roof["frac_note"] = ("frac = frac_executed: f64 MFMA flops the kernel issued / its HIP-event time / 78.6 TFLOP/s.  frac_on_F: "
                         "SURVEY 8d's dense F of that phase x rows / the same time / the same peak"
                         + ("; pruned: rows and pairs proven irrelevant are skipped, so the F-basis fraction is not a utilisation"
                            if pruned else ""))
    return roof, groups, fl_pair, timed_counts


_STDOUT_FD = None


def hide_stdout():
    """The JSON line must be the only thing on stdout, and libraries print there from C (RCCL's version banner, gloo's
    "connected to peer ranks" lines): file descriptor 1 points at stderr until emit() writes the line."""
    global _STDOUT_FD
    if _STDOUT_FD is None:
        sys.stdout.flush()
        _STDOUT_FD = os.dup(1)
        os.dup2(2, 1)


def show_stdout(on):
    if _STDOUT_FD is not None:
        flush_c_stdio()
        os.dup2(_STDOUT_FD if on else 2, 1)


def flush_c_stdio():
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:       # noqa: BLE001
        pass
    sys.stdout.flush()


def _pick(d, keys):
    return {k: d[k] for k in keys if d is not None and k in d} if d is not None else None


def compact_line(out):
    """The ONE line the driver parses (< 6 KB): the contract's keys + roofline + cpu_baseline + both parity legs + the
    scalars that say what was measured.  Everything else (per-step lists, kernel groups, legs in full) is in the
    detail record (--detail)."""
    r = out["roofline"]
    roof = _pick(r, ("bound", "kernel", "achieved", "peak", "unit", "frac", "frac_executed", "frac_on_F", "step_frac_on_F", "pruned",
                     "traffic", "hbm_frac", "spans", "step_hbm_frac", "f64_mfma_ceiling_samples_per_s", "frac_note"))
    roof["pairs_per_sample"] = {k: round(v, 4) for k, v in r["pairs_per_sample"].items()}
    roof["kernel_groups_ms"] = {g: round(v["ms"], 4) for g, v in r["kernel_groups"].items()}
    line = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                "scaling", "vs_baseline", "dtype", "data")}
    line["config"] = _pick(out["config"], ("workload", "classes", "degree", "rows_per_gpu", "rows_total", "x_storage",
                                           "cluster_spread", "parallelism", "row_tiles"))
    line["window"] = out["window"]
    line["roofline"] = roof
    line["cpu_baseline"] = _pick(out["cpu_baseline"], ("value", "unit", "cores", "kind", "sample", "threads_swept"))
    for k in ("parity", "parity_sparse_path"):
        line[k] = _pick(out[k], ("max_rel_err", "tolerance", "passed", "rows", "iterations", "path"))
    w = out["per_step"]["wall_ms"]
    line["first_step_ms"], line["last_step_ms"] = w[0], w[-1]
    if out.get("full_fit"):
        line["full_fit_seconds"] = out["full_fit"]["seconds"]
        line["full_fit_samples_per_s_per_data_pass"] = out["full_fit"]["samples_per_s_per_data_pass"]
        line["full_fit_what"] = out["full_fit"]["what"]
    if out.get("dense"):
        line["dense_floor_samples_per_s"] = out["dense"]["kernel_only_samples_per_s"]
        line["dense_floor_f64_mfma_frac"] = {"estep": out["dense"]["estep"]["frac_of_f64_mfma_peak"],
                                             "mstep": out["dense"]["mstep"]["frac_of_f64_mfma_peak"]}
        line["sparse_vs_dense_statistics_max_rel_diff"] = out["dense"]["sparse_vs_dense_statistics"]["max_rel_diff"]
    h = out.get("hmm_c5")
    if h:
        line["hmm_c5"] = dict(_pick(h, ("value", "unit", "steps", "warmup", "ms_per_step")),
                              workload=h["config"]["workload"], window=h.get("window"),
                              roofline=_pick(h["roofline"], ("bound", "achieved", "peak", "unit", "frac", "frac_executed", "frac_on_F",
                                                             "hbm_frac", "traffic")),
                              parity_max_rel_err=(h["parity"] or {}).get("max_rel_err"),
                              cpu_baseline_value=(h["cpu_baseline"] or {}).get("value"),
                              boundary_pass=h.get("boundary_pass"),
                              viterbi_ms=(h.get("viterbi") or {}).get("ms"))
    for k in ("hard_workload", "spread_sweep", "offpath"):
        if out.get(k):
            line[k] = out[k].get("summary")
    for k in ("c2", "c4_shard", "c4_strong"):
        if out.get(k):
            line[k] = {f: (round(v, 4) if isinstance(v, float) and f != "samples_per_s" else v) for f, v in out[k].items()
                       if f not in ("kernel_groups_ms", "steps", "warmup", "step_hbm_frac")}
    if out["n_gpus"] > 1 or out.get("allreduce"):
        line["ranks"], line["backend"] = out["ranks"], out["backend"]
        line["allreduce"] = out["allreduce"]
        line["estep_kinds_identical_across_ranks"] = out["estep_kinds_identical_across_ranks"]
        line["per_rank_ms_per_step"] = [round(p["ms_per_step"], 3) for p in out["per_rank"]]
    return line


def emit(out, args):
    """Full record to the detail file, the compact line - and nothing else - on stdout."""
    line = compact_line(out)
    if args.detail == "-":
        show_stdout(True)
        print(json.dumps(out), flush=True)
        show_stdout(False)
    elif args.detail:
        try:
            os.makedirs(os.path.dirname(os.path.abspath(args.detail)), exist_ok=True)
            with open(args.detail, "w") as f:
                json.dump(out, f)
            line["detail"] = args.detail
        except OSError:
            pass
    # the driver's record keeps a bounded tail of stdout: never let the line outgrow it (drop optional blocks first)
    for drop in (None, "full_fit_what", "offpath", "spread_sweep", "hard_workload", "per_rank_ms_per_step", "window", "hmm_c5"):
        if drop:
            line.pop(drop, None)
        text = json.dumps(line)
        if len(text) < 6000:
            break
    show_stdout(True)
    print(text, flush=True)
    show_stdout(False)


def dense_leg_run(w, K, D, n_local, fl_pair):
    """The last iteration's data pass again, every pair in f64 with the dense kernels, on the same parameters: the
    floor the policy falls back to, its MFMA utilisation, and the difference between the two paths' statistics."""
    from bayesml_amd._engine import DataPass
    eng, xd, q = w.eng, w.xd, w.q
    with env_vars(GMMVB_ESTEP_PRUNE="0", GMMVB_MSTEP_SPARSE="0"):
        ref = DataPass(K, D, xd.dtype, n_local, xd.device)
    ref.profile(True)
    ref.set_pivot(eng.pivot)
    ref.prepare_rows(xd)
    ref.set_params(q.c, q.m, q.u)
    st_ref = ref.estep_mstep(xd)
    st_ref = ref.estep_mstep(xd)            # second pass: warm instruction caches / clocks
    e_ms, m_ms = ref.last_kernel_ms()
    ref_info = ref.launch_info
    eng.set_params(q.c, q.m, q.u)
    st_new = eng.estep_mstep(xd)
    # one definition: per block of the statistics (ns, h, a, B) max|sparse - dense| / max|dense|; the line's
    # max_rel_diff is the largest of the four
    blocks_rel = [float((a_ - b_).abs().max() / b_.abs().max())
                  for a_, b_ in zip(eng.split_stats(st_new), ref.split_stats(st_ref))]
    ex = fl_pair * n_local * K
    leg = {"estep": {"kernel": ref_info.split("|")[0].strip(), "ms": e_ms, "executed_f64_tflops": ex / e_ms / 1e9,
                     "frac_of_f64_mfma_peak": ex / e_ms / 1e9 / PEAK_F64_MFMA_TFLOPS},
           "mstep": {"kernel": ref_info.split("|")[1].strip(), "ms": m_ms, "executed_f64_tflops": ex / m_ms / 1e9,
                     "frac_of_f64_mfma_peak": ex / m_ms / 1e9 / PEAK_F64_MFMA_TFLOPS},
           "peak_f64_mfma_tflops": PEAK_F64_MFMA_TFLOPS,
           "kernel_only_samples_per_s": n_local / ((e_ms + m_ms) * 1e-3),
           "sparse_vs_dense_statistics": {"max_rel_diff": max(blocks_rel), "per_block_ns_h_a_B": blocks_rel,
                                          "sparse_kernels": eng.launch_info}}
    ref.close()
    return leg


def hard_workload_leg(K, D, n_local, tdtype, ndtype, dev, warmup=2, steps=3, parity=True):
    """Same shape, cluster means 0.3 * randn: the components overlap, (almost) every pair matters and the policy has to
    stay on (or fall back to) the dense kernels."""
    import torch
    x = device_rows(K, D, n_local, tdtype, dev, SEED + 77, 0.3)
    w = Workload(K, D, x, dev, None)
    for _ in range(warmup):
        w.step()
    c0 = w.eng.pass_counts()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    snaps = []
    for _ in range(steps):
        w.step()
        snaps.append(w.snapshot())
    torch.cuda.synchronize()
    dt_s = time.perf_counter() - t0
    c1 = w.eng.pass_counts()
    out = {"workload": f"K={K} D={D} N={n_local}, cluster means 0.3 * randn (overlapping)", "steps": steps, "warmup": warmup,
           "samples_per_s": n_local * steps / dt_s, "ms_per_step": dt_s / steps * 1e3,
           "kernel_launches": {k: c1[k] - c0[k] for k in c1}, "per_step": snaps}
    out["summary"] = {"samples_per_s": out["samples_per_s"], "ms_per_step": out["ms_per_step"]}
    w.close()
    if parity:       # a bounded oracle run on overlapping rows (6 iterations over 6000 rows), both policies
        x_ref = recipe_rows_host(K, D, min(6000, n_local), ndtype, 0.3)
        _base, out["parity"], out["parity_sparse_path"] = cpu_baseline_and_parity(K, D, x_ref, dev, iters=6)
    return out


def policy_run(K, D, x, dev, warmup, steps, **switches):
    """`steps` timed VB iterations after `warmup` on x under the library switches given: ms per step, pass kinds, pairs."""
    import torch
    with env_vars(**switches):
        w = Workload(K, D, x, dev, None)
    for _ in range(warmup):
        w.step()
    c0 = w.eng.pass_counts()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    walls, act, ev, kinds = [], [], [], []
    for _ in range(steps):
        ts = time.perf_counter()
        w.step()
        walls.append((time.perf_counter() - ts) * 1e3)
        a, e = w.eng.sparsity()
        act.append(a / w.n if a >= 0 else float(K))
        ev.append(e / w.n)
        kinds.append(kernel_name(w.eng.launch_info.split("|")[0]))
    torch.cuda.synchronize()
    dt_s = time.perf_counter() - t0
    c1 = w.eng.pass_counts()
    out = {"ms_per_step": dt_s / steps * 1e3, "wall_ms": [round(v, 2) for v in walls],
           "kernel_launches": {k: c1[k] - c0[k] for k in c1 if c1[k] - c0[k]},
           "estep_kernel": kinds, "active_pairs_per_row": [round(v, 2) for v in act],
           "evaluated_pairs_per_row": [round(v, 2) for v in ev],
           "mean_active_pairs_per_row": float(np.mean(act)), "mean_evaluated_pairs_per_row": float(np.mean(ev))}
    w.close()
    return out


def spread_sweep_leg(K, D, n, tdtype, ndtype, dev, spreads=(0.5, 0.75, 1.0, 1.5), warmup=2, steps=20, parity=True):
    """The middle of the separation spectrum (the headline's recipe puts the cluster means 2 * randn apart, the hard
    workload 0.3 * randn): same shape, cluster means spread * randn, iterations 3-22 of one restart under the default
    policy and with the dense kernels only; the default policy must never be slower than 1.1 x the dense step."""
    import torch
    rows = []
    for sp in spreads:
        x = device_rows(K, D, n, tdtype, dev, SEED + 101, sp)
        r = {"spread": sp, "rows": n, "default": policy_run(K, D, x, dev, warmup, steps),
             "dense": policy_run(K, D, x, dev, warmup, steps, GMMVB_ESTEP_PRUNE="0", GMMVB_MSTEP_SPARSE="0")}
        for k in ("wall_ms", "estep_kernel", "active_pairs_per_row", "evaluated_pairs_per_row"):
            r["dense"].pop(k, None)
        r["default_over_dense"] = r["default"]["ms_per_step"] / r["dense"]["ms_per_step"]
        del x
        torch.cuda.empty_cache()
        if parity:       # a bounded oracle run on rows of the same mixture (6 iterations over 6000 rows), both policies
            x_ref = recipe_rows_host(K, D, min(6000, n), ndtype, sp)
            _b, p0, p1 = cpu_baseline_and_parity(K, D, x_ref, dev, iters=6)
            r["parity"] = {k: p0[k] for k in ("max_rel_err", "passed", "rows", "iterations", "path")}
            r["parity_sparse_path"] = {k: p1[k] for k in ("max_rel_err", "passed", "rows", "iterations", "path")}
        rows.append(r)
    return {"workload": f"K={K} D={D} N={n}, cluster means spread * randn; iterations {warmup + 1}-{warmup + steps} of one restart",
            "steps": steps, "warmup": warmup, "spreads": rows,
            "worst_default_over_dense": max(r["default_over_dense"] for r in rows),
            "summary": {"default_over_dense": {str(r["spread"]): round(r["default_over_dense"], 3) for r in rows},
                        "parity_max_rel_err": max([max(r["parity"]["max_rel_err"], r["parity_sparse_path"]["max_rel_err"])
                                                   for r in rows if "parity" in r] or [None])}}


def offpath_mixture(kind, K, D, spread=2.0):
    """The three workloads off the headline's recipe (equal weights, unit covariances, K_model = K_data):
    kdata   the rows come from K / 2 clusters: half of the model's components end up empty or as duplicates
    weights mixing weights ~ Dirichlet(0.3): a few big clusters, many tiny ones
    aniso   every cluster has its own per-feature standard deviations, log-uniform in [0.3, 3]
    Returns (means [Kd, D], weights [Kd] or None, scales [Kd, D] or None), drawn on the host from SEED."""
    rng = np.random.default_rng(SEED + {"kdata": 11, "weights": 12, "aniso": 13}[kind])
    Kd = K // 2 if kind == "kdata" else K
    mu = spread * rng.standard_normal((Kd, D))
    w = rng.dirichlet(np.full(Kd, 0.3)) if kind == "weights" else None
    sc = np.exp(rng.uniform(np.log(0.3), np.log(3.0), (Kd, D))) if kind == "aniso" else None
    return mu, w, sc


def offpath_rows_host(mix, n, dtype, seed):
    mu, w, sc = mix
    rng = np.random.default_rng(seed)
    z = rng.integers(0, mu.shape[0], n) if w is None else rng.choice(mu.shape[0], size=n, p=w)
    e = rng.standard_normal((n, mu.shape[1]))
    return (mu[z] + (e if sc is None else e * sc[z])).astype(dtype)


def offpath_rows_device(mix, n, dtype, dev, seed):
    import torch
    mu, w, sc = (None if a is None else torch.from_numpy(a).to(dev) for a in mix)
    x = torch.empty((n, mu.shape[1]), dtype=dtype, device=dev)
    gen = torch.Generator(device=dev).manual_seed(seed)
    step = 1 << 20
    for lo in range(0, n, step):
        hi = min(n, lo + step)
        if w is None:
            z = torch.randint(0, mu.shape[0], (hi - lo,), device=dev, generator=gen)
        else:
            z = torch.multinomial(w, hi - lo, replacement=True, generator=gen)
        e = torch.randn(hi - lo, mu.shape[1], dtype=torch.float64, device=dev, generator=gen)
        x[lo:hi] = (mu[z] + (e if sc is None else e * sc[z])).to(dtype)
    return x


def offpath_leg(K, D, n, tdtype, ndtype, dev, warmup=2, steps=20, parity=True):
    """Perf and parity off the happy path: the default policy against the dense kernels (it must not lose more than 10 %),
    and a 6000-row oracle parity run under both policies, on the three mixtures of offpath_mixture()."""
    import torch
    rows = []
    for kind in ("kdata", "weights", "aniso"):
        mix = offpath_mixture(kind, K, D)
        x = offpath_rows_device(mix, n, tdtype, dev, SEED + 301)
        r = {"kind": kind, "rows": n, "model_components": K, "data_clusters": int(mix[0].shape[0]),
             "default": policy_run(K, D, x, dev, warmup, steps),
             "dense": policy_run(K, D, x, dev, warmup, steps, GMMVB_ESTEP_PRUNE="0", GMMVB_MSTEP_SPARSE="0")}
        for k in ("wall_ms", "estep_kernel", "active_pairs_per_row", "evaluated_pairs_per_row"):
            r["dense"].pop(k, None)
        r["default_over_dense"] = r["default"]["ms_per_step"] / r["dense"]["ms_per_step"]
        del x
        torch.cuda.empty_cache()
        if parity:
            x_ref = offpath_rows_host(mix, min(6000, n), ndtype, SEED + 302)
            _b, p0, p1 = cpu_baseline_and_parity(K, D, x_ref, dev, iters=6)
            r["parity"] = {k: p0[k] for k in ("max_rel_err", "passed", "rows", "iterations", "path")}
            r["parity_sparse_path"] = {k: p1[k] for k in ("max_rel_err", "passed", "rows", "iterations", "path")}
        rows.append(r)
    return {"workload": f"K={K} D={D} N={n}: model components = 2 x data clusters | mixing weights ~ Dirichlet(0.3) | per-feature "
                        f"scales in [0.3, 3]; iterations {warmup + 1}-{warmup + steps} of one restart",
            "steps": steps, "warmup": warmup, "mixtures": rows,
            "worst_default_over_dense": max(r["default_over_dense"] for r in rows),
            "summary": {"default_over_dense": {r["kind"]: round(r["default_over_dense"], 3) for r in rows},
                        "default_ms_per_step": {r["kind"]: round(r["default"]["ms_per_step"], 2) for r in rows},
                        "parity_max_rel_err": max([max(r["parity"]["max_rel_err"], r["parity_sparse_path"]["max_rel_err"])
                                                   for r in rows if "parity" in r] or [None])}}


def full_fit_leg(K, D, x, dev, max_itr=25, num_init=2):
    """A whole fit through the public API on the resident matrix: update_posterior(max_itr, num_init, tolerance=0)."""
    import torch
    from bayesml_amd import gaussianmixture as gm
    m = gm.LearnModel(K, D, seed=0, device=dev, verbose=False)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m.update_posterior(x, max_itr=1, num_init=1, tolerance=0.0)      # first call: workspace allocation + three data passes
        torch.cuda.synchronize()
        first = time.perf_counter() - t0
        c0 = m._engine.pass_counts()
        t0 = time.perf_counter()
        m.update_posterior(x, max_itr=max_itr, num_init=num_init, tolerance=0.0)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        c1 = m._engine.pass_counts()
    passes = num_init * (max_itr + 1) + 1
    hn = m.get_hn_params()
    out = {"what": f"update_posterior(max_itr={max_itr}, num_init={num_init}, tolerance=0) through the public API, x resident",
           "seconds": el, "data_passes": passes, "ms_per_data_pass": el / passes * 1e3,
           "samples_per_s_per_data_pass": x.shape[0] * passes / el,
           "first_call_seconds_workspace_plus_3_passes": first,
           "kernel_launches": {k: c1[k] - c0[k] for k in c1 if c1[k] - c0[k]},
           "checksum_hn_m_vecs": float(np.abs(hn["hn_m_vecs"]).sum()), "final_vl": float(m.vl)}
    m._engine.close()
    m._engine = None
    return out


def small_c1_leg(dev, reps=5):
    """BASELINE.json configs[0] (K=3, D=2, N=1000) through the public API with the reference's defaults (10 restarts, up
    to 100 iterations each, tolerance 1e-8): one launch runs all of it (csrc/small.hip).  The reference takes 0.16 s on the
    survey container's CPU."""
    import contextlib
    import io
    import torch
    from bayesml_amd import gaussianmixture as gm
    gen = gm.GenModel(3, 2, pi_vec=np.array([0.3, 0.3, 0.4]), mu_vecs=np.array([[-4.0, 0.0], [4.0, 0.0], [0.0, 5.0]]), seed=123)
    x1, _ = gen.gen_sample(1000)
    out = {}
    for label, flag in (("one_launch", "1"), ("general_engine", "0")):
        old = os.environ.get("BAYESML_AMD_SMALL")
        os.environ["BAYESML_AMD_SMALL"] = flag
        times, iters, vl = [], 0, None
        try:
            for _ in range(reps):                     # (the first call pays library / code-object loads)
                m = gm.LearnModel(3, 2, seed=0, device=dev)
                buf = io.StringIO()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                with warnings.catch_warnings(), contextlib.redirect_stdout(buf):
                    warnings.simplefilter("ignore")
                    m.update_posterior(x1)
                torch.cuda.synchronize()
                times.append(time.perf_counter() - t0)
                iters, vl = buf.getvalue().count("t="), float(m.vl)
                if m._engine is not None:
                    m._engine.close()
        finally:
            os.environ.pop("BAYESML_AMD_SMALL", None)
            if old is not None:
                os.environ["BAYESML_AMD_SMALL"] = old
        out[label] = {"seconds_first_call": times[0], "seconds": min(times[1:]), "vb_iterations": iters, "final_vl": vl}
    out["workload"] = "gaussianmixture.LearnModel K=3 D=2 N=1000, update_posterior() defaults (10 restarts, tolerance 1e-8)"
    out["reference_seconds_survey_container_cpu"] = 0.16
    return out


def hmm_c5_leg(dev, steps=10, warmup=5, cpu=True):
    """BASELINE.json configs[4] (hiddenmarkovnormal.LearnModel K=32, D=16, T=1e7): tools/bench_hmm.py's measurement."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_hmm
    a = argparse.Namespace(steps=steps, warmup=warmup, rows=10_000_000, classes=32, degree=16, ref_rows=4000, no_cpu=not cpu)
    return bench_hmm.measure(a, dev)


if __name__ == "__main__":
    main()
