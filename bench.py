#!/usr/bin/env python3
"""GMM-VB E+M samples/sec at K=64, D=128 (BASELINE.json metric), one process per GPU.

A "step" is one full VB iteration of ``gaussianmixture.LearnModel.update_posterior``'s inner loop
(reference ``_gaussianmixture.py:864-867``): K-side posterior update -> parameter packing ->
E-step kernel -> row log-normaliser -> M-step kernel -> slab reduction -> (all-reduce over row
shards) -> moments -> variational lower bound (one host sync).  x is resident in HBM before the
timed region.  ``value`` = rows processed by all ranks per second.

Single GPU (default): N = 1e7 rows of f32 (BASELINE.json configs[2], the config the metric is quoted
on; it fits one GPU).  N GPUs: every rank holds its own 1e7 rows (weak scaling), statistics are
combined by ONE all-reduce(sum, f64) of K(2 + D + D^2) doubles per step over RCCL.

Also reported on the same JSON line:
  roofline      dominant kernel against the f64 MFMA peak, duration from HIP events recorded in the
                library around that kernel's launch on its stream (gmmvb_profile_last_ms)
  cpu_baseline  the oracle (NumPy port of the reference's formulation) timed on this host's cores
                over 10 VB iterations of the first N_ref = 20000 rows (rank 0, N = 1 only)
  parity        max relative error of the posterior hyper-parameters after those 10 iterations, GPU
                path vs oracle on the same N_ref rows (north_star tolerance 1e-5)
"""
import argparse
import json
import os
import sys
import time
import warnings

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from bayesml_amd import RowShard, _kside                      # noqa: E402
from bayesml_amd import gaussianmixture as gm                 # noqa: E402

SEED = 20250711
PEAK_F64_MFMA_TFLOPS = 78.6      # MI355X datasheet FP64 matrix (MI355X_MICROARCH.md lists no f64 row)


def recipe_means(K, D):
    """First draw of the synthetic recipe: mu = 2 * standard_normal((K, D))."""
    return 2.0 * np.random.default_rng(SEED).standard_normal((K, D))


def recipe_rows_host(K, D, n, dtype):
    """First n rows of the recipe, drawn on the host exactly like oracle.synth_gmm (chunk boundary 2**20 > n)."""
    rng = np.random.default_rng(SEED)
    mu = 2.0 * rng.standard_normal((K, D))
    z = rng.integers(0, K, n)
    return (mu[z] + rng.standard_normal((n, D))).astype(dtype)


def device_rows(K, D, n, dtype, dev, seed, head=None):
    """Same mixture drawn with the device generator, in chunks; the first len(head) rows are `head`."""
    mu = torch.from_numpy(recipe_means(K, D)).to(dev)
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


def cpu_baseline_and_parity(K, D, x_ref, dev, iters=10):
    """Oracle (test infrastructure) on the host cores vs the GPU driver on the same rows."""
    from oracle import gmm_vb_oracle as orc
    x64 = x_ref.astype(np.float64)
    p = orc.Prior.default(K, D)
    q = orc.Posterior.from_prior(p)
    orc.init_subsampling(x64, q, np.random.default_rng(0))
    st = orc.data_pass(x64, q)
    t0 = time.perf_counter()
    for _ in range(iters):
        orc.update_q_mu_lambda(p, q, st)
        orc.update_q_pi(p, q, st)
        st = orc.data_pass(x64, q, st.s)
        orc.lower_bound(p, q, st)
    cpu_s = time.perf_counter() - t0
    try:
        from threadpoolctl import threadpool_info
        threads = max([i.get("num_threads", 1) for i in threadpool_info()] or [1])
    except Exception:      # noqa: BLE001
        threads = os.cpu_count() or 1
    m = gm.LearnModel(K, D, seed=0, device=dev, verbose=False)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m.update_posterior(x_ref, max_itr=iters, num_init=1, tolerance=0.0)

    def rel(a, b):
        return float(np.max(np.abs(a - b)) / np.max(np.abs(b)))

    errs = dict(hn_alpha_vec=rel(m.hn_alpha_vec, q.alpha), hn_m_vecs=rel(m.hn_m_vecs, q.m),
                hn_kappas=rel(m.hn_kappas, q.kappa), hn_nus=rel(m.hn_nus, q.nu),
                hn_w_mats=rel(m.hn_w_mats, q.w), hn_w_mats_inv=rel(m.hn_w_mats_inv, q.w_inv))
    base = dict(value=x_ref.shape[0] * iters / cpu_s, unit="samples/s", cores=int(threads), kind="port",
                sample=f"{iters} VB iterations (K-side + E + M + lower bound, fp64 NumPy/OpenBLAS, "
                       f"{threads} threads of {os.cpu_count()} cores) over the first {x_ref.shape[0]} rows of the workload",
                seconds=cpu_s)
    par = dict(max_rel_err=max(errs.values()), tolerance=1e-5, passed=max(errs.values()) < 1e-5, per_array=errs,
               rows=int(x_ref.shape[0]), iterations=iters)
    m._engine.close()
    return base, par


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--rows", type=int, default=10_000_000, help="rows per GPU")
    ap.add_argument("--classes", type=int, default=64)
    ap.add_argument("--degree", type=int, default=128)
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"], help="storage dtype of x in HBM")
    ap.add_argument("--ref-rows", type=int, default=20_000)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline / parity leg")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    comm = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        comm = RowShard()
    K, D, n_local = args.classes, args.degree, args.rows
    tdtype = torch.float32 if args.dtype == "f32" else torch.float64
    ndtype = np.float32 if args.dtype == "f32" else np.float64

    # ---- workload, resident in HBM before anything is timed
    x_ref = recipe_rows_host(K, D, min(args.ref_rows, n_local), ndtype)
    x = device_rows(K, D, n_local, tdtype, dev, SEED + 1 + rank, head=x_ref if rank == 0 else None)

    cpu_base = parity = None
    if rank == 0 and world == 1 and not args.no_cpu:
        cpu_base, parity = cpu_baseline_and_parity(K, D, x_ref, dev)

    # ---- the model, driven through the same internals update_posterior uses
    m = gm.LearnModel(K, D, seed=0, device=dev, comm=comm, verbose=False)
    eng, xd = m._open(x)
    eng.profile(True)
    prior = m._prior_tensors(dev)
    q = _kside.post_from_prior(prior)
    q = m._init_subsampling(eng, xd, q, m._comm.global_rows)
    s_prev = torch.zeros(K, D, D, dtype=torch.float64, device=dev)
    ns, x_bar, s, h = m._pass(eng, xd, q, s_prev)

    def step():
        nonlocal q, ns, x_bar, s, h
        q = _kside.update_q(prior, ns, x_bar, s)
        ns, x_bar, s, h = m._pass(eng, xd, q, s)
        return float(_kside.lower_bound(prior, q, ns, x_bar, s, h)["vl"])

    for _ in range(args.warmup):
        step()

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    ker, launches, spars = [], [], []
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        vl = step()
        ker.append(eng.last_kernel_ms())        # events already complete: step() ended with a host sync
        launches.append(eng.launch_info)
        spars.append(eng.sparsity())
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        n_total = n_local * world
        e_ms = float(np.mean([k[0] for k in ker]))
        m_ms = float(np.mean([k[1] for k in ker]))
        # algorithmic flops per sample: E = K(2D^2+3D) + 6K, M = K(2D^2+2D) + 2KD
        fl_e = K * (2 * D * D + 3 * D) + 6 * K
        fl_m = K * (2 * D * D + 2 * D) + 2 * K * D
        names = [part.strip().split("<")[0] for part in eng.launch_info.split("|")]      # kernels actually launched
        # executed MFMA flops: T(T+1)/2 tile pairs of 16x16, x4 (E: per 16 samples) or x1 (M: per 4 samples) MFMAs of 2048 flops
        tiles = (D + 15) // 16
        fl_exec = K * 512 * tiles * (tiles + 1) // 2
        kern = {names[0]: dict(ms=e_ms, algorithmic_tflops=fl_e * n_local / (e_ms * 1e-3) / 1e12,
                               executed_mfma_tflops=fl_exec * n_local / (e_ms * 1e-3) / 1e12),
                names[1]: dict(ms=m_ms, algorithmic_tflops=fl_m * n_local / (m_ms * 1e-3) / 1e12,
                               executed_mfma_tflops=fl_exec * n_local / (m_ms * 1e-3) / 1e12)}
        dom = max(kern, key=lambda k: kern[k]["ms"])
        ach = kern[dom]["algorithmic_tflops"]
        # HBM-side bytes per launch of that kernel: PMC counters cannot be read from inside this process, so
        # take them from the committed rocprofv3 --pmc passes of this same command (tools/summarize_pmc.py)
        traffic = traffic_src = None
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                pm = json.load(f).get(dom)
            if pm and K == 64 and D == 128 and n_local == 10_000_000 and args.dtype == "f32":
                traffic = pm["fetch_bytes"] + pm["write_bytes"]
                traffic_src = "profiles/pmc_traffic.json: " + pm["note"]
        except (OSError, ValueError, KeyError):
            pass
        bytes_per_sample = D * x.element_size()
        out = {
            "metric": "GMM-VB E+M samples/sec at K=64,D=128,N=1e7; 1/2/4/8-GPU scaling",
            "value": n_total * args.steps / elapsed, "unit": "samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"GMM-VB K={K} D={D} N={n_local} rows/GPU x {world} GPU, x stored {args.dtype}, "
                                   "one VB iteration per step (configs[2] of BASELINE.json)",
                       "classes": K, "degree": D, "rows_per_gpu": n_local, "x_storage": args.dtype,
                       "parallelism": f"rows{world}"},
            "roofline": {"bound": "mfma", "kernel": dom, "achieved": ach, "peak": PEAK_F64_MFMA_TFLOPS,
                         "unit": "TFLOP/s", "frac": ach / PEAK_F64_MFMA_TFLOPS, "traffic": traffic,
                         "traffic_source": traffic_src,
                         "kernels": kern,
                         "hbm_algorithmic_GBps": bytes_per_sample * n_local / ((e_ms + m_ms) * 1e-3) / 1e9,
                         "note": "achieved = algorithmic (dense) flops of SURVEY 8d / HIP-event kernel time; the kernels "
                                 "execute ~0.56x of them (triangular whitening factor, symmetric second moment)"},
            "cpu_baseline": cpu_base, "parity": parity, "final_vl": vl, "launch": eng.launch_info,
            "per_step": {"estep_ms": [round(k[0], 2) for k in ker], "mstep_ms": [round(k[1], 2) for k in ker],
                         "estep_kernel": [l.split("<")[0] for l in launches],
                         "active_components_per_sample": [round(a / n_local, 2) for a, _ in spars],
                         "evaluated_components_per_sample": [round(e / n_local, 2) for _, e in spars]},
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
