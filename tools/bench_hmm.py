#!/usr/bin/env python3
"""HMM-VB time steps/sec at K=32, D=16, T=1e7 (BASELINE.json configs[4]) on one MI355X.

Not the headline metric (bench.py is); this is the measurement for SURVEY.md section 8 row a11.
A "step" = one VB iteration of hiddenmarkovnormal.LearnModel.update_posterior's inner loop
(reference ``_hiddenmarkovnormal.py:1099-1105``): K-side update -> emission E-step -> forward-backward ->
statistics -> lower bound.  Prints one JSON line; ``cpu_baseline`` times the oracle (NumPy restatement of
the reference's Python loops over T) on the first ``--ref-rows`` steps and the parity of the posterior
after 5 iterations on those steps.
"""
import argparse
import json
import os
import sys
import time
import warnings

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from bayesml_amd import _kside                                   # noqa: E402
from bayesml_amd import hiddenmarkovnormal as hmm                # noqa: E402


def synth_device(K, D, T, dtype, dev, seed=20250711, stay=0.9, head=None):
    """Sticky chain drawn on the device (state sequence via blocks of geometric run lengths is overkill:
    a jump mask + cumulative 'last jump' gather gives the same process)."""
    gen = torch.Generator(device=dev).manual_seed(seed)
    mu = 3.0 * torch.randn(K, D, dtype=torch.float64, device=dev, generator=gen)
    jump = torch.rand(T, device=dev, generator=gen) >= stay
    jump[0] = True
    target = torch.randint(0, K, (T,), device=dev, generator=gen)
    idx = torch.arange(T, device=dev)
    last = torch.cummax(torch.where(jump, idx, torch.zeros_like(idx)), dim=0).values
    z = target[last]
    x = torch.empty((T, D), dtype=dtype, device=dev)
    step = 1 << 20
    for lo in range(0, T, step):
        hi = min(T, lo + step)
        x[lo:hi] = (mu[z[lo:hi]] + torch.randn(hi - lo, D, dtype=torch.float64, device=dev, generator=gen)).to(dtype)
    if head is not None:
        x[: head.shape[0]] = torch.from_numpy(head).to(dev)
    return x


def measure(args, dev=None):
    """The measurement itself (also bench.py's ``hmm_c5`` leg): returns the JSON line as a dict."""
    K, D, T = args.classes, args.degree, args.rows
    if dev is None:
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(0)

    cpu = parity = None
    x_ref = None
    if not args.no_cpu:
        # the oracle is only touched by this cpu_baseline / parity leg (it also supplies the rows both sides see)
        from oracle import hmm_vb_oracle as orc
        x_ref, _ = orc.synth_hmm(K, D, min(args.ref_rows, T), np.float32)
    x = synth_device(K, D, T, torch.float32, dev, head=x_ref)

    if not args.no_cpu:
        x64 = x_ref.astype(np.float64)
        p = orc.HmmPrior.default(K, D)
        q = orc.HmmPosterior.from_prior(p)
        orc.init_subsampling(x64, q, np.random.default_rng(0))
        st = orc.data_pass(x64, q)
        iters = 5
        t0 = time.perf_counter()
        for _ in range(iters):
            orc.update_q(p, q, st)
            st = orc.data_pass(x64, q, st.s)
            orc.lower_bound(p, q, st)
        cpu_s = time.perf_counter() - t0
        m = hmm.LearnModel(K, D, seed=0, device=dev, verbose=False)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            m.update_posterior(x_ref, max_itr=iters, num_init=1, tolerance=0.0)

        def rel(a, b):
            return float(np.max(np.abs(a - b)) / np.max(np.abs(b)))

        errs = dict(hn_eta_vec=rel(m.hn_eta_vec, q.eta), hn_zeta_vecs=rel(m.hn_zeta_vecs, q.zeta),
                    hn_m_vecs=rel(m.hn_m_vecs, q.m), hn_kappas=rel(m.hn_kappas, q.kappa), hn_nus=rel(m.hn_nus, q.nu),
                    hn_w_mats=rel(m.hn_w_mats, q.w))
        cpu = dict(value=x_ref.shape[0] * iters / cpu_s, unit="time steps/s", cores=1, kind="port",
                   sample=f"{iters} VB iterations over the first {x_ref.shape[0]} steps (Python loop over T like the "
                          "reference; the K x K mat-vec per step does not thread)", seconds=cpu_s)
        parity = dict(max_rel_err=max(errs.values()), tolerance=1e-5, passed=max(errs.values()) < 1e-5, per_array=errs)
        m._engine.close()

    m = hmm.LearnModel(K, D, seed=0, device=dev, verbose=False)
    eng, xd = m._open(x)
    eng.enable_hmm()
    eng.hmm_skip_h(True)                                                        # as update_posterior does
    eng.emission_target(os.environ.get("BENCH_HMM_LN_RHO_ARRAY") is None)       # (developer switch: the ln rho array + hmm_prep_kernel)
    prior = m._prior_tensors(dev)
    q = _kside.hmm_post_from_prior(prior)
    size, a, B = m._subsample_moments(eng, xd, T)
    q = _kside.subsample_moments_init(q, size, a, B, eng.pivot, _kside.hmm_features)
    # update_posterior's loop (ref:1099-1105): the data pass writes into the K-side stepper's buffers, the stepper's one unit
    # (a replayed hipGraph) gives the lower bound and the next posterior, one device-to-host copy per iteration
    ks = _kside.HmmKStepper(prior, eng.pivot, eng.stats_len)
    ks.load(q)
    m._stepper_pass(eng, xd, ks)
    ks.step()
    ks.read()

    whole = m._whole_iteration_graph(xd)

    def step():
        ks.iterate(lambda: m._stepper_pass(eng, xd, ks), whole)
        return ks.read()["vl"]

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        vl = step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    boundary_pass = {-1: "chunk products", 0: "forgetting pass stood", 1: "forgetting pass, then chunk products"}.get(
        eng.last_boundary_pass(), "?")
    # HBM roofline (the chunked scan is bandwidth / latency bound, K^2 flops per byte of state are far below the MFMA
    # balance): algorithmic bytes per time step = the row of x read once (D * 4) + one [K] f64 state vector written by the
    # forward recursion and read back by the backward one (the reference materialises ln_rho, alpha, beta, gamma [T, K]
    # and xi [T, K, K]); the kernels of DESIGN.md section 4c sweep about twelve [T][16 ceil(K/16)] f64 arrays per iteration.
    # Viterbi path of the whole sequence under the last posterior (chunked max-plus scan, hmm.h hmm_vit_*)
    qf = ks.current()
    eng.emission_target(False)              # (the Viterbi pass reads the ln rho array)
    eng.set_params(qf.c, qf.m, qf.u)
    eng.estep(xd)
    torch.cuda.synchronize()
    tv = time.perf_counter()
    if getattr(args, "no_viterbi", False):      # (more than 64 states: the sequential Viterbi kernel, seconds per million steps)
        z = torch.zeros(1, dtype=torch.int32)
    else:
        z = eng.viterbi(qf.ln_pi_tilde, qf.ln_a_tilde)
    torch.cuda.synchronize()
    viterbi_ms = (time.perf_counter() - tv) * 1e3
    Kp = 16 * ((K + 15) // 16)
    alg = D * 4 + 2 * K * 8
    swept = 12 * Kp * 8 + D * 4 + 16 * ((D + 15) // 16) * 8
    ms = el / args.steps * 1e3
    roofline = {"bound": "hbm", "kernel": "whole HMM iteration (emission, boundary sweeps / chunk products, replays with the xi sum, M-step)",
                "achieved": alg * T / (ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                "frac": alg * T / (ms * 1e-3) / 1e9 / 8000.0, "algorithmic_bytes_per_time_step": alg,
                "estimated_swept_bytes_per_time_step": swept, "estimated_swept_GBps": swept * T / (ms * 1e-3) / 1e9,
                "hbm_roofline_time_steps_per_s": 8e12 / alg, "traffic": None}
    # the other yardstick: the f64 matrix pipe.  Algorithmic flop per time step, as the reference computes it: emission
    # K(2 D^2 + 3 D) (ref:988-997), statistics K(2 D^2 + 2 D) + 2 K D (ref:837-845), forward, backward and the xi sum
    # 2 K^2 each (ref:999-1018), gamma K.  At K = 32, D = 16: 42.5 kflop against 576 B = 74 flop/B, above the pipe's
    # balance of 78.6 TFLOP/s / 8 TB/s = 9.8 flop/B -> the pipe binds; `bound` names the larger fraction.
    flop = K * (2 * D * D + 3 * D) + K * (2 * D * D + 2 * D) + 2 * K * D + 3 * 2 * K * K + K
    tf = flop * T / (ms * 1e-3) / 1e12
    roofline["hbm_frac"] = roofline["frac"]
    roofline["hbm_achieved_GBps"] = roofline["achieved"]
    roofline["algorithmic_flop_per_time_step"] = flop
    roofline["f64_mfma_frac"] = tf / 78.6
    # one meaning per name, as in bench.py's roofline: frac_on_F = the reference's dense flop count (above) / time / peak;
    # frac_executed would need the flops the kernels really issue, and the M-step's bit walk (csrc/hmm.h) skips an uncounted
    # share of its MFMA steps - not reported rather than guessed
    roofline["frac_on_F"] = tf / 78.6
    roofline["frac_executed"] = None
    if roofline["f64_mfma_frac"] > roofline["hbm_frac"]:
        roofline.update(bound="mfma", achieved=tf, peak=78.6, unit="TFLOP/s", frac=tf / 78.6)
    try:            # measured HBM bytes per iteration (tools/hmm_pmc_total.py), only if made for this very workload
        with open(os.path.join(ROOT, "profiles", "hmm_pmc_traffic.json")) as f:
            pm = json.load(f)
        if pm.get("config") == f"K{K} D{D} T{T}":
            roofline["traffic"] = pm["bytes_per_iteration"]
            roofline["traffic_GBps"] = pm["bytes_per_iteration"] / (ms * 1e-3) / 1e9
            roofline["traffic_frac_of_peak"] = roofline["traffic_GBps"] / 8000.0
            roofline["traffic_source"] = "profiles/hmm_pmc_traffic.json: " + pm["note"]
    except (OSError, ValueError, KeyError):
        pass
    eng.close()
    m._engine = None
    del x, xd
    return ({
        "metric": "HMM-VB time steps/sec at K=32,D=16,T=1e7 (BASELINE.json configs[4])", "value": T * args.steps / el,
        "unit": "time steps/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": el / args.steps * 1e3, "higher_is_better": True, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"HMM-VB K={K} D={D} T={T}, x stored f32, one VB iteration per step"},
        "window": f"VB iterations {args.warmup + 1}-{args.warmup + args.steps} of one restart",
        "roofline": roofline, "cpu_baseline": cpu, "parity": parity, "final_vl": vl, "boundary_pass": boundary_pass,
        "viterbi": None if getattr(args, "no_viterbi", False) else {"ms": viterbi_ms, "time_steps_per_s": T / (viterbi_ms * 1e-3), "states_visited": int(torch.unique(z).numel()),
                    "note": "hmmvb_viterbi over all T steps after the emission E-step (round 2: one sequential wave, ~10 s)"}})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--classes", type=int, default=32)
    ap.add_argument("--degree", type=int, default=16)
    ap.add_argument("--ref-rows", type=int, default=20_000)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-viterbi", action="store_true")
    print(json.dumps(measure(ap.parse_args())))


if __name__ == "__main__":
    main()
