#!/usr/bin/env python3
"""Micro-benchmark of the proof round's kernel (estep_i8_proof over stored digit planes) at C3's shape: every row against
one component (N pairs), HIP-event time per launch, against the exact f64 gather's 0.31 ns per pair."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import bench  # noqa: E402
from bayesml_amd import _kside  # noqa: E402
from bayesml_amd._engine import DataPass  # noqa: E402


def main():
    K, D, N = 64, 128, int(os.environ.get("ROWS", "10000000"))
    dev = torch.device("cuda", 0)
    x = bench.device_rows(K, D, N, torch.float32, dev, bench.SEED + 1, 2.0)
    mu = torch.from_numpy(bench.recipe_means(K, D, 2.0)).to(dev)
    p = _kside.prior_from_numpy(np.full(K, 0.5), np.zeros((K, D)), np.ones(K), np.full(K, float(D)),
                                np.tile(np.eye(D), (K, 1, 1)), dev)
    q = _kside.post_from_prior(p)
    q.m = mu.clone()
    q.nu = q.nu + 1000.0
    q.kappa = q.kappa + 1000.0
    q = _kside.features(q)
    eng = DataPass(K, D, x.dtype, N, dev)
    eng.set_pivot(x[:4096].to(torch.float64).mean(dim=0))
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    eng.prepare_rows(x)
    t1.record()
    torch.cuda.synchronize()
    prep_ms = t0.elapsed_time(t1)
    eng.set_params(q.c, q.m, q.u)
    times = []
    for k in (0, 1, 2, 3, 4):
        t0.record()
        ub, lb = eng.debug_proof(k, N)
        t1.record()
        torch.cuda.synchronize()
        times.append(t0.elapsed_time(t1))
    gap = float((ub.double() - lb).max())
    print(json.dumps(dict(rows=N, prepare_rows_ms=prep_ms, proof_ms_per_launch=times, ns_per_pair=min(times) * 1e6 / N,
                          digit_bytes_per_pair=384, GBps=384 * N / (min(times) * 1e-3) / 1e9, max_gap_nats=gap,
                          note="each launch: N pairs (every row x one component), incl. two device-to-device copies of the "
                               "outputs (12 B per pair)")))
    eng.close()


if __name__ == "__main__":
    main()
