#!/usr/bin/env python3
"""The launch-bound configurations through the public API (needs a GPU):
  C1  gaussianmixture.LearnModel K=3, D=2, N=1000, update_posterior() with the reference's defaults (10 restarts, up
      to 100 iterations each, tolerance 1e-8) - the reference takes 0.16 s on the survey container's CPU;
  C2  K=16, D=32, N=1e6 f64 rows: ms per VB iteration over 30 iterations of one restart.
Prints one JSON line."""
import contextlib
import io
import json
import os
import sys
import time
import warnings

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesml_amd import gaussianmixture as gm      # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    out = {}
    gen = gm.GenModel(3, 2, pi_vec=np.array([0.3, 0.3, 0.4]), mu_vecs=np.array([[-4.0, 0.0], [4.0, 0.0], [0.0, 5.0]]), seed=123)
    x1, _ = gen.gen_sample(1000)
    times, iters = [], 0
    for rep in range(4):                     # the first call pays library / graph warm-up
        m = gm.LearnModel(3, 2, seed=0, device=dev)
        buf = io.StringIO()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with warnings.catch_warnings(), contextlib.redirect_stdout(buf):
            warnings.simplefilter("ignore")
            m.update_posterior(x1)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
        iters = buf.getvalue().count("t=")
    out["c1"] = {"workload": "K=3 D=2 N=1000, update_posterior() defaults (10 restarts)", "seconds_first_call": times[0],
                 "seconds": min(times[1:]), "vb_iterations": iters, "ms_per_iteration": min(times[1:]) / max(iters, 1) * 1e3,
                 "reference_seconds_survey_container_cpu": 0.16}
    rng = np.random.default_rng(1)
    mu = 2.0 * rng.standard_normal((16, 32))
    x2 = torch.from_numpy(mu[rng.integers(0, 16, 1_000_000)] + rng.standard_normal((1_000_000, 32))).to(dev)
    m = gm.LearnModel(16, 32, seed=0, device=dev, verbose=False)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m.update_posterior(x2, max_itr=3, num_init=1, tolerance=0.0)      # warm-up (workspace, x copy)
    for itr in (5, 35):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            m.update_posterior(x2, max_itr=itr, num_init=1, tolerance=0.0)
        torch.cuda.synchronize()
        out.setdefault("c2_runs", []).append((itr, time.perf_counter() - t0))
    (i0, t0_), (i1, t1_) = out["c2_runs"]
    out["c2"] = {"workload": "K=16 D=32 N=1e6 f64, one restart", "ms_per_iteration": (t1_ - t0_) / (i1 - i0) * 1e3,
                 "samples_per_s": 1e6 * (i1 - i0) / (t1_ - t0_)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
