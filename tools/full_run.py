#!/usr/bin/env python3
"""Whole update_posterior() at the benchmark configuration (K=64, D=128, N rows of f32 on one GPU): restarts x
iterations through the public LearnModel API, default policy vs dense kernels only.  Prints one JSON line."""
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
import bench                                                     # noqa: E402
from bayesml_amd import gaussianmixture as gm                    # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--classes", type=int, default=64)
    ap.add_argument("--degree", type=int, default=128)
    ap.add_argument("--max-itr", type=int, default=25)
    ap.add_argument("--num-init", type=int, default=2)
    ap.add_argument("--dense", action="store_true")
    args = ap.parse_args()
    if args.dense:
        os.environ["GMMVB_ESTEP_PRUNE"] = "0"
        os.environ["GMMVB_MSTEP_SPARSE"] = "0"
    dev = torch.device("cuda", 0)
    K, D, N = args.classes, args.degree, args.rows
    x = bench.device_rows(K, D, N, torch.float32, dev, bench.SEED + 1, 2.0)
    m = gm.LearnModel(K, D, seed=0, device=dev, verbose=False)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m.update_posterior(x, max_itr=1, num_init=1, tolerance=0.0)      # first call: allocations, code-object loads
        torch.cuda.synchronize()
        first = time.perf_counter() - t0
        t0 = time.perf_counter()
        m.update_posterior(x, max_itr=args.max_itr, num_init=args.num_init, tolerance=0.0)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
    passes = args.num_init * (args.max_itr + 1) + 1
    hn = m.get_hn_params()
    print(json.dumps({"what": "update_posterior(max_itr=%d, num_init=%d, tolerance=0)" % (args.max_itr, args.num_init),
                      "dense_only": args.dense, "first_call_3_passes_seconds": first, "K": K, "D": D, "rows": N, "seconds": el, "data_passes": passes,
                      "ms_per_data_pass": el / passes * 1e3, "samples_per_s": N * passes / el,
                      "checksum_hn_m_vecs": float(np.abs(hn["hn_m_vecs"]).sum()),
                      "checksum_hn_w_mats": float(np.abs(hn["hn_w_mats"]).sum())}))


if __name__ == "__main__":
    main()
