#!/usr/bin/env python3
"""How much tighter would bounds carried from a fixed anchor be than bounds carried step by step?  Runs C3's first 26 VB
iterations, keeps every posterior's (u, u^-1, m), and prints for anchors a = 5, 10, 15 and every later iteration t the
direct drift gamma(a -> t) = lower bound of sigma_min(u_t u_a^-1) next to the product of the per-step gammas."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from bayesml_amd import _kside  # noqa: E402


def main():
    K, D, N = 64, 128, int(os.environ.get("ROWS", "2000000"))
    dev = torch.device("cuda", 0)
    x = bench.device_rows(K, D, N, torch.float32, dev, bench.SEED + 1, 2.0)
    w = bench.Workload(K, D, x, dev, None)
    qs = [_kside._clone_post(w.ks.q)]
    for _ in range(26):
        w.step()
        qs.append(_kside._clone_post(w.ks.q))
    out = {}
    for a in (5, 10, 15):
        prod = torch.ones(K, dtype=torch.float64, device=dev)
        rows = []
        for t in range(a + 1, len(qs)):
            g_step, d_step, _G = _kside.drift(qs[t - 1], qs[t])
            prod = prod * g_step
            g_dir, d_dir, G_dir = _kside.drift(qs[a], qs[t])
            rows.append(dict(t=t, direct_min=float(g_dir.min()), direct_med=float(g_dir.median()),
                             product_min=float(prod.min()), product_med=float(prod.median()),
                             delta_direct_max=float(d_dir.max()), big_gamma_direct_max=float(G_dir.max()),
                             worst5_direct=[round(float(v), 3) for v in torch.sort(g_dir).values[:5]],
                             worst5_product=[round(float(v), 3) for v in torch.sort(prod).values[:5]]))
        out[f"anchor_{a}"] = rows
    print(json.dumps(out))
    w.close()


if __name__ == "__main__":
    main()
