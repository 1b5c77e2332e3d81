#!/usr/bin/env python3
"""Diagnostics of the carried E-step: per VB iteration the drift hint's spread over the components and what the
records made of it (needs a GPU; GMMVB_DEBUG=2 adds the library's own mode line on stderr)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    K, D = 64, 128
    dev = torch.device("cuda", 0)
    x = bench.device_rows(K, D, n, torch.float32, dev, 1, 2.0)
    w = bench.Workload(K, D, x, dev, None)
    for it in range(iters):
        ks = w.ks
        g, dl, G = ks.gamma.cpu().numpy(), ks.delta.cpu().numpy(), ks.big_gamma.cpu().numpy()
        ns = ks.ns.cpu().numpy()
        order = np.argsort(g * 25.0 - dl)
        w.step()
        a, e = w.eng.sparsity()
        counts = w.eng.pass_counts()
        print(f"it {it:2d} gamma min {g.min():.3f} med {np.median(g):.3f} | delta max {dl.max():.2f} med {np.median(dl):.3f} | "
              f"Gamma max {G.max():.3f} med {np.median(G):.3f} | worst comps {order[:4].tolist()} ns {ns[order[:4]].round(1).tolist()} "
              f"g {g[order[:4]].round(3).tolist()} d {dl[order[:4]].round(2).tolist()} | active {a / n:.2f} evaluated {e / n:.2f} "
              f"{w.eng.launch_info.split(' ')[0]} E {w.eng.last_kernel_ms()[0]:.1f} ms", flush=True)


if __name__ == "__main__":
    main()
