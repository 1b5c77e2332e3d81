#!/usr/bin/env python3
"""How far below the row's log-normaliser do the ACTIVE pairs of a pass lie?  C3 shape (or CLASSES / DEGREE / ROWS / SPREAD):
at iterations ITERS (default 6, 8, 10, 15, 25) the responsibilities of every row are read back and the non-dominant pairs
with r >= 2^-100 are binned by -log2 r.  Decides whether the relevance line (2^-100 today: a pair below it cannot change
any f64 sum of fewer than 2^40 terms) could sit higher: the pairs between two candidate lines are f64 work (evaluation in
the E-step, accumulation in the M-step) whose contribution is below the rounding of the sums they enter."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    K, D, N = int(os.environ.get("CLASSES", 64)), int(os.environ.get("DEGREE", 128)), int(os.environ.get("ROWS", "4000000"))
    spread = float(os.environ.get("SPREAD", "2.0"))
    iters = [int(v) for v in os.environ.get("ITERS", "6,8,10,15,25").split(",")]
    dev = torch.device("cuda", 0)
    x = bench.device_rows(K, D, N, torch.float32, dev, bench.SEED + 1, spread)
    w = bench.Workload(K, D, x, dev, None)
    edges = torch.tensor([0.0, 1.0, 8.0, 16.0, 24.0, 32.0, 40.0, 48.0, 53.0, 64.0, 80.0, 100.0], dtype=torch.float64, device=dev)
    out = {"rows": N, "K": K, "D": D, "spread": spread, "bin_edges_minus_log2_r": edges.tolist(), "iterations": {}}
    it = 1
    for target in iters:
        while it < target:
            w.step()
            it += 1
        hist = torch.zeros(edges.numel() - 1, dtype=torch.float64, device=dev)
        multi = 0
        for lo in range(0, N, 500000):
            r = w.eng.responsibilities(lo, min(500000, N - lo))
            top = r.max(dim=1, keepdim=True).values
            act = (r >= 2.0 ** -100) & (r < top)             # active, not the row's dominant pair
            bits = -torch.log2(r[act])
            hist += torch.stack([((bits >= edges[i]) & (bits < edges[i + 1])).sum().double() for i in range(edges.numel() - 1)])
            multi += int((act.sum(dim=1) > 0).sum())
        a, e = w.eng.sparsity()
        out["iterations"][str(target)] = {"non_dominant_active_pairs_per_row": float(hist.sum()) / N,
                                          "rows_with_more_than_one_active": multi / N,
                                          "active_pairs_per_row_engine": a / N, "evaluated_pairs_per_row": e / N,
                                          "share_by_bin": [round(float(v), 4) for v in (hist / hist.sum().clamp_min(1.0))]}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
