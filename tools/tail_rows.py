#!/usr/bin/env python3
"""Would sorting the rows of a component group by their margin at the time of the regroup concentrate the rows that stay
multi-component later?  C3 (or ROWS): run to iteration A (default 4: the pass before the regrouping bound pass), take every
row's gap between its best and second-best ln rho (bounds included, as the device has them), run on to iteration B (25),
mark the rows with more than one active component there, and report how many tiles of 256 / waves of 64 rows would be free
of such rows (a) in the present order by dominant component only, (b) with the rows of each component sorted by that gap."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def free_share(flag_sorted, width):
    n = flag_sorted.numel() // width * width
    return float((flag_sorted[:n].view(-1, width).sum(dim=1) == 0).double().mean())


def main():
    K, D, N = int(os.environ.get("CLASSES", 64)), int(os.environ.get("DEGREE", 128)), int(os.environ.get("ROWS", "4000000"))
    A, B = int(os.environ.get("ITER_A", 4)), int(os.environ.get("ITER_B", 25))
    dev = torch.device("cuda", 0)
    x = bench.device_rows(K, D, N, torch.float32, dev, bench.SEED + 1, 2.0)
    w = bench.Workload(K, D, x, dev, None)
    for _ in range(A - 1):
        w.step()
    gaps, dom = [], []
    for lo in range(0, N, 500000):
        L = w.eng.ln_rho(lo, min(500000, N - lo))
        top = torch.topk(L, 2, dim=1)
        gaps.append(top.values[:, 0] - top.values[:, 1])
        dom.append(top.indices[:, 0])
    gap, dom = torch.cat(gaps), torch.cat(dom)
    for _ in range(B - A):
        w.step()
    multi = []
    for lo in range(0, N, 500000):
        r = w.eng.responsibilities(lo, min(500000, N - lo))
        multi.append(((r > 2.0 ** -80).sum(dim=1) > 1))
    multi = torch.cat(multi)
    out = {"rows": N, "multi_share_at_B": float(multi.double().mean()), "gap_quantiles_at_A": [float(v) for v in torch.quantile(gap[::37].double(), torch.tensor([0.05, 0.25, 0.5, 0.75, 0.95], dtype=torch.float64, device=dev))]}
    order_now = torch.argsort(dom, stable=True)
    key = dom.double() * 1e7 - torch.clamp(gap, max=9.9e6)          # by component, then descending gap
    order_gap = torch.argsort(key, stable=True)
    for name, order in (("by_component", order_now), ("by_component_then_gap", order_gap)):
        f = multi[order].to(torch.int32)
        out[name] = {"tiles256_free": free_share(f, 256), "waves64_free": free_share(f, 64)}
    # a threshold on the gap instead of a full sort: rows with gap < G last
    for G in (100.0, 200.0, 400.0):
        tail = (gap < G)
        key2 = dom.double() * 2 + tail.double()
        f = multi[torch.argsort(key2, stable=True)].to(torch.int32)
        out[f"tail_if_gap_below_{int(G)}"] = {"tail_share": float(tail.double().mean()), "tiles256_free": free_share(f, 256),
                                              "waves64_free": free_share(f, 64), "multi_caught": float((multi & tail).double().sum() / multi.double().sum())}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
