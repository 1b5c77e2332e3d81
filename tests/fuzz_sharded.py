#!/usr/bin/env python3
"""[test utility, run by hand on a GPU box] Random row-sharded fits: python tests/fuzz_sharded.py [--cases 16] [--seed 1] [--world 3]

`world` processes share cuda:0 and exchange the statistics block over gloo (RCCL wants a device per rank; the collective
is the same torch.distributed.all_reduce call bench.py issues over "nccl").  Every case draws a shape, a data recipe,
uneven cuts (a rank may hold a handful of rows), the pass policy (default or forced pruning), restarts and optionally
row tiles inside the shards; all ranks fit it with `RowShard`, then one process fits the whole matrix, and the posteriors
are compared (the sums are taken in another order: rounding, amplified by the fit).  Flagged above 1e-8 relative (1e-5 for
samples with fewer rows than eight times c_degree), when the ranks' posteriors are not bit-identical, or
when the ranks did not run the same kernels in the last pass."""
import argparse
import json
import os
import re
import socket
import sys
import tempfile
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GMMVB_DEBUG", "1")
KEYS = ("GMMVB_ESTEP_PRUNE", "BAYESML_AMD_TILE_ROWS", "BAYESML_AMD_TILE_RESIDENT")


def draw_cases(n, seed, world):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        K = int(rng.choice([2, 5, 16, 33, 64, 100, 256]))
        D = int(rng.choice([8, 16, 33, 49, 64, 65, 100, 128, 160]))
        N = int(rng.choice([300, 2049, 10_000, 40_001, 120_000]))
        if K * N > 6e6:
            N = int(6e6 // K)
        cuts = sorted(int(c) for c in rng.integers(1, N, world - 1))
        if rng.random() < 0.2:
            cuts[0] = min(cuts[0], int(rng.integers(1, 70)))          # a rank with a handful of rows
        c = dict(K=K, D=D, N=N, dtype=str(rng.choice(["float32", "float64"])), iters=int(rng.integers(2, 12)),
                 num_init=int(rng.choice([1, 1, 2])), K_data=int(max(1, min(K, rng.choice([K, max(1, K // 2), 3])))),
                 spread=float(rng.choice([2.0, 1.0, 0.5])), seed=int(rng.integers(0, 1000)), cuts=[0] + cuts + [N],
                 prune=str(rng.choice(["default", "force"])))
        if rng.random() < 0.3 and N > 3000:
            c["tile_rows"] = int(rng.choice([512, 1000, 4096]))
            c["tile_resident"] = int(rng.integers(0, 2))
        out.append(c)
    return out


def set_env(c):
    for k in KEYS:
        os.environ.pop(k, None)
    if c["prune"] == "force":
        os.environ["GMMVB_ESTEP_PRUNE"] = "force"
    if "tile_rows" in c:
        os.environ["BAYESML_AMD_TILE_ROWS"] = str(c["tile_rows"])
        os.environ["BAYESML_AMD_TILE_RESIDENT"] = str(c["tile_resident"])


def data(c):
    from oracle import gmm_vb_oracle as orc
    return orc.synth_gmm(c["K_data"], c["D"], c["N"], np.dtype(c["dtype"]), seed=c["seed"], spread=c["spread"])


def fit(c, x, comm=None):
    import torch
    from bayesml_amd import gaussianmixture as gm
    kw = dict(comm=comm) if comm is not None else {}
    m = gm.LearnModel(c["K"], c["D"], seed=c["seed"], device=torch.device("cuda", 0), verbose=False, **kw)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m.update_posterior(x, max_itr=c["iters"], num_init=c["num_init"], tolerance=0.0)
    out = dict(vl=float(m.vl), ns=np.array(m.ns), info=str(m._engine.launch_info), **{k: np.array(v) for k, v in m.get_hn_params().items()})
    m._engine.close()
    return out


def worker(rank, world, port, cases, out_dir):
    import torch
    import torch.distributed as dist
    from bayesml_amd import RowShard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    for i, c in enumerate(cases):
        set_env(c)
        x = data(c)
        try:
            res = fit(c, x[c["cuts"][rank]:c["cuts"][rank + 1]], RowShard())
            info = res.pop("info")
            np.savez(os.path.join(out_dir, f"case{i}_rank{rank}.npz"), info=info, **res)
        except Exception as e:                                         # noqa: BLE001  (the case is the finding; keep the ranks in step)
            np.savez(os.path.join(out_dir, f"case{i}_rank{rank}.npz"), error=repr(e)[:400])
            raise
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=16)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--world", type=int, default=3)
    a = ap.parse_args()
    import torch.multiprocessing as mp
    cases = draw_cases(a.cases, a.seed, a.world)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    t0 = time.time()
    flagged = 0
    with tempfile.TemporaryDirectory() as td:
        try:
            mp.spawn(worker, args=(a.world, port, cases, td), nprocs=a.world, join=True)
        except Exception as e:                                         # noqa: BLE001
            print(json.dumps(dict(spawn_error=repr(e)[:600])), flush=True)
        for i, c in enumerate(cases):
            files = [os.path.join(td, f"case{i}_rank{r}.npz") for r in range(a.world)]
            if not all(os.path.exists(f) for f in files):
                print(json.dumps(dict(case=c, error="no result from the ranks")), flush=True)
                flagged += 1
                continue
            ranks = [dict(np.load(f)) for f in files]
            if any("error" in r for r in ranks):
                print(json.dumps(dict(case=c, error=[str(r.get("error")) for r in ranks])), flush=True)
                flagged += 1
                continue
            set_env(c)
            one = fit(c, data(c))
            keys = [k for k in one if k.startswith("hn_")] + ["ns"]
            d = max(float(np.max(np.abs(r[k] - one[k])) / max(1e-300, float(np.max(np.abs(one[k]))))) for r in ranks for k in keys)
            dvl = max(abs(float(r["vl"]) - one["vl"]) / max(1.0, abs(one["vl"])) for r in ranks)
            same_between_ranks = all(np.array_equal(ranks[0][k], r[k]) for r in ranks[1:] for k in keys)
            estep = [(re.search(r"estep_\w+", str(r["info"])) or [""])[0] for r in ranks]       # (the last pass's E-step kernel)
            # (a few rows per component and feature: the other summation order's rounding is amplified by ill-conditioned
            # scatter matrices - measured 1e-8 .. 2e-6 at N = 300 with K x D = 3300 .. 6400)
            tol = 1e-8 if c["N"] >= 8 * c["D"] else 1e-5
            bad = not (d < tol and dvl < 0.1 * tol) or not same_between_ranks or len(set(estep)) > 1
            flagged += bad
            print(json.dumps(dict(case=c, diff=float(f"{d:.1e}"), dvl=float(f"{dvl:.1e}"), ranks_identical=bool(same_between_ranks),
                                  estep=estep, single=one["info"][:40], flag=bool(bad))), flush=True)
    print(json.dumps(dict(cases=len(cases), flagged=int(flagged), seconds=round(time.time() - t0, 1))), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
