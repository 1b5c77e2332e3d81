#!/usr/bin/env python3
"""[test utility, run by hand on a GPU box] Random-shape stress of the pruned data pass: python tests/fuzz_sparse.py [--cases 60] [--seed 1]

Every case draws a shape (K, D, N, dtype), a data recipe (cluster spread, unequal weights, anisotropic scales, fewer true
clusters than components) and an iteration count, fits it three times through the public driver - dense kernels only,
pruning forced, the default policy - and compares posterior hyper-parameters, responsibilities and hard assignments.
The three runs follow the same trajectory up to rounding, and a fit amplifies rounding differences (overlapping clusters
most), so the line printed per case carries the differences themselves; a case is flagged when they exceed what
rounding explains (1e-8 relative on the posterior, 1e-7 on a responsibility) or when a pruned pair's bound is not one.
The oracle is used as the data generator only."""
import argparse
import json
import os
import sys
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GMMVB_DEBUG", "1")

KEYS = ("GMMVB_ESTEP_PRUNE", "GMMVB_MSTEP_SPARSE", "BAYESML_AMD_TILE_ROWS", "BAYESML_AMD_TILE_RESIDENT")
VARIANTS = (("dense", dict(GMMVB_ESTEP_PRUNE="0", GMMVB_MSTEP_SPARSE="0")), ("forced", dict(GMMVB_ESTEP_PRUNE="force")),
            ("default", {}))


def random_prior(K, D, seed):
    """Non-default hyper-parameters (h0_alpha_vec, h0_m_vecs, h0_kappas, h0_nus, h0_w_mats) for a case."""
    rng = np.random.default_rng(seed + 77)
    a = rng.standard_normal((K, D, D)) * (0.3 / np.sqrt(D))
    w = np.eye(D)[None] * rng.uniform(0.5, 2.0, (K, 1, 1)) + a @ a.transpose(0, 2, 1)
    return dict(h0_alpha_vec=rng.uniform(0.2, 3.0, K), h0_m_vecs=rng.standard_normal((K, D)) * 0.5, h0_kappas=rng.uniform(0.2, 3.0, K),
                h0_nus=D - 1 + rng.uniform(0.1, 4.0, K), h0_w_mats=0.5 * (w + w.transpose(0, 2, 1)))


def fit(x, K, iters, env, seed, num_init=1, prior=False, init="subsampling"):
    import torch
    from bayesml_amd import gaussianmixture as gm
    old = {k: os.environ.pop(k, None) for k in KEYS}
    os.environ.update(env)
    try:
        m = gm.LearnModel(K, x.shape[1], seed=seed, device=torch.device("cuda", 0), verbose=False,
                          **(random_prior(K, x.shape[1], seed) if prior else {}))
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            m.update_posterior(x, max_itr=iters, num_init=num_init, tolerance=0.0, init_type=init)
    finally:
        for k in KEYS:
            os.environ.pop(k, None)
            if old[k] is not None:
                os.environ[k] = old[k]
    n = min(x.shape[0], 50_000)
    out = dict(hn=m.get_hn_params(), r=m._engine.responsibilities(0, n).cpu().numpy(), z=m._engine.argmax(0, n).cpu().numpy(),
               ln_rho=m._engine.ln_rho(0, min(n, 4000)).cpu().numpy(), info=str(m._engine.launch_info), vl=float(m.vl))
    m._engine.close()
    return out


def draw_case(rng):
    K = int(rng.choice([2, 3, 5, 8, 16, 24, 33, 64, 65, 100, 128, 200, 256]))
    D = int(rng.choice([49, 50, 63, 64, 65, 80, 96, 100, 112, 127, 128]))
    N = int(rng.choice([70, 513, 2049, 4097, 10_000, 30_001, 65_536, 120_000]))
    if K * N > 6e6:
        N = int(6e6 // K)
    case = dict(K=K, D=D, N=N, dtype=str(rng.choice(["float32", "float64"])), iters=int(rng.integers(3, 15)),
                K_data=int(max(1, min(K, rng.choice([K, K, max(1, K // 2), max(1, K // 4), 3])))),
                spread=float(rng.choice([2.0, 2.0, 1.0, 0.6, 0.3])), seed=int(rng.integers(0, 1000)))
    if rng.random() < 0.25:
        case["weights_alpha"] = 0.3
    if rng.random() < 0.25:
        case["scale_range"] = (0.3, 3.0)
    if rng.random() < 0.3:
        case["num_init"] = 2
    if rng.random() < 0.3:
        case["prior"] = True
    if rng.random() < 0.2:
        case["init"] = "random_responsibility"
    if rng.random() < 0.35 and case["N"] > 600:
        case["tile_rows"] = int(rng.choice([256, 320, 1000, 4096, 10_000, case["N"] // 2 + 1, case["N"] - 1]))
        case["tile_resident"] = int(rng.integers(0, 2))
    return case


def rel(a, b):
    return float(np.max(np.abs(np.asarray(a, dtype=np.float64) - b)) / max(1e-300, float(np.max(np.abs(b)))))


def run(cases, seed, seconds=1e9, emit=print, max_pairs=6e6, scale=1):
    """`cases` random cases from `seed`; returns (cases run, the flagged lines).  tests/test_gpu_fuzz.py runs a few."""
    from oracle import gmm_vb_oracle as orc
    rng = np.random.default_rng(seed)
    t0, flagged, i = time.time(), [], -1
    for i in range(cases):
        if time.time() - t0 > seconds:
            break
        c = draw_case(rng)
        c["N"] *= scale                     # (--scale: the same draws with more rows, for the many-tile machinery)
        if c["K"] * c["N"] > max_pairs:
            c["N"] = int(max_pairs // c["K"])
        x = orc.synth_gmm(c["K_data"], c["D"], c["N"], np.dtype(c["dtype"]), seed=c["seed"], spread=c["spread"],
                          weights_alpha=c.get("weights_alpha"), scale_range=c.get("scale_range"))
        variants = list(VARIANTS)
        if "tile_rows" in c:      # the same fit through row tiles (resident workspaces per tile, or one workspace for all)
            variants.append(("tiled", dict(BAYESML_AMD_TILE_ROWS=str(c["tile_rows"]), BAYESML_AMD_TILE_RESIDENT=str(c["tile_resident"]))))
        try:
            res = {tag: fit(x, c["K"], c["iters"], env, c["seed"], c.get("num_init", 1), c.get("prior", False), c.get("init", "subsampling"))
                   for tag, env in variants}
        except Exception as e:                                         # noqa: BLE001  (the case is the finding)
            flagged.append(dict(case=c, error=repr(e)[:400]))
            emit(json.dumps(flagged[-1]))
            continue
        d = res["dense"]
        line = dict(case=c)
        bad = False
        for tag in [t for t in ("forced", "default", "tiled") if t in res]:
            s = res[tag]
            hn = max(rel(s["hn"][k], d["hn"][k]) for k in d["hn"])
            dr = float(np.max(np.abs(s["r"] - d["r"])))
            same_z = float(np.mean(s["z"] == d["z"]))
            la, lb = d["ln_rho"], s["ln_rho"]
            same = np.abs(la - lb) <= 1e-6 * np.maximum(1.0, np.abs(la))
            mx = la.max(axis=1, keepdims=True)
            lse = mx + np.log(np.exp(la - mx).sum(axis=1, keepdims=True))
            # a pair that differs was pruned: an upper bound of ln rho, and below the relevance line
            bound_ok = bool(np.all(lb[~same] >= la[~same] - 1e-6 * np.abs(la[~same])) and np.all((lb <= lse - 55.0) | same))
            line[tag] = dict(hn=float(f"{hn:.2e}"), dr=float(f"{dr:.2e}"), same_z=same_z, bound_ok=bound_ok,
                             pruned=float(f"{1.0 - same.mean():.3f}"), vl=float(f"{abs(s['vl'] - d['vl']) / abs(d['vl']):.1e}"),
                             info=s["info"][:90])
            bad |= hn > 1e-8 or dr > 1e-7 or not bound_ok or not np.isfinite(hn)
        line["flag"] = bool(bad)
        if bad:
            flagged.append(line)
        emit(json.dumps(line))
    emit(json.dumps(dict(cases=i + 1, flagged=len(flagged), seconds=round(time.time() - t0, 1))))
    return i + 1, flagged


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=60)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--seconds", type=float, default=1e9, help="stop drawing new cases after this long")
    ap.add_argument("--scale", type=int, default=1, help="multiply every case's row count")
    ap.add_argument("--max-pairs", type=float, default=6e6, help="cap on rows x components of a case")
    a = ap.parse_args()
    run(a.cases, a.seed, a.seconds, emit=lambda t: print(t, flush=True), max_pairs=a.max_pairs, scale=a.scale)
    return 0


if __name__ == "__main__":
    sys.exit(main())
