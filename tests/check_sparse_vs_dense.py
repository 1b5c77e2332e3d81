#!/usr/bin/env python3
"""[test utility, run by hand on a GPU box: python tests/check_sparse_vs_dense.py]
Pruned E-step / sparse M-step against the dense kernels: same model, same data, a few VB iterations; the
posterior hyper-parameters, the responsibilities and the hard assignments must agree to rounding.
Each configuration runs in a child process (the switches are environment variables read at workspace creation)."""
import json
import os
import subprocess
import sys
import tempfile
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))      # repo root (this file lives in tests/)
sys.path.insert(0, ROOT)

CASES = [(64, 128, 200_000, "float32", 6), (16, 64, 100_000, "float64", 5), (32, 96, 50_000, "float32", 5),
         (8, 128, 30_001, "float32", 4)]


def child(K, D, N, dt, iters, out):
    import torch
    from oracle import gmm_vb_oracle as orc
    from bayesml_amd import gaussianmixture as gm
    x = orc.synth_gmm(K, D, N, np.dtype(dt))
    m = gm.LearnModel(K, D, seed=0, device=torch.device("cuda", 0), verbose=False)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m.update_posterior(x, max_itr=iters, num_init=1, tolerance=0.0)
    hn = m.get_hn_params()
    r = m._engine.responsibilities(0, min(N, 20000)).cpu().numpy()
    z = m._engine.argmax(0, min(N, 20000)).cpu().numpy()
    np.savez(out, r=r, z=z, info=m._engine.launch_info, **hn)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        K, D, N = map(int, sys.argv[2:5])
        child(K, D, N, sys.argv[5], int(sys.argv[6]), sys.argv[7])
        return 0
    bad = 0
    for K, D, N, dt, iters in CASES:
        res = {}
        with tempfile.TemporaryDirectory() as td:
            for tag, env in (("dense", dict(GMMVB_ESTEP_PRUNE="0", GMMVB_MSTEP_SPARSE="0")),
                             ("sparse", dict(GMMVB_ESTEP_PRUNE="force"))):
                out = os.path.join(td, tag + ".npz")
                r = subprocess.run([sys.executable, __file__, "child", str(K), str(D), str(N), dt, str(iters), out],
                                   env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
                if r.returncode != 0:
                    print(tag, "FAILED", r.stderr[-3000:])
                    return 1
                res[tag] = dict(np.load(out))
        a, b = res["dense"], res["sparse"]
        errs = {k: float(np.max(np.abs(a[k] - b[k])) / np.max(np.abs(a[k]))) for k in a if k.startswith("hn_")}
        r_err = float(np.max(np.abs(a["r"] - b["r"])))
        z_same = float(np.mean(a["z"] == b["z"]))
        worst = max(errs.values())
        ok = worst < 1e-11 and r_err < 1e-11 and z_same == 1.0
        bad += not ok
        print(f"K={K} D={D} N={N} {dt} {iters} it: hn max rel diff {worst:.2e}, |dr| {r_err:.2e}, same argmax {z_same:.6f} "
              f"{'OK' if ok else 'MISMATCH'}   [{str(b['info'])[:120]}]", flush=True)
    return bad


if __name__ == "__main__":
    sys.exit(main())
