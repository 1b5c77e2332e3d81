#!/usr/bin/env python3
"""[test utility, run by hand on a GPU box: python tests/check_i8_estep.py]
int8-digit E-step against the f64 E-step on the same parameters: ln rho differences and kernel times.

The variant is chosen per process (env GMMVB_ESTEP_VARIANT), so each case runs in a child process and hands
its ln rho back through a file."""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))      # repo root (this file lives in tests/)
sys.path.insert(0, ROOT)

CASES = [(3, 2, 1000, "float64"), (16, 32, 4096, "float64"), (5, 33, 777, "float32"), (7, 100, 3000, "float32"),
         (8, 64, 5000, "float32"), (64, 128, 20000, "float32"), (6, 96, 2049, "float64")]


def child(K, D, N, dt, out):
    import torch
    from oracle import gmm_vb_oracle as orc
    from bayesml_amd import _kside
    from bayesml_amd._engine import DataPass
    dev = torch.device("cuda", 0)
    x = orc.synth_gmm(K, D, N, np.dtype(dt))
    p = orc.Prior.default(K, D)
    q = orc.Posterior.from_prior(p)
    orc.init_subsampling(x.astype(np.float64), q, np.random.default_rng(0))
    t = lambda a: torch.as_tensor(a, dtype=torch.float64, device=dev)      # noqa: E731
    qt = _kside.features(_kside.PostT(t(q.alpha), t(q.m), t(q.kappa), t(q.nu), t(q.w_inv)))
    xd = torch.from_numpy(x).to(dev)
    eng = DataPass(K, D, xd.dtype, N, dev)
    eng.set_pivot(xd[:4096].to(torch.float64).mean(dim=0))
    eng.set_params(qt.c, qt.m, qt.u)
    eng.profile(True)
    eng.estep(xd)
    torch.cuda.synchronize()
    ms = eng.last_kernel_ms()[0]
    np.save(out, eng.ln_rho(0, N).cpu().numpy())
    print(json.dumps({"ms": ms, "info": eng.launch_info}))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        K, D, N = map(int, sys.argv[2:5])
        child(K, D, N, sys.argv[5], sys.argv[6])
        return
    worst = 0.0
    for K, D, N, dt in CASES:
        res = {}
        with tempfile.TemporaryDirectory() as td:
            for var in ("lds8", "i8"):
                out = os.path.join(td, var + ".npy")
                env = dict(os.environ, GMMVB_ESTEP_VARIANT=var)
                r = subprocess.run([sys.executable, __file__, "child", str(K), str(D), str(N), dt, out], env=env,
                                   capture_output=True, text=True, timeout=600)
                if r.returncode != 0:
                    print(var, "FAILED", r.stderr[-2000:])
                    return 1
                res[var] = (np.load(out), json.loads(r.stdout.strip().splitlines()[-1]))
        a, b = res["lds8"][0], res["i8"][0]
        near = a > a.max(axis=1, keepdims=True) - 40.0
        err_near = float(np.abs(a - b)[near].max())
        err_rel = float(np.abs(a - b).max() / np.abs(a).max())
        worst = max(worst, err_near)
        print(f"K={K} D={D} N={N} {dt}: |d ln rho| near the row max {err_near:.3e}, overall relative {err_rel:.3e}; "
              f"f64 {res['lds8'][1]['ms']:.3f} ms, i8 {res['i8'][1]['ms']:.3f} ms", flush=True)
    print("worst near-max abs error", worst)
    return 0


if __name__ == "__main__":
    sys.exit(main())
