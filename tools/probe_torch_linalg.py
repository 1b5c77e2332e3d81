#!/usr/bin/env python3
"""[developer tool, GPU box] Is the framework's batched f64 linear algebra on the GPU reliable?  python tools/probe_torch_linalg.py [n_lo n_hi]
Batched inverse / slogdet / Cholesky / triangular solve of well-conditioned SPD matrices of every order in [n_lo, n_hi],
repeated, against numpy.  (Round 6: torch.linalg.inv and solve_triangular return wrong entries at order 65 with 24 or more
matrices in the batch on this image - bayesml_amd/_kside.py keeps the device path on the library's own factorisation.)"""
import sys

import numpy as np
import torch

dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)
lo, hi = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2, 260)
REPS = 4
for n in range(lo, hi + 1):
    for K in (24, 64):
        a = rng.standard_normal((K, n, n)) * 0.05
        mats = np.eye(n)[None] + a @ a.transpose(0, 2, 1)
        want_inv = np.linalg.inv(mats)
        want_ld = np.linalg.slogdet(mats)[1]
        want_g = np.linalg.cholesky(mats)
        want_gi = np.linalg.inv(want_g)
        t = torch.as_tensor(mats, dtype=torch.float64, device=dev)
        eye = torch.eye(n, dtype=torch.float64, device=dev).expand(K, n, n)
        bad = {"inv": 0, "slogdet": 0, "cholesky_ex": 0, "solve_triangular": 0}
        for _ in range(REPS):
            bad["inv"] += np.abs(torch.linalg.inv(t).cpu().numpy() - want_inv).max() > 1e-9
            bad["slogdet"] += np.abs(torch.linalg.slogdet(t)[1].cpu().numpy() - want_ld).max() > 1e-9
            g = torch.linalg.cholesky_ex(t)[0]
            bad["cholesky_ex"] += np.abs(g.cpu().numpy() - want_g).max() > 1e-9
            gi = torch.linalg.solve_triangular(torch.as_tensor(want_g, device=dev), eye, upper=False)
            bad["solve_triangular"] += np.abs(gi.cpu().numpy() - want_gi).max() > 1e-9
        if any(bad.values()):
            print("n", n, "K", K, f"bad runs of {REPS}:", {k: int(v) for k, v in bad.items() if v}, flush=True)
print("done", lo, hi)
