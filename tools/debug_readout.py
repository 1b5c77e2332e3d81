import os, sys, warnings
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from oracle import gmm_vb_oracle as orc
from test_gpu_sparse import _fit, DENSE, SPARSE
K, K_data, D, N, iters = 24, 6, 128, 20001, 6
x = orc.synth_gmm(K_data, D, N, np.float32)
a = _fit(x, K, iters, DENSE); b = _fit(x, K, iters, SPARSE)
la = a._engine.ln_rho().cpu().numpy(); lb = b._engine.ln_rho().cpu().numpy()
ra = a._engine.responsibilities().cpu().numpy(); rb = b._engine.responsibilities().cpu().numpy()
same = np.abs(la - lb) <= 1e-9 * np.maximum(1.0, np.abs(la))
best = la.max(axis=1, keepdims=True)
bad = ~((lb <= best - 69.0) | same)
print("bad", bad.sum(), "of", bad.size, "rows", np.unique(np.nonzero(bad)[0]).size, b._engine.launch_info, b._engine.pass_counts())
rows, ks = np.nonzero(bad)
for r, k in list(zip(rows, ks))[:12]:
    print(r, k, "la", la[r, k], "lb", lb[r, k], "best", best[r, 0], "ra", ra[r, k], "rb", rb[r, k], "argbest", la[r].argmax(),
          "n within 69:", int((la[r] > best[r, 0] - 69).sum()))
print("max |ra-rb|", np.abs(ra - rb).max())
