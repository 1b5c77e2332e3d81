#!/usr/bin/env python3
"""[test utility, run by hand on a GPU box] One case of tests/fuzz_sparse.py again, iteration by iteration: python tests/fuzz_case.py '<case json>'
prints, for max_itr = 1..iters (or, with a second argument <iters>, for that max_itr under each developer switch), the pruned pairs whose reported bound is not an upper bound of the dense ln rho or is not
below the relevance line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np                                   # noqa: E402
import fuzz_sparse as fz                             # noqa: E402
from oracle import gmm_vb_oracle as orc              # noqa: E402

c = json.loads(sys.argv[1])
x = orc.synth_gmm(c["K_data"], c["D"], c["N"], np.dtype(c["dtype"]), seed=c["seed"], spread=c["spread"],
                  weights_alpha=c.get("weights_alpha"), scale_range=c.get("scale_range"))
SWITCHES = [{}, {"GMMVB_GATHER_EXIT": "0"}, {"GMMVB_PROOF": "0"}, {"GMMVB_SWEEP_LAZY": "0"}, {"GMMVB_SORT_ROWS": "0"},
            {"GMMVB_SETTLE_MARGIN": "-1"}, {"GMMVB_MSTEP_CACHE": "0"}, {"GMMVB_PROOF": "settled"}, {"GMMVB_REGROUP_MARGIN": "0"}]
runs = [(i, {}) for i in range(1, c["iters"] + 1)] if len(sys.argv) < 3 else [(int(sys.argv[2]), sw) for sw in SWITCHES]
for iters, sw in runs:
    os.environ.update(sw)
    other = os.environ.get("FUZZ_VARIANT", "forced")            # or "default"
    res = {tag: fz.fit(x, c["K"], iters, env, c["seed"], c.get("num_init", 1), c.get("prior", False), c.get("init", "subsampling")) for tag, env in fz.VARIANTS if tag in ("dense", other)}
    res["forced"] = res[other]
    for k in sw:
        os.environ.pop(k)
    if sw:
        print(sw, end=" ")
    la, lb = res["dense"]["ln_rho"], res["forced"]["ln_rho"]
    same = np.abs(la - lb) <= 1e-6 * np.maximum(1.0, np.abs(la))
    mx = la.max(axis=1, keepdims=True)
    lse = mx + np.log(np.exp(la - mx).sum(axis=1, keepdims=True))
    lo = (~same) & (lb < la - 1e-6 * np.abs(la))
    hi = (~same) & (lb > lse - 55.0)
    hn = {k: fz.rel(res["forced"]["hn"][k], res["dense"]["hn"][k]) for k in res["dense"]["hn"]}
    print("iters", iters, "hn", {k: float(f"{v:.1e}") for k, v in hn.items() if v > 1e-10}, "dr", float(np.max(np.abs(res["forced"]["r"] - res["dense"]["r"]))),
          "dense:", res["dense"]["info"][:40])
    print("iters", iters, "not upper bound:", int(lo.sum()), "above line:", int(hi.sum()), "of", int((~same).sum()), "|",
          res["forced"]["info"][:60], flush=True)
    for n, k in list(zip(*np.nonzero(lo)))[:6] + list(zip(*np.nonzero(hi)))[:6]:
        print("   row", n, "k", k, "dense", la[n, k], "sparse", lb[n, k], "lse", lse[n, 0], "dense gap", lse[n, 0] - la[n, k],
              "r_dense", res["dense"]["r"][n, k], "r_sparse", res["forced"]["r"][n, k])
