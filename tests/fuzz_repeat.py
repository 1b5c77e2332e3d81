#!/usr/bin/env python3
"""[test utility, run by hand on a GPU box] Run one fuzz case's variant several times and compare the runs with each other (determinism):
python tests/fuzz_repeat.py '<case json>' <variant> <iters> [repeats]"""
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
env = dict(fz.VARIANTS)[sys.argv[2]]
iters = int(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 4
x = orc.synth_gmm(c["K_data"], c["D"], c["N"], np.dtype(c["dtype"]), seed=c["seed"], spread=c["spread"],
                  weights_alpha=c.get("weights_alpha"), scale_range=c.get("scale_range"))
runs = [fz.fit(x, c["K"], iters, env, c["seed"], c.get("num_init", 1), c.get("prior", False), c.get("init", "subsampling")) for _ in range(reps)]
a = runs[0]
for i, b in enumerate(runs[1:], 1):
    print("run", i, "vs 0:", {k: float(f"{fz.rel(b['hn'][k], a['hn'][k]):.1e}") for k in a["hn"]},
          "dr", float(np.max(np.abs(a["r"] - b["r"]))), "dlnrho", float(np.nanmax(np.abs(a["ln_rho"] - b["ln_rho"]))), "vl", a["vl"], b["vl"])
print(a["info"])
print("hn_alpha", a["hn"]["hn_alpha_vec"][:8] if "hn_alpha_vec" in a["hn"] else list(a["hn"])[:8])
