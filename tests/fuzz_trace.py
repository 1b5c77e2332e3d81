#!/usr/bin/env python3
"""[test utility, run by hand on a GPU box] Where do two runs of the same fit part?  python tests/fuzz_trace.py '<case json>' <variant> <iters>
Records, after every K-side step of two identical fits, the statistics block the data pass produced and the next
posterior's fields, and prints the first quantities that differ between the runs."""
import json
import os
import sys
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np                                   # noqa: E402
import torch                                         # noqa: E402
import fuzz_sparse as fz                             # noqa: E402
from oracle import gmm_vb_oracle as orc              # noqa: E402
from bayesml_amd import _kside                       # noqa: E402
from bayesml_amd import gaussianmixture as gm        # noqa: E402

c = json.loads(sys.argv[1])
env = dict(fz.VARIANTS)[sys.argv[2]]
iters = int(sys.argv[3])
x = orc.synth_gmm(c["K_data"], c["D"], c["N"], np.dtype(c["dtype"]), seed=c["seed"], spread=c["spread"],
                  weights_alpha=c.get("weights_alpha"), scale_range=c.get("scale_range"))
FIELDS = [f for f in _kside._POST_FIELDS]
log = []
orig_step = _kside.KStepper.step


def step(self):
    before = self.stats.clone()
    orig_step(self)
    torch.cuda.synchronize()
    rec = {"stats": before.cpu().numpy()}
    for f in FIELDS:
        v = getattr(self.q_next, f, None)
        if isinstance(v, torch.Tensor):
            rec["q_next." + f] = v.detach().cpu().numpy().copy()
    for f in ("ns", "x_bar", "s"):
        rec[f] = getattr(self, f).cpu().numpy().copy()
    log.append(rec)


_kside.KStepper.step = step
runs = []
for _ in range(2):
    log = []
    for k in fz.KEYS:
        os.environ.pop(k, None)
    os.environ.update(env)
    m = gm.LearnModel(c["K"], c["D"], seed=c["seed"], device=torch.device("cuda", 0), verbose=False)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m.update_posterior(x, max_itr=iters, num_init=1, tolerance=0.0)
    m._engine.close()
    runs.append(log)
K, D = c["K"], c["D"]
for i, (a, b) in enumerate(zip(*runs)):
    for k in a:
        d = np.abs(a[k] - b[k])
        bad = ~(d <= 0)            # also NaN
        if bad.any():
            idx = np.argwhere(bad)
            print("step", i, k, "shape", a[k].shape, "differs at", int(bad.sum()), "entries; first", idx[:6].tolist(), "max", float(np.nanmax(d)),
                  "values", a[k][tuple(idx[0])], b[k][tuple(idx[0])])
    if any((~(np.abs(a[k] - b[k]) <= 0)).any() for k in a):
        break
else:
    print("no difference in", len(runs[0]), "steps")
print("stats layout: ns[K] h[K] a[K][D] B[K][D][D]?  len", runs[0][0]["stats"].shape, "K", K, "D", D)
if os.environ.get("FUZZ_TRACE_WINV"):
    # the posterior's W^-1 again on the host from the statistics block (ref:745-770), against both runs' device values
    for i, (a, b) in enumerate(zip(*runs)):
        st = a["stats"]
        ns, aa, B = st[:K], st[2 * K:2 * K + K * D].reshape(K, D), st[2 * K + K * D:2 * K + K * D + K * D * D].reshape(K, D, D)
        for run, r in (("run0", a), ("run1", b)):
            xb, s = r["x_bar"], r["s"]
            kap0 = 1.0
            dev0 = xb - 0.0
            want = np.eye(D)[None] + ns[:, None, None] * s + (kap0 * ns / (kap0 + ns))[:, None, None] * dev0[:, :, None] * dev0[:, None, :]
            d = np.abs(want - r["q_next.w_inv"])
            k, p, q = np.unravel_index(np.argmax(d), d.shape)
            print("step", i, run, "max |w_inv - host| =", d.max(), "at", (k, p, q), "device", r["q_next.w_inv"][k, p, q], "host", want[k, p, q],
                  "s", s[k, p, q], "ns", ns[k])
