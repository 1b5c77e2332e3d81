#!/usr/bin/env python3
"""[test utility, run by hand on a GPU box] One hidden-Markov case of tests/fuzz_oracle.py again: the marginals and the
Viterbi path of fresh sequences of several lengths under the fitted model's own posterior, against the oracle.
python tests/fuzz_hmm_readout.py '<case json>'"""
import json
import os
import sys
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GMMVB_DEBUG", "1")
import numpy as np                                   # noqa: E402
import torch                                         # noqa: E402
from oracle import hmm_vb_oracle as orc              # noqa: E402
from bayesml_amd import hiddenmarkovnormal as hm     # noqa: E402

c = json.loads(sys.argv[1])
K, D, N = c["K"], c["D"], c["N"]
x = orc.synth_hmm(max(1, min(K, 8)), D, N, np.dtype(c["dtype"]), seed=c["seed"], stay=c["stay"])[0]
m = hm.LearnModel(K, D, seed=c["seed"], device=torch.device("cuda", 0), verbose=False)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    m.update_posterior(x, max_itr=c["iters"], num_init=c["num_init"], tolerance=0.0, init_type=c["init"])
hn = m.get_hn_params()
own = orc.HmmPosterior(hn["hn_eta_vec"].copy(), hn["hn_zeta_vecs"].copy(), hn["hn_m_vecs"].copy(), hn["hn_kappas"].copy(),
                       hn["hn_nus"].copy(), hn["hn_w_mats"].copy(), np.array(m.hn_w_mats_inv)).refresh()
for n in (2, 3, 64, 65, 300, 1500, 5000):
    xs = orc.synth_hmm(max(1, min(K, 8)), D, n, np.dtype(c["dtype"]), seed=c["seed"] + 1, stay=c["stay"])[0]
    with np.errstate(all="ignore"):
        st = orc.data_pass(xs.astype(np.float64), own)
    got = m.estimate_latent_vars(xs, loss="squared", viterbi=False)
    d = np.abs(got - st.gamma)
    if np.isnan(d).all():
        print("n", n, "the oracle's pass is NaN throughout (every state's exp(ln rho) underflows at some step)")
        continue
    t, k = np.unravel_index(np.nanargmax(d), d.shape)
    print("n", n, "max |gamma - oracle|", float(np.nanmax(d)), "at step", int(t), "state", int(k), "device", got[t, k], "oracle", st.gamma[t, k],
          "row sums", float(got[t].sum()), float(st.gamma[t].sum()), "oracle nan", bool(np.isnan(st.gamma).any()),
          "lowest row maximum of ln rho", float(orc.emission_ln_rho(xs.astype(np.float64), own).max(axis=1).min()), "|", str(m._engine.launch_info)[:80], flush=True)
