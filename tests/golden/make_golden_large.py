#!/usr/bin/env python3
"""Reference-generated driver fixtures at the shapes where the engine's DEFAULT policy takes the sparse path
(N K >= 2^23: int8 bound pass, carried bounds, candidate gathers, list M-step), and one on heavily overlapping
clusters (the policy's fall-backs).  Run in the build container only (imports /root/reference):

    MPLBACKEND=Agg python tests/golden/make_golden_large.py

x is not stored (77 MB): it is the seeded recipe ``oracle.synth_gmm`` (checksum stored).  The K D D matrices are
stored as functionals (two components in full, every diagonal, three fixed projections, log-determinants) so the
fixtures stay small; tests/test_gpu_sparse_parity.py applies the same functionals to the engine's output.
"""
import json
import os
import sys
import time
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import ROOT, full_driver_state, sha  # noqa: E402,F401
from oracle.gmm_vb_oracle import synth_gmm            # noqa: E402


def compact_mats(name, a, out):
    """[K, D, D] -> the stored functionals (see tests/conftest.py::mat_functionals)."""
    K, D, _ = a.shape
    v = np.random.default_rng(7).standard_normal((D, 3))
    out[name + "_head"] = a[:2].copy()
    out[name + "_diag"] = np.diagonal(a, axis1=1, axis2=2).copy()
    out[name + "_proj"] = a @ v
    out[name + "_logabsdet"] = np.linalg.slogdet(a)[1]


def run(name, K, D, N, spread=2.0, K_data=None, weights_alpha=None, scale_range=None, **kw):
    """K: the MODEL's number of components; K_data (default K): the mixture the rows are drawn from."""
    K_data = K if K_data is None else K_data
    x = synth_gmm(K_data, D, N, np.float32, spread=spread, weights_alpha=weights_alpha, scale_range=scale_range)
    t0 = time.time()
    st = full_driver_state(K, D, x, seed=0, **kw)
    st.pop("_model")
    out = {k: v for k, v in st.items() if k not in ("hn_w_mats", "hn_w_mats_inv", "s_mats", "e_lambda_mats")}
    for key in ("hn_w_mats", "hn_w_mats_inv", "s_mats"):
        compact_mats(key, st[key], out)
    out.update(K_data=K_data, spread=spread)
    if weights_alpha is not None:
        out["weights_alpha"] = weights_alpha
    if scale_range is not None:
        out["scale_range"] = np.array(scale_range, dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, name), **out)
    print("wrote", name, "winner", st["winner"], "vl", st["final_vl"], "%.0f s" % (time.time() - t0), flush=True)


def main():
    which = sys.argv[1:] or ["overlap", "mid", "k64", "k256", "offpath"]
    if "overlap" in which:
        run("gmm_f3_k16_d64_n32768_f32_overlap.npz", 16, 64, 32768, spread=0.3, num_init=1, max_itr=12, tolerance=0.0)
    if "mid" in which:       # the middle of the separation spectrum: 2-10 components active per row for most of the run
        run("gmm_f3_k16_d64_n32768_f32_spread1.npz", 16, 64, 32768, spread=1.0, num_init=1, max_itr=12, tolerance=0.0)
    if "offpath" in which:
        # off the benchmark's recipe (round 5): more components in the model than in the data (empty and duplicate
        # components: an empty one's c_k sits ~170 nats above the others), unequal mixing weights, anisotropic clusters
        run("gmm_f3_k16_d64_n32768_f32_kdata8.npz", 16, 64, 32768, K_data=8, num_init=1, max_itr=12, tolerance=0.0)
        run("gmm_f3_k16_d64_n32768_f32_weights.npz", 16, 64, 32768, weights_alpha=0.3, num_init=1, max_itr=12, tolerance=0.0)
        run("gmm_f3_k16_d64_n32768_f32_aniso.npz", 16, 64, 32768, scale_range=(0.3, 3.0), num_init=1, max_itr=12, tolerance=0.0)
    if "k64" in which:
        run("gmm_f3_k64_d128_n140000_f32.npz", 64, 128, 140000, num_init=1, max_itr=12, tolerance=0.0)
    if "k256" in which:
        run("gmm_f3_k256_d64_n36000_f32.npz", 256, 64, 36000, num_init=1, max_itr=10, tolerance=0.0)


if __name__ == "__main__":
    main()
