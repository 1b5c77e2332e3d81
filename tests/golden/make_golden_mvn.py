#!/usr/bin/env python3
"""Golden fixtures for multivariate_normal.LearnModel, produced by the REFERENCE (build container only):

    MPLBACKEND=Agg python tests/golden/make_golden_mvn.py
"""
import json
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")
os.environ.setdefault("MPLBACKEND", "Agg")

from bayesml import multivariate_normal as ref_mvn      # noqa: E402


def state(m):
    return dict(hn_m_vec=m.hn_m_vec.copy(), hn_kappa=float(m.hn_kappa), hn_nu=float(m.hn_nu),
                hn_w_mat=m.hn_w_mat.copy(), hn_w_mat_inv=m.hn_w_mat_inv.copy())


def case(name, D, n, seed, dtype=np.float64, prior=None, batches=1):
    gen = ref_mvn.GenModel(D, seed=seed)
    gen.gen_params()
    x = gen.gen_sample(n).astype(dtype)
    m = ref_mvn.LearnModel(D, **(prior or {}))
    out = dict(D=D, N=n, x=x, mu_vec=gen.mu_vec.copy(), lambda_mat=gen.lambda_mat.copy(), batches=batches,
               prior=json.dumps({k: (v.tolist() if hasattr(v, "tolist") else v) for k, v in (prior or {}).items()}))
    for i, part in enumerate(np.array_split(x.astype(np.float64), batches)):
        m.update_posterior(part)                 # the reference is fed the float64 widening of the same values
        out.update({f"b{i}_{k}": v for k, v in state(m).items()})
    out.update(state(m))
    mu, lam = m.estimate_params("squared")
    out.update(est_sq_mu=mu.copy(), est_sq_lambda=lam.copy())
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        mu01, lam01 = m.estimate_params("0-1")
    out.update(est_01_lambda=np.full((D, D), np.nan) if lam01 is None else lam01)
    m.calc_pred_dist()
    out.update(p_m_vec=m.p_m_vec.copy(), p_nu=float(m.p_nu), p_v_mat=m.p_v_mat.copy(), p_v_mat_inv=m.p_v_mat_inv.copy())
    # sequential prediction of two further points
    nxt = gen.gen_sample(2)
    preds = [m.pred_and_update(nxt[0]).copy(), m.pred_and_update(nxt[1], loss="0-1").copy()]
    out.update(next_x=nxt, preds=np.array(preds), **{"after_" + k: v for k, v in state(m).items()})
    np.savez_compressed(os.path.join(HERE, name), **out)
    print("wrote", name, "kappa", m.hn_kappa)


def errors():
    cases = {
        "ctor_float_degree": lambda: ref_mvn.LearnModel(2.0),
        "h0_m_vec_wrong_dim": lambda: ref_mvn.LearnModel(2, h0_m_vec=np.zeros(3)),
        "h0_kappa_nonpos": lambda: ref_mvn.LearnModel(2, h0_kappa=0.0),
        "h0_nu_too_small": lambda: ref_mvn.LearnModel(3, h0_nu=2.0),
        "h0_w_mat_not_pd": lambda: ref_mvn.LearnModel(2, h0_w_mat=np.array([[1.0, 2.0], [2.0, 1.0]])),
        "h0_w_mat_wrong_dim": lambda: ref_mvn.LearnModel(2, h0_w_mat=np.eye(3)),
        "x_wrong_last_dim": lambda: ref_mvn.LearnModel(2).update_posterior(np.zeros((5, 3))),
        "x_not_ndarray": lambda: ref_mvn.LearnModel(2).update_posterior([[0.0, 1.0]]),
        "bad_loss_estimate": lambda: ref_mvn.LearnModel(2).estimate_params("L1"),
        "bad_loss_prediction": lambda: ref_mvn.LearnModel(2).make_prediction("L1"),
        "pred_and_update_wrong_shape": lambda: ref_mvn.LearnModel(2).pred_and_update(np.zeros((1, 2))),
        "gen_sample_float": lambda: ref_mvn.GenModel(2).gen_sample(3.0),
        "x_int_ok": lambda: ref_mvn.LearnModel(2).update_posterior(np.arange(10).reshape(5, 2)),
        "x_3d_ok": lambda: ref_mvn.LearnModel(2).update_posterior(np.zeros((3, 4, 2))),
    }
    res = {}
    for name, fn in cases.items():
        try:
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                fn()
            res[name] = None
        except Exception as e:      # noqa: BLE001
            res[name] = type(e).__name__
    with open(os.path.join(HERE, "mvn_errors.json"), "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)
    print(res)


if __name__ == "__main__":
    case("mvn_d2_n100.npz", 2, 100, seed=1)
    case("mvn_d5_n1.npz", 5, 1, seed=2)
    case("mvn_d32_n5000_f32_batches3.npz", 32, 5000, seed=3, dtype=np.float32, batches=3,
         prior=dict(h0_m_vec=np.full(32, 0.5), h0_kappa=2.0, h0_nu=40.0, h0_w_mat=np.eye(32) * 0.5))
    case("mvn_d128_n3000_f32.npz", 128, 3000, seed=4, dtype=np.float32)
    errors()
