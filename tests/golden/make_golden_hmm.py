#!/usr/bin/env python3
"""Golden fixtures for the HMM path (family F6), produced by the REFERENCE.

Build container only (imports /root/reference):  MPLBACKEND=Agg python tests/golden/make_golden_hmm.py
"""
import contextlib
import io
import json
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
sys.path.insert(0, "/root/reference")
os.environ.setdefault("MPLBACKEND", "Agg")

from bayesml import hiddenmarkovnormal as ref_hmm       # noqa: E402  (the reference)
from bayesml._exceptions import ResultWarning           # noqa: E402
from make_golden import parse_trace, quiet, sha         # noqa: E402
from oracle.hmm_vb_oracle import synth_hmm              # noqa: E402


def hn_state(m):
    return dict(hn_eta_vec=m.hn_eta_vec.copy(), hn_zeta_vecs=m.hn_zeta_vecs.copy(), hn_m_vecs=m.hn_m_vecs.copy(),
                hn_kappas=m.hn_kappas.copy(), hn_nus=m.hn_nus.copy(), hn_w_mats=m.hn_w_mats.copy(),
                hn_w_mats_inv=m.hn_w_mats_inv.copy())


def vl_terms(m):
    return dict(vl=m.vl, vl_p_x=m._vl_p_x, vl_p_z=m._vl_p_z, vl_p_pi=m._vl_p_pi, vl_p_a=m._vl_p_a,
                vl_p_mu_lambda=m._vl_p_mu_lambda, vl_q_z=m._vl_q_z, vl_q_pi=m._vl_q_pi, vl_q_a=m._vl_q_a,
                vl_q_mu_lambda=m._vl_q_mu_lambda)


def alloc(m, T):
    K = m.c_num_classes
    m._length = T
    m._ln_rho = np.zeros([T, K])
    m._rho = np.ones([T, K])
    m.alpha_vecs = np.ones([T, K]) / K
    m.beta_vecs = np.ones([T, K])
    m.gamma_vecs = np.ones([T, K]) / K
    m.xi_mats = np.zeros([T, K, K])
    m._cs = np.ones([T])


def single_step(name, K, D, x, seed, store_x, keep=None, warm=2):
    m = ref_hmm.LearnModel(K, D, seed=seed)
    T = x.shape[0]
    alloc(m, T)
    m.s_mats[:] = 0.0
    m._init_fb_params()
    m.reset_hn_params()
    m._init_subsampling(x)
    for _ in range(warm):
        m._update_q_z(x)
        m._update_q_mu_lambda()
        m._update_q_pi()
        m._update_q_a()
    out = {"in_" + k: v for k, v in hn_state(m).items()}
    m._update_q_z(x)
    m._calc_vl()
    sl = slice(None) if keep is None else slice(0, keep)
    out.update(ln_rho=m._ln_rho[sl].copy(), alpha_vecs=m.alpha_vecs[sl].copy(), beta_vecs=m.beta_vecs[sl].copy(),
               gamma_vecs=m.gamma_vecs[sl].copy(), cs=m._cs[sl].copy(), ln_cs_sum=float(np.log(m._cs).sum()),
               gamma_last=m.gamma_vecs[-1].copy(), ns=m.ns.copy(), ms=m.ms.copy(), x_bar_vecs=m.x_bar_vecs.copy(),
               s_mats=m.s_mats.copy(), rho_min=float(m._rho.min()), cs_min=float(m._cs.min()))
    out.update(vl_terms(m))
    m._update_q_mu_lambda()
    m._update_q_pi()
    m._update_q_a()
    out.update({"out_" + k: v for k, v in hn_state(m).items()})
    out.update(out_ln_pi_tilde=m._ln_pi_tilde_vec.copy(), out_ln_a_tilde=m._ln_a_tilde_mat.copy(),
               out_a_tilde=m._a_tilde_mat.copy(), out_pi_tilde=m._pi_tilde_vec.copy(),
               out_e_ln_lambda_dets=m._e_ln_lambda_dets.copy())
    out.update(K=K, D=D, N=T, seed=seed, x_sha256=sha(x), x_dtype=str(x.dtype), warm_iters=warm)
    if store_x:
        out["x"] = x
    np.savez_compressed(os.path.join(HERE, name), **out)
    print("wrote", name, "rho_min", out["rho_min"], "cs_min", out["cs_min"], "vl", m.vl)


def full_driver(name, K, D, x, seed, store_x, viterbi_rows=0, **kw):
    m = ref_hmm.LearnModel(K, D, seed=seed)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        _, text = quiet(m.update_posterior, x.astype(np.float64), **kw)
    warned = any(issubclass(i.category, ResultWarning) for i in w)
    traces, winners, converged = parse_trace(text)
    winner = max(i for i, s in enumerate(winners) if s)
    L = max(len(t) for t in traces)
    tr = np.full((len(traces), L), np.nan)
    for i, t in enumerate(traces):
        tr[i, :len(t)] = t
    out = dict(K=K, D=D, N=x.shape[0], seed=seed, x_sha256=sha(x), x_dtype=str(x.dtype), vl_trace=tr, winner=winner,
               converged=np.array(converged), result_warning=warned, final_vl=m.vl, ns=m.ns.copy(), ms=m.ms.copy(),
               x_bar_vecs=m.x_bar_vecs.copy(), s_mats=m.s_mats.copy(), gamma_head=m.gamma_vecs[:64].copy(),
               gamma_last=m.gamma_vecs[-1].copy(), kw=json.dumps(kw))
    out.update(hn_state(m))
    pi, a, mu, lam = m.estimate_params("squared")
    out.update(est_sq_pi=pi, est_sq_a=a, est_sq_lambda=np.array(lam))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        pi01, a01, _, lam01 = m.estimate_params("0-1")
    out.update(est_01_pi=pi01, est_01_a=a01, est_01_lambda=lam01)
    out.update({"stale_" + k: np.array(v) for k, v in m.get_p_params().items()})
    m.calc_pred_dist()
    out.update({k: np.array(v) for k, v in m.get_p_params().items()})
    out.update(pred_squared=m.make_prediction("squared"), pred_01=m.make_prediction("0-1"))
    if viterbi_rows:
        xs = x[:viterbi_rows].astype(np.float64)
        out.update(viterbi_01=m.estimate_latent_vars(xs, "0-1", viterbi=True),
                   marginal_01=m.estimate_latent_vars(xs, "0-1", viterbi=False),
                   marginal_sq=m.estimate_latent_vars(xs, "squared", viterbi=False).copy())
    if store_x:
        out["x"] = x
    np.savez_compressed(os.path.join(HERE, name), **out)
    print("wrote", name, "winner", winner, "vl", m.vl, "warned", warned, "iters", [len(t) for t in traces])


def boundary_errors():
    cases = {
        "ctor_positional_h0": lambda: ref_hmm.LearnModel(3, 2, np.ones(3)),
        "ctor_float_degree": lambda: ref_hmm.LearnModel(3, 2.0),
        "h0_zeta_nonpos": lambda: ref_hmm.LearnModel(3, 2, h0_zeta_vecs=np.zeros((3, 3))),
        "h0_nus_all_small": lambda: ref_hmm.LearnModel(3, 2, h0_nus=np.array([1.0, 1.0, 1.0])),
        "h0_nus_some_small": lambda: ref_hmm.LearnModel(3, 2, h0_nus=np.array([1.0, 3.0, 3.0])),
        "h0_w_not_pd": lambda: ref_hmm.LearnModel(3, 2, h0_w_mats=np.array([[1.0, 2.0], [2.0, 1.0]])),
        "h0_m_wrong_dim": lambda: ref_hmm.LearnModel(3, 2, h0_m_vecs=np.zeros((3, 3))),
        "x_wrong_last_dim": lambda: quiet(ref_hmm.LearnModel(3, 2).update_posterior, np.zeros((10, 3))),
        "x_not_ndarray": lambda: quiet(ref_hmm.LearnModel(3, 2).update_posterior, [[0.0, 1.0]]),
        "bad_init_type": lambda: quiet(ref_hmm.LearnModel(3, 2, seed=0).update_posterior,
                                       np.random.default_rng(0).standard_normal((50, 2)), init_type="kmeans"),
        "bad_loss_estimate_params": lambda: ref_hmm.LearnModel(3, 2).estimate_params("L1"),
        "viterbi_bad_loss": lambda: ref_hmm.LearnModel(3, 2).estimate_latent_vars(np.zeros((4, 2)), "squared", viterbi=True),
        "marginal_bad_loss": lambda: ref_hmm.LearnModel(3, 2).estimate_latent_vars(np.zeros((4, 2)), "L1", viterbi=False),
        "gen_a_not_sum1": lambda: ref_hmm.GenModel(2, 1, a_mat=np.array([[0.5, 0.4], [0.5, 0.5]])),
        "gen_positional": lambda: ref_hmm.GenModel(2, 1, np.array([0.5, 0.5])),
        "scalar_broadcast_ok": lambda: ref_hmm.LearnModel(3, 2, h0_eta_vec=2.0, h0_zeta_vecs=1.5, h0_kappas=2.0, h0_nus=3.0),
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
    with open(os.path.join(HERE, "hmm_errors.json"), "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)
    print("wrote hmm_errors.json", res)


def main():
    gen = ref_hmm.GenModel(4, 2, a_mat=np.array([[0.85, 0.05, 0.05, 0.05], [0.05, 0.85, 0.05, 0.05],
                                                  [0.05, 0.05, 0.85, 0.05], [0.05, 0.05, 0.05, 0.85]]),
                           mu_vecs=np.array([[-4.0, 0.0], [4.0, 0.0], [0.0, 5.0], [0.0, -5.0]]), seed=321)
    x1, z1 = gen.gen_sample(500)
    np.savez_compressed(os.path.join(HERE, "hmm_c1_sample.npz"), x=x1, z=z1)
    single_step("hmm_f6_k4_d2_t500.npz", 4, 2, x1, seed=0, store_x=False)
    x2, _ = synth_hmm(32, 16, 4096, np.float64)
    single_step("hmm_f6_k32_d16_t4096.npz", 32, 16, x2, seed=0, store_x=False, keep=512)
    full_driver("hmm_f3_k4_subsampling.npz", 4, 2, x1, seed=0, store_x=False, viterbi_rows=200, num_init=4)
    full_driver("hmm_f3_k4_random_resp.npz", 4, 2, x1, seed=3, store_x=False, num_init=3, max_itr=40,
                init_type="random_responsibility")
    x3, _ = synth_hmm(8, 16, 8192, np.float32)
    full_driver("hmm_f3_k8_d16_t8192_f32.npz", 8, 16, x3, seed=0, store_x=False, num_init=1, max_itr=10, tolerance=0.0)
    full_driver("hmm_f3_t1.npz", 2, 1, x1[:1, :1].copy(), seed=0, store_x=True, num_init=2, max_itr=4,
                init_type="random_responsibility")
    boundary_errors()


if __name__ == "__main__":
    main()
