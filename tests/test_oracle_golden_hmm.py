"""Pin the HMM oracle (oracle/hmm_vb_oracle.py) to outputs of the reference (tests/golden/hmm_*.npz)."""
import hashlib
import json

import numpy as np
import pytest

from conftest import load_golden, rel_err
from oracle import hmm_vb_oracle as orc


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def fixture_x(g):
    if "x" in g:
        return g["x"]
    K, D, T = int(g["K"]), int(g["D"]), int(g["N"])
    if D == 2:
        x = load_golden("hmm_c1_sample.npz")["x"]
    else:
        x, _ = orc.synth_hmm({16: 32 if K == 32 else 8}[D], D, T, np.dtype(str(g["x_dtype"])))
    assert sha(x) == str(g["x_sha256"])
    return x


def posterior_from(g, prefix):
    return orc.HmmPosterior(*(g[prefix + k].copy() for k in ("hn_eta_vec", "hn_zeta_vecs", "hn_m_vecs", "hn_kappas",
                                                            "hn_nus", "hn_w_mats", "hn_w_mats_inv"))).refresh()


@pytest.mark.parametrize("name", ["hmm_f6_k4_d2_t500.npz", "hmm_f6_k32_d16_t4096.npz"])
def test_single_pass_and_k_side(name):
    g = load_golden(name)
    x = fixture_x(g).astype(np.float64)
    K, D = int(g["K"]), int(g["D"])
    p = orc.HmmPrior.default(K, D)
    q = posterior_from(g, "in_")
    st = orc.data_pass(x, q, np.zeros((K, D, D)))
    n = g["ln_rho"].shape[0]
    assert rel_err(st.ln_rho[:n], g["ln_rho"]) < 1e-13
    assert np.max(np.abs(st.alpha[:n] - g["alpha_vecs"])) < 1e-12
    assert rel_err(st.beta[:n], g["beta_vecs"]) < 1e-10
    assert np.max(np.abs(st.gamma[:n] - g["gamma_vecs"])) < 1e-11
    assert abs(np.log(st.cs).sum() - float(g["ln_cs_sum"])) < 1e-9 * abs(float(g["ln_cs_sum"]))
    assert rel_err(st.ns, g["ns"]) < 1e-11 and rel_err(st.ms, g["ms"]) < 1e-10
    assert rel_err(st.x_bar, g["x_bar_vecs"]) < 1e-11 and rel_err(st.s, g["s_mats"]) < 1e-10
    t = orc.lower_bound(p, q, st)
    for key in ("p_x", "p_z", "p_pi", "p_a", "p_mu_lambda", "q_z", "q_pi", "q_a", "q_mu_lambda", "vl"):
        ref = float(g["vl" if key == "vl" else "vl_" + key])
        assert abs(t[key] - ref) <= 1e-9 * max(1.0, abs(ref)), key
    orc.update_q(p, q, st)
    for mine, key, tol in ((q.eta, "hn_eta_vec", 1e-12), (q.zeta, "hn_zeta_vecs", 1e-11), (q.m, "hn_m_vecs", 1e-11),
                           (q.w_inv, "hn_w_mats_inv", 1e-11), (q.w, "hn_w_mats", 1e-9)):
        assert rel_err(mine, g["out_" + key]) < tol, key
    assert rel_err(q.ln_a_tilde, g["out_ln_a_tilde"]) < 1e-11
    assert rel_err(q.a_tilde, g["out_a_tilde"]) < 1e-11
    assert rel_err(q.pi_tilde, g["out_pi_tilde"]) < 1e-11


@pytest.mark.parametrize("name", ["hmm_f3_k4_subsampling.npz", "hmm_f3_k4_random_resp.npz",
                                  "hmm_f3_k8_d16_t8192_f32.npz", "hmm_f3_t1.npz"])
def test_full_driver(name):
    g = load_golden(name)
    x = fixture_x(g).astype(np.float64)
    K, D = int(g["K"]), int(g["D"])
    kw = json.loads(str(g["kw"]))
    p = orc.HmmPrior.default(K, D)
    res = orc.update_posterior(x, p, orc.HmmPosterior.from_prior(p), np.random.default_rng(int(g["seed"])), **kw)
    assert res.winner == int(g["winner"])
    assert (not res.converged_any) == bool(g["result_warning"])
    tr = g["vl_trace"]
    for i, t in enumerate(res.vl_trace):
        ref = tr[i][~np.isnan(tr[i])]
        assert len(t) == len(ref) and np.allclose(t, ref, rtol=1e-8)
    q = res.posterior
    for mine, key in ((q.eta, "hn_eta_vec"), (q.zeta, "hn_zeta_vecs"), (q.m, "hn_m_vecs"), (q.kappa, "hn_kappas"),
                      (q.nu, "hn_nus"), (q.w, "hn_w_mats")):
        assert rel_err(mine, g[key]) < 1e-7, key
    assert rel_err(res.stats.ms, g["ms"]) < 1e-7
    assert np.max(np.abs(res.stats.gamma[:64] - g["gamma_head"])) < 1e-7
    if "viterbi_01" in g:
        xs = x[:g["viterbi_01"].shape[0]]
        assert np.array_equal(orc.viterbi(xs, q), g["viterbi_01"])
