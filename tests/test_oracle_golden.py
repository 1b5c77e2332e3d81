"""Pin the CPU oracle (oracle/gmm_vb_oracle.py) to outputs of the reference itself.

The fixtures were produced by tests/golden/make_golden.py, which imports /root/reference in
the build container.  Nothing here reads /root/reference.
"""
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_golden, rel_err
from oracle import gmm_vb_oracle as orc

F1 = ["gmm_f1_c1_k3_d2_n1000.npz", "gmm_f1_k16_d32_n2048.npz", "gmm_f1_k4_d128_n32768_f32.npz",
      "gmm_f1_k8_d64_n1024_f32_illcond.npz"]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def fixture_x(g):
    """x is stored for small cases, regenerated from the recipe (and checksummed) otherwise."""
    if "x" in g:
        return g["x"]
    K_data = {("float32", 128): 8, ("float64", 32): 16}[(str(g["x_dtype"]), int(g["D"]))]
    x = orc.synth_gmm(K_data, int(g["D"]), int(g["N"]), np.dtype(str(g["x_dtype"])))
    assert sha(x) == str(g["x_sha256"]), "synthetic recipe drifted from the one the fixture was made with"
    return x


def posterior_from(g, prefix):
    q = orc.Posterior(alpha=g[prefix + "hn_alpha_vec"].copy(), m=g[prefix + "hn_m_vecs"].copy(),
                      kappa=g[prefix + "hn_kappas"].copy(), nu=g[prefix + "hn_nus"].copy(),
                      w=g[prefix + "hn_w_mats"].copy(), w_inv=g[prefix + "hn_w_mats_inv"].copy())
    q.refresh_pi()
    q.refresh_lambda()
    return q


@pytest.mark.parametrize("name", F1)
def test_single_data_pass_and_lower_bound(name):
    g = load_golden(name)
    x = fixture_x(g)
    K, D = int(g["K"]), int(g["D"])
    q = posterior_from(g, "in_")
    # derived features (F2 of the previous step)
    assert rel_err(q.e_ln_pi, g["in_e_ln_pi_vec"]) < 1e-13
    assert rel_err(q.e_ln_lambda_det, g["in_e_ln_lambda_dets"]) < 1e-12
    assert rel_err(q.e_lambda, g["in_e_lambda_mats"]) < 1e-14
    assert rel_err(q.ln_b_w_nu, g["in_ln_b_hn_w_nus"]) < 1e-12
    st = orc.data_pass(x, q, np.zeros((K, D, D)))
    n = g["ln_rho"].shape[0]
    assert rel_err(st.ln_rho[:n], g["ln_rho"]) < 1e-13
    assert np.max(np.abs(st.r[:n] - g["r_vecs"])) < 1e-10
    assert rel_err(st.ns, g["ns"]) < 1e-12
    assert rel_err(st.r.sum(axis=0), g["r_colsum"]) < 1e-12
    assert rel_err(st.x_bar, g["x_bar_vecs"]) < 1e-12
    assert rel_err(st.s, g["s_mats"]) < 1e-11
    p = orc.Prior.default(K, D)
    t = orc.lower_bound(p, q, st)
    for key in ("p_x", "p_z", "p_pi", "p_mu_lambda", "q_z", "q_pi", "q_mu_lambda"):
        ref = float(g["vl_" + key])
        assert abs(t[key] - ref) <= 1e-10 * max(1.0, abs(ref)), key
    assert abs(t["vl"] - float(g["vl"])) <= 1e-10 * abs(float(g["vl"]))


@pytest.mark.parametrize("name", F1)
def test_k_side_step(name):
    g = load_golden(name)
    K, D = int(g["K"]), int(g["D"])
    p = orc.Prior.default(K, D)
    q = posterior_from(g, "in_")
    st = orc.Stats(None, None, g["ns"], g["x_bar_vecs"], g["s_mats"])
    orc.update_q_mu_lambda(p, q, st)
    orc.update_q_pi(p, q, st)
    for mine, key, tol in ((q.alpha, "hn_alpha_vec", 1e-14), (q.m, "hn_m_vecs", 1e-13), (q.kappa, "hn_kappas", 1e-14),
                           (q.nu, "hn_nus", 1e-14), (q.w_inv, "hn_w_mats_inv", 1e-13), (q.w, "hn_w_mats", 1e-9),
                           (q.e_ln_pi, "e_ln_pi_vec", 1e-13), (q.e_ln_lambda_det, "e_ln_lambda_dets", 1e-11),
                           (q.e_lambda, "e_lambda_mats", 1e-9), (q.ln_b_w_nu, "ln_b_hn_w_nus", 1e-11)):
        assert rel_err(mine, g["out_" + key]) < tol, key


DRIVER = [("gmm_f3_c1_subsampling.npz", "c1"), ("gmm_f3_c1_random_resp.npz", "c1"), ("gmm_f3_c1_noconv.npz", "c1"),
          ("gmm_f3_k16_d32_n16384.npz", None), ("gmm_f3_k8_d128_n32768_f32.npz", None), ("gmm_f3_n1.npz", None)]


@pytest.mark.parametrize("name,xsrc", DRIVER)
def test_full_driver(name, xsrc):
    g = load_golden(name)
    x = load_golden("gmm_c1_sample.npz")["x"] if xsrc == "c1" else fixture_x(g)
    assert sha(x) == str(g["x_sha256"])
    K, D = int(g["K"]), int(g["D"])
    kw = json.loads(str(g["kw"]))
    p = orc.Prior.default(K, D)
    q0 = orc.Posterior.from_prior(p)
    # the fixture was made by feeding the reference x.astype(float64) (see make_golden.full_driver)
    res = orc.update_posterior(x.astype(np.float64), p, q0, np.random.default_rng(int(g["seed"])), **kw)
    assert res.winner == int(g["winner"])
    assert (not res.converged_any) == bool(g["result_warning"])
    tr = g["vl_trace"]
    assert len(res.vl_trace) == tr.shape[0]
    for i, t in enumerate(res.vl_trace):
        ref = tr[i][~np.isnan(tr[i])]
        assert len(t) == len(ref), (i, len(t), len(ref))
        assert np.allclose(t, ref, rtol=1e-9, atol=0)
    q = res.posterior
    tol_w = 1e-7 if D >= 64 else 1e-9
    assert rel_err(q.alpha, g["hn_alpha_vec"]) < 1e-9
    assert rel_err(q.m, g["hn_m_vecs"]) < 1e-9
    assert rel_err(q.kappa, g["hn_kappas"]) < 1e-9
    assert rel_err(q.nu, g["hn_nus"]) < 1e-9
    assert rel_err(q.w, g["hn_w_mats"]) < tol_w
    assert rel_err(q.w_inv, g["hn_w_mats_inv"]) < tol_w
    assert rel_err(res.stats.ns, g["ns"]) < 1e-9
    assert rel_err(res.stats.s, g["s_mats"]) < 1e-8
    assert np.max(np.abs(res.stats.r[:64] - g["r_head"])) < 1e-8
    if "est_sq_pi" in g:
        pi, mu, lam = orc.estimate_params(q, "squared")
        assert rel_err(pi, g["est_sq_pi"]) < 1e-9 and rel_err(lam, g["est_sq_lambda"]) < 1e-8
        pi, mu, lam = orc.estimate_params(q, "0-1")
        assert np.allclose(pi, g["est_01_pi"], rtol=1e-9, equal_nan=True)
        assert np.allclose(lam, g["est_01_lambda"], rtol=1e-7, equal_nan=True)
        pp = orc.predictive_params(q)
        stale = orc.predictive_params(orc.Posterior.from_prior(p))   # what update_posterior leaves behind
        for key in ("p_mu_vecs", "p_nus", "p_lambda_mats", "p_pi_vec"):
            assert rel_err(pp[key], g[key]) < 1e-8, key
            assert np.allclose(stale[key], g["stale_" + key], rtol=1e-12, atol=1e-300), key
        assert rel_err((pp["p_pi_vec"][:, None] * pp["p_mu_vecs"]).sum(axis=0), g["pred_squared"]) < 1e-8
        xs = x.reshape(-1, D)[:128]
        assert np.array_equal(orc.estimate_latent_vars(xs, q, "0-1"), g["latent_01"])
        assert np.max(np.abs(orc.estimate_latent_vars(xs, q, "squared") - g["latent_sq"])) < 1e-8


def test_error_fixture_is_present_and_sane():
    with open(os.path.join(GOLDEN, "gmm_errors.json")) as f:
        e = json.load(f)
    assert e["ctor_float_degree"] == "ParameterFormatError"
    assert e["x_wrong_last_dim"] == "DataFormatError"
    assert e["bad_init_type"] == "ValueError"
    assert e["x_int_dtype"] is None


@pytest.mark.parametrize("kind", ["kdata8", "weights", "aniso"])
def test_full_driver_off_the_benchmark_recipe(kind):
    """The round-5 fixtures off the benchmark's recipe (K_model = 2 K_data; Dirichlet(0.3) mixing weights; anisotropic
    clusters): the oracle's driver against the reference's, matrices through the stored functionals."""
    from conftest import mat_functionals
    g = load_golden(f"gmm_f3_k16_d64_n32768_f32_{kind}.npz")
    K, D, N = int(g["K"]), int(g["D"]), int(g["N"])
    x = orc.synth_gmm(int(g["K_data"]), D, N, np.float32, spread=float(g["spread"]),
                      weights_alpha=float(g["weights_alpha"]) if "weights_alpha" in g else None,
                      scale_range=tuple(g["scale_range"]) if "scale_range" in g else None)
    assert sha(x) == str(g["x_sha256"])
    p = orc.Prior.default(K, D)
    res = orc.update_posterior(x.astype(np.float64), p, orc.Posterior.from_prior(p),
                               np.random.default_rng(int(g["seed"])), **json.loads(str(g["kw"])))
    ref = g["vl_trace"][0]
    assert np.allclose(res.vl_trace[0], ref[~np.isnan(ref)], rtol=1e-9, atol=0)
    q = res.posterior
    assert rel_err(q.alpha, g["hn_alpha_vec"]) < 1e-9 and rel_err(q.m, g["hn_m_vecs"]) < 1e-9
    for key, got in (("hn_w_mats", q.w), ("hn_w_mats_inv", q.w_inv)):
        for fn, val in mat_functionals(got).items():
            assert rel_err(val, g[f"{key}_{fn}"]) < 1e-7, (key, fn)


def test_full_driver_overlapping_clusters_compact_fixture():
    """tests/golden/make_golden_large.py fixture on heavily overlapping clusters (means 0.3 * randn): the oracle's
    driver against the reference's, matrices compared through the stored functionals.  (The two larger fixtures of
    that script - K=64 D=128 N=140000 and K=256 D=64 N=36000 - take the oracle minutes; they pin the GPU path in
    tests/test_gpu_sparse_parity.py, and the oracle is pinned to the reference at those K, D by the F1/F3 cases.)"""
    from conftest import mat_functionals
    g = load_golden("gmm_f3_k16_d64_n32768_f32_overlap.npz")
    K, D, N = int(g["K"]), int(g["D"]), int(g["N"])
    x = orc.synth_gmm(int(g["K_data"]), D, N, np.float32, spread=float(g["spread"]))
    assert sha(x) == str(g["x_sha256"])
    p = orc.Prior.default(K, D)
    res = orc.update_posterior(x.astype(np.float64), p, orc.Posterior.from_prior(p),
                               np.random.default_rng(int(g["seed"])), **json.loads(str(g["kw"])))
    ref = g["vl_trace"][0]
    assert np.allclose(res.vl_trace[0], ref[~np.isnan(ref)], rtol=1e-9, atol=0)
    q = res.posterior
    assert rel_err(q.alpha, g["hn_alpha_vec"]) < 1e-9 and rel_err(q.m, g["hn_m_vecs"]) < 1e-9
    for key, got in (("hn_w_mats", q.w), ("hn_w_mats_inv", q.w_inv), ("s_mats", res.stats.s)):
        for fn, val in mat_functionals(got).items():
            assert rel_err(val, g[f"{key}_{fn}"]) < 1e-7, (key, fn)
