"""multivariate_normal.LearnModel: oracle and host logic against reference-generated fixtures
on the CPU (the data pass through tests/fake_engine.py), and the real engine on the GPU."""
import json
import os
import warnings

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_golden, rel_err
from oracle import mvn_oracle

CASES = ["mvn_d2_n100.npz", "mvn_d5_n1.npz", "mvn_d32_n5000_f32_batches3.npz", "mvn_d128_n3000_f32.npz"]


def make_model(g, fake):
    from bayesml_amd import multivariate_normal as mvn
    prior = {k: (np.array(v) if isinstance(v, list) else v) for k, v in json.loads(str(g["prior"])).items()}
    m = mvn.LearnModel(int(g["D"]), **prior)
    if fake:
        from fake_engine import cpu_factory
        m._data_pass_factory = cpu_factory
    return m


def check_model(g, m, tol):
    x = g["x"]
    for i, part in enumerate(np.array_split(x, int(g["batches"]))):
        m.update_posterior(part)
        assert rel_err(m.hn_m_vec, g[f"b{i}_hn_m_vec"]) < tol
        assert rel_err(m.hn_w_mat_inv, g[f"b{i}_hn_w_mat_inv"]) < tol
        assert m.hn_kappa == float(g[f"b{i}_hn_kappa"]) and m.hn_nu == float(g[f"b{i}_hn_nu"])
    assert rel_err(m.hn_w_mat, g["hn_w_mat"]) < 100 * tol
    assert list(m.get_hn_params()) == ["hn_m_vec", "hn_kappa", "hn_nu", "hn_w_mat"]
    mu, lam = m.estimate_params("squared")
    assert rel_err(mu, g["est_sq_mu"]) < tol and rel_err(lam, g["est_sq_lambda"]) < 100 * tol
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        _mu, lam01 = m.estimate_params("0-1")
    if np.isnan(g["est_01_lambda"]).all():
        assert lam01 is None
    else:
        assert rel_err(lam01, g["est_01_lambda"]) < 100 * tol
    assert set(m.estimate_params("squared", dict_out=True)) == {"mu_vec", "lambda_mat"}
    m.calc_pred_dist()
    assert rel_err(m.p_v_mat, g["p_v_mat"]) < 100 * tol and rel_err(m.p_v_mat_inv, g["p_v_mat_inv"]) < tol
    assert m.p_nu == float(g["p_nu"])
    preds = [m.pred_and_update(g["next_x"][0]).copy(), m.pred_and_update(g["next_x"][1], loss="0-1").copy()]
    assert rel_err(np.array(preds), g["preds"]) < tol
    assert rel_err(m.hn_w_mat_inv, g["after_hn_w_mat_inv"]) < tol and rel_err(m.hn_m_vec, g["after_hn_m_vec"]) < tol
    assert m.make_prediction("KL").df == m.p_nu


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference(name):
    g = load_golden(name)
    D = int(g["D"])
    prior = json.loads(str(g["prior"]))
    m = np.array(prior.get("h0_m_vec", np.zeros(D)), dtype=float)
    kappa, nu = float(prior.get("h0_kappa", 1.0)), float(prior.get("h0_nu", D))
    w_inv = np.linalg.inv(np.array(prior.get("h0_w_mat", np.eye(D)), dtype=float))
    for i, part in enumerate(np.array_split(g["x"], int(g["batches"]))):
        m, kappa, nu, w, w_inv = mvn_oracle.update(m, kappa, nu, w_inv, part)
        assert rel_err(m, g[f"b{i}_hn_m_vec"]) < 1e-13 and rel_err(w_inv, g[f"b{i}_hn_w_mat_inv"]) < 1e-13
        assert rel_err(w, g[f"b{i}_hn_w_mat"]) < 1e-10
    pm, pnu, pv, pvi = mvn_oracle.pred_params(m, kappa, nu, w, w_inv)
    assert pnu == float(g["p_nu"]) and rel_err(pv, g["p_v_mat"]) < 1e-10 and rel_err(pvi, g["p_v_mat_inv"]) < 1e-13


@pytest.mark.parametrize("name", CASES)
def test_host_logic_with_cpu_stand_in(name):
    g = load_golden(name)
    check_model(g, make_model(g, fake=True), 1e-11)


def test_boundary_errors_match_reference():
    from bayesml_amd import multivariate_normal as mvn
    from fake_engine import cpu_factory
    with open(os.path.join(GOLDEN, "mvn_errors.json")) as f:
        expected = json.load(f)

    def lm(*a, **k):
        m = mvn.LearnModel(*a, **k)
        m._data_pass_factory = cpu_factory
        return m

    cases = {
        "ctor_float_degree": lambda: lm(2.0),
        "h0_m_vec_wrong_dim": lambda: lm(2, h0_m_vec=np.zeros(3)),
        "h0_kappa_nonpos": lambda: lm(2, h0_kappa=0.0),
        "h0_nu_too_small": lambda: lm(3, h0_nu=2.0),
        "h0_w_mat_not_pd": lambda: lm(2, h0_w_mat=np.array([[1.0, 2.0], [2.0, 1.0]])),
        "h0_w_mat_wrong_dim": lambda: lm(2, h0_w_mat=np.eye(3)),
        "x_wrong_last_dim": lambda: lm(2).update_posterior(np.zeros((5, 3))),
        "x_not_ndarray": lambda: lm(2).update_posterior([[0.0, 1.0]]),
        "bad_loss_estimate": lambda: lm(2).estimate_params("L1"),
        "bad_loss_prediction": lambda: lm(2).make_prediction("L1"),
        "pred_and_update_wrong_shape": lambda: lm(2).pred_and_update(np.zeros((1, 2))),
        "gen_sample_float": lambda: mvn.GenModel(2).gen_sample(3.0),
        "x_int_ok": lambda: lm(2).update_posterior(np.arange(10).reshape(5, 2)),
        "x_3d_ok": lambda: lm(2).update_posterior(np.zeros((3, 4, 2))),
    }
    assert set(cases) == set(expected)
    for name, fn in cases.items():
        try:
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                fn()
            got = None
        except Exception as e:      # noqa: BLE001
            got = type(e).__name__
        assert got == expected[name], name


def test_gen_model_stream_and_pickle_round_trip(tmp_path):
    """gen_params / gen_sample consume the Generator like the reference (the fixture's x came from the reference's
    GenModel with the same seed); h0/hn dicts survive the positional pickle round trip (base.py:191,251)."""
    from bayesml_amd import multivariate_normal as mvn
    g = load_golden("mvn_d2_n100.npz")
    gen = mvn.GenModel(2, seed=1)
    gen.gen_params()
    assert np.allclose(gen.mu_vec, g["mu_vec"], rtol=1e-12) and np.allclose(gen.lambda_mat, g["lambda_mat"], rtol=1e-12)
    assert np.allclose(gen.gen_sample(100), g["x"], rtol=1e-12, atol=1e-14)
    m = make_model(g, fake=True)
    m.update_posterior(g["x"])
    f = str(tmp_path / "hn.pkl")
    m.save_hn_params(f)
    m2 = make_model(g, fake=True)
    m2.load_hn_params(f)
    assert np.array_equal(m2.hn_w_mat, m.hn_w_mat) and m2.hn_kappa == m.hn_kappa
    m2.overwrite_h0_params()
    assert np.array_equal(m2.h0_m_vec, m.hn_m_vec) and m2.h0_nu == m.hn_nu


def test_no_cpu_fallback():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from bayesml_amd import multivariate_normal as mvn
    from bayesml_amd._engine import EngineUnavailableError
    with pytest.raises(EngineUnavailableError):
        mvn.LearnModel(2).update_posterior(np.zeros((4, 2)))


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_gpu_engine_matches_reference(name):
    g = load_golden(name)
    m = make_model(g, fake=False)
    check_model(g, m, 1e-10)
    assert "mstep" in m._engine.launch_info


@pytest.mark.gpu
def test_gpu_full_size_linearity():
    """N = 2e6 rows of D = 64 on the device: batch update == two sequential half updates (size-independent property
    of the conjugate update), and the sample mean is recovered."""
    from bayesml_amd import multivariate_normal as mvn
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev).manual_seed(5)
    x = (torch.randn(2_000_000, 64, device=dev, generator=gen) * 1.5 + 0.25).to(torch.float32)
    a = mvn.LearnModel(64, device=dev).update_posterior(x)
    b = mvn.LearnModel(64, device=dev)
    b.update_posterior(x[:900_001])
    b.update_posterior(x[900_001:])
    assert rel_err(a.hn_m_vec, b.hn_m_vec) < 1e-12 and rel_err(a.hn_w_mat_inv, b.hn_w_mat_inv) < 1e-11
    assert np.max(np.abs(a.hn_m_vec - x.to(torch.float64).mean(dim=0).cpu().numpy())) < 1e-5
