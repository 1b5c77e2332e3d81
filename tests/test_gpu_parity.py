"""GPU parity tests proper: the HIP data pass (through the C ABI) and the LearnModel driver against
the committed golden fixtures (reference outputs) and against the oracle on seeded inputs.

Tolerances (all relative, metric = max|a-b| / max|b| per array):
  * north_star target on posterior hyper-parameters: 1e-5.  The engine computes in f64, so the
    tests hold it to much tighter bounds (1e-9 .. 1e-7, written at each assert).
"""
import io
import json
import warnings
from contextlib import redirect_stdout

import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err
from oracle import gmm_vb_oracle as orc

pytestmark = pytest.mark.gpu

F1 = ["gmm_f1_c1_k3_d2_n1000.npz", "gmm_f1_k16_d32_n2048.npz", "gmm_f1_k4_d128_n32768_f32.npz",
      "gmm_f1_k8_d64_n1024_f32_illcond.npz"]


def fixture_x(g):
    if "x" in g:
        return g["x"]
    K_data = {("float32", 128): 8, ("float64", 32): 16}[(str(g["x_dtype"]), int(g["D"]))]
    return orc.synth_gmm(K_data, int(g["D"]), int(g["N"]), np.dtype(str(g["x_dtype"])))


def device_post(g, prefix, dev):
    from bayesml_amd import _kside
    t = lambda a: torch.as_tensor(a, dtype=torch.float64, device=dev)   # noqa: E731
    return _kside.features(_kside.PostT(t(g[prefix + "hn_alpha_vec"]), t(g[prefix + "hn_m_vecs"]),
                                        t(g[prefix + "hn_kappas"]), t(g[prefix + "hn_nus"]),
                                        t(g[prefix + "hn_w_mats_inv"])))


@pytest.mark.parametrize("name", F1)
def test_single_data_pass_matches_reference(name):
    from bayesml_amd import _kside
    from bayesml_amd._engine import DataPass
    g = load_golden(name)
    x = fixture_x(g)
    K, D, N = int(g["K"]), int(g["D"]), int(g["N"])
    dev = torch.device("cuda", 0)
    xd = torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    q = device_post(g, "in_", dev)
    # The rank-deficient fixture (sqrt(N) < D: W^-1 = singular covariance + 1e-5 I, cond ~ 1e9) is
    # inherently sensitive: LU (reference) and Cholesky (engine) legitimately differ by cond * eps.
    loose = 1e4 if "illcond" in name else 1.0
    # K-side features against the reference's
    assert rel_err(q.e_ln_lambda_det.cpu().numpy(), g["in_e_ln_lambda_dets"]) < 1e-11 * loose
    assert rel_err((q.nu[:, None, None] * q.w).cpu().numpy(), g["in_e_lambda_mats"]) < 1e-9 * loose
    eng = DataPass(K, D, xd.dtype, N, dev)
    eng.set_pivot(xd[:4096].to(torch.float64).mean(dim=0))
    eng.set_params(q.c, q.m, q.u)
    stats = eng.estep_mstep(xd)
    ns, h, a, B = eng.split_stats(stats)
    x_bar, s = _kside.moments_from_stats(ns, a, B, eng.pivot, torch.zeros(K, D, D, dtype=torch.float64, device=dev))
    n = g["ln_rho"].shape[0]
    ln_rho = eng.ln_rho(0, n).cpu().numpy()
    r = eng.responsibilities(0, n).cpu().numpy()
    # ln rho spans up to 1e7 in the ill-conditioned fixture: relative to its own scale
    assert rel_err(ln_rho, g["ln_rho"]) < 1e-11 * loose
    assert np.max(np.abs(r - g["r_vecs"])) < 1e-8 * loose
    assert rel_err(ns.cpu().numpy(), g["ns"]) < 1e-10 * loose
    assert rel_err(x_bar.cpu().numpy(), g["x_bar_vecs"]) < 1e-10 * loose
    assert rel_err(s.cpu().numpy(), g["s_mats"]) < 1e-9 * loose
    B_np = B.cpu().numpy()
    assert np.array_equal(B_np, np.swapaxes(B_np, 1, 2)), "B must be exactly symmetric"
    assert abs(float(-h.sum()) - float(g["vl_q_z"])) <= 1e-9 * loose * max(1.0, abs(float(g["vl_q_z"])))
    # argmax read-out = numpy argmax of the reference's r (first maximiser)
    z = eng.argmax(0, n).cpu().numpy()
    assert np.mean(z == np.argmax(g["ln_rho"], axis=1)) >= (0.999 if loose > 1 else 1.0)
    # lower bound through the K-side code
    p = _kside.prior_from_numpy(*(v for v in _prior_arrays(K, D)), dev)
    terms = _kside.lower_bound(p, q, ns, x_bar, s, h.sum())
    for key in ("p_x", "p_z", "p_pi", "p_mu_lambda", "q_z", "q_pi", "q_mu_lambda", "vl"):
        ref = float(g["vl" if key == "vl" else "vl_" + key])
        assert abs(float(terms[key]) - ref) <= 1e-9 * loose * max(1.0, abs(ref)), key
    # run-to-run determinism: fixed reduction order, no atomics
    stats2 = eng.estep_mstep(xd)
    assert torch.equal(stats, stats2)
    # the centred-f64-copy form of the M-step (what LearnModel uses) is the same arithmetic in the same
    # order up to the compiler's fma contraction of the first-moment sum
    eng.prepare_rows(xd)
    stats3 = eng.estep_mstep(xd)
    assert "centred-f64" in eng.launch_info
    assert rel_err(stats3.cpu().numpy(), stats.cpu().numpy()) < 1e-13
    assert torch.equal(stats3, eng.estep_mstep(xd))
    eng.close()


def _prior_arrays(K, D):
    return (np.full(K, 0.5), np.zeros((K, D)), np.ones(K), np.full(K, float(D)), np.tile(np.eye(D), (K, 1, 1)))


DRIVER = [("gmm_f3_c1_subsampling.npz", "c1"), ("gmm_f3_c1_random_resp.npz", "c1"), ("gmm_f3_c1_noconv.npz", "c1"),
          ("gmm_f3_k16_d32_n16384.npz", None), ("gmm_f3_k8_d128_n32768_f32.npz", None), ("gmm_f3_n1.npz", None)]


@pytest.mark.parametrize("name,xsrc", DRIVER)
def test_full_driver_matches_reference(name, xsrc):
    from bayesml_amd import gaussianmixture as gm
    from bayesml_amd import ResultWarning
    g = load_golden(name)
    x = load_golden("gmm_c1_sample.npz")["x"] if xsrc == "c1" else fixture_x(g)
    K, D = int(g["K"]), int(g["D"])
    kw = json.loads(str(g["kw"]))
    m = gm.LearnModel(K, D, seed=int(g["seed"]))
    buf = io.StringIO()
    with warnings.catch_warnings(record=True) as w, redirect_stdout(buf):
        warnings.simplefilter("always")
        m.update_posterior(x, **kw)
    warned = any(issubclass(i.category, ResultWarning) for i in w)
    assert warned == bool(g["result_warning"])
    # stdout protocol: one line per restart, '*' marks a new best, same VL trace
    lines = [ln for ln in buf.getvalue().split("\n") if ln.strip()]
    tr = g["vl_trace"]
    assert len(lines) == tr.shape[0]
    stars = [ln.endswith("*") for ln in lines]
    assert max(i for i, s in enumerate(stars) if s) == int(g["winner"])
    for i, ln in enumerate(lines):
        vals = [float(seg.split("VL: ")[1].split(" ")[0].rstrip("*").replace("(converged)", ""))
                for seg in ln.split("\r") if seg]
        ref = tr[i][~np.isnan(tr[i])]
        assert len(vals) == len(ref), (i, len(vals), len(ref))
        assert np.allclose(vals, ref, rtol=1e-8, atol=0)
    tol = 1e-6 if D >= 64 else 1e-8        # north_star asks for 1e-5
    hn = m.get_hn_params()
    for key in ("hn_alpha_vec", "hn_m_vecs", "hn_kappas", "hn_nus", "hn_w_mats"):
        assert rel_err(hn[key], g[key]) < tol, key
    assert rel_err(m.hn_w_mats_inv, g["hn_w_mats_inv"]) < tol
    assert rel_err(m.ns, g["ns"]) < tol
    assert rel_err(m.s_mats, g["s_mats"]) < 10 * tol
    assert np.max(np.abs(m.r_vecs[:64] - g["r_head"])) < 1e-6
    assert abs(m.vl - float(g["final_vl"])) <= 1e-8 * abs(float(g["final_vl"]))
    if "est_sq_pi" in g:
        pi, mu, lam = m.estimate_params("squared")
        assert rel_err(pi, g["est_sq_pi"]) < tol and rel_err(lam, g["est_sq_lambda"]) < tol
        for key in ("p_mu_vecs", "p_nus", "p_lambda_mats"):          # stale until calc_pred_dist, like the reference
            assert np.allclose(m.get_p_params()[key], g["stale_" + key], rtol=1e-12, atol=1e-300), key
        m.calc_pred_dist()
        for key in ("p_mu_vecs", "p_nus", "p_lambda_mats"):
            assert rel_err(m.get_p_params()[key], g[key]) < tol, key
        assert rel_err(m.make_prediction("squared"), g["pred_squared"]) < tol
        assert rel_err(m.make_prediction("0-1"), g["pred_01"]) < tol
        xs = x.reshape(-1, D)[:128]
        with redirect_stdout(io.StringIO()):
            assert np.array_equal(m.estimate_latent_vars(xs, "0-1"), g["latent_01"])
            assert np.max(np.abs(m.estimate_latent_vars(xs, "squared") - g["latent_sq"])) < 1e-7


@pytest.mark.parametrize("K,D,N,dtype", [(5, 7, 1003, np.float64), (3, 20, 517, np.float32), (9, 48, 4100, np.float32),
                                         (2, 1, 65, np.float64), (6, 100, 2500, np.float32), (4, 16, 64, np.float64)])
def test_ragged_shapes_against_oracle(K, D, N, dtype):
    """Masked (D % 16 != 0) and tail (N % 64 != 0) paths, random posterior state, vs the oracle."""
    from bayesml_amd import _kside
    from bayesml_amd._engine import DataPass
    rng = np.random.default_rng(K * 1000 + D)
    x = (rng.standard_normal((N, D)) * 1.5 + rng.standard_normal(D)).astype(dtype)
    p = orc.Prior.default(K, D)
    q = orc.Posterior.from_prior(p)
    q.m = rng.standard_normal((K, D))
    a = rng.standard_normal((K, D, D))
    q.w_inv = a @ np.swapaxes(a, 1, 2) + D * np.eye(D)
    q.w = np.linalg.inv(q.w_inv)
    q.nu = q.nu + rng.uniform(0, 5, K)
    q.kappa = q.kappa + rng.uniform(0, 5, K)
    q.alpha = q.alpha + rng.uniform(0, 5, K)
    q.refresh_pi()
    q.refresh_lambda()
    st = orc.data_pass(x.astype(np.float64), q)
    dev = torch.device("cuda", 0)
    t = lambda v: torch.as_tensor(v, dtype=torch.float64, device=dev)   # noqa: E731
    qd = _kside.features(_kside.PostT(t(q.alpha), t(q.m), t(q.kappa), t(q.nu), t(q.w_inv)))
    xd = torch.from_numpy(x).to(dev)
    eng = DataPass(K, D, xd.dtype, N, dev)
    eng.set_pivot(xd.to(torch.float64).mean(dim=0))
    eng.set_params(qd.c, qd.m, qd.u)
    ns, h, av, B = eng.split_stats(eng.estep_mstep(xd))
    x_bar, s = _kside.moments_from_stats(ns, av, B, eng.pivot, torch.zeros(K, D, D, dtype=torch.float64, device=dev))
    assert rel_err(eng.ln_rho().cpu().numpy(), st.ln_rho) < 1e-11
    assert np.max(np.abs(eng.responsibilities().cpu().numpy() - st.r)) < 1e-9
    assert rel_err(ns.cpu().numpy(), st.ns) < 1e-10
    assert rel_err(x_bar.cpu().numpy(), st.x_bar) < 1e-10
    assert rel_err(s.cpu().numpy(), st.s) < 1e-9
    from scipy.special import xlogy
    assert abs(float(h.sum()) - float(np.sum(xlogy(st.r, st.r)))) < 1e-9 * max(1.0, N)
    eng.close()


def test_linearity_over_row_shards_full_width():
    """Size-independent property at the benchmark's K, D: statistics of a row-sharded matrix add up
    to the statistics of the whole (this is what the multi-GPU all-reduce relies on)."""
    from bayesml_amd import _kside
    from bayesml_amd._engine import DataPass
    K, D, N = 64, 128, 200_000
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev).manual_seed(1)
    mu = 2.0 * torch.randn(K, D, device=dev, generator=gen)
    z = torch.randint(0, K, (N,), device=dev, generator=gen)
    x = (mu[z] + torch.randn(N, D, device=dev, generator=gen)).to(torch.float32)
    p = _kside.prior_from_numpy(*_prior_arrays(K, D), dev)
    q = _kside.post_from_prior(p)
    q.m = mu.to(torch.float64) + 0.1
    q = _kside.features(q)
    piv = x[:4096].to(torch.float64).mean(dim=0)
    whole = DataPass(K, D, x.dtype, N, dev)
    whole.set_pivot(piv)
    whole.set_params(q.c, q.m, q.u)
    full = whole.estep_mstep(x).clone()
    acc = torch.zeros_like(full)
    cuts = [0, 70_001, 123_456, N]
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        part = DataPass(K, D, x.dtype, hi - lo, dev)
        part.set_pivot(piv)
        part.set_params(q.c, q.m, q.u)
        acc += part.estep_mstep(x[lo:hi])
        part.close()
    assert rel_err(acc.cpu().numpy(), full.cpu().numpy()) < 1e-12
    ns = whole.split_stats(full)[0]
    assert abs(float(ns.sum()) - N) < 1e-6          # responsibilities sum to one per row
    whole.close()
