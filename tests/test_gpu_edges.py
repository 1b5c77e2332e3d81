"""Edge cases of the GPU boundary: row-strided x (ldx > D), K = 1, N around the 64-row granule, f64 rows at
D = 128, torch-tensor input, latent-variable read-outs and the sequential-update entry points."""
import io
import warnings
from contextlib import redirect_stdout

import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err
from oracle import gmm_vb_oracle as orc

pytestmark = pytest.mark.gpu


def _random_posterior(K, D, rng):
    p = orc.Prior.default(K, D)
    q = orc.Posterior.from_prior(p)
    q.m = 1.5 * rng.standard_normal((K, D))
    a = rng.standard_normal((K, D, D))
    q.w_inv = a @ np.swapaxes(a, 1, 2) + D * np.eye(D)
    q.w = np.linalg.inv(q.w_inv)
    q.nu = q.nu + rng.uniform(0, 5, K)
    q.kappa = q.kappa + rng.uniform(0, 5, K)
    q.alpha = q.alpha + rng.uniform(0, 5, K)
    q.refresh_pi()
    q.refresh_lambda()
    return q


def _engine_pass(xd, q, prepare):
    from bayesml_amd import _kside
    from bayesml_amd._engine import DataPass
    K, D = q.m.shape
    dev = xd.device
    t = lambda v: torch.as_tensor(v, dtype=torch.float64, device=dev)   # noqa: E731
    qd = _kside.features(_kside.PostT(t(q.alpha), t(q.m), t(q.kappa), t(q.nu), t(q.w_inv)))
    eng = DataPass(K, D, xd.dtype, xd.shape[0], dev)
    eng.set_pivot(xd.to(torch.float64).mean(dim=0))
    if prepare:
        eng.prepare_rows(xd)
    eng.set_params(qd.c, qd.m, qd.u)
    ns, h, a, B = eng.split_stats(eng.estep_mstep(xd))
    x_bar, s = _kside.moments_from_stats(ns, a, B, eng.pivot, torch.zeros(K, D, D, dtype=torch.float64, device=dev))
    out = (eng.ln_rho().cpu().numpy(), ns.cpu().numpy(), x_bar.cpu().numpy(), s.cpu().numpy(), eng.launch_info)
    eng.close()
    return out


@pytest.mark.parametrize("K,D,N,dtype,pad,prepare", [
    (4, 32, 1000, np.float32, 32, True),      # ldx = 64: still vector-aligned
    (4, 32, 1000, np.float32, 3, False),      # ldx = 35: misaligned rows -> masked scalar loads
    (3, 128, 300, np.float64, 0, True),       # f64 rows at the widest D
    (1, 16, 63, np.float64, 0, True), (1, 16, 64, np.float32, 0, False), (2, 48, 65, np.float32, 16, True),
    (70, 8, 129, np.float64, 0, True),        # more components than a 64-wide anything
])
def test_strided_and_small_shapes(K, D, N, dtype, pad, prepare):
    rng = np.random.default_rng(K + 10 * D + N)
    wide = (rng.standard_normal((N, D + pad)) * 1.3 + 0.7).astype(dtype)
    q = _random_posterior(K, D, rng)
    st = orc.data_pass(wide[:, :D].astype(np.float64), q)
    xd = torch.from_numpy(wide).to("cuda:0")[:, :D]            # a row-strided view when pad > 0
    ln_rho, ns, x_bar, s, info = _engine_pass(xd, q, prepare)
    assert rel_err(ln_rho, st.ln_rho) < 1e-11, info
    assert rel_err(ns, st.ns) < 1e-10 and rel_err(x_bar, st.x_bar) < 1e-10 and rel_err(s, st.s) < 1e-9, info
    if pad == 3:
        assert "masked" in info


def test_tensor_input_and_latent_readouts():
    from bayesml_amd import gaussianmixture as gm
    g = load_golden("gmm_f3_c1_subsampling.npz")
    x = load_golden("gmm_c1_sample.npz")["x"]
    m = gm.LearnModel(3, 2, seed=0, verbose=False)
    m.update_posterior(torch.from_numpy(x).to("cuda:0"))         # a device tensor skips the host copy
    assert rel_err(m.hn_m_vecs, g["hn_m_vecs"]) < 1e-8
    assert m.r_vecs.shape == (1000, 3) and abs(m.r_vecs.sum() - 1000) < 1e-9
    z = m.estimate_latent_vars(x[:128], "0-1")
    assert np.array_equal(z, g["latent_01"])
    kl = m.estimate_latent_vars(x[:128], "KL")
    assert np.max(np.abs(kl - g["latent_sq"])) < 1e-8
    with pytest.raises(gm.CriteriaError if hasattr(gm, "CriteriaError") else Exception):
        m.estimate_latent_vars(x[:8], "L1")


def test_sequential_update_entry_points():
    from bayesml_amd import gaussianmixture as gm
    x = load_golden("gmm_c1_sample.npz")["x"]
    m = gm.LearnModel(3, 2, seed=0, verbose=False)
    with warnings.catch_warnings(), redirect_stdout(io.StringIO()):
        warnings.simplefilter("ignore")
        m.update_posterior(x[:300], num_init=2, max_itr=30)
        before = m.hn_kappas.sum()
        pred = m.pred_and_update(x[300], num_init=1, max_itr=5)          # N = 1 through the GPU path
        assert pred.shape == (2,) and abs(m.hn_kappas.sum() - before - 1.0) < 1e-9
        z = m.estimate_latent_vars_and_update(x[301:365], num_init=1, max_itr=5)
    assert z.shape == (64, 3) and np.all(z.sum(axis=1) == 1)
    assert abs(m.hn_kappas.sum() - before - 65.0) < 1e-8


def test_component_below_the_relevance_line_is_pinned():
    """A component whose largest responsibility over all rows lies below 2^-100 (but above the f64 underflow): the
    reference gives it a tiny positive ns[k], so its x_bar_vecs[k] / s_mats[k] are a weighted mean / scatter (ref :729-732).
    The engine's behaviour, documented in INTEGRATION.md:
      dense pass   - ns[k] is that tiny number (to rounding), x_bar / s as in the reference;
      pruned pass  - every pair of the component is PROVEN below 2^-100 and skipped: ns[k] == 0.0 exactly, and the
                     reference's own ns[k] > 0 guard then leaves x_bar_vecs[k] = 0 and s_mats[k] at its previous value.
    Either way the posterior is the reference's: hn_* change by ns[k] ~ 1e-150 relative."""
    import os
    from bayesml_amd import gaussianmixture as gm
    K, D, N = 3, 64, 40000
    rng = np.random.default_rng(11)
    mu = np.zeros((K, D))
    mu[1, 0] = 6.0
    mu[2, :] = 3.0                                  # no data near it: ln r ~ -300 for every row
    z = rng.integers(0, 2, N)
    x = (mu[z] + rng.standard_normal((N, D))).astype(np.float32)
    p = orc.Prior.default(K, D)
    q = orc.Posterior.from_prior(p)
    q.m = mu.copy()
    q.kappa = np.full(K, 50.0)
    q.nu = np.full(K, float(D) + 50.0)
    q.w = np.tile(np.eye(D) / q.nu[0], (K, 1, 1))
    q.w_inv = np.linalg.inv(q.w)
    q.alpha = np.array([20.0, 20.0, 1.0])
    q.refresh_pi()
    q.refresh_lambda()
    st = orc.data_pass(x.astype(np.float64), q)
    assert 0.0 < st.ns[2] < 2.0 ** -100 * N and np.max(st.r[:, 2]) < 2.0 ** -100      # the case under test

    def run(force):
        old = os.environ.get("GMMVB_ESTEP_PRUNE")
        os.environ.pop("GMMVB_ESTEP_PRUNE", None)
        if force:
            os.environ["GMMVB_ESTEP_PRUNE"] = "force"
        try:
            m = gm.LearnModel(K, D, seed=0, device=torch.device("cuda", 0), verbose=False)
            m.set_hn_params(q.alpha, q.m, q.kappa, q.nu, q.w)
            m.s_mats[2] = 7.0                       # "previous value" of the stale scatter
            m.estimate_latent_vars(x, loss="squared")
            counts = m._engine.pass_counts()
        finally:
            os.environ.pop("GMMVB_ESTEP_PRUNE", None)
            if old is not None:
                os.environ["GMMVB_ESTEP_PRUNE"] = old
        return m, counts

    dense, c0 = run(False)
    assert c0["estep_dense"] >= 1 and c0["estep_bound"] == 0
    assert dense.ns[2] > 0.0 and abs(dense.ns[2] / st.ns[2] - 1.0) < 1e-6
    assert rel_err(dense.x_bar_vecs[2], st.x_bar[2]) < 1e-6
    pruned, c1 = run(True)
    assert c1["estep_bound"] >= 1
    assert pruned.ns[2] == 0.0                                           # proven irrelevant, never accumulated
    assert np.all(pruned.x_bar_vecs[2] == 0.0) and np.all(pruned.s_mats[2] == 7.0)
    for m in (dense, pruned):                                            # the components that hold the data: unaffected
        assert rel_err(m.ns[:2], st.ns[:2]) < 1e-10 and rel_err(m.x_bar_vecs[:2], st.x_bar[:2]) < 1e-10
        assert np.all(m.r_vecs[:, 2] < 2.0 ** -100)
    # no posterior-level gap: the closed-form update with ns[2] = 0 and with the reference's tiny ns[2]
    q_ref, q_zero = q.copy(), q.copy()
    orc.update_q_mu_lambda(p, q_ref, st)
    st0 = orc.Stats(st.ln_rho, st.r, st.ns.copy(), st.x_bar.copy(), st.s.copy())
    st0.ns[2], st0.x_bar[2], st0.s[2] = 0.0, 0.0, 7.0
    orc.update_q_mu_lambda(p, q_zero, st0)
    for a, b in ((q_ref.m, q_zero.m), (q_ref.w_inv, q_zero.w_inv), (q_ref.kappa, q_zero.kappa), (q_ref.nu, q_zero.nu)):
        assert np.max(np.abs(a - b)) <= 1e-15 * np.max(np.abs(a))


@pytest.mark.parametrize("K,D,N,dtype", [(24, 65, 513, np.float32), (64, 65, 2049, np.float64)])
def test_degree_65_fit_is_reproducible_and_follows_the_oracle(K, D, N, dtype):
    """A fit at c_degree = 65 (one feature in the fifth tile) with enough components to reach the batch sizes at which
    this image's batched GPU inverse is unsound: two runs agree bit for bit and the posterior follows the oracle's.
    Found by tests/fuzz_sparse.py (round 6): the prior's W^-1 had come from torch.linalg.inv."""
    from bayesml_amd import gaussianmixture as gm
    x = orc.synth_gmm(K, D, N, dtype, seed=826, spread=0.6)
    runs = []
    for _ in range(2):
        m = gm.LearnModel(K, D, seed=826, device="cuda:0", verbose=False)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            m.update_posterior(x, max_itr=4, num_init=1, tolerance=0.0)
        runs.append((m.get_hn_params(), np.array(m.hn_w_mats_inv)))
        m._engine.close()
    for key in runs[0][0]:
        assert np.array_equal(runs[0][0][key], runs[1][0][key]), key
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref = orc.update_posterior(x.astype(np.float64), orc.Prior.default(K, D), orc.Posterior.from_prior(orc.Prior.default(K, D)),
                                   np.random.default_rng(826), max_itr=4, num_init=1, tolerance=0.0)
    hn = runs[0][0]
    for key, val in (("hn_alpha_vec", ref.posterior.alpha), ("hn_m_vecs", ref.posterior.m), ("hn_kappas", ref.posterior.kappa),
                     ("hn_nus", ref.posterior.nu), ("hn_w_mats", ref.posterior.w)):
        assert rel_err(hn[key], val) < 1e-8, key
    assert rel_err(runs[0][1], ref.posterior.w_inv) < 1e-9
