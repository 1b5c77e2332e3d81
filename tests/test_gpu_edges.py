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
