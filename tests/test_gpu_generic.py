"""c_degree > 128: up to 256 the data pass runs on dense MFMA kernels of their own (csrc/estep_rows.h: U's block rows streamed
through LDS; mstep.h: a component's tile pairs spread over T / 2 waves), beyond on the plain f64 kernels of csrc/generic.h.
Same contract: the public driver must reproduce the oracle's posterior (the reference accepts any positive c_degree,
``_gaussianmixture.py:433``)."""
import warnings

import numpy as np
import pytest
import torch

from conftest import rel_err
from oracle import gmm_vb_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("K,D,N,dtype", [(4, 160, 3000, np.float32), (3, 129, 1000, np.float64), (2, 330, 700, np.float32),
                                         (5, 200, 2500, np.float32), (2, 256, 900, np.float64), (3, 241, 777, np.float32)])
def test_wide_rows_through_the_driver(K, D, N, dtype):
    from bayesml_amd import gaussianmixture as gm
    x = orc.synth_gmm(K, D, N, dtype)
    m = gm.LearnModel(K, D, seed=0, device="cuda:0", verbose=False)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m.update_posterior(x, max_itr=5, num_init=2, tolerance=0.0)
        ref = orc.update_posterior(x.astype(np.float64), orc.Prior.default(K, D), orc.Posterior.from_prior(orc.Prior.default(K, D)),
                                   np.random.default_rng(0), max_itr=5, num_init=2, tolerance=0.0)
    info = m._engine.launch_info
    if D > 256:
        assert "generic" in info, info
    else:           # T = 10, 12, 14 or 16 feature tiles (odd counts rounded up)
        t = 2 * (((D + 15) // 16 + 1) // 2)
        assert f"estep_rows_f64<T={t}," in info and f"mstep_wide_f64<T={t},centred-f64" in info, info
    hn = m.get_hn_params()
    for key, val in (("hn_alpha_vec", ref.posterior.alpha), ("hn_m_vecs", ref.posterior.m), ("hn_kappas", ref.posterior.kappa),
                     ("hn_nus", ref.posterior.nu), ("hn_w_mats", ref.posterior.w)):
        assert rel_err(hn[key], val) < 1e-7, key
    st = orc.data_pass(x.astype(np.float64), ref.posterior)
    assert np.max(np.abs(m.r_vecs - st.r)) < 1e-5
    assert rel_err(m.ns, st.ns) < 1e-6 and rel_err(m.s_mats, st.s) < 1e-5
    # ... and to rounding for the posterior the model itself holds (the final E-step of ref:895)
    own = orc.Posterior(alpha=hn["hn_alpha_vec"].copy(), m=hn["hn_m_vecs"].copy(), kappa=hn["hn_kappas"].copy(),
                        nu=hn["hn_nus"].copy(), w=hn["hn_w_mats"].copy(), w_inv=m.hn_w_mats_inv.copy())
    own.refresh_pi()
    own.refresh_lambda()
    st2 = orc.data_pass(x.astype(np.float64), own)
    assert np.max(np.abs(m.r_vecs - st2.r)) < 1e-9
    assert rel_err(m.ns, st2.ns) < 1e-10 and rel_err(m.s_mats, st2.s) < 1e-9
    st = st2
    z = m.estimate_latent_vars(x[:500])
    assert np.array_equal(z.argmax(axis=1), st.r[:500].argmax(axis=1))


def test_wide_rows_random_responsibility_and_mvn():
    from bayesml_amd import gaussianmixture as gm
    from bayesml_amd import multivariate_normal as mvn
    K, D, N = 3, 144, 1500
    x = orc.synth_gmm(K, D, N, np.float64)
    m = gm.LearnModel(K, D, seed=3, device="cuda:0", verbose=False)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m.update_posterior(x, max_itr=4, num_init=1, tolerance=0.0, init_type="random_responsibility")
        ref = orc.update_posterior(x, orc.Prior.default(K, D), orc.Posterior.from_prior(orc.Prior.default(K, D)),
                                   np.random.default_rng(3), max_itr=4, num_init=1, tolerance=0.0,
                                   init_type="random_responsibility")
    assert rel_err(m.hn_m_vecs, ref.posterior.m) < 1e-7 and rel_err(m.hn_w_mats, ref.posterior.w) < 1e-7
    # the single Gaussian (K = 1, unit responsibilities) at the same width: exact conjugate update
    g = mvn.LearnModel(D, device="cuda:0")
    g.update_posterior(x)
    n = x.shape[0]
    x_bar = x.mean(axis=0)
    kappa = 1.0 + n
    m_n = n * x_bar / kappa
    dx = x - x_bar
    w_inv = np.eye(D) + dx.T @ dx + (1.0 * n / kappa) * np.outer(x_bar, x_bar)
    hn = g.get_hn_params()
    assert rel_err(hn["hn_m_vec"], m_n) < 1e-10 and abs(hn["hn_kappa"] - kappa) < 1e-9 and abs(hn["hn_nu"] - (D + n)) < 1e-9
    assert rel_err(np.linalg.inv(hn["hn_w_mat"]), w_inv) < 1e-9
