"""``GenModel.gen_sample(..., device=...)`` (reference ``_gaussianmixture.py:241-264``,
``_hiddenmarkovnormal.py:344-358``): the batched device samplers draw from the reference's distributions and are
reproducible per seed.  CPU tests run the same torch code on the "cpu" device; the ``gpu`` tests run it at 1e6 rows on
the MI355X."""
import numpy as np
import pytest
import torch

from bayesml_amd import _sample
from bayesml_amd import gaussianmixture as gm
from bayesml_amd import hiddenmarkovnormal as hm


def _gmm(K, D, seed):
    rng = np.random.default_rng(100 + seed)
    lam = np.stack([np.linalg.inv(np.cov(rng.standard_normal((D, 4 * D))) + 0.2 * np.eye(D)) for _ in range(K)])
    lam = (lam + lam.transpose(0, 2, 1)) / 2
    return gm.GenModel(K, D, pi_vec=rng.dirichlet(np.ones(K) * 2), mu_vecs=4 * rng.standard_normal((K, D)),
                       lambda_mats=lam, seed=seed)


def _hmm(K, D, seed):
    rng = np.random.default_rng(200 + seed)
    lam = np.stack([np.linalg.inv(np.cov(rng.standard_normal((D, 4 * D))) + 0.2 * np.eye(D)) for _ in range(K)])
    lam = (lam + lam.transpose(0, 2, 1)) / 2
    return hm.GenModel(K, D, pi_vec=rng.dirichlet(np.ones(K)), a_mat=rng.dirichlet(np.ones(K) * 0.7, K),
                       mu_vecs=4 * rng.standard_normal((K, D)), lambda_mats=lam, seed=seed)


def _check_emissions(x, z, mu, lam, n_min=2000):
    """Per class: sample mean within 5 sigma of mu_k, sample covariance close to Lambda_k^-1."""
    x = x.double().cpu().numpy()
    z = z.cpu().numpy()
    for k in range(mu.shape[0]):
        rows = x[z == k]
        if rows.shape[0] < n_min:
            continue
        cov = np.linalg.inv(lam[k])
        se = np.sqrt(np.diag(cov) / rows.shape[0])
        assert np.all(np.abs(rows.mean(axis=0) - mu[k]) < 5 * se), k
        emp = np.cov(rows.T)
        assert np.max(np.abs(emp - cov)) < 8 * np.max(np.abs(cov)) / np.sqrt(rows.shape[0]), k


def test_markov_chain_equals_the_sequential_recursion():
    rng = np.random.default_rng(0)
    K, T = 5, 7001
    pi, a = rng.dirichlet(np.ones(K)), rng.dirichlet(np.ones(K) * 0.5, K)
    for chunk in (1, 64, 999, 7001, 20000):
        gen = torch.Generator().manual_seed(42)
        z = _sample.markov_chain(torch.tensor(pi), torch.tensor(a), T, gen, chunk=chunk).numpy()
        gen = torch.Generator().manual_seed(42)
        u = torch.rand(T, dtype=torch.float64, generator=gen).numpy()
        cp, ca = np.cumsum(pi)[:-1], np.cumsum(a, axis=1)[:, :-1]
        s = int((u[0] >= cp).sum())
        seq = [s]
        for t in range(1, T):
            s = int((u[t] >= ca[s]).sum())
            seq.append(s)
        assert np.array_equal(z, np.array(seq)), chunk
    one = _sample.markov_chain(torch.ones(1, dtype=torch.float64), torch.ones(1, 1, dtype=torch.float64), 10,
                               torch.Generator().manual_seed(0))
    assert one.tolist() == [0] * 10


def test_gmm_device_sampler_on_cpu():
    g = _gmm(3, 4, seed=1)
    x, z = g.gen_sample(60000, device="cpu", dtype=torch.float32)
    assert x.shape == (60000, 4) and x.dtype == torch.float32 and z.dtype == torch.int64
    freq = np.bincount(z.numpy(), minlength=3) / 60000
    assert np.max(np.abs(freq - g.pi_vec)) < 5 * np.sqrt(0.25 / 60000)
    _check_emissions(x, z, g.mu_vecs, g.lambda_mats)
    x2, z2 = _gmm(3, 4, seed=1).gen_sample(60000, device="cpu", dtype=torch.float32)
    assert torch.equal(x, x2) and torch.equal(z, z2)
    x3, _z3 = _gmm(3, 4, seed=2).gen_sample(60000, device="cpu", dtype=torch.float32)
    assert not torch.equal(x, x3)


def test_hmm_device_sampler_on_cpu():
    g = _hmm(4, 3, seed=1)
    T = 80000
    x, z = g.gen_sample(T, device="cpu")
    assert x.shape == (T, 3) and x.dtype == torch.float64 and z.shape == (T,)
    zz = z.numpy()
    cnt = np.zeros((4, 4))
    np.add.at(cnt, (zz[:-1], zz[1:]), 1)
    emp = cnt / cnt.sum(axis=1, keepdims=True)
    assert np.max(np.abs(emp - g.a_mat)) < 6 * np.sqrt(0.25 / cnt.sum(axis=1).min())
    _check_emissions(x, z, g.mu_vecs, g.lambda_mats)
    x2, z2 = _hmm(4, 3, seed=1).gen_sample(T, device="cpu")
    assert torch.equal(x, x2) and torch.equal(z, z2)
    # the host path is still the reference's stream (tests/test_host_logic_hmm.py); both accept the same arguments
    xh, zh = _hmm(4, 3, seed=1).gen_sample(50)
    assert xh.shape == (50, 3) and zh.shape == (50, 4) and np.all(zh.sum(axis=1) == 1)


@pytest.mark.gpu
def test_gmm_device_sampler_on_gpu():
    g = _gmm(16, 32, seed=3)
    n = 1_000_000
    x, z = g.gen_sample(n, device="cuda:0", dtype=torch.float32)
    assert x.is_cuda and x.shape == (n, 32) and x.dtype == torch.float32
    freq = torch.bincount(z, minlength=16).double().cpu().numpy() / n
    assert np.max(np.abs(freq - g.pi_vec)) < 5 * np.sqrt(0.25 / n)
    _check_emissions(x, z, g.mu_vecs, g.lambda_mats)
    x2, z2 = _gmm(16, 32, seed=3).gen_sample(n, device="cuda:0", dtype=torch.float32)
    assert torch.equal(x, x2) and torch.equal(z, z2)
    # the sample feeds the learner without leaving the device
    m = gm.LearnModel(16, 32, seed=0, device="cuda:0", verbose=False)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m.update_posterior(x, max_itr=5, num_init=1, tolerance=0.0)
    assert abs(m.ns.sum() - n) < 1e-6 * n


@pytest.mark.gpu
def test_hmm_device_sampler_on_gpu():
    g = _hmm(8, 16, seed=4)
    T = 1_000_000
    x, z = g.gen_sample(T, device="cuda:0", dtype=torch.float32)
    assert x.is_cuda and x.shape == (T, 16) and z.shape == (T,)
    zz = z.cpu().numpy()
    cnt = np.zeros((8, 8))
    np.add.at(cnt, (zz[:-1], zz[1:]), 1)
    emp = cnt / cnt.sum(axis=1, keepdims=True)
    assert np.max(np.abs(emp - g.a_mat)) < 6 * np.sqrt(0.25 / cnt.sum(axis=1).min())
    _check_emissions(x, z, g.mu_vecs, g.lambda_mats)
    x2, z2 = _hmm(8, 16, seed=4).gen_sample(T, device="cuda:0", dtype=torch.float32)
    assert torch.equal(x, x2) and torch.equal(z, z2)
