"""Size-independent properties at BASELINE.json's FULL sizes (the oracle cannot run there):
  C3  GMM K=64, D=128, N=1e7 (f32 rows):  sum_k r_nk = 1 per row (=> sum ns = N), exact symmetry of B,
      run-to-run determinism, linearity over row shards (two halves add up to the whole to 1e-12),
      the centred second moments are shift-consistent (trace identity against a direct torch reduction).
  C5  HMM K=32, D=16, T=1e7:  sum ns = T, sum ms = T - 1 (every xi_t sums to one), rows of gamma sum to one.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _gmm_state(K, D, dev, seed=3):
    from bayesml_amd import _kside
    gen = torch.Generator(device=dev).manual_seed(seed)
    mu = 2.0 * torch.randn(K, D, dtype=torch.float64, device=dev, generator=gen)
    p = _kside.prior_from_numpy(np.full(K, 0.5), np.zeros((K, D)), np.ones(K), np.full(K, float(D)),
                                np.tile(np.eye(D), (K, 1, 1)), dev)
    q = _kside.post_from_prior(p)
    q.m = mu + 0.05 * torch.randn(K, D, dtype=torch.float64, device=dev, generator=gen)
    q.nu = q.nu + 50.0
    q.kappa = q.kappa + 50.0
    return mu, _kside.features(q)


def test_gmm_c3_full_size_properties():
    from bayesml_amd._engine import DataPass
    K, D, N = 64, 128, 10_000_000
    dev = torch.device("cuda", 0)
    mu, q = _gmm_state(K, D, dev)
    gen = torch.Generator(device=dev).manual_seed(11)
    x = torch.empty((N, D), dtype=torch.float32, device=dev)
    for lo in range(0, N, 1 << 20):
        hi = min(N, lo + (1 << 20))
        z = torch.randint(0, K, (hi - lo,), device=dev, generator=gen)
        x[lo:hi] = (mu[z] + torch.randn(hi - lo, D, dtype=torch.float64, device=dev, generator=gen)).to(torch.float32)
    piv = x[:4096].to(torch.float64).mean(dim=0)
    eng = DataPass(K, D, x.dtype, N, dev)
    eng.set_pivot(piv)
    eng.prepare_rows(x)
    eng.set_params(q.c, q.m, q.u)
    full = eng.estep_mstep(x)
    ns, h, a, B = eng.split_stats(full)
    assert abs(float(ns.sum()) - N) < 1e-5                      # responsibilities sum to one on every row
    assert torch.equal(B, B.transpose(1, 2))                    # exact symmetry
    assert float(h.sum()) <= 0.0
    assert torch.equal(full, eng.estep_mstep(x))                # deterministic
    # trace identity: sum_k tr B_k = sum_n |x_n - pivot|^2 (because sum_k r_nk = 1), against torch in f64 chunks
    tr = 0.0
    for lo in range(0, N, 1 << 21):
        d = x[lo:lo + (1 << 21)].to(torch.float64) - piv
        tr += float((d * d).sum())
    assert abs(float(torch.diagonal(B, dim1=1, dim2=2).sum()) - tr) < 1e-9 * tr
    r_head = eng.responsibilities(0, 4096)
    assert float((r_head.sum(dim=1) - 1.0).abs().max()) < 1e-12
    # linearity over row shards: what the multi-GPU all-reduce relies on
    half = N // 2 + 12345
    acc = torch.zeros_like(full)
    for lo, hi in ((0, half), (half, N)):
        part = DataPass(K, D, x.dtype, hi - lo, dev)
        part.set_pivot(piv)
        part.prepare_rows(x[lo:hi])
        part.set_params(q.c, q.m, q.u)
        acc += part.estep_mstep(x[lo:hi])
        part.close()
    rel = float((acc - full).abs().max() / full.abs().max())
    assert rel < 1e-12, rel
    eng.close()


def test_hmm_c5_full_size_properties():
    from bayesml_amd import _kside
    from bayesml_amd._engine import DataPass
    K, D, T = 32, 16, 10_000_000
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev).manual_seed(5)
    mu = 3.0 * torch.randn(K, D, dtype=torch.float64, device=dev, generator=gen)
    jump = torch.rand(T, device=dev, generator=gen) >= 0.9
    jump[0] = True
    idx = torch.arange(T, device=dev)
    last = torch.cummax(torch.where(jump, idx, torch.zeros_like(idx)), dim=0).values
    z = torch.randint(0, K, (T,), device=dev, generator=gen)[last]
    x = (mu[z] + torch.randn(T, D, dtype=torch.float64, device=dev, generator=gen)).to(torch.float32)
    p = _kside.hmm_prior_from_numpy(np.full(K, 0.5), np.full((K, K), 0.5), np.zeros((K, D)), np.ones(K),
                                    np.full(K, float(D)), np.tile(np.eye(D), (K, 1, 1)), dev)
    q = _kside.hmm_post_from_prior(p)
    q.m = mu + 0.05
    q.nu = q.nu + 20.0
    q.kappa = q.kappa + 20.0
    q.zeta = q.zeta + 5.0 * torch.eye(K, dtype=torch.float64, device=dev)
    q = _kside.hmm_features(q)
    eng = DataPass(K, D, x.dtype, T, dev)
    eng.set_pivot(x[:4096].to(torch.float64).mean(dim=0))
    eng.prepare_rows(x)
    eng.enable_hmm()
    eng.set_params(q.c, q.m, q.u)
    eng.estep(x)
    ms, g0, gl, lnc = eng.forward_backward(q.pi_tilde, q.a_tilde)
    ns, h, a, B = eng.split_stats(eng.mstep(x))
    assert abs(float(ns.sum()) - T) < 1e-5
    assert abs(float(ms.sum()) - (T - 1)) < 1e-5
    assert abs(float(g0.sum()) - 1.0) < 1e-12 and abs(float(gl.sum()) - 1.0) < 1e-12
    assert np.isfinite(float(lnc)) and float(h.sum()) < 0.0
    g = eng.responsibilities(T - 4096, 4096)
    assert float((g.sum(dim=1) - 1.0).abs().max()) < 1e-12
    # transition counts are consistent with state occupancies: row sums of ms = ns - gamma_last, column sums = ns - gamma_0
    assert float((ms.sum(dim=1) - (ns - gl)).abs().max()) < 1e-6
    assert float((ms.sum(dim=0) - (ns - g0)).abs().max()) < 1e-6
    eng.close()
