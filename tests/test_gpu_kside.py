"""The fused K-side kernel (gmmvb_kside_step, csrc/kside.hip) against the torch functions of bayesml_amd._kside, which
restate the reference's closed forms (_gaussianmixture.py:671-770) and are themselves pinned to reference fixtures
(tests/test_host_logic.py F2, tests/test_gpu_parity.py).  Also the standalone factorisation and drift kernels."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _steppers(K, D, want_drift, seed):
    from bayesml_amd import _kside
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(seed)
    a = rng.standard_normal((K, D, D))
    prior = _kside.prior_from_numpy(rng.uniform(0.3, 2.0, K), rng.standard_normal((K, D)), rng.uniform(0.5, 2.0, K),
                                    D + rng.uniform(0.0, 3.0, K), np.linalg.inv(a @ a.transpose(0, 2, 1) + D * np.eye(D)), dev)
    pivot = torch.from_numpy(rng.standard_normal(D)).to(dev)
    L = K * (2 + D + D * D)
    out = []
    for fused in ("1", "0"):
        os.environ["BAYESML_AMD_KSIDE_FUSED"] = fused
        os.environ["BAYESML_AMD_KSIDE_GRAPH"] = "0"
        try:
            out.append(_kside.KStepper(prior, pivot, L, want_drift))
        finally:
            os.environ.pop("BAYESML_AMD_KSIDE_FUSED", None)
            os.environ.pop("BAYESML_AMD_KSIDE_GRAPH", None)
    return out, rng, dev


def _random_stats(K, D, n, rng, dev, dead=()):
    x = rng.standard_normal((n, D)) * 1.3 + 0.4
    r = rng.dirichlet(np.ones(K) * 0.3, n)
    for k in dead:
        r[:, k] = 0.0
    ns = r.sum(0)
    h = np.where(r > 0, r * np.log(np.where(r > 0, r, 1.0)), 0.0).sum(0)
    a = r.T @ x
    B = np.stack([(x * r[:, k, None]).T @ x for k in range(K)])       # (einsum "nk,ni,nj->kij" takes a minute at K = 256)
    B = 0.5 * (B + B.transpose(0, 2, 1))
    return torch.from_numpy(np.concatenate([ns, h, a.ravel(), B.ravel()])).to(dev)


@pytest.mark.parametrize("K,D,drift", [(3, 2, False), (16, 32, True), (64, 128, True), (5, 1, False), (7, 100, True), (256, 64, True)])
def test_fused_step_matches_torch_functions(K, D, drift):
    (fu, ea), rng, dev = _steppers(K, D, drift, K * 100 + D)
    assert fu._fused and not ea._fused
    for it in range(3):                         # chained: q_next of one step is the q of the next
        st = _random_stats(K, D, 40 * K + 5 * D, rng, dev, dead=(1,) if (it == 1 and K > 2) else ())
        for s in (fu, ea):
            s.stats.copy_(st)
            s.step()
        tf, gf = fu.read()
        te, ge = ea.read()
        for key in te:
            assert abs(tf[key] - te[key]) <= 1e-10 * max(1.0, abs(te[key])), (it, key, tf[key], te[key])
        for name in ("ns", "x_bar", "s", "s_prev"):
            a, b = getattr(fu, name), getattr(ea, name)
            assert float((a - b).abs().max()) <= 1e-12 * max(1.0, float(b.abs().max())), (it, name)
        for f in ("alpha", "m", "kappa", "nu", "w_inv", "w", "u", "u_inv", "e_ln_pi", "e_ln_lambda_det", "ln_b_w_nu", "c"):
            a, b = getattr(fu.q_next, f), getattr(ea.q_next, f)
            assert float((a - b).abs().max()) <= 1e-10 * max(1.0, float(b.abs().max())), (it, f)
        w = fu.q_next.w
        assert torch.equal(w, w.transpose(1, 2))
        if drift:
            assert abs(gf - ge) <= 1e-9
            for name in ("gamma", "delta", "big_gamma"):
                a, b = getattr(fu, name), getattr(ea, name)
                assert float((a - b).abs().max()) <= 1e-9 * max(1.0, float(b.abs().max())), (it, name)
        for s in (fu, ea):
            s.advance()


def test_factor_and_drift_kernels_against_linalg():
    from bayesml_amd import _engine, _kside
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev).manual_seed(3)
    for K, D in ((4, 3), (64, 128), (9, 77)):
        a = torch.randn(K, D, D, dtype=torch.float64, device=dev, generator=gen)
        w_inv = a @ a.transpose(1, 2) + D * torch.eye(D, dtype=torch.float64, device=dev)
        g, g_inv, logdet = _engine.kside_factor(w_inv)
        ref = torch.linalg.cholesky(w_inv)
        assert float((g - ref).abs().max()) < 1e-11 * float(ref.abs().max())
        eye = torch.eye(D, dtype=torch.float64, device=dev)
        assert float((g_inv @ ref - eye).abs().max()) < 1e-10
        assert float((logdet - torch.linalg.slogdet(w_inv)[1]).abs().max()) < 1e-10
    # drift bounds against the SVD
    K, D = 12, 96
    q = []
    for scale in (0.0, 0.03):
        a = torch.randn(K, D, D, dtype=torch.float64, device=dev, generator=torch.Generator(device=dev).manual_seed(5))
        w_inv = a @ a.transpose(1, 2) + D * torch.eye(D, dtype=torch.float64, device=dev)
        p = torch.randn(K, D, D, dtype=torch.float64, device=dev, generator=gen) * scale
        w_inv = w_inv + p @ p.transpose(1, 2) * D
        m = torch.randn(K, D, dtype=torch.float64, device=dev, generator=torch.Generator(device=dev).manual_seed(6)) + scale
        q.append(_kside.features(_kside.PostT(torch.ones(K, dtype=torch.float64, device=dev), m, torch.ones(K, dtype=torch.float64, device=dev),
                                              torch.full((K,), D + 2.0, dtype=torch.float64, device=dev), w_inv)))
    g, d, big = _kside.drift(q[0], q[1])
    sv = torch.linalg.svdvals(q[1].u @ q[0].u_inv)
    assert bool(torch.all(g <= sv[:, -1])) and bool(torch.all(g >= 0.97 * sv[:, -1]))
    assert bool(torch.all(big >= sv[:, 0])) and bool(torch.all(big <= 1.05 * sv[:, 0]))
    dd = torch.linalg.vector_norm((q[1].u @ (q[1].m - q[0].m)[:, :, None])[:, :, 0], dim=1)
    assert bool(torch.all(d >= dd)) and bool(torch.all(d <= dd * (1 + 1e-6) + 1e-12))


@pytest.mark.parametrize("K,D", [(24, 65), (64, 65), (64, 128), (5, 129), (3, 200)])
def test_prior_inverse_never_uses_the_frameworks_batched_inverse(K, D):
    """_kside.spd_inverse against numpy, repeatedly.  Order 65 with 24 or more matrices is where this image's
    torch.linalg.inv / solve_triangular return an O(1) error in a last diagonal element, differently from run to run
    (tools/probe_torch_linalg.py): with it the prior's W^-1 - and every posterior built on it - was wrong for c_degree = 65."""
    from bayesml_amd import _kside
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(K * 1000 + D)
    a = rng.standard_normal((K, D, D)) * 0.05
    for w in (np.broadcast_to(np.eye(D), (K, D, D)).copy(), np.eye(D)[None] + a @ a.transpose(0, 2, 1)):
        want, want_ld = np.linalg.inv(w), -np.linalg.slogdet(w)[1]
        for _ in range(6):
            got, ld = _kside.spd_inverse(w, dev)
            assert float(np.abs(got.cpu().numpy() - want).max()) < 1e-12
            assert float(np.abs(ld.cpu().numpy() - want_ld).max()) < 1e-10
            assert torch.equal(got, got.transpose(1, 2))
        p = _kside.prior_from_numpy(np.ones(K), np.zeros((K, D)), np.ones(K), np.full(K, float(D)), w, dev)
        assert float(np.abs(p.w_inv.cpu().numpy() - want).max()) < 1e-12
