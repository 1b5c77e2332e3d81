"""The sparse-responsibility M-step and the pruned E-step against the dense kernels.

Both only drop work that cannot change the f64 results (responsibilities below 2^-80 of the component's / the
sample's largest one), so everything that leaves the data pass has to agree with the dense path to rounding:
statistics, responsibilities, hard assignments, posterior hyper-parameters.  ln rho itself is allowed to be an
upper bound for pruned pairs, at least 80 ln 2 below the sample's best component.
The switches are environment variables read when a workspace is created."""
import os
import warnings

import numpy as np
import pytest
import torch

from conftest import rel_err
from oracle import gmm_vb_oracle as orc

pytestmark = pytest.mark.gpu

DENSE = dict(GMMVB_ESTEP_PRUNE="0", GMMVB_MSTEP_SPARSE="0")
SPARSE = dict(GMMVB_ESTEP_PRUNE="force")          # M-step: sparse whenever at most 35 % of the pairs are active


class _env:
    def __init__(self, kv):
        self.kv = kv

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in ("GMMVB_ESTEP_PRUNE", "GMMVB_MSTEP_SPARSE")}
        for k in self.old:
            os.environ.pop(k, None)
        os.environ.update(self.kv)

    def __exit__(self, *exc):
        for k, v in self.old.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v


def _fit(x, K, iters, env, seed=0):
    from bayesml_amd import gaussianmixture as gm
    with _env(env):
        m = gm.LearnModel(K, x.shape[1], seed=seed, device=torch.device("cuda", 0), verbose=False)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            m.update_posterior(x, max_itr=iters, num_init=1, tolerance=0.0)
    return m


@pytest.mark.parametrize("K,K_data,D,N,dtype,iters", [
    (32, 32, 96, 40_000, np.float32, 5),
    (16, 16, 64, 30_000, np.float64, 4),
    (24, 6, 128, 20_001, np.float32, 6),        # four components per true cluster: most of them die
])
def test_driver_sparse_equals_dense(K, K_data, D, N, dtype, iters):
    x = orc.synth_gmm(K_data, D, N, dtype)
    a = _fit(x, K, iters, DENSE)
    b = _fit(x, K, iters, SPARSE)
    assert "_bound" in b._engine.launch_info and "_bound" not in a._engine.launch_info
    ha, hb = a.get_hn_params(), b.get_hn_params()
    for k in ha:
        assert rel_err(hb[k], ha[k]) < 1e-11, k
    ra = a._engine.responsibilities().cpu().numpy()
    rb = b._engine.responsibilities().cpu().numpy()
    assert np.max(np.abs(ra - rb)) < 1e-10      # five iterations amplify rounding differences of the two paths
    assert np.array_equal(a._engine.argmax().cpu().numpy(), b._engine.argmax().cpu().numpy())
    # pruned pairs hold upper bounds of ln rho, far below the row's best; exact pairs agree to rounding
    la = a._engine.ln_rho().cpu().numpy()
    lb = b._engine.ln_rho().cpu().numpy()
    same = np.abs(la - lb) <= 1e-9 * np.maximum(1.0, np.abs(la))
    assert np.all(lb[~same] >= la[~same])
    mx = la.max(axis=1, keepdims=True)
    lse = mx + np.log(np.exp(la - mx).sum(axis=1, keepdims=True))
    assert np.all((lb <= lse - 55.4) | same)          # 80 ln 2 = 55.45
    if K == K_data:        # (with several components per cluster every pair may be a candidate)
        assert same.mean() < 0.9, "the pruned path did not prune anything"


def _pass(xd, qd, env, pivot):
    from bayesml_amd._engine import DataPass
    K, D = qd.m.shape
    with _env(env):
        eng = DataPass(K, D, xd.dtype, xd.shape[0], xd.device)
    eng.set_pivot(pivot)
    eng.prepare_rows(xd)
    eng.set_params(qd.c, qd.m, qd.u)
    out = []
    for _ in range(2):           # the second pass is the one that may prune (it needs the first one's sparsity count)
        stats = eng.estep_mstep(xd).clone()
        out.append((stats.cpu().numpy(), eng.responsibilities().cpu().numpy(), eng.launch_info))
    eng.close()
    return out


def test_data_pass_statistics_and_nan_rows():
    from bayesml_amd import _kside
    K, D, N = 32, 96, 32_000
    x = orc.synth_gmm(K, D, N, np.float32)
    m = _fit(x, K, 6, DENSE)
    dev = torch.device("cuda", 0)
    hn = m.get_hn_params()
    t = lambda v: torch.as_tensor(v, dtype=torch.float64, device=dev)   # noqa: E731
    qd = _kside.features(_kside.PostT(t(hn["hn_alpha_vec"]), t(hn["hn_m_vecs"]), t(hn["hn_kappas"]), t(hn["hn_nus"]),
                                      t(np.linalg.inv(hn["hn_w_mats"]))))
    xd = torch.from_numpy(x).to(dev)
    pivot = xd[:4096].to(torch.float64).mean(dim=0)
    dense = _pass(xd, qd, DENSE, pivot)
    sparse = _pass(xd, qd, SPARSE, pivot)
    assert "mstep_list_f64" in sparse[1][2] and "_bound" in sparse[1][2]
    assert "mstep_mfma_f64" in dense[1][2]
    for (sa, ra, _), (sb, rb, _) in zip(dense, sparse):
        assert rel_err(sb, sa) < 1e-12
        assert np.max(np.abs(ra - rb)) < 1e-12
    # a non-finite sample poisons the statistics on both paths alike (nothing is silently skipped)
    xd[1234, 5] = float("nan")
    for env in (DENSE, SPARSE):
        out = _pass(xd, qd, env, pivot)
        assert np.isnan(out[1][0][:K]).all()


def test_restarts_and_default_policy():
    """The default policy over several restarts: every restart begins dense (broad initial components), turns sparse
    after a few iterations, and the next restart's first pass must notice that nothing can be pruned any more.
    N K is just above the library's 2^23 floor for pruning."""
    K, D, N = 48, 64, 180_000
    x = orc.synth_gmm(K, D, N, np.float32)
    from bayesml_amd import gaussianmixture as gm
    out = {}
    for tag, env in (("dense", DENSE), ("default", {})):
        with _env(env):
            m = gm.LearnModel(K, D, seed=3, device=torch.device("cuda", 0), verbose=False)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                m.update_posterior(x, max_itr=7, num_init=3, tolerance=0.0)
        out[tag] = (m.get_hn_params(), m._engine.launch_info)
    assert "_bound" in out["default"][1] and "mstep_list_f64" in out["default"][1]
    for k in out["dense"][0]:
        assert rel_err(out["default"][0][k], out["dense"][0][k]) < 1e-11, k


def test_int8_digit_estep_variant():
    """GMMVB_ESTEP_VARIANT=i8 (opt-in): six base-128 digits per operand on the int8 matrix pipe.  Not bit-equivalent
    to f64: fixed point relative to (row max of U) x (sample max), ~1e-8 absolute on ln rho for well-conditioned
    components, a few 1e-12 of the largest |ln rho| in general."""
    from bayesml_amd import _kside
    from bayesml_amd._engine import DataPass
    dev = torch.device("cuda", 0)
    for K, D, N, dt in ((16, 32, 4096, np.float64), (8, 64, 5000, np.float32), (12, 128, 6000, np.float32), (5, 33, 777, np.float32)):
        x = orc.synth_gmm(K, D, N, dt)
        p = orc.Prior.default(K, D)
        q = orc.Posterior.from_prior(p)
        orc.init_subsampling(x.astype(np.float64), q, np.random.default_rng(0))
        t = lambda v: torch.as_tensor(v, dtype=torch.float64, device=dev)   # noqa: E731
        qd = _kside.features(_kside.PostT(t(q.alpha), t(q.m), t(q.kappa), t(q.nu), t(q.w_inv)))
        xd = torch.from_numpy(x).to(dev)
        res = {}
        for tag, env in (("f64", {}), ("i8", {"GMMVB_ESTEP_VARIANT": "i8"})):
            os.environ.pop("GMMVB_ESTEP_VARIANT", None)
            os.environ.update(env)
            try:
                eng = DataPass(K, D, xd.dtype, N, dev)
            finally:
                os.environ.pop("GMMVB_ESTEP_VARIANT", None)
            eng.set_pivot(xd[:4096].to(torch.float64).mean(dim=0))
            eng.set_params(qd.c, qd.m, qd.u)
            eng.estep(xd)
            res[tag] = (eng.ln_rho().cpu().numpy(), eng.launch_info)
            eng.close()
        assert "estep_i8" in res["i8"][1]
        a, b = res["f64"][0], res["i8"][0]
        assert np.max(np.abs(a - b)) < 2e-11 * np.max(np.abs(a))
        if N >= D * D:       # sqrt(N) >= D: full-rank initial covariances, well conditioned
            assert np.max(np.abs(a - b)) < 1e-7


@pytest.mark.parametrize("K,D,N,dtype,pad", [
    (3, 49, 5000, np.float32, 0),        # smallest D that prunes (two 32-row blocks, masked loads)
    (70, 64, 3000, np.float64, 0),       # more than 64 components: two mask words
    (130, 80, 2049, np.float32, 3),      # three mask words, T32 = 3, misaligned rows (ldx = 83)
    (9, 100, 4097, np.float32, 0),       # D % 16 != 0 at T32 = 4
    (17, 128, 2500, np.float64, 0),      # f64 rows at the widest D
    (2, 96, 70, np.float32, 0),          # fewer rows than one selection block
    (256, 64, 1500, np.float32, 0),      # the largest K the lists support
])
def test_sparse_path_shapes(K, D, N, dtype, pad):
    """Forced pruning on ragged shapes: the statistics of one data pass (second call, so that the first one's
    sparsity count is there) equal the dense kernels' for parameters a few iterations into a fit."""
    from bayesml_amd import _kside
    rng = np.random.default_rng(K * 1000 + D)
    K_data = max(2, min(K, 12))
    wide = np.zeros((N, D + pad), dtype=dtype)
    wide[:, :D] = orc.synth_gmm(K_data, D, N, dtype)
    x = wide[:, :D]
    dev = torch.device("cuda", 0)
    xd = torch.from_numpy(wide).to(dev)[:, :D]
    # parameters: components on the data's clusters (some duplicated / empty when K > K_data), unit-ish covariances
    means = 2.0 * np.random.default_rng(20250711).standard_normal((K_data, D))
    p = orc.Prior.default(K, D)
    q = orc.Posterior.from_prior(p)
    q.m = means[rng.integers(0, K_data, K)] + 0.05 * rng.standard_normal((K, D))
    a = 0.05 * rng.standard_normal((K, D, D))
    q.nu = q.nu + 50.0
    q.w_inv = (np.eye(D) + a @ np.swapaxes(a, 1, 2)) * q.nu[:, None, None]
    q.w = np.linalg.inv(q.w_inv)
    q.kappa = q.kappa + 50.0
    q.alpha = q.alpha + rng.uniform(1, 50, K)
    q.refresh_pi()
    q.refresh_lambda()
    t = lambda v: torch.as_tensor(v, dtype=torch.float64, device=dev)   # noqa: E731
    qd = _kside.features(_kside.PostT(t(q.alpha), t(q.m), t(q.kappa), t(q.nu), t(q.w_inv)))
    pivot = xd[:4096].to(torch.float64).mean(dim=0)
    dense = _pass(xd, qd, DENSE, pivot)
    sparse = _pass(xd, qd, SPARSE, pivot)
    assert "_bound" in sparse[1][2], sparse[1][2]
    for (sa, ra, _), (sb, rb, _) in zip(dense, sparse):
        assert rel_err(sb, sa) < 1e-12
        assert np.max(np.abs(ra - rb)) < 1e-12
    # and against the oracle's formulation
    st = orc.data_pass(x.astype(np.float64), q)
    assert np.max(np.abs(sparse[1][1] - st.r)) < 1e-9


def test_carried_bounds_pass():
    """gmmvb_set_drift: after a parameter update the E-step carries the previous pass's ln rho values / bounds over
    (one elementwise pass) instead of bounding every pair again.  Three successive updates of a fitted posterior
    (shrinking steps, like VB iterations), each pass compared with the dense kernels on the same parameters."""
    from bayesml_amd import _kside
    from bayesml_amd._engine import DataPass
    K, D, N = 32, 96, 32_000
    x = orc.synth_gmm(K, D, N, np.float32)
    m = _fit(x, K, 6, DENSE)
    dev = torch.device("cuda", 0)
    hn = m.get_hn_params()
    t = lambda v: torch.as_tensor(v, dtype=torch.float64, device=dev)   # noqa: E731
    base = (t(hn["hn_alpha_vec"]), t(hn["hn_m_vecs"]), t(hn["hn_kappas"]), t(hn["hn_nus"]), t(np.linalg.inv(hn["hn_w_mats"])))
    gen = torch.Generator(device=dev).manual_seed(1)

    def perturbed(scale):
        a, mm, kap, nu, winv = (v.clone() for v in base)
        mm += scale * torch.randn(mm.shape, dtype=torch.float64, device=dev, generator=gen)
        rot = torch.eye(D, dtype=torch.float64, device=dev) + scale / np.sqrt(D) * torch.randn(
            winv.shape, dtype=torch.float64, device=dev, generator=gen)
        winv = rot @ winv @ rot.transpose(1, 2)                      # stays symmetric positive definite
        winv = 0.5 * (winv + winv.transpose(1, 2))
        return _kside.features(_kside.PostT(a, mm, kap, nu, winv))

    qs = [_kside.features(_kside.PostT(*(v.clone() for v in base)))] + [perturbed(s) for s in (0.05, 0.02, 0.01)]
    xd = torch.from_numpy(x).to(dev)
    pivot = xd[:4096].to(torch.float64).mean(dim=0)
    engines = {}
    for tag, env in (("dense", DENSE), ("sparse", SPARSE)):
        with _env(env):
            eng = DataPass(K, D, xd.dtype, N, dev)
        eng.set_pivot(pivot)
        eng.prepare_rows(xd)
        engines[tag] = eng
    carried = 0
    for i, q in enumerate(qs):
        out = {}
        for tag, eng in engines.items():
            if i > 0:
                eng.set_drift(*_kside.drift(qs[i - 1], q))
            eng.set_params(q.c, q.m, q.u)
            out[tag] = (eng.estep_mstep(xd).cpu().numpy(), eng.responsibilities().cpu().numpy(), eng.launch_info)
        carried += "estep_carried_bounds" in out["sparse"][2] or "estep_sweep_bounds" in out["sparse"][2]
        assert rel_err(out["sparse"][0], out["dense"][0]) < 1e-12, (i, out["sparse"][2])
        assert np.max(np.abs(out["sparse"][1] - out["dense"][1])) < 1e-12
    assert carried >= 2, "the drift hint was not used"
    for eng in engines.values():
        eng.close()
    # the hint itself: gamma <= sigma_min(u_new u_old^-1) (checked against the SVD), tight to a few per cent
    g, d, big = _kside.drift(qs[0], qs[1])
    a = torch.linalg.solve_triangular(qs[0].u, qs[1].u, upper=False, left=False)       # u_new u_old^-1
    sv = torch.linalg.svdvals(a)
    smin, smax = sv[:, -1], sv[:, 0]
    assert bool(torch.all(g <= smin)) and bool(torch.all(g >= 0.95 * smin))
    assert bool(torch.all(big >= smax)) and bool(torch.all(big <= 1.08 * smax))


def test_sparse_path_ill_conditioned_components():
    """Rank-deficient initial covariances (sqrt(N) < D: W^-1 = singular sample covariance + 1e-5 I, condition ~1e9,
    whitening factors with entries in the hundreds): the int8 bound pass has to stay a bound there (its error term
    scales with the row and sample exponents), so the sparse path must still reproduce the dense statistics."""
    from bayesml_amd import _kside
    K, D, N = 8, 64, 3000
    x = orc.synth_gmm(K, D, N, np.float32)
    p = orc.Prior.default(K, D)
    q = orc.Posterior.from_prior(p)
    orc.init_subsampling(x.astype(np.float64), q, np.random.default_rng(0))
    dev = torch.device("cuda", 0)
    t = lambda v: torch.as_tensor(v, dtype=torch.float64, device=dev)   # noqa: E731
    qd = _kside.features(_kside.PostT(t(q.alpha), t(q.m), t(q.kappa), t(q.nu), t(q.w_inv)))
    assert float(qd.u.abs().max()) > 50.0            # the case is what it claims to be
    xd = torch.from_numpy(x).to(dev)
    pivot = xd.to(torch.float64).mean(dim=0)
    dense = _pass(xd, qd, DENSE, pivot)
    sparse = _pass(xd, qd, SPARSE, pivot)
    assert "_bound" in sparse[1][2]
    for (sa, ra, _), (sb, rb, _) in zip(dense, sparse):
        assert rel_err(sb, sa) < 1e-11
        assert np.max(np.abs(ra - rb)) < 1e-11


def test_profile_levels_record_the_same_results():
    """gmmvb_profile_enable: level 1 records a span per kernel group and the E/M phase events, level 2 only the groups that
    can dominate a step (and no phase events); neither changes a bit of what the pass returns."""
    from bayesml_amd._engine import DataPass
    K, D, N = 32, 96, 32_000
    x = orc.synth_gmm(K, D, N, np.float32)
    m = _fit(x, K, 6, DENSE)
    dev = torch.device("cuda", 0)
    from bayesml_amd import _kside
    hn = m.get_hn_params()
    t = lambda v: torch.as_tensor(v, dtype=torch.float64, device=dev)   # noqa: E731
    q = _kside.features(_kside.PostT(t(hn["hn_alpha_vec"]), t(hn["hn_m_vecs"]), t(hn["hn_kappas"]), t(hn["hn_nus"]),
                                     t(np.linalg.inv(hn["hn_w_mats"]))))
    xd = torch.from_numpy(x).to(dev)
    pivot = xd[:4096].to(torch.float64).mean(dim=0)
    got = {}
    for level in (0, 1, 2):
        with _env(SPARSE):
            eng = DataPass(K, D, xd.dtype, N, dev)
        eng.set_pivot(pivot)
        eng.prepare_rows(xd)
        eng.profile(level)
        eng.set_params(q.c, q.m, q.u)
        for _ in range(2):              # the second pass runs on carried records / lists
            stats = eng.estep_mstep(xd).clone()
        torch.cuda.synchronize()
        got[level] = (stats, eng.kernel_spans() if level else {}, eng.last_kernel_ms() if level else None)
        eng.close()
    assert torch.equal(got[0][0], got[1][0]) and torch.equal(got[0][0], got[2][0])
    full, light = got[1][1], got[2][1]
    assert set(light) <= {"estep_main", "estep_gather", "mstep_main"} and "mstep_main" in light
    assert set(light) < set(full) and "mstep_reduce" in full
    for g in light:
        assert light[g][1] == full[g][1] and light[g][0] > 0
    assert min(got[1][2]) > 0 and got[2][2] == (-1.0, -1.0)
