"""GPU parity of the HMM forward-backward kernels (through the C ABI) against the reference fixtures
(tests/golden/hmm_f6_*.npz) and against the oracle on seeded ragged shapes.  Tolerances: the engine is f64;
alpha/gamma are probabilities (absolute 1e-10), sums relative 1e-9 (north_star asks 1e-5 on hyper-parameters)."""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err
from oracle import hmm_vb_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["steps above 2^-80", "every step"])
def hmm_mstep_mode(request, monkeypatch):
    """The D <= 16 M-step of an HMM pass walks only the (four steps, state) blocks that hold a gamma above the relevance line
    (mstep.h, hmm_mstep_small_kernel<.., SPARSE>); GMMVB_HMM_MSTEP_DENSE walks all of them.  Reference fixtures and oracle
    comparisons run both ways."""
    if request.param == "every step":
        monkeypatch.setenv("GMMVB_HMM_MSTEP_DENSE", "1")
    else:
        monkeypatch.delenv("GMMVB_HMM_MSTEP_DENSE", raising=False)
    return request.param


def fixture_x(g):
    K, D, T = int(g["K"]), int(g["D"]), int(g["N"])
    if D == 2:
        return load_golden("hmm_c1_sample.npz")["x"]
    return orc.synth_hmm({16: 32 if K == 32 else 8}[D], D, T, np.dtype(str(g["x_dtype"])))[0]


def device_pass(x, q, dev):
    """One HMM data pass through the C ABI for oracle posterior `q`; returns numpy results."""
    from bayesml_amd import _kside
    from bayesml_amd._engine import DataPass
    K, D = q.m.shape
    T = x.shape[0]
    t = lambda a: torch.as_tensor(a, dtype=torch.float64, device=dev)   # noqa: E731
    f = _kside.features(_kside.PostT(torch.ones(K, dtype=torch.float64, device=dev), t(q.m), t(q.kappa), t(q.nu),
                                     t(q.w_inv)))
    c = (f.e_ln_lambda_det - D * _kside.LN_2PI - D / f.kappa) / 2.0
    xd = torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    eng = DataPass(K, D, xd.dtype, T, dev)
    eng.set_pivot(xd[:4096].to(torch.float64).mean(dim=0))
    eng.prepare_rows(xd)
    eng.set_params(c, f.m, f.u)
    eng.estep(xd)
    ln_rho = eng.ln_rho().cpu().numpy()
    eng.enable_hmm()
    ms, g0, gl, lnc = eng.forward_backward(t(q.pi_tilde), t(q.a_tilde))
    ns, h, a, B = eng.split_stats(eng.mstep(xd))
    x_bar, s = _kside.moments_from_stats(ns, a, B, eng.pivot, torch.zeros(K, D, D, dtype=torch.float64, device=dev))
    out = dict(ln_rho=ln_rho, ms=ms.cpu().numpy(), g0=g0.cpu().numpy(), gl=gl.cpu().numpy(), lnc=float(lnc),
               ns=ns.cpu().numpy(), h=float(h.sum()), x_bar=x_bar.cpu().numpy(), s=s.cpu().numpy(),
               gamma=eng.responsibilities().cpu().numpy(), alpha=eng.hmm_debug(0).cpu().numpy(),
               alpha_nat=eng.hmm_readout("alpha").cpu().numpy(), beta=eng.hmm_readout("beta").cpu().numpy(),
               xi=eng.hmm_readout("xi", 0, min(T, 300), t(q.a_tilde)).cpu().numpy(),
               xi_tail=eng.hmm_readout("xi", max(0, T - 7), min(T, 7), t(q.a_tilde)).cpu().numpy(),
               argmax=eng.argmax().cpu().numpy())
    eng.close()
    return out


@pytest.mark.parametrize("name", ["hmm_f6_k4_d2_t500.npz", "hmm_f6_k32_d16_t4096.npz"])
def test_forward_backward_matches_reference(name, hmm_mstep_mode):
    g = load_golden(name)
    x = fixture_x(g)
    q = orc.HmmPosterior(*(g["in_" + k].copy() for k in ("hn_eta_vec", "hn_zeta_vecs", "hn_m_vecs", "hn_kappas",
                                                         "hn_nus", "hn_w_mats", "hn_w_mats_inv"))).refresh()
    r = device_pass(x, q, torch.device("cuda", 0))
    n = g["ln_rho"].shape[0]
    assert rel_err(r["ln_rho"][:n], g["ln_rho"]) < 1e-11
    assert np.max(np.abs(r["alpha"][:n] - g["alpha_vecs"])) < 1e-10
    assert np.max(np.abs(r["gamma"][:n] - g["gamma_vecs"])) < 1e-10
    assert np.max(np.abs(r["gl"] - g["gamma_last"])) < 1e-10
    assert np.max(np.abs(r["g0"] - g["gamma_vecs"][0])) < 1e-10
    assert abs(r["lnc"] - float(g["ln_cs_sum"])) < 1e-9 * abs(float(g["ln_cs_sum"]))
    assert rel_err(r["ms"], g["ms"]) < 1e-9
    assert rel_err(r["ns"], g["ns"]) < 1e-9
    assert rel_err(r["x_bar"], g["x_bar_vecs"]) < 1e-9
    assert rel_err(r["s"], g["s_mats"]) < 1e-8
    # h = sum gamma ln rho is the first term of _vl_q_z; rebuild that term from the fixture's pieces
    K = int(g["K"])
    assert abs(r["ms"].sum() - (int(g["N"]) - 1)) < 1e-7          # every xi_t sums to one
    assert abs(r["ns"].sum() - int(g["N"])) < 1e-7
    # row-range read-outs of what the reference keeps as [T, K] / [T, K, K] attributes (ref:1063-1069)
    assert np.max(np.abs(r["alpha_nat"][:n] - g["alpha_vecs"])) < 1e-10
    big = g["alpha_vecs"] > 1e-200
    assert np.max(np.abs(r["beta"][:n][big] / g["beta_vecs"][big] - 1.0)) < 1e-8
    # xi_t = alpha_{t-1} rho_t a~ beta_t / c_t (ref:1016-1018), rebuilt from the fixture's own arrays
    m = r["xi"].shape[0]
    rho = np.exp(g["ln_rho"][:m])
    xi_ref = g["alpha_vecs"][:m - 1, :, None] * q.a_tilde[None] * (rho[1:] * g["beta_vecs"][1:m])[:, None, :] / g["cs"][1:m, None, None]
    assert np.all(r["xi"][0] == 0.0)
    assert np.max(np.abs(r["xi"][1:] - xi_ref)) < 1e-9
    assert np.max(np.abs(r["xi"][1:].sum(axis=(1, 2)) - 1.0)) < 1e-9
    assert np.max(np.abs(r["xi_tail"].sum(axis=1)[1:] - r["gamma"][-6:])) < 1e-9        # marginals: sum_i xi_t[i, j] = gamma_t[j]


@pytest.mark.parametrize("K,D,T,dtype", [(3, 2, 1, np.float64), (5, 3, 2, np.float64), (7, 4, 17, np.float32),
                                         (16, 8, 1000, np.float64), (20, 5, 3001, np.float32),
                                         (33, 6, 777, np.float64), (64, 4, 5000, np.float32), (2, 1, 40000, np.float64),
                                         (8, 4, 300001, np.float32), (33, 3, 270000, np.float64),
                                         (65, 3, 1, np.float64), (70, 3, 900, np.float64), (130, 2, 400, np.float32),
                                         (150, 2, 300, np.float64),
                                         # 65 .. 128 states over 2048 steps or more: the chunk-parallel kernels of hmm_wide.h
                                         (70, 3, 5000, np.float64), (96, 2, 2048, np.float32), (81, 2, 2305, np.float64),
                                         (112, 3, 3000, np.float32), (128, 2, 4100, np.float64),
                                         # ... and past 128 chunks of 256 steps their two-level boundary pass
                                         (70, 2, 40000, np.float64), (128, 2, 33500, np.float32)])
def test_ragged_shapes_against_oracle(K, D, T, dtype, hmm_mstep_mode):
    """K % 16 != 0 (padded states), T = 1, partial chunks, several chunk lengths - and, past 2^15 steps, the
    two-level boundary pass (chunks of 256 steps, super-chunk products); random posterior.  More than 64 states: the
    sequential kernels of csrc/hmm_generic.h (transition matrix in LDS up to K = 128, in L2 beyond) - and for 65 .. 128
    states over at least 2048 steps the chunk-parallel ones of csrc/hmm_wide.h (5 .. 8 tiles of 16 states, ragged last
    chunk, ragged last tile)."""
    rng = np.random.default_rng(100 * K + D)
    x, _ = orc.synth_hmm(max(2, K // 2), D, T, dtype, seed=K + T, stay=0.8)
    p = orc.HmmPrior.default(K, D)
    q = orc.HmmPosterior.from_prior(p)
    q.m = 3.0 * rng.standard_normal((K, D))
    a = rng.standard_normal((K, D, D))
    q.w_inv = a @ np.swapaxes(a, 1, 2) + D * np.eye(D)
    q.w = np.linalg.inv(q.w_inv)
    q.nu = q.nu + rng.uniform(0, 3, K)
    q.kappa = q.kappa + rng.uniform(0, 3, K)
    q.eta = q.eta + rng.uniform(0, 5, K)
    q.zeta = q.zeta + rng.uniform(0, 5, (K, K)) + 4 * np.eye(K)
    q.refresh()
    x64 = x.astype(np.float64)
    ln_rho = orc.emission_ln_rho(x64, q)
    mx = ln_rho.max(axis=1, keepdims=True)
    alpha, beta, cs = orc.forward_backward(np.exp(ln_rho - mx), q.pi_tilde, q.a_tilde)   # shifted: no underflow
    gamma = alpha * beta
    ms = np.zeros((K, K))
    rho = np.exp(ln_rho - mx)
    for t in range(1, T):
        ms += alpha[t - 1][:, None] * rho[t][None, :] * q.a_tilde * beta[t][None, :] / cs[t]
    r = device_pass(x, q, torch.device("cuda", 0))
    assert np.max(np.abs(r["alpha"] - alpha)) < 1e-10
    assert np.max(np.abs(r["gamma"] - gamma)) < 1e-10
    assert np.max(np.abs(r["g0"] - gamma[0])) < 1e-10 and np.max(np.abs(r["gl"] - gamma[-1])) < 1e-10
    assert abs(r["lnc"] - float((np.log(cs) + mx[:, 0]).sum())) < 1e-9 * max(1.0, abs(float((np.log(cs) + mx[:, 0]).sum())))
    if T > 1:
        assert rel_err(r["ms"], ms) < 1e-9
    else:
        assert np.all(r["ms"] == 0.0)
    assert rel_err(r["ns"], gamma.sum(axis=0)) < 1e-9
    assert abs(r["h"] - float((gamma * ln_rho).sum())) < 1e-9 * max(1.0, abs(float((gamma * ln_rho).sum())))
    assert np.array_equal(r["argmax"], np.argmax(gamma, axis=1)) or np.mean(r["argmax"] == np.argmax(gamma, axis=1)) > 0.999


DRIVER = ["hmm_f3_k4_subsampling.npz", "hmm_f3_k4_random_resp.npz", "hmm_f3_k8_d16_t8192_f32.npz", "hmm_f3_t1.npz"]


@pytest.mark.parametrize("name", DRIVER)
def test_full_driver_matches_reference(name, hmm_mstep_mode):
    import io
    import json
    import warnings
    from contextlib import redirect_stdout
    from bayesml_amd import ResultWarning
    from bayesml_amd import hiddenmarkovnormal as hmm
    g = load_golden(name)
    if "x" in g:
        x = g["x"]
    elif int(g["D"]) == 2:
        x = load_golden("hmm_c1_sample.npz")["x"]
    else:
        x = orc.synth_hmm(8, 16, int(g["N"]), np.float32)[0]
    K, D = int(g["K"]), int(g["D"])
    kw = json.loads(str(g["kw"]))
    m = hmm.LearnModel(K, D, seed=int(g["seed"]))
    buf = io.StringIO()
    with warnings.catch_warnings(record=True) as w, redirect_stdout(buf):
        warnings.simplefilter("always")
        m.update_posterior(x, **kw)
    assert any(issubclass(i.category, ResultWarning) for i in w) == bool(g["result_warning"])
    lines = [ln for ln in buf.getvalue().split("\n") if ln.strip()]
    tr = g["vl_trace"]
    assert len(lines) == tr.shape[0]
    assert max(i for i, ln in enumerate(lines) if ln.endswith("*")) == int(g["winner"])
    for i, ln in enumerate(lines):
        vals = [float(s.split("VL: ")[1].split(" ")[0].rstrip("*")) for s in ln.split("\r") if s]
        ref = tr[i][~np.isnan(tr[i])]
        assert len(vals) == len(ref) and np.allclose(vals, ref, rtol=1e-8)
    tol = 1e-7                                   # north_star asks for 1e-5
    for key in ("hn_eta_vec", "hn_zeta_vecs", "hn_m_vecs", "hn_kappas", "hn_nus", "hn_w_mats"):
        assert rel_err(m.get_hn_params()[key], g[key]) < tol, key
    assert rel_err(m.ns, g["ns"]) < tol
    if float(np.abs(g["ms"]).max()) > 0:
        assert rel_err(m.ms, g["ms"]) < tol
    assert np.max(np.abs(m.gamma_vecs[:64] - g["gamma_head"])) < 1e-7
    assert abs(m.vl - float(g["final_vl"])) <= 1e-8 * abs(float(g["final_vl"]))
    m.calc_pred_dist()
    assert rel_err(m.make_prediction("squared"), g["pred_squared"]) < 1e-6
    if "viterbi_01" in g:
        xs = x[:g["viterbi_01"].shape[0]]
        with redirect_stdout(io.StringIO()):
            assert np.array_equal(m.estimate_latent_vars(xs, "0-1", viterbi=True), g["viterbi_01"])
            assert np.array_equal(m.estimate_latent_vars(xs, "0-1", viterbi=False), g["marginal_01"])
            assert np.max(np.abs(m.estimate_latent_vars(xs, "squared", viterbi=False) - g["marginal_sq"])) < 1e-7


# T < 512: the single sequential wave; 512 <= T < 65536: the chunked max-plus scan with chunks of 32 steps; beyond: 256
@pytest.mark.parametrize("K,D,T", [(5, 3, 1), (12, 4, 511), (12, 4, 700), (40, 2, 3000), (64, 2, 1537), (32, 16, 200001),
                                   (66, 2, 1), (72, 2, 800), (140, 2, 500), (133, 2, 20000),
                                   # 65 .. 128 states over 2048 steps or more: the chunked pass of hmm_wide.h (two end states per
                                   # lane, four start states per wave, ragged last chunk / tile)
                                   (72, 2, 2048), (100, 3, 5001), (128, 2, 2700), (81, 2, 20000)])
def test_viterbi_kernel_against_oracle(K, D, T):
    from bayesml_amd import hiddenmarkovnormal as hmm
    x, _ = orc.synth_hmm(max(2, K // 2), D, T, np.float64, seed=7 * K + T)
    m = hmm.LearnModel(K, D, seed=1, verbose=False)
    rng = np.random.default_rng(K)
    m.set_hn_params(hn_eta_vec=rng.uniform(0.5, 5, K), hn_zeta_vecs=rng.uniform(0.5, 5, (K, K)) + 3 * np.eye(K),
                    hn_m_vecs=3.0 * rng.standard_normal((K, D)))
    q = orc.HmmPosterior(m.hn_eta_vec.copy(), m.hn_zeta_vecs.copy(), m.hn_m_vecs.copy(), m.hn_kappas.copy(),
                         m.hn_nus.copy(), m.hn_w_mats.copy(), m.hn_w_mats_inv.copy()).refresh()
    assert np.array_equal(m.estimate_latent_vars(x, "0-1", viterbi=True), orc.viterbi(x, q))


@pytest.mark.parametrize("K,D,T,dtype", [(32, 16, 4096, np.float32), (5, 3, 777, np.float64), (16, 16, 70001, np.float32),
                                         (17, 9, 1, np.float64), (32, 12, 263000, np.float32), (40, 16, 5000, np.float64),
                                         (64, 7, 3001, np.float32)])
def test_emission_into_the_forward_backward_buffers(K, D, T, dtype, hmm_mstep_mode):
    """hmmvb_emission_target(1): rho' and the row maxima straight from the emission kernel instead of the ln rho array
    followed by hmm_prep_kernel; no ln rho array (Viterbi / ln rho read-out refuse), h = 0 in the statistics and
    sum gamma ln rho from the moments (ref:905 against ref:871-877) to rounding."""
    from bayesml_amd import _kside
    from bayesml_amd._engine import DataPass, EngineError
    dev = torch.device("cuda", 0)
    x, _ = orc.synth_hmm(K, D, T, np.dtype(dtype), seed=3)
    rng = np.random.default_rng(5)
    t = lambda a: torch.as_tensor(a, dtype=torch.float64, device=dev)   # noqa: E731
    m = t(x[rng.integers(0, T, K)].astype(np.float64))
    g = rng.normal(size=(K, D, D)) * 0.2 + np.eye(D)
    w_inv = t(g @ g.transpose(0, 2, 1) * (D + 3.0))
    f = _kside.features(_kside.PostT(torch.ones(K, dtype=torch.float64, device=dev), m, t(rng.uniform(1, 3, K)),
                                     t(rng.uniform(D + 1, D + 6, K)), w_inv))
    c = (f.e_ln_lambda_det - D * _kside.LN_2PI - D / f.kappa) / 2.0
    f.c = c
    pi = t(rng.dirichlet(np.ones(K)))
    a = t(rng.dirichlet(np.ones(K), K))
    xd = torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    outs = []
    for fused in (False, True):
        eng = DataPass(K, D, xd.dtype, T, dev)
        eng.set_pivot(xd[:4096].to(torch.float64).mean(dim=0))
        eng.prepare_rows(xd)
        eng.enable_hmm()
        assert eng.emission_target(fused) == (fused and K <= 32)         # (in effect up to 32 states, D <= 16)
        fused = eng.emission_fused
        eng.set_params(c, f.m, f.u)
        eng.estep(xd)
        if fused:
            with pytest.raises(EngineError):
                eng.viterbi(torch.log(pi), torch.log(a))
            with pytest.raises(EngineError):
                eng.ln_rho(0, 1)
        ms, g0, gl, lnc = eng.forward_backward(pi, a)
        stats = eng.mstep(xd).clone()
        ns, h, am, B = eng.split_stats(stats)
        x_bar, s = _kside.moments_from_stats(ns, am, B, eng.pivot, torch.zeros(K, D, D, dtype=torch.float64, device=dev))
        closed = float(_kside.sum_gamma_ln_rho(f, ns, x_bar, s))
        if not fused:                  # hmmvb_skip_h: the same statistics with h = 0
            eng.hmm_skip_h(True)
            st2 = eng.mstep(xd).clone()
            eng.hmm_skip_h(False)
            assert torch.equal(st2[:K], stats[:K]) and torch.equal(st2[2 * K:], stats[2 * K:]) and float(st2[K:2 * K].abs().sum()) == 0.0
        outs.append(dict(ms=ms.clone(), g0=g0.clone(), gl=gl.clone(), lnc=float(lnc), stats=stats, h=float(h.sum()),
                         closed=closed, gamma=eng.responsibilities(max(0, T - 500), min(T, 500)).clone(),
                         alpha=eng.hmm_readout("alpha", 0, min(T, 300)).clone()))
        if fused:                      # back to the array: Viterbi works again on the same workspace
            assert eng.emission_target(False) is False
            eng.estep(xd)
            z = eng.viterbi(torch.log(pi), torch.log(a))
            assert z.shape[0] == T
        eng.close()
    u, v = outs
    # (the fused emission runs on the matrix pipe, the array form on the vector ALU: same values to rounding)
    for k in ("ms", "g0", "gl", "gamma", "alpha"):
        assert float((u[k] - v[k]).abs().max()) <= 1e-11 * max(1.0, float(u[k].abs().max())), k
    assert abs(u["lnc"] - v["lnc"]) <= 1e-12 * abs(u["lnc"]) + 1e-9
    assert v["h"] == 0.0 or K > 32
    hs = 2 * K
    for lo, hi in ((0, K), (hs, None)):
        a_, b_ = u["stats"][lo:hi], v["stats"][lo:hi]
        assert float((a_ - b_).abs().max()) <= 1e-10 * float(a_.abs().max())
    assert abs(u["closed"] - u["h"]) <= 1e-11 * abs(u["h"]) + 1e-9
    assert abs(v["closed"] - u["h"]) <= 1e-11 * abs(u["h"]) + 1e-9


@pytest.mark.parametrize("K,D,flat", [(32, 16, False), (8, 3, False), (5, 2, True), (48, 8, False), (64, 5, False),
                                      (96, 8, False), (72, 4, True), (130, 4, False), (150, 3, True), (6, 2, "cycle")])
def test_boundary_vectors_by_forgetting(K, D, flat, monkeypatch):
    """Sequences past 2^15 steps: chunk boundary vectors from sweeps started at the uniform vector (the scaled recursions
    forget their start), checked against the replays' own and replaced by the chunk-product path when they do not stand.
    Informative emissions: the pass stands (0) and the results are those of the products path; flat emissions and a sticky
    chain: the gate opens (1), the products path runs behind it - same results -, and the next calls go straight to it (-1)."""
    from bayesml_amd import _kside
    from bayesml_amd._engine import DataPass
    dev = torch.device("cuda", 0)
    T = 270001
    x, _ = orc.synth_hmm(K, D, T, np.float32, seed=11)
    rng = np.random.default_rng(13)
    t = lambda a: torch.as_tensor(a, dtype=torch.float64, device=dev)   # noqa: E731
    m = t(x[rng.integers(0, T, K)].astype(np.float64))
    cycle = flat == "cycle"                             # (a nearly deterministic cycle under flat emissions: no forgetting)
    flat = bool(flat)
    scale = 4000.0 if flat else 1.0                     # (flat: every component covers the whole data set)
    w_inv = t(np.broadcast_to(np.eye(D) * (D + 3.0) * scale, (K, D, D)).copy())
    f = _kside.features(_kside.PostT(torch.ones(K, dtype=torch.float64, device=dev), m, t(np.full(K, 2.0)),
                                     t(np.full(K, D + 3.0)), w_inv))
    c = (f.e_ln_lambda_det - D * _kside.LN_2PI - D / f.kappa) / 2.0
    stay = 0.9999 if flat else 0.9
    a = t(np.eye(K) * stay + (1.0 - stay) / K)
    if cycle:
        a = t(np.roll(np.eye(K), 1, axis=1) * 0.9999 + 0.0001 / K)
    pi = t(rng.dirichlet(np.ones(K)))
    xd = torch.from_numpy(np.ascontiguousarray(x)).to(dev)

    def run(forgetting, calls=1):
        if forgetting:
            monkeypatch.delenv("GMMVB_HMM_FORGETTING_OFF", raising=False)
        else:
            monkeypatch.setenv("GMMVB_HMM_FORGETTING_OFF", "1")
        eng = DataPass(K, D, xd.dtype, T, dev)
        eng.set_pivot(xd[:4096].to(torch.float64).mean(dim=0))
        eng.prepare_rows(xd)
        eng.enable_hmm()
        eng.set_params(c, f.m, f.u)
        how = []
        for _ in range(calls):
            eng.estep(xd)
            ms, g0, gl, lnc = eng.forward_backward(pi, a)
            how.append(eng.last_boundary_pass())
        stats = eng.mstep(xd).clone()
        out = dict(ms=ms.clone(), g0=g0.clone(), gl=gl.clone(), lnc=float(lnc), stats=stats,
                   gamma=torch.cat([eng.responsibilities(0, 600), eng.responsibilities(T // 2, 600),
                                    eng.responsibilities(T - 600, 600)]).clone(),
                   alpha=eng.hmm_readout("alpha", 255, 600).clone())
        eng.close()
        return out, how

    ref, how0 = run(False)
    got, how1 = run(True, calls=4)
    assert how0 == [-1]
    # Up to 64 states the chunks start at 32 steps (a sequence of this length leaves the CUs idle on longer ones) and double
    # after a pass that did not stand, up to 256, where the hold-off begins; from 128 steps on the pass has two stages (sweeps
    # over the 64 steps next to every boundary, then whole chunks behind a gate of its own).  "Stood" = any of that.
    if K <= 64:
        assert all(v in (0, 1) for v in how1) and how1[0] == (1 if flat else how1[0]), how1
        assert (how1[-1] == 1) if flat else (how1[-1] == 0), how1
    else:
        assert how1 == ([1, -1, -1, -1] if flat else [0, 0, 0, 0]), how1
    for k in ("ms", "g0", "gl", "gamma", "alpha", "stats"):
        scale_k = max(1.0, float(ref[k].abs().max()))
        assert float((ref[k] - got[k]).abs().max()) <= 1e-10 * scale_k, k
    assert abs(ref["lnc"] - got["lnc"]) <= 1e-11 * abs(ref["lnc"])


def test_forgetting_gate_sees_an_unreachable_state(monkeypatch):
    """Round-4 advisor's counter-example to an ABSOLUTE boundary test: two states, a~_01 = 1e-40 (exp(psi(0.01)) ~ 1e-44 is
    reachable with a sparse h0_zeta prior), a~_10 = 0.02, emission means 0 and 0.8.  The exact alpha_t(1) is ~1e-40 while the
    chain sits in state 0; a sweep restarted from the uniform vector leaves ~1e-19 there after 128 steps - invisible to an
    absolute 2e-14 test - and a following stretch of 110 steps at x = 1.14 (likelihood ratio e^0.59 per step: 1e28 in all)
    amplifies exactly that entry to O(1): gamma wrong by O(1) with the gate shut.  The gate's test is relative per entry
    (Hilbert metric, hmm.h hmm_boundary_check_kernel<true>): it must open, and the result must be the chunk-product path's."""
    from bayesml_amd import _kside
    from bayesml_amd._engine import DataPass
    dev = torch.device("cuda", 0)
    K, D, T = 2, 1, 270001
    rng = np.random.default_rng(5)
    x = rng.standard_normal((T, D))
    for c0 in (300, 600, 900):                           # stretches that start on chunk boundaries of 128 and 256 steps
        x[1 + 256 * c0: 1 + 256 * c0 + 110] = 1.14
    x = x.astype(np.float32)
    t = lambda a: torch.as_tensor(a, dtype=torch.float64, device=dev)   # noqa: E731
    f = _kside.features(_kside.PostT(torch.ones(K, dtype=torch.float64, device=dev), t([[0.0], [0.8]]), t(np.full(K, 2.0)),
                                     t(np.full(K, D + 3.0)), t(np.full((K, 1, 1), D + 3.0))))
    c = (f.e_ln_lambda_det - D * _kside.LN_2PI - D / f.kappa) / 2.0
    a = t([[1.0, 1e-40], [0.02, 0.98]])
    pi = t([0.999, 0.001])
    xd = torch.from_numpy(x).to(dev)
    res = []
    for off in (True, False):
        if off:
            monkeypatch.setenv("GMMVB_HMM_FORGETTING_OFF", "1")
        else:
            monkeypatch.delenv("GMMVB_HMM_FORGETTING_OFF", raising=False)
        eng = DataPass(K, D, xd.dtype, T, dev)
        eng.set_pivot(xd[:4096].to(torch.float64).mean(dim=0))
        eng.prepare_rows(xd)
        eng.enable_hmm()
        eng.set_params(c, f.m, f.u)
        eng.estep(xd)
        ms, g0, gl, lnc = eng.forward_backward(pi, a)
        how = eng.last_boundary_pass()
        seg = torch.cat([eng.responsibilities(1 + 256 * c0 - 50, 400) for c0 in (300, 600, 900)]).clone()
        res.append((how, ms.clone(), seg, float(lnc)))
        eng.close()
    (how0, ms0, seg0, lnc0), (how1, ms1, seg1, lnc1) = res
    assert how0 == -1 and how1 == 1, (how0, how1)        # the pass must NOT stand here
    # the exact answer: the chain stays in state 0 through the stretches (1e-40 x 1e28 is still 1e-12)
    assert float(seg0[:, 1].max()) < 1e-6
    assert float((seg0 - seg1).abs().max()) <= 1e-12
    assert float((ms0 - ms1).abs().max()) <= 1e-9 * float(ms0.abs().max())
    assert abs(lnc0 - lnc1) <= 1e-12 * abs(lnc0)


def test_viterbi_invalidates_the_forward_backward_read_outs():
    """hmmvb_viterbi takes the forward-backward pass's buffers as scratch: a read-out of that pass afterwards is refused
    (explicitly: gamma_rows = 0), whatever state the E-step flags are in."""
    from bayesml_amd import _kside
    from bayesml_amd._engine import DataPass, EngineError
    dev = torch.device("cuda", 0)
    K, D, T = 4, 2, 5000
    x, _ = orc.synth_hmm(K, D, T, np.float32, seed=3)
    t = lambda a: torch.as_tensor(a, dtype=torch.float64, device=dev)   # noqa: E731
    rng = np.random.default_rng(1)
    f = _kside.features(_kside.PostT(torch.ones(K, dtype=torch.float64, device=dev), t(x[rng.integers(0, T, K)].astype(np.float64)),
                                     t(np.full(K, 2.0)), t(np.full(K, D + 3.0)),
                                     t(np.broadcast_to(np.eye(D) * (D + 3.0), (K, D, D)).copy())))
    c = (f.e_ln_lambda_det - D * _kside.LN_2PI - D / f.kappa) / 2.0
    a = np.eye(K) * 0.9 + 0.1 / K
    xd = torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    eng = DataPass(K, D, xd.dtype, T, dev)
    eng.set_pivot(xd[:4096].to(torch.float64).mean(dim=0))
    eng.prepare_rows(xd)
    eng.enable_hmm()
    eng.set_params(c, f.m, f.u)
    eng.estep(xd)
    eng.forward_backward(t(np.full(K, 1.0 / K)), t(a))
    assert eng.hmm_readout("xi", 10, 5, a_tilde=t(a)).shape == (5, K, K)
    eng.estep(xd)
    eng.viterbi(t(np.log(np.full(K, 1.0 / K))), t(np.log(a)))
    with pytest.raises(EngineError):
        eng.hmm_readout("xi", 10, 5, a_tilde=t(a))
    eng.close()


def test_update_posterior_through_the_forgetting_pass(monkeypatch):
    """The whole VB loop of a long sequence (public API: fused emission, closed-form sum gamma ln rho, xi in the backward replay,
    boundary vectors by forgetting) against the same loop on the chunk-product path."""
    from bayesml_amd import hiddenmarkovnormal as hmm
    K, D, T = 5, 3, 270001
    x = orc.synth_hmm(K, D, T, np.float64, seed=21)[0]
    res = []
    for off in (True, False):
        if off:
            monkeypatch.setenv("GMMVB_HMM_FORGETTING_OFF", "1")
        else:
            monkeypatch.delenv("GMMVB_HMM_FORGETTING_OFF", raising=False)
        m = hmm.LearnModel(K, D, seed=3, device="cuda:0", verbose=False)
        import io
        from contextlib import redirect_stdout
        with redirect_stdout(io.StringIO()):
            m.update_posterior(x, max_itr=6, num_init=1, tolerance=0.0)
        res.append((m.get_hn_params(), m._engine.last_boundary_pass() if getattr(m, "_engine", None) is not None else None))
    (p0, how0), (p1, how1) = res
    assert how0 == -1 and how1 == 0, (how0, how1)
    for k in p0:
        a, b = np.asarray(p0[k]), np.asarray(p1[k])
        assert np.max(np.abs(a - b)) <= 1e-9 * max(1.0, float(np.max(np.abs(a)))), k


@pytest.mark.parametrize("K,D,flat", [(32, 16, False), (6, 2, True), (48, 3, False), (12, 4, "cycle"), (96, 3, False),
                                      (128, 2, False), (72, 2, True), (130, 2, False), (140, 2, True)])
def test_viterbi_chunk_starts_by_coalescence(K, D, flat, monkeypatch):
    """hmmvb_viterbi on 65536 steps or more: chunk start vectors from a sweep of the max-plus recursion started at zero (best paths
    merge inside a chunk), checked against the replay's own; the chunk-matrix path behind a gate otherwise.  Same path either way."""
    from bayesml_amd import _kside
    from bayesml_amd._engine import DataPass
    dev = torch.device("cuda", 0)
    T = 150001
    x, _ = orc.synth_hmm(K, D, T, np.float32, seed=17)
    rng = np.random.default_rng(19)
    t = lambda a: torch.as_tensor(a, dtype=torch.float64, device=dev)   # noqa: E731
    cycle = flat == "cycle"
    flat = bool(flat)
    m = t(x[rng.integers(0, T, K)].astype(np.float64))
    w_inv = t(np.broadcast_to(np.eye(D) * (D + 3.0) * (1e7 if flat else 1.0), (K, D, D)).copy())
    f = _kside.features(_kside.PostT(torch.ones(K, dtype=torch.float64, device=dev), m, t(np.full(K, 2.0)),
                                     t(np.full(K, D + 3.0)), w_inv))
    c = (f.e_ln_lambda_det - D * _kside.LN_2PI - D / f.kappa) / 2.0
    a = np.eye(K) * 0.9 + 0.1 / K
    if cycle:
        a = np.roll(np.eye(K), 1, axis=1) * 0.999999 + 1e-6 / K
    ln_a, ln_pi = t(np.log(a)), t(np.log(rng.dirichlet(np.ones(K))))
    xd = torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    out = []
    for off in (True, False):
        if off:
            monkeypatch.setenv("GMMVB_HMM_FORGETTING_OFF", "1")
        else:
            monkeypatch.delenv("GMMVB_HMM_FORGETTING_OFF", raising=False)
        eng = DataPass(K, D, xd.dtype, T, dev)
        eng.set_pivot(xd[:4096].to(torch.float64).mean(dim=0))
        eng.prepare_rows(xd)
        eng.enable_hmm()
        eng.set_params(c, f.m, f.u)
        eng.estep(xd)
        z = eng.viterbi(ln_pi, ln_a).clone()
        out.append((z, eng.last_viterbi_pass()))
        eng.close()
    (z0, how0), (z1, how1) = out
    assert how0 == -1 and how1 == (1 if flat else 0), (how0, how1)
    assert torch.equal(z0, z1)


@pytest.mark.parametrize("K,D", [(32, 16), (5, 3), (64, 7), (40, 33)])
def test_fused_k_side_against_the_torch_specification(K, D, monkeypatch):
    """HmmKStepper on the GPU - gmmvb_kside_step on views of the HMM posterior for the Normal-Wishart half,
    hmmvb_kside_dirichlet for eta / zeta - against the torch functions of _kside.py (BAYESML_AMD_KSIDE_FUSED=0): the ten terms
    of the lower bound and every field of the next posterior, over chained steps (an empty state among them)."""
    from bayesml_amd import _kside
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(K * 100 + D)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64, device=dev)   # noqa: E731
    g = rng.normal(size=(K, D, D)) * 0.3 + np.eye(D)
    prior = _kside.hmm_prior_from_numpy(rng.uniform(0.3, 2.0, K), rng.uniform(0.3, 2.0, (K, K)), rng.normal(size=(K, D)),
                                        rng.uniform(0.5, 2.0, K), rng.uniform(D, D + 4.0, K),
                                        np.linalg.inv(g @ g.transpose(0, 2, 1)), dev)
    pivot = t(rng.normal(size=D))
    stats_len = K * (2 + D + D * D)
    steppers = []
    for fused in ("1", "0"):
        monkeypatch.setenv("BAYESML_AMD_KSIDE_FUSED", fused)
        steppers.append(_kside.HmmKStepper(prior, pivot, stats_len))
    assert steppers[0]._fused and not steppers[1]._fused
    for it in range(4):
        # consistent statistics: weighted moments of real points about the pivot (S = B / ns - (a / ns)(a / ns)^T must be PSD)
        pts = rng.normal(size=(6 * D + 40, D)) * 2.0 + rng.normal(size=D)
        r = rng.uniform(0.0, 1.0, (pts.shape[0], K)) * (rng.uniform(size=(pts.shape[0], K)) < 0.4)
        if it == 1:
            r[:, K // 2] = 0.0                        # an empty state: the ns > 0 guard and the stale S
        xc = pts - pivot.cpu().numpy()
        ns = r.sum(axis=0)
        a = r.T @ xc
        B = np.einsum("nk,ni,nj->kij", r, xc, xc)
        stats = np.concatenate([ns, np.zeros(K), a.reshape(-1), B.reshape(-1)])
        ms = rng.uniform(0.0, 5.0, (K, K))
        fb = np.concatenate([ms.reshape(-1), rng.dirichlet(np.ones(K)), rng.dirichlet(np.ones(K)), [rng.normal() * 100.0]])
        for ks in steppers:
            ks.stats.copy_(t(stats))
            ks.fb.copy_(t(fb))
            ks.h_scale.fill_(0.0 if it == 2 else 1.0)
            ks.step()
        u, v = steppers
        su, sv = u.scal.cpu().numpy(), v.scal.cpu().numpy()
        # (p_x and q_z cancel in vl - sum gamma ln rho IS E[ln p(x|z)] in closed form -: rounding is relative to the largest term)
        assert np.max(np.abs(su - sv)) <= 1e-12 * np.max(np.abs(sv)), (it, su, sv)
        for f in _kside._HMM_POST_FIELDS:
            x, y = getattr(u.q_next, f), getattr(v.q_next, f)
            assert float((x - y).abs().max()) <= 1e-10 * max(1.0, float(y.abs().max())), (it, f)
        for name in ("ns", "x_bar", "s"):
            x, y = getattr(u, name), getattr(v, name)
            assert float((x - y).abs().max()) <= 1e-11 * max(1.0, float(y.abs().max())), (it, name)
        for ks in steppers:
            ks.advance()


@pytest.mark.parametrize("T", [4096, 20000, 32768, 70001])
def test_forgetting_pass_equals_chunk_products_with_sparse_transitions(T, monkeypatch):
    """The forgetting pass's start vectors are only proven within chunks x tolerance in the Hilbert metric (relative per
    entry): with near-absorbing rows of a~ (exp(psi) of a sparse h0_zeta prior), two DUPLICATE states
    (same emission: the recursion cannot tell them apart, only a~ does) and lengths on both sides of the 2^15-step switch
    between its chunk plans, gamma, the xi sum and sum ln c must still equal the chunk-product path's
    (GMMVB_HMM_FORGETTING_OFF) far inside the 1e-5 contract: 1e-10 relative."""
    from bayesml_amd import _kside
    from bayesml_amd._engine import DataPass
    dev = torch.device("cuda", 0)
    K, D = 6, 3
    rng = np.random.default_rng(11)
    mu = 2.5 * rng.standard_normal((K, D))
    mu[5] = mu[2]                                         # duplicate states
    a_np = rng.dirichlet(np.full(K, 0.05), K) + 1e-30
    a_np[np.arange(K), np.arange(K)] += 3.0               # sticky, rows far from normalised: a~ need not be
    z = np.zeros(T, dtype=np.int64)
    for i in range(1, T):
        z[i] = z[i - 1] if rng.random() < 0.97 else rng.integers(0, K)
    x = (mu[z] + rng.standard_normal((T, D))).astype(np.float32)
    t = lambda v: torch.as_tensor(v, dtype=torch.float64, device=dev)   # noqa: E731
    f = _kside.features(_kside.PostT(torch.ones(K, dtype=torch.float64, device=dev), t(mu), t(np.full(K, 2.0)),
                                     t(np.full(K, D + 3.0)), t(np.tile(np.eye(D) * (D + 3.0), (K, 1, 1)))))
    c = (f.e_ln_lambda_det - D * _kside.LN_2PI - D / f.kappa) / 2.0
    a, pi = t(a_np), t(np.full(K, 1.0 / K))
    xd = torch.from_numpy(x).to(dev)
    res = []
    for off in (True, False):
        if off:
            monkeypatch.setenv("GMMVB_HMM_FORGETTING_OFF", "1")
        else:
            monkeypatch.delenv("GMMVB_HMM_FORGETTING_OFF", raising=False)
        eng = DataPass(K, D, xd.dtype, T, dev)
        eng.set_pivot(xd[:4096].to(torch.float64).mean(dim=0))
        eng.prepare_rows(xd)
        eng.enable_hmm()
        eng.set_params(c, f.m, f.u)
        eng.estep(xd)
        ms, g0, gl, lnc = eng.forward_backward(pi, a)
        res.append((eng.last_boundary_pass(), ms.clone(), eng.responsibilities().clone(), float(lnc)))
        eng.close()
    (how0, ms0, gam0, lnc0), (_how1, ms1, gam1, lnc1) = res
    assert how0 == -1
    assert float((ms0 - ms1).abs().max()) <= 1e-10 * float(ms0.abs().max())
    assert float((gam0 - gam1).abs().max()) <= 1e-10
    assert abs(lnc0 - lnc1) <= 1e-10 * abs(lnc0)
