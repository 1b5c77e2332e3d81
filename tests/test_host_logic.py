"""Host-side logic of the drop-in on CPU: validators/exceptions, K-side math, restart driver, RNG
consumption order and stdout protocol.  The N-sized data pass is supplied by tests/fake_engine.py
(a CPU stand-in injected through the private test seam) so these run without a GPU; the HIP path
itself is covered by tests/test_gpu_parity.py (-m gpu)."""
import io
import json
import os
import pickle
import warnings
from contextlib import redirect_stdout

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_golden, rel_err
from fake_engine import cpu_factory
from oracle import gmm_vb_oracle as orc

import bayesml_amd
from bayesml_amd import _kside
from bayesml_amd import gaussianmixture as gm


def cpu_model(*a, **k):
    m = gm.LearnModel(*a, **k)
    m._data_pass_factory = cpu_factory
    return m


def quiet(fn, *a, **k):
    with redirect_stdout(io.StringIO()), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return fn(*a, **k)


# ------------------------------------------------------------------ F5: boundary errors
def _error_cases():
    rng = np.random.default_rng(0)
    return {
        "ctor_float_degree": lambda: gm.LearnModel(3, 2.0),
        "ctor_zero_classes": lambda: gm.LearnModel(0, 2),
        "ctor_bool_like_negative": lambda: gm.LearnModel(3, -1),
        "h0_m_vecs_wrong_dim": lambda: gm.LearnModel(3, 2, h0_m_vecs=np.zeros((3, 3))),
        "h0_w_mats_not_pd": lambda: gm.LearnModel(3, 2, h0_w_mats=np.array([[1.0, 2.0], [2.0, 1.0]])),
        "h0_w_mats_not_sym": lambda: gm.LearnModel(3, 2, h0_w_mats=np.array([[1.0, 0.5], [0.0, 1.0]])),
        "h0_nus_too_small": lambda: gm.LearnModel(3, 2, h0_nus=1.0),
        "h0_alpha_nonpos": lambda: gm.LearnModel(3, 2, h0_alpha_vec=np.array([1.0, 0.0, 1.0])),
        "h0_kappas_negative": lambda: gm.LearnModel(3, 2, h0_kappas=-1.0),
        "x_wrong_last_dim": lambda: quiet(cpu_model(3, 2).update_posterior, np.zeros((10, 3))),
        "x_not_ndarray": lambda: quiet(cpu_model(3, 2).update_posterior, [[0.0, 1.0]]),
        "x_complex": lambda: quiet(cpu_model(3, 2).update_posterior, np.zeros((4, 2), dtype=complex)),
        "bad_init_type": lambda: quiet(cpu_model(3, 2, seed=0).update_posterior, rng.standard_normal((50, 2)),
                                       init_type="kmeans"),
        "bad_loss_estimate_params": lambda: gm.LearnModel(3, 2).estimate_params("L1"),
        "bad_loss_make_prediction": lambda: gm.LearnModel(3, 2).make_prediction("KL"),
        "bad_loss_latent": lambda: cpu_model(3, 2).estimate_latent_vars(np.zeros((4, 2)), "L1"),
        "pred_and_update_wrong_shape": lambda: quiet(cpu_model(3, 2).pred_and_update, np.zeros((1, 2))),
        "gen_pi_not_sum1": lambda: gm.GenModel(3, 2, pi_vec=np.array([0.5, 0.4, 0.2])),
        "gen_sample_size_float": lambda: gm.GenModel(3, 2).gen_sample(10.0),
        "visualize_d3": lambda: quiet(gm.LearnModel(2, 3).visualize_posterior),
        "ctor_numpy_int": lambda: gm.LearnModel(np.int64(3), np.int32(2)),
        "h0_scalar_broadcast": lambda: gm.LearnModel(3, 2, h0_kappas=2.0, h0_nus=3, h0_w_mats=np.eye(2) * 2),
        "x_int_dtype": lambda: quiet(cpu_model(2, 2, seed=0).update_posterior,
                                     np.random.default_rng(0).integers(-5, 5, (40, 2)), num_init=1, max_itr=2),
        "x_3d_reshaped": lambda: quiet(cpu_model(2, 2, seed=0).update_posterior,
                                       np.random.default_rng(0).standard_normal((5, 8, 2)), num_init=1, max_itr=2),
    }


def test_boundary_errors_match_reference_classes():
    with open(os.path.join(GOLDEN, "gmm_errors.json")) as f:
        expected = json.load(f)
    cases = _error_cases()
    assert set(cases) == set(expected)
    for name, fn in cases.items():
        try:
            fn()
            got = None
        except Exception as e:      # noqa: BLE001
            got = type(e).__name__
        assert got == expected[name], (name, got, expected[name])


def test_exception_classes_and_str():
    e = bayesml_amd.ParameterFormatError("bad")
    assert str(e) == "'bad'" and e.value == "bad"
    assert issubclass(bayesml_amd.ResultWarning, UserWarning)


# ------------------------------------------------------------------ API surface
def test_constructor_defaults_and_dict_key_order():
    m = gm.LearnModel(4, 3)
    assert list(m.get_h0_params()) == ["h0_alpha_vec", "h0_m_vecs", "h0_kappas", "h0_nus", "h0_w_mats"]
    assert list(m.get_hn_params()) == ["hn_alpha_vec", "hn_m_vecs", "hn_kappas", "hn_nus", "hn_w_mats"]
    assert list(m.get_p_params()) == ["p_mu_vecs", "p_nus", "p_lambda_mats"]
    assert m.get_constants() == {"c_num_classes": 4, "c_degree": 3}
    assert np.all(m.h0_alpha_vec == 0.5) and np.all(m.h0_nus == 3) and np.all(m.h0_kappas == 1)
    assert np.array_equal(m.h0_w_mats, np.tile(np.eye(3), (4, 1, 1)))
    assert m.get_hn_params()["hn_m_vecs"] is m.hn_m_vecs          # getters return live arrays
    assert m.r_vecs is None
    g = gm.GenModel(4, 3)
    assert list(g.get_h_params()) == ["h_alpha_vec", "h_m_vecs", "h_kappas", "h_nus", "h_w_mats"]
    assert list(g.get_params()) == ["pi_vec", "mu_vecs", "lambda_mats"]


def test_pickle_roundtrips_are_positional(tmp_path):
    m = gm.LearnModel(3, 2, h0_kappas=2.0, h0_nus=np.array([3.0, 4.0, 5.0]))
    f = str(tmp_path / "h0.pkl")
    m.save_h0_params(f)
    m2 = gm.LearnModel(3, 2).load_h0_params(f)
    assert np.array_equal(m2.h0_nus, [3.0, 4.0, 5.0]) and np.all(m2.h0_kappas == 2.0)
    m2.load_hn_params(f)                   # an h0 dict loads into hn positionally (reference base.py:251)
    assert np.array_equal(m2.hn_nus, [3.0, 4.0, 5.0])
    with open(f, "wb") as fh:
        pickle.dump([1, 2], fh)
    with pytest.raises(bayesml_amd.ParameterFormatError):
        m2.load_h0_params(f)
    g = gm.GenModel(3, 2, seed=1)
    g.gen_params()
    fp = str(tmp_path / "p.pkl")
    g.save_params(fp)
    g2 = gm.GenModel(3, 2).load_params(fp)
    assert np.array_equal(g2.mu_vecs, g.mu_vecs)


def test_gen_sample_reproduces_reference_stream():
    """Same seed, same per-row call order => the fixture the reference's GenModel generated."""
    gen = gm.GenModel(3, 2, pi_vec=np.array([0.3, 0.3, 0.4]),
                      mu_vecs=np.array([[-4.0, 0.0], [4.0, 0.0], [0.0, 5.0]]),
                      lambda_mats=np.array([[[1.0, 0.3], [0.3, 2.0]], [[2.0, 0.0], [0.0, 0.5]], [[1.0, -0.4], [-0.4, 1.0]]]),
                      seed=123)
    x, z = gen.gen_sample(1000)
    ref = load_golden("gmm_c1_sample.npz")
    assert np.array_equal(z, ref["z"])
    assert np.allclose(x, ref["x"], rtol=1e-13, atol=1e-13)


def test_subsample_index_draw_consumes_stream_like_row_draw():
    """rng.choice(x, n, replace=False, axis=0, shuffle=False) == x[rng.choice(N, n, replace=False, shuffle=False)]."""
    x = np.random.default_rng(3).standard_normal((1000, 4))
    a, b = np.random.default_rng(7), np.random.default_rng(7)
    for _ in range(5):
        rows = a.choice(x, size=31, replace=False, axis=0, shuffle=False)
        idx = b.choice(1000, size=31, replace=False, shuffle=False)
        assert np.array_equal(rows, x[idx])
    assert a.random() == b.random()


# ------------------------------------------------------------------ F2: K-side math on CPU tensors
F1 = ["gmm_f1_c1_k3_d2_n1000.npz", "gmm_f1_k16_d32_n2048.npz", "gmm_f1_k4_d128_n32768_f32.npz"]


@pytest.mark.parametrize("name", F1)
def test_k_side_update_matches_reference(name):
    g = load_golden(name)
    K, D = int(g["K"]), int(g["D"])
    m = gm.LearnModel(K, D)
    p = m._prior_tensors("cpu")
    assert abs(p.ln_c_alpha - float(orc.Prior.default(K, D).ln_c_alpha)) < 1e-12
    t = lambda a: torch.as_tensor(a, dtype=torch.float64)   # noqa: E731
    q = _kside.update_q(p, t(g["ns"]), t(g["x_bar_vecs"]), t(g["s_mats"]))
    for mine, key, tol in ((q.alpha, "hn_alpha_vec", 1e-14), (q.m, "hn_m_vecs", 1e-13), (q.kappa, "hn_kappas", 1e-14),
                           (q.nu, "hn_nus", 1e-14), (q.w_inv, "hn_w_mats_inv", 1e-13), (q.w, "hn_w_mats", 1e-9),
                           (q.e_ln_pi, "e_ln_pi_vec", 1e-12), (q.e_ln_lambda_det, "e_ln_lambda_dets", 1e-11),
                           (q.nu[:, None, None] * q.w, "e_lambda_mats", 1e-9), (q.ln_b_w_nu, "ln_b_hn_w_nus", 1e-11)):
        assert rel_err(mine.numpy(), g["out_" + key]) < tol, key
    # whitening factor: u^T u = E[Lambda]
    assert rel_err((q.u.transpose(1, 2) @ q.u).numpy(), g["out_e_lambda_mats"]) < 1e-9
    assert float(torch.triu(q.u, diagonal=1).abs().max()) == 0.0


@pytest.mark.parametrize("name", F1[:2])
def test_lower_bound_matches_reference(name):
    g = load_golden(name)
    K, D = int(g["K"]), int(g["D"])
    m = gm.LearnModel(K, D)
    p = m._prior_tensors("cpu")
    t = lambda a: torch.as_tensor(a, dtype=torch.float64)   # noqa: E731
    q = _kside.features(_kside.PostT(t(g["in_hn_alpha_vec"]), t(g["in_hn_m_vecs"]), t(g["in_hn_kappas"]),
                                     t(g["in_hn_nus"]), t(g["in_hn_w_mats_inv"])))
    terms = _kside.lower_bound(p, q, t(g["ns"]), t(g["x_bar_vecs"]), t(g["s_mats"]), -t(g["vl_q_z"]))
    for key in ("p_x", "p_z", "p_pi", "p_mu_lambda", "q_z", "q_pi", "q_mu_lambda", "vl"):
        ref = float(g["vl" if key == "vl" else "vl_" + key])
        assert abs(float(terms[key]) - ref) <= 1e-10 * max(1.0, abs(ref)), key


def test_moments_guard_for_empty_components():
    ns = torch.tensor([2.0, 0.0], dtype=torch.float64)
    a = torch.tensor([[2.0, 4.0], [0.0, 0.0]], dtype=torch.float64)
    B = torch.stack([torch.eye(2, dtype=torch.float64) * 10, torch.zeros(2, 2, dtype=torch.float64)])
    prev = torch.full((2, 2, 2), 7.0, dtype=torch.float64)
    x_bar, s = _kside.moments_from_stats(ns, a, B, torch.tensor([1.0, 1.0], dtype=torch.float64), prev)
    assert torch.equal(x_bar[0], torch.tensor([2.0, 3.0], dtype=torch.float64))
    assert torch.equal(x_bar[1], torch.zeros(2, dtype=torch.float64))        # raw zero sum (ref:727,729)
    assert torch.equal(s[1], prev[1])                                         # untouched (ref:729)
    assert torch.allclose(s[0], torch.tensor([[4.0, -2.0], [-2.0, 1.0]], dtype=torch.float64))


# ------------------------------------------------------------------ F3: the restart driver on the CPU stand-in
DRIVER = ["gmm_f3_c1_subsampling.npz", "gmm_f3_c1_random_resp.npz", "gmm_f3_c1_noconv.npz", "gmm_f3_n1.npz"]


@pytest.mark.parametrize("name", DRIVER)
def test_driver_protocol_and_rng_order(name):
    g = load_golden(name)
    x = g["x"] if "x" in g else load_golden("gmm_c1_sample.npz")["x"]
    K, D = int(g["K"]), int(g["D"])
    kw = json.loads(str(g["kw"]))
    m = cpu_model(K, D, seed=int(g["seed"]))
    buf = io.StringIO()
    with warnings.catch_warnings(record=True) as w, redirect_stdout(buf):
        warnings.simplefilter("always")
        ret = m.update_posterior(x, **kw)
    assert ret is m
    assert any(issubclass(i.category, bayesml_amd.ResultWarning) for i in w) == bool(g["result_warning"])
    lines = [ln for ln in buf.getvalue().split("\n") if ln.strip()]
    tr = g["vl_trace"]
    assert len(lines) == tr.shape[0]
    assert max(i for i, ln in enumerate(lines) if ln.endswith("*")) == int(g["winner"])
    conv = ["(converged)" in ln for ln in lines]
    assert conv == [bool(c) for c in g["converged"]]
    for i, ln in enumerate(lines):
        segs = [s for s in ln.split("\r") if s]
        assert segs[0].startswith(f"{i}. VL: ")
        ref = tr[i][~np.isnan(tr[i])]
        assert len(segs) == len(ref)
        for j, s in enumerate(segs[1:]):
            assert f" t={j} " in s
        vals = [float(s.split("VL: ")[1].split(" ")[0].rstrip("*")) for s in segs]
        assert np.allclose(vals, ref, rtol=1e-8)
    for key in ("hn_alpha_vec", "hn_m_vecs", "hn_kappas", "hn_nus", "hn_w_mats"):
        assert rel_err(m.get_hn_params()[key], g[key]) < 1e-7, key
    assert rel_err(m.ns, g["ns"]) < 1e-7
    assert abs(m.vl - float(g["final_vl"])) <= 1e-8 * abs(float(g["final_vl"]))
    if "est_sq_pi" in g:
        for key in ("p_mu_vecs", "p_nus", "p_lambda_mats"):      # stale until calc_pred_dist, like the reference
            assert np.allclose(m.get_p_params()[key], g["stale_" + key], rtol=1e-12, atol=1e-300), key
        assert np.allclose(m.make_prediction("squared"), g["stale_pred_squared"], atol=1e-12)
        m.calc_pred_dist()
        assert rel_err(m.make_prediction("squared"), g["pred_squared"]) < 1e-7
        pi01, _, lam01 = quiet(m.estimate_params, "0-1")
        assert np.allclose(pi01, g["est_01_pi"], rtol=1e-7, equal_nan=True)
        assert np.allclose(lam01, g["est_01_lambda"], rtol=1e-6, equal_nan=True)
        kl = m.estimate_params("KL")
        assert len(kl[1]) == K and len(kl[2]) == K


@pytest.mark.parametrize("name", DRIVER)
def test_small_problem_path_host_side(name):
    """The small-problem path (gaussianmixture/_small.py: every restart and iteration in one launch) with the launch
    replaced by its CPU stand-in: the reference's draws in the reference's order, winner, progress lines, posterior and
    the attributes update_posterior leaves - against the REFERENCE's fixtures."""
    from fake_engine import cpu_small_fit
    g = load_golden(name)
    x = g["x"] if "x" in g else load_golden("gmm_c1_sample.npz")["x"]
    K, D = int(g["K"]), int(g["D"])
    kw = json.loads(str(g["kw"]))
    m = gm.LearnModel(K, D, seed=int(g["seed"]))
    m._small_fit_impl = cpu_small_fit
    buf = io.StringIO()
    with warnings.catch_warnings(record=True) as w, redirect_stdout(buf):
        warnings.simplefilter("always")
        ret = m.update_posterior(x, **kw)
    assert ret is m and m._engine is None                      # no workspace was opened
    assert any(issubclass(i.category, bayesml_amd.ResultWarning) for i in w) == bool(g["result_warning"])
    lines = [ln for ln in buf.getvalue().split("\n") if ln.strip()]
    tr = g["vl_trace"]
    assert len(lines) == tr.shape[0]
    assert max(i for i, ln in enumerate(lines) if ln.endswith("*")) == int(g["winner"])
    assert ["(converged)" in ln for ln in lines] == [bool(c) for c in g["converged"]]
    for i, ln in enumerate(lines):
        segs = [s for s in ln.split("\r") if s]
        ref = tr[i][~np.isnan(tr[i])]
        assert len(segs) == len(ref) and segs[0].startswith(f"{i}. VL: ")
        vals = [float(s.split("VL: ")[1].split(" ")[0].rstrip("*")) for s in segs]
        assert np.allclose(vals, ref, rtol=1e-10)
    for key in ("hn_alpha_vec", "hn_m_vecs", "hn_kappas", "hn_nus", "hn_w_mats"):
        assert rel_err(m.get_hn_params()[key], g[key]) < 1e-9, key
    assert rel_err(m.hn_w_mats_inv, g["hn_w_mats_inv"]) < 1e-9
    assert rel_err(m.ns, g["ns"]) < 1e-9 and m.r_vecs.shape == (x.reshape(-1, D).shape[0], K)
    assert abs(m.vl - float(g["final_vl"])) <= 1e-10 * abs(float(g["final_vl"]))
    # the general driver on the same seed consumes the Generator identically: both leave it in the same state
    m2 = cpu_model(K, D, seed=int(g["seed"]))
    quiet(m2.update_posterior, x, **kw)
    assert m.rng.bit_generator.state == m2.rng.bit_generator.state
    assert rel_err(m.hn_m_vecs, m2.hn_m_vecs) < 1e-7


def test_num_init_zero_keeps_posterior_and_warns():
    x = load_golden("gmm_c1_sample.npz")["x"]
    m = cpu_model(3, 2, seed=0)
    before = {k: v.copy() for k, v in m.get_hn_params().items()}
    with pytest.warns(bayesml_amd.ResultWarning):
        quiet_out = io.StringIO()
        with redirect_stdout(quiet_out):
            m.update_posterior(x, num_init=0)
    for k, v in before.items():
        assert np.array_equal(m.get_hn_params()[k], v)
    assert m.r_vecs.shape == (1000, 3) and abs(m.ns.sum() - 1000) < 1e-9


def test_pred_and_update_and_latent_update_paths():
    x = load_golden("gmm_c1_sample.npz")["x"]
    m = cpu_model(3, 2, seed=0)
    quiet(m.update_posterior, x[:200], num_init=2, max_itr=20)
    pred = quiet(m.pred_and_update, x[200], num_init=1, max_itr=5)
    assert pred.shape == (2,)
    assert np.allclose(m.h0_m_vecs, m.h0_m_vecs) and m.hn_kappas.sum() > m.h0_kappas.sum() - 1e-9
    z = quiet(m.estimate_latent_vars_and_update, x[201:260], num_init=1, max_itr=5)
    assert z.shape == (59, 3) and np.all(z.sum(axis=1) == 1)


# ---- row tiles (bayesml_amd/_engine.py: TiledDataPass) without a GPU: the bookkeeping around the per-tile workspaces -------
class _FakeTile:
    """Stands in for _engine.DataPass: statistics = [sum of the rows' first column, row count, parameter tag]."""
    PASS_NAMES = ("estep_dense", "estep_bound", "estep_carried", "estep_fell_back_dense", "estep_sweep", "mstep_dense",
                  "mstep_list", "estep_gather")
    made = []

    def __init__(self, K, D, x_dtype, max_rows, device=None, tile_of=None):
        self.K, self.D, self.x_dtype, self.max_rows, self.device = K, D, x_dtype, max_rows, torch.device("cpu")
        self.tile_of, self.lib, self.stats_len = tile_of, None, 3
        self.closed, self.params, self.pivot, self.prepared, self.held, self.esteps, self.drift = False, None, None, None, None, 0, None
        self.shard, self.rows = None, 0
        _FakeTile.made.append(self)

    workspace_bytes = property(lambda self: 1000 if self.tile_of is None else 300)
    launch_info = property(lambda self: "estep_sweep_bounds | mstep_list")
    regroup_count = property(lambda self: 1)

    def close(self):
        self.closed = True

    def _stats_out(self, out):
        return out if out is not None else torch.zeros(3, dtype=torch.float64)

    def set_pivot(self, p):
        self.pivot = torch.as_tensor(p)

    def prepare_rows(self, x):
        self.prepared = x

    def set_params(self, c, m, u):
        self.params = float(c)

    def wants_drift(self, n):
        return True

    def set_drift(self, *a, **k):
        self.drift = a

    def forget(self):
        pass

    def set_shard(self, rows, ranks):
        self.shard = (rows, ranks)

    def policy_export(self, tail):
        tail.fill_(1.0)

    def policy_import(self, tail):
        self.imported = tail.clone()

    def estep(self, x):
        self.held, self.rows = x, x.shape[0]
        self.esteps += 1

    def estep_mstep(self, x, out=None):
        self.estep(x)
        out.copy_(torch.tensor([float(x[:, 0].sum()), float(x.shape[0]), self.params], dtype=torch.float64))
        return out

    def load_responsibilities(self, r):
        self.held, self.rows = ("r", r), r.shape[0]

    def mstep(self, x, out=None):
        out.copy_(torch.tensor([float(self.held[1].sum()), float(x.shape[0]), -1.0], dtype=torch.float64))
        return out

    def pass_counts(self):
        return dict(zip(self.PASS_NAMES, [0, 0, 0, 0, self.esteps, 0, self.esteps, 0]))

    def profile(self, on=True):
        pass

    def sparsity(self):
        return float(self.rows), 2.0 * self.rows

    def work(self):
        return dict(active=float(self.rows), evaluated=2.0 * self.rows, accumulated=0.5 * self.rows, settled_rows=0.25 * self.rows,
                    early_exits=0.0, proof_pairs=3.0 * self.rows, sweep_share=0.5)

    def last_kernel_ms(self):
        return 1.0, 0.5

    def kernel_spans(self):
        return {"estep_select": (0.25, 2)}

    def responsibilities(self, row0, n):
        assert not isinstance(self.held, tuple)
        return self.held[row0:row0 + n, :self.K].to(torch.float64)

    def split_stats(self, s):
        return s


@pytest.mark.parametrize("resident", [True, False])
def test_row_tiles_host_logic(monkeypatch, resident):
    from bayesml_amd import _engine
    monkeypatch.setattr(_engine, "DataPass", _FakeTile)
    _FakeTile.made = []
    K, D, N = 2, 3, 1000
    x = torch.arange(N * D, dtype=torch.float32).reshape(N, D)
    eng = _engine.TiledDataPass(K, D, torch.float32, N, "cpu", 300, resident=resident)
    tiles = _FakeTile.made
    assert eng.n_tiles == 4 and len(tiles) == (4 if resident else 1)
    if resident:            # further tiles share the first one's pass-local buffers and are sized for their own rows
        assert all(t.tile_of is tiles[0] for t in tiles[1:]) and [t.max_rows for t in tiles] == [300, 300, 300, 100]
        assert eng.workspace_bytes == 1000 + 3 * 300
        assert eng.wants_drift(N)
    else:
        assert tiles[0].shard == (N, 4) and not eng.wants_drift(N)        # one workspace: the tiles decide from job-wide sums
    eng.set_pivot(torch.zeros(D, dtype=torch.float64))
    eng.prepare_rows(x)
    if resident:
        assert [t.prepared.shape[0] for t in tiles] == [300, 300, 300, 100]
    eng.set_params(torch.tensor(7.0), None, None)
    eng.set_drift(1, 2, 3)
    assert all((t.drift == (1, 2, 3)) == resident for t in tiles)
    eng.profile(True)
    stats = eng.estep_mstep(x)
    assert float(stats[0]) == float(x[:, 0].sum()) and float(stats[1]) == N and float(stats[2]) == 4 * 7.0
    wk = eng.work()                           # summed over the tiles (resident: read back after the pass)
    assert wk["active"] == N and wk["evaluated"] == 2 * N and wk["proof_pairs"] == 3 * N and wk["settled_rows"] == 0.25 * N
    assert eng.sparsity() == (float(N), 2.0 * N)
    assert eng.last_kernel_ms() == (4.0, 2.0) and eng.kernel_spans() == {"estep_select": (1.0, 8)}
    assert eng.pass_counts()["estep_sweep"] == 4
    # read-outs: the last tile's E-step output is what the buffers hold; any other tile is evaluated again
    before = [t.esteps for t in tiles]
    r = eng.responsibilities(290, 20)          # rows 290..309: tiles 0 and 1
    assert r.shape == (20, K) and torch.equal(r[:, 0], x[290:310, 0].to(torch.float64))
    after = [t.esteps for t in tiles]
    assert sum(after) - sum(before) == 2
    # a sharded job: the tiles of all ranks share one policy, the tail is the sum over this rank's tiles
    eng.set_shard(5 * N, 5)
    assert all(t.shard == (5 * N, 20) for t in tiles)
    eng.estep_mstep(x)
    tail = torch.zeros(_engine.POLICY_LEN, dtype=torch.float64)
    eng.policy_export(tail)
    assert float(tail[0]) == 4.0
    # loaded responsibilities go tile by tile too
    rr = torch.ones(N, K, dtype=torch.float64)
    eng.load_responsibilities(rr)
    st = eng.mstep(x)
    assert float(st[0]) == N * K and float(st[1]) == N
    eng.close()
    assert all(t.closed for t in tiles)


def test_wire_block_packs_the_upper_triangles():
    """The block a row-sharded job all-reduces: [ns | h | a | upper triangles of B] (gmmvb_stats_pack's layout, here the
    torch index map the CPU tests run) - half of the full block's bytes, and the round trip restores a mirrored block."""
    import torch
    from bayesml_amd import _kside
    K, D = 3, 5
    g = torch.Generator().manual_seed(0)
    head = torch.randn(K * (2 + D), dtype=torch.float64, generator=g)
    B = torch.randn(K, D, D, dtype=torch.float64, generator=g)
    B = B + B.transpose(1, 2)
    full = torch.cat([head, B.reshape(-1)])
    n = _kside.packed_stats_len(K, D)
    assert n == K * (2 + D) + K * D * (D + 1) // 2
    wire = torch.zeros(n, dtype=torch.float64)
    _kside.stats_triangle(True, K, D, full, wire)
    assert torch.equal(wire[:K * (2 + D)], head)
    assert torch.equal(wire[K * (2 + D):K * (2 + D) + D], B[0, 0])              # row 0 of the first block, whole
    assert wire[K * (2 + D) + D] == B[0, 1, 1]                                    # row 1 starts on the diagonal
    back = torch.full_like(full, float("nan"))
    _kside.stats_triangle(False, K, D, wire, back)
    assert torch.equal(back, full)
    # a block whose lower triangle disagrees comes back mirrored from the UPPER one
    lop = full.clone()
    lop[K * (2 + D) + D] += 1.0                                                   # entry (1, 0) of block 0
    _kside.stats_triangle(True, K, D, lop, wire)
    _kside.stats_triangle(False, K, D, wire, back)
    assert torch.equal(back, full)


def test_factorisation_outside_the_library_kernel_is_verified(monkeypatch):
    """_kside._factor_checked (CPU tensors, GPU past 128 features): a wrong entry from the framework's batched triangular
    solve - what this image's GPU routines return at order 65 - is caught by the residual check and redone on the host;
    NaN inputs still propagate.  Also _kside.spd_inverse on the host path."""
    import torch
    from bayesml_amd import _kside
    rng = np.random.default_rng(5)
    K, D = 5, 37
    a = rng.standard_normal((K, D, D))
    w_inv = a @ a.transpose(0, 2, 1) + D * np.eye(D)
    t = torch.as_tensor(w_inv)
    want_g = np.linalg.cholesky(w_inv)
    g, g_inv, logdet = _kside._factor_checked(t)
    assert np.abs(g.numpy() - want_g).max() < 1e-12 and np.abs(g_inv.numpy() - np.linalg.inv(want_g)).max() < 1e-12
    assert np.abs(logdet.numpy() - np.linalg.slogdet(w_inv)[1]).max() < 1e-11
    real = torch.linalg.solve_triangular

    def broken(*args, **kw):
        out = real(*args, **kw).clone()
        out[2, D - 1, D - 1] = 0.0           # "an O(1) error in a last diagonal element"
        return out

    monkeypatch.setattr(torch.linalg, "solve_triangular", broken)
    g2, g_inv2, _ = _kside._factor_checked(t)
    assert np.abs(g_inv2.numpy() - np.linalg.inv(want_g)).max() < 1e-12
    bad = t.clone()
    bad[1, 3, 3] = float("nan")
    g3, g_inv3, _ = _kside._factor_checked(bad)
    assert torch.isnan(g3[1]).any() and not torch.isnan(g3[0]).any()
    monkeypatch.undo()
    w = np.linalg.inv(w_inv)
    got, ld = _kside.spd_inverse(w, "cpu")
    assert np.abs(got.numpy() - w_inv).max() < 1e-9 * np.abs(w_inv).max() and np.abs(ld.numpy() + np.linalg.slogdet(w)[1]).max() < 1e-10
    p = _kside.prior_from_numpy(np.ones(K), np.zeros((K, D)), np.ones(K), np.full(K, float(D)), w, "cpu")
    assert torch.equal(p.w_inv, got)
