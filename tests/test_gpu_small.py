"""Small problems on the device: ``gmmvb_small_fit`` (csrc/small.hip) runs every restart and every VB iteration of
``update_posterior`` (reference ``_gaussianmixture.py:846-872``) in one launch.  Checked against the REFERENCE's driver
fixtures (winner, VL trace, convergence flags, posterior), against the oracle-built stand-in of the same ABI on other shapes,
and against the general engine path on the same seed."""
import io
import json
import os
import warnings
from contextlib import redirect_stdout

import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err
from fake_engine import cpu_small_fit
from oracle import gmm_vb_oracle as orc

pytestmark = pytest.mark.gpu

DRIVER = ["gmm_f3_c1_subsampling.npz", "gmm_f3_c1_random_resp.npz", "gmm_f3_c1_noconv.npz", "gmm_f3_n1.npz"]


def _run(K, D, x, seed, kw, small=True):
    from bayesml_amd import gaussianmixture as gm
    old = os.environ.get("BAYESML_AMD_SMALL")
    os.environ["BAYESML_AMD_SMALL"] = "1" if small else "0"
    try:
        m = gm.LearnModel(K, D, seed=seed, device=torch.device("cuda", 0))
        buf = io.StringIO()
        with warnings.catch_warnings(record=True) as w, redirect_stdout(buf):
            warnings.simplefilter("always")
            m.update_posterior(x, **kw)
    finally:
        os.environ.pop("BAYESML_AMD_SMALL", None)
        if old is not None:
            os.environ["BAYESML_AMD_SMALL"] = old
    return m, buf.getvalue(), w


@pytest.mark.parametrize("name", DRIVER)
def test_small_fit_matches_reference_fixtures(name):
    import bayesml_amd
    g = load_golden(name)
    x = g["x"] if "x" in g else load_golden("gmm_c1_sample.npz")["x"]
    K, D = int(g["K"]), int(g["D"])
    kw = json.loads(str(g["kw"]))
    m, text, w = _run(K, D, x, int(g["seed"]), kw)
    assert m._engine is None and m._small_r is not None          # the one-launch path ran, no workspace was opened
    assert any(issubclass(i.category, bayesml_amd.ResultWarning) for i in w) == bool(g["result_warning"])
    lines = [ln for ln in text.split("\n") if ln.strip()]
    tr = g["vl_trace"]
    assert len(lines) == tr.shape[0]
    assert max(i for i, ln in enumerate(lines) if ln.endswith("*")) == int(g["winner"])
    assert ["(converged)" in ln for ln in lines] == [bool(c) for c in g["converged"]]
    for i, ln in enumerate(lines):
        segs = [s for s in ln.split("\r") if s]
        ref = tr[i][~np.isnan(tr[i])]
        assert len(segs) == len(ref)                              # the same number of iterations in every restart
        vals = [float(s.split("VL: ")[1].split(" ")[0].rstrip("*")) for s in segs]
        assert np.allclose(vals, ref, rtol=1e-8, atol=0)
    for key in ("hn_alpha_vec", "hn_m_vecs", "hn_kappas", "hn_nus", "hn_w_mats"):
        assert rel_err(m.get_hn_params()[key], g[key]) < 1e-7, key
    assert rel_err(m.hn_w_mats_inv, g["hn_w_mats_inv"]) < 1e-7
    assert rel_err(m._e_ln_pi_vec, g["e_ln_pi_vec"]) < 1e-7 and rel_err(m._e_ln_lambda_dets, g["e_ln_lambda_dets"]) < 1e-7
    assert rel_err(m._ln_b_hn_w_nus, g["ln_b_hn_w_nus"]) < 1e-7
    assert rel_err(m.ns, g["ns"]) < 1e-7 and rel_err(m.x_bar_vecs, g["x_bar_vecs"]) < 1e-7
    assert rel_err(m.s_mats, g["s_mats"]) < 1e-6
    r = m.r_vecs
    assert np.max(np.abs(r[: g["r_head"].shape[0]] - g["r_head"])) < 1e-7
    assert np.max(np.abs(r.sum(axis=0) - g["r_colsum"])) < 1e-6
    assert abs(m.vl - float(g["final_vl"])) <= 1e-8 * abs(float(g["final_vl"]))
    # same Generator state as the general path leaves, and the same posterior
    m2, _t, _w = _run(K, D, x, int(g["seed"]), kw, small=False)
    assert m2._engine is not None
    assert m.rng.bit_generator.state == m2.rng.bit_generator.state
    for key in ("hn_alpha_vec", "hn_m_vecs", "hn_w_mats"):
        assert rel_err(m.get_hn_params()[key], m2.get_hn_params()[key]) < 1e-7, key


SHAPES = [(5, 8, 3000, np.float32, "subsampling"), (16, 2, 16384, np.float64, "subsampling"),
          (32, 2, 700, np.float32, "random_responsibility"), (4, 5, 1, np.float64, "subsampling"),
          (7, 3, 513, np.float32, "random_responsibility"), (1, 1, 64, np.float64, "subsampling")]


@pytest.mark.parametrize("K,D,N,dtype,init_type", SHAPES)
def test_small_fit_kernel_against_the_abi_stand_in(K, D, N, dtype, init_type):
    """The launch's whole output block (traces, flags, posterior, moments, responsibilities) against the oracle-built
    stand-in of the same ABI, restart by restart, across the kernel's range (chunked rows, ragged last chunk, K = 32,
    D = 8, a single row, both initialisations)."""
    from bayesml_amd import gaussianmixture as gm
    x = orc.synth_gmm(max(K, 2), D, N, dtype, spread=3.0)
    kw = dict(num_init=3, max_itr=15, tolerance=1e-9, init_type=init_type)
    got, ref = {}, {}

    def spy(store, impl):
        def f(*a):
            out, r = impl(*a)
            store["out"], store["r"] = np.array(out), (r.cpu().numpy() if isinstance(r, torch.Tensor) else np.array(r))
            return out, r
        return f

    from bayesml_amd.gaussianmixture import _small
    for store, impl in ((got, None), (ref, cpu_small_fit)):
        m = gm.LearnModel(K, D, seed=5, device=torch.device("cuda", 0), verbose=False)
        if impl is None:
            m._small_fit_impl = spy(store, lambda K_, D_, xh, pivot, prior, init, R, code, mi, tol:
                                    _small._device_fit(m, xh, pivot, prior, init, R, code, mi, tol))
        else:
            m._small_fit_impl = spy(store, impl)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            m.update_posterior(x, **kw)
    a, b = got["out"], ref["out"]
    assert a.shape == b.shape
    assert np.array_equal(a[:, :2], b[:, :2])                    # iterations run and convergence flag of every restart
    t0 = 10
    for i in range(a.shape[0]):
        n_vl = int(a[i, 0])
        assert np.allclose(a[i, t0:t0 + n_vl], b[i, t0:t0 + n_vl], rtol=1e-9, atol=0), i
        assert np.allclose(a[i, 2:10], b[i, 2:10], rtol=1e-8, atol=1e-8 * abs(b[i, 9])), i
        post_a, post_b = a[i, t0 + 16:], b[i, t0 + 16:]
        scale = np.maximum(np.abs(post_b), 1e-3 * np.max(np.abs(post_b)))
        assert np.max(np.abs(post_a - post_b) / scale) < 1e-6, i
    assert np.max(np.abs(got["r"] - ref["r"])) < 1e-8


def test_small_fit_is_bypassed_outside_its_range():
    from bayesml_amd import _engine
    lib = _engine.load_library()
    assert lib.gmmvb_small_supported(3, 2, 1000) == 1 and lib.gmmvb_small_supported(32, 2, 16384) == 1
    assert lib.gmmvb_small_supported(3, 9, 1000) == 0 and lib.gmmvb_small_supported(33, 2, 10) == 0
    assert lib.gmmvb_small_supported(8, 8, 100) == 0              # 8 * 45 statistics > 256
    assert lib.gmmvb_small_supported(3, 2, 16385) == 0
    assert lib.gmmvb_small_out_len(3, 2, 100) == 2 + 8 + 101 + 7 * 3 + 2 * 6 + 3 * 12
    x = orc.synth_gmm(3, 2, 20000, np.float64)
    m, _t, _w = _run(3, 2, x, 0, dict(num_init=1, max_itr=3))
    assert m._engine is not None and m._small_r is None


def test_small_fit_after_a_large_one_leaves_one_consistent_pass():
    """A small-problem fit releases the workspace of an earlier, larger fit: r_vecs (from the launch) and _ln_rho (the final
    E-step of ref:895, re-run on demand) both describe the small fit's rows."""
    from bayesml_amd import gaussianmixture as gm
    rng = np.random.default_rng(4)
    mu = np.array([[-4.0, 0.0], [4.0, 0.0], [0.0, 5.0]])
    big = (mu[rng.integers(0, 3, 40000)] + rng.standard_normal((40000, 2)))
    small = (mu[rng.integers(0, 3, 700)] + rng.standard_normal((700, 2)))
    m = gm.LearnModel(3, 2, seed=1, device="cuda:0", verbose=False)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m.update_posterior(big, max_itr=5, num_init=1)
        assert m._engine is not None and m.r_vecs.shape == (40000, 3)
        m.update_posterior(small, max_itr=20, num_init=2)
    assert m._engine is None and m._x_dev is None
    r = m.r_vecs
    assert r.shape == (700, 3)
    ln_rho = m._ln_rho
    assert ln_rho.shape == (700, 3)
    soft = np.exp(ln_rho - ln_rho.max(axis=1, keepdims=True))
    soft /= soft.sum(axis=1, keepdims=True)
    assert np.max(np.abs(soft - r)) < 1e-9
    assert m.r_vecs is r and m._engine is None          # the read-out left the fit's own arrays in place
