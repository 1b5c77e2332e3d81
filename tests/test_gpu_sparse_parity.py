"""The sparse data pass (int8 bound pass -> candidate lists -> exact f64 gathers -> carried bounds -> M-step over
active-row lists) against the REFERENCE's outputs - not against the repo's own dense kernels.

Every case runs the public driver (reference ``_gaussianmixture.py:802-896``) on a fixture the reference itself
produced (tests/golden/make_golden.py, make_golden_large.py), asserts from the library's launch counters that the
kernels under test really ran, and compares posterior, VL trace and responsibilities with the fixture.  Tolerances are
relative (max|a-b| / max|b|); north_star asks for 1e-5 on the posterior.
"""
import io
import json
import os
import warnings
from contextlib import redirect_stdout

import numpy as np
import pytest
import torch

from conftest import load_golden, mat_functionals, rel_err
from oracle import gmm_vb_oracle as orc

pytestmark = pytest.mark.gpu

ENV_KEYS = ("GMMVB_ESTEP_PRUNE", "GMMVB_MSTEP_SPARSE", "GMMVB_ESTEP_CARRY_OFF", "GMMVB_SORT_ROWS", "GMMVB_SETTLE_MARGIN",
            "GMMVB_MSTEP_CACHE", "GMMVB_PROOF", "GMMVB_GATHER_EXIT", "GMMVB_SWEEP_LAZY", "GMMVB_PROOF_BLOCKED", "GMMVB_REGROUP_MARGIN",
            "GMMVB_PROJECT")
VARIANTS = {
    "default": {},
    # Rows with a single active component are settled (left out of the E-step).  Default: in every pruned pass, without
    # slack - a settled row whose carried bounds leave candidates goes through the int8 proof round (estep_i8_proof), as do
    # the candidates of every other row before anything is evaluated in f64.  "settle": 5 nats of slack (fewer rows settle,
    # fewer come back); the read-outs below need the settled rows' values re-evaluated
    "settle": {"GMMVB_SETTLE_MARGIN": "5"},
    "force_settle": {"GMMVB_ESTEP_PRUNE": "force", "GMMVB_SETTLE_MARGIN": "3"},
    "force_noproof": {"GMMVB_ESTEP_PRUNE": "force", "GMMVB_PROOF": "0"},
    # only the settled rows' pairs go through the proof round (default: every candidate does before anything is evaluated in f64)
    "force_proof_settled": {"GMMVB_ESTEP_PRUNE": "force", "GMMVB_PROOF": "settled"},
    "proof_settled": {"GMMVB_PROOF": "settled"},
    # ... without the proof round rows never settle (a settled row that came loose would cost exact evaluations)
    "noproof": {"GMMVB_PROOF": "0"},
    "nosettle": {"GMMVB_SETTLE_MARGIN": "-1"},
    "nocache": {"GMMVB_MSTEP_CACHE": "0"},
    "noexit": {"GMMVB_GATHER_EXIT": "0"},
    # every sweep carries all K bounds of every row (default: per tile of 256 rows only the components within reach)
    "nolazy": {"GMMVB_SWEEP_LAZY": "0"},
    "force_nolazy": {"GMMVB_ESTEP_PRUNE": "force", "GMMVB_SWEEP_LAZY": "0"},
    # the proof round component after component (default: by row superblocks, estep_i8_proof_blocked)
    "proof_by_component": {"GMMVB_PROOF_BLOCKED": "0"},
    "force_proof_by_component": {"GMMVB_ESTEP_PRUNE": "force", "GMMVB_PROOF_BLOCKED": "0"},
    "force": {"GMMVB_ESTEP_PRUNE": "force"},
    # The stateless table of csrc/project.h (bounds from the parameters in force and the rows' int8 digit planes), opt-in:
    # "project_filter": it takes pairs off the carried sweep's proof lists; "project_only": it replaces the carried per-pair
    # bounds altogether (rec_project_kernel).  Default: no table - all three must agree with the reference
    "project_filter": {"GMMVB_PROJECT": "filter"},
    "force_project_filter": {"GMMVB_ESTEP_PRUNE": "force", "GMMVB_PROJECT": "filter"},
    "project_only": {"GMMVB_PROJECT": "only"},
    "force_project_only": {"GMMVB_ESTEP_PRUNE": "force", "GMMVB_PROJECT": "only"},
    # the rows regrouped by best component only (default: within a component by how firmly they sit in it)
    "force_nomargin": {"GMMVB_ESTEP_PRUNE": "force", "GMMVB_REGROUP_MARGIN": "0"},
    "force_nocache": {"GMMVB_ESTEP_PRUNE": "force", "GMMVB_MSTEP_CACHE": "0"},
    "force_nocarry": {"GMMVB_ESTEP_PRUNE": "force", "GMMVB_ESTEP_CARRY_OFF": "1"},
    "dense": {"GMMVB_ESTEP_PRUNE": "0", "GMMVB_MSTEP_SPARSE": "0"},
}


class env:
    def __init__(self, kv):
        self.kv = kv

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in ENV_KEYS}
        for k in ENV_KEYS:
            os.environ.pop(k, None)
        os.environ.update(self.kv)

    def __exit__(self, *exc):
        for k, v in self.old.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v


def run_driver(g, x, variant, extra_env=None):
    from bayesml_amd import gaussianmixture as gm
    K, D = int(g["K"]), int(g["D"])
    kw = json.loads(str(g["kw"]))
    buf = io.StringIO()
    with env({**VARIANTS[variant], **(extra_env or {})}):
        m = gm.LearnModel(K, D, seed=int(g["seed"]), device=torch.device("cuda", 0))
        with warnings.catch_warnings(), redirect_stdout(buf):
            warnings.simplefilter("ignore")
            m.update_posterior(x, **kw)
            counts = m._engine.pass_counts()
    lines = [ln for ln in buf.getvalue().split("\n") if ln.strip()]
    trace = [[float(seg.split("VL: ")[1].split(" ")[0].rstrip("*").replace("(converged)", ""))
              for seg in ln.split("\r") if seg] for ln in lines]
    return m, counts, trace


def check_trace(trace, g, rtol):
    tr = g["vl_trace"]
    assert len(trace) == tr.shape[0]
    for i, vals in enumerate(trace):
        ref = tr[i][~np.isnan(tr[i])]
        assert len(vals) == len(ref)
        assert np.allclose(vals, ref, rtol=rtol, atol=0), (i, np.max(np.abs(np.array(vals) / ref - 1)))


def expect_kernels(counts, variant, min_carried=1, lists=True):
    if variant == "dense":
        assert counts["estep_bound"] == counts["estep_carried"] == counts["estep_sweep"] == counts["mstep_list"] == 0, counts
        return
    assert counts["estep_bound"] >= 1 and counts["estep_gather"] >= 2, counts
    if lists:           # (the M-step runs over lists only while at most 35 % of the pairs are active)
        assert counts["mstep_list"] >= 1, counts
    if variant == "force_nocarry":
        assert counts["estep_carried"] == counts["estep_sweep"] == 0, counts
    elif variant in ("settle", "force_settle", "force_proof_settled"):
        assert counts["estep_sweep"] >= 2, counts
    else:       # carried over the parameter update by a sweep of the per-pair bound array
        assert counts["estep_sweep"] >= min_carried, counts


@pytest.mark.parametrize("variant", ["force", "force_nocarry"])
def test_small_fixture_forced_sparse_matches_reference(variant):
    """K=8, D=128, N=32768 (f32 rows), 10 iterations: the reference's posterior through the forced sparse path."""
    g = load_golden("gmm_f3_k8_d128_n32768_f32.npz")
    x = orc.synth_gmm(8, 128, 32768, np.float32)
    m, counts, trace = run_driver(g, x, variant)
    expect_kernels(counts, variant, min_carried=0, lists=False)   # 8 broad components: over 35 % of the pairs stay active
    check_trace(trace, g, 1e-8)
    hn = m.get_hn_params()
    for key in ("hn_alpha_vec", "hn_m_vecs", "hn_kappas", "hn_nus", "hn_w_mats"):
        assert rel_err(hn[key], g[key]) < 1e-6, key
    assert rel_err(m.hn_w_mats_inv, g["hn_w_mats_inv"]) < 1e-6
    assert rel_err(m.ns, g["ns"]) < 1e-6 and rel_err(m.s_mats, g["s_mats"]) < 1e-5
    assert np.max(np.abs(m.r_vecs[:64] - g["r_head"])) < 1e-6
    assert abs(m.vl - float(g["final_vl"])) <= 1e-8 * abs(float(g["final_vl"]))


LARGE = [("gmm_f3_k64_d128_n140000_f32.npz", "default"), ("gmm_f3_k64_d128_n140000_f32.npz", "force_nocarry"),
         ("gmm_f3_k64_d128_n140000_f32.npz", "settle"), ("gmm_f3_k64_d128_n140000_f32.npz", "noproof"),
         ("gmm_f3_k64_d128_n140000_f32.npz", "nosettle"), ("gmm_f3_k64_d128_n140000_f32.npz", "nocache"),
         ("gmm_f3_k64_d128_n140000_f32.npz", "noexit"), ("gmm_f3_k64_d128_n140000_f32.npz", "nolazy"),
         ("gmm_f3_k64_d128_n140000_f32.npz", "proof_settled"), ("gmm_f3_k64_d128_n140000_f32.npz", "proof_by_component"),
         ("gmm_f3_k64_d128_n140000_f32.npz", "project_filter"), ("gmm_f3_k256_d64_n36000_f32.npz", "project_filter"),
         ("gmm_f3_k256_d64_n36000_f32.npz", "force_project_filter"), ("gmm_f3_k16_d64_n32768_f32_spread1.npz", "force_project_filter"),
         ("gmm_f3_k64_d128_n140000_f32.npz", "project_only"), ("gmm_f3_k256_d64_n36000_f32.npz", "project_only"),
         ("gmm_f3_k256_d64_n36000_f32.npz", "force_project_only"), ("gmm_f3_k16_d64_n32768_f32_spread1.npz", "force_project_only"),
         ("gmm_f3_k256_d64_n36000_f32.npz", "force_proof_by_component"),
         ("gmm_f3_k256_d64_n36000_f32.npz", "nolazy"),
         ("gmm_f3_k256_d64_n36000_f32.npz", "settle"),
         ("gmm_f3_k256_d64_n36000_f32.npz", "default"), ("gmm_f3_k256_d64_n36000_f32.npz", "force"),
         ("gmm_f3_k16_d64_n32768_f32_overlap.npz", "force"), ("gmm_f3_k16_d64_n32768_f32_overlap.npz", "dense"),
         # the middle of the separation spectrum (cluster means 1.0 * randn: 2-10 components active per row): where the
         # policy's choice between dense, bound pass and sweep is closest
         ("gmm_f3_k16_d64_n32768_f32_spread1.npz", "force"), ("gmm_f3_k16_d64_n32768_f32_spread1.npz", "force_nolazy"),
         ("gmm_f3_k16_d64_n32768_f32_spread1.npz", "force_nocarry"), ("gmm_f3_k16_d64_n32768_f32_spread1.npz", "default"),
         ("gmm_f3_k16_d64_n32768_f32_spread1.npz", "force_nomargin")]
# off the benchmark's recipe (round 5): twice as many components in the model as in the data (empty and duplicate
# components), mixing weights ~ Dirichlet(0.3), anisotropic clusters (per-feature scales 0.3 ... 3) - reference fixtures, the
# pruned path forced (N K = 2^19: the default policy stays dense at this size, which is the fourth variant)
OFFPATH = [(f"gmm_f3_k16_d64_n32768_f32_{kind}.npz", variant) for kind in ("kdata8", "weights", "aniso")
           for variant in ("force", "force_nocache", "force_noproof", "force_project_filter", "force_project_only", "default")]
LARGE += OFFPATH


@pytest.mark.parametrize("name,variant", LARGE)
def test_large_fixture_matches_reference(name, variant):
    """The benchmark's shape (K=64, D=128) and config 4's (K=256, D=64) at the smallest N where the DEFAULT policy
    prunes and carries (N K >= 2^23), and overlapping clusters (spread 0.3) where it has to fall back."""
    g = load_golden(name)
    K, D, N = int(g["K"]), int(g["D"]), int(g["N"])
    x = orc.synth_gmm(int(g["K_data"]), D, N, np.float32, spread=float(g["spread"]),
                      weights_alpha=float(g["weights_alpha"]) if "weights_alpha" in g else None,
                      scale_range=tuple(g["scale_range"]) if "scale_range" in g else None)
    offpath = (name, variant) in OFFPATH
    m, counts, trace = run_driver(g, x, variant)
    if offpath:
        if variant.startswith("force"):     # the pruned kernels really ran: bound pass, carried sweeps, gathers, list M-step
            assert counts["estep_bound"] >= 1 and counts["estep_sweep"] >= 1 and counts["estep_gather"] >= 2 and counts["mstep_list"] >= 1, counts
    elif "overlap" in name or "spread1" in name:
        if variant.startswith("force"):     # (nearly) everything is a candidate: the pruned E-step must still be exact
            assert counts["estep_bound"] >= 1 and counts["estep_gather"] >= 2, counts
    else:
        expect_kernels(counts, variant)
    check_trace(trace, g, 1e-8)
    hn = m.get_hn_params()
    for key in ("hn_alpha_vec", "hn_m_vecs", "hn_kappas", "hn_nus"):
        assert rel_err(hn[key], g[key]) < 1e-6, key
    for key, got in (("hn_w_mats", hn["hn_w_mats"]), ("hn_w_mats_inv", m.hn_w_mats_inv), ("s_mats", m.s_mats)):
        for fn, val in mat_functionals(got).items():
            ref = g[f"{key}_{fn}"]
            if fn == "logabsdet" and key == "s_mats":
                continue        # scatter matrices of (nearly) empty components are singular: their determinant is noise
            if fn == "logabsdet":
                assert np.max(np.abs(val - ref)) < 1e-6 * max(1.0, float(np.max(np.abs(ref)))), (key, fn)
            else:
                assert rel_err(val, ref) < (1e-5 if key == "s_mats" else 1e-6), (key, fn)
    assert rel_err(m.ns, g["ns"]) < 1e-6 and rel_err(m.x_bar_vecs, g["x_bar_vecs"]) < 1e-6
    assert np.max(np.abs(m.r_vecs[:64] - g["r_head"])) < 1e-6
    assert np.max(np.abs(m._engine.responsibilities().sum(dim=0).cpu().numpy() - g["r_colsum"])) < 1e-6 * N / K
    assert abs(m.vl - float(g["final_vl"])) <= 1e-8 * abs(float(g["final_vl"]))
    if m._engine.launch_info.startswith("estep_sweep"):
        # which sweep it was, and whether the table took part: as a filter (GMMVB_PROJECT=filter), as the sweep itself
        # (GMMVB_PROJECT=only), or - the default - not at all
        wk = m._engine.work()
        if variant in ("project_filter", "force_project_filter"):
            assert m._engine.launch_info.startswith("estep_sweep_bounds") and wk["table_left"] >= 0, (m._engine.launch_info, wk)
        elif variant in ("project_only", "force_project_only"):
            assert m._engine.launch_info.startswith("estep_sweep_projected") and wk["table_left"] >= 0, m._engine.launch_info
        else:
            assert m._engine.launch_info.startswith("estep_sweep_bounds") and wk["table_left"] < 0, m._engine.launch_info
    if not offpath and "overlap" not in name and "spread1" not in name and variant in ("default", "settle", "noproof", "nosettle", "nocache", "noexit", "nolazy", "proof_settled", "project_filter", "project_only"):
        # the M-step's cache of single-component rows: in use by default, its rows are not accumulated
        # again; settled rows are not even evaluated
        wk = m._engine.work()
        swept = m._engine.launch_info.startswith("estep_sweep")
        if variant == "nocache":
            assert wk["accumulated"] == wk["active"] and wk["settled_rows"] == 0, wk
        else:       # (the cache lives through every pruned pass; after a sweep most single-component rows are in it)
            assert 0 <= wk["accumulated"] <= wk["active"], wk
            if swept:
                assert wk["accumulated"] < wk["active"], wk
        if variant in ("default", "settle", "project_filter", "project_only") and swept:
            assert wk["settled_rows"] > 0.2 * N and wk["evaluated"] < wk["active"], wk
        if variant in ("nosettle", "noproof"):      # (without settled rows the proof round still serves the bound passes)
            assert wk["settled_rows"] == 0 and (wk["proof_pairs"] == 0 or variant == "nosettle"), wk
    if variant == "default" and not offpath and "overlap" not in name and "spread1" not in name:
        # the workspace regrouped its internal row order by dominant component on the way: every
        # read-out above - responsibilities of the first rows, their column sums, hard assignments - is nevertheless
        # in the caller's row order
        assert m._engine.regroup_count >= 1
        z = m._engine.argmax().cpu().numpy()
        r_all = m._engine.responsibilities().cpu().numpy()
        assert np.array_equal(z, r_all.argmax(axis=1))
        assert np.array_equal(z[:64], g["r_head"].argmax(axis=1))
        m2, _c, _t = run_driver(g, x, variant, {"GMMVB_SORT_ROWS": "0"})
        assert m2._engine.regroup_count == 0
        assert np.max(np.abs(m2._engine.responsibilities().cpu().numpy() - r_all)) < 1e-9
        for key in ("hn_m_vecs", "hn_w_mats"):
            assert rel_err(m2.get_hn_params()[key], hn[key]) < 1e-10, key


def _oracle_post(q):
    """Device posterior (bayesml_amd._kside.PostT) -> the oracle's Posterior with its own derived features."""
    n = lambda t: t.detach().cpu().numpy().copy()   # noqa: E731
    o = orc.Posterior(alpha=n(q.alpha), m=n(q.m), kappa=n(q.kappa), nu=n(q.nu), w=n(q.w), w_inv=n(q.w_inv))
    o.refresh_pi()
    o.refresh_lambda()
    return o


@pytest.mark.parametrize("variant", ["force", "force_settle", "force_noproof", "force_proof_settled", "force_nolazy",
                                     "force_nomargin", "force_project_filter", "force_project_only"])
def test_carried_bounds_are_upper_bounds_of_the_oracle(variant):
    """Property behind gmmvb_set_drift, checked right after E-steps that lived on carried bounds: every value in
    the workspace is either the exact ln rho - as the ORACLE computes it for the same posterior - or an upper
    bound of it lying at least 80 ln 2 below the row's best component; responsibilities and statistics equal the
    oracle's.  The loop is update_posterior's (K-side update -> drift hint -> data pass)."""
    from bayesml_amd import _kside
    from bayesml_amd import gaussianmixture as gm
    K, D, N = 24, 64, 24000
    x = orc.synth_gmm(K, D, N, np.float32)
    x64 = x.astype(np.float64)
    dev = torch.device("cuda", 0)
    with env(VARIANTS[variant]):
        m = gm.LearnModel(K, D, seed=0, device=dev, verbose=False)
        eng, xd = m._open(x)
    settled_seen = cached_seen = proof_seen = 0.0
    prior = m._prior_tensors(dev)
    q = m._init_subsampling(eng, xd, _kside.post_from_prior(prior), N)
    s = torch.zeros(K, D, D, dtype=torch.float64, device=dev)
    ns, x_bar, s, _h = m._pass(eng, xd, q, s)
    checked = 0
    for it in range(14):
        q_new = _kside.update_q(prior, ns, x_bar, s)
        hint = m._drift_hint(eng, xd, q, q_new)
        assert hint is not None
        before = eng.pass_counts()["estep_sweep"]
        q = q_new
        ns, x_bar, s, _h = m._pass(eng, xd, q, s, hint=(*hint, float((hint[0] - hint[1] / 30.0).min())))
        if eng.pass_counts()["estep_sweep"] == before:
            continue
        wk = eng.work()
        settled_seen = max(settled_seen, wk["settled_rows"])
        proof_seen = max(proof_seen, wk["proof_pairs"])
        if it >= 6 and wk["accumulated"] >= 0:
            cached_seen = max(cached_seen, wk["active"] - wk["accumulated"])
        lb = eng.ln_rho().cpu().numpy()
        oq = _oracle_post(q)
        st = orc.data_pass(x64, oq)
        la = st.ln_rho
        same = np.abs(la - lb) <= 1e-8 * np.maximum(1.0, np.abs(la))
        assert same.mean() < 0.9, "nothing was pruned"
        assert np.all(lb[~same] >= la[~same]), (it, "a carried value is not an upper bound")
        mx = la.max(axis=1, keepdims=True)
        lse = mx + np.log(np.exp(la - mx).sum(axis=1, keepdims=True))
        assert np.all((lb <= lse - 55.4) | same), it          # 80 ln 2 = 55.45
        assert np.max(np.abs(eng.responsibilities().cpu().numpy() - st.r)) < 1e-9
        assert rel_err(ns.cpu().numpy(), st.ns) < 1e-10 and rel_err(s.cpu().numpy(), st.s) < 1e-9
        checked += 1
    assert checked >= 3, eng.pass_counts()
    assert cached_seen > 0.1 * N, cached_seen       # single-component rows the M-step did not accumulate again
    if variant in ("force", "force_settle", "force_nolazy", "force_project_filter", "force_project_only"):     # rows that were not evaluated at all: read out exactly all the same -
        assert settled_seen > 0.1 * N, settled_seen          # many of them on the strength of the int8 proof round
        if variant in ("force", "force_settle", "force_nolazy"):   # (carried bounds erode into the proof round; with the
            assert proof_seen > 0.01 * N, proof_seen               # table in front of it next to nothing is left)
    if variant == "force_noproof":
        assert proof_seen == 0 and settled_seen == 0


@pytest.mark.parametrize("variant", ["force", "force_settle", "force_noproof"])
def test_cache_survives_unusual_call_orders(variant):
    """The cache of single-component rows is internal state of the workspace: whatever order the entry points are
    called in - an M-step twice, an E-step twice without an M-step, read-outs between the two - the statistics and
    the responsibilities stay the oracle's."""
    from bayesml_amd import _kside
    from bayesml_amd import gaussianmixture as gm
    K, D, N = 24, 64, 24000
    x = orc.synth_gmm(K, D, N, np.float32)
    x64 = x.astype(np.float64)
    dev = torch.device("cuda", 0)
    with env(VARIANTS[variant]):
        m = gm.LearnModel(K, D, seed=0, device=dev, verbose=False)
        eng, xd = m._open(x)
    prior = m._prior_tensors(dev)
    q = m._init_subsampling(eng, xd, _kside.post_from_prior(prior), N)
    s = torch.zeros(K, D, D, dtype=torch.float64, device=dev)
    ns, x_bar, s, _h = m._pass(eng, xd, q, s)

    def check(ns, s, q):
        st = orc.data_pass(x64, _oracle_post(q))
        assert rel_err(ns.cpu().numpy(), st.ns) < 1e-10 and rel_err(s.cpu().numpy(), st.s) < 1e-9
        return st

    cached = 0.0
    for it in range(16):
        q_new = _kside.update_q(prior, ns, x_bar, s)
        hint = m._drift_hint(eng, xd, q, q_new)
        q = q_new
        full = (*hint, float((hint[0] - hint[1] / 30.0).min()))
        kind = it % 4
        if kind == 0:           # the ordinary iteration
            ns, x_bar, s, _h = m._pass(eng, xd, q, s, hint=full)
        elif kind == 1:         # M-step twice on one E-step
            ns, x_bar, s, _h = m._pass(eng, xd, q, s, hint=full)
            ns2, _xb, s2, _h2 = m._pass(eng, xd, q, s, estep=False)
            assert torch.equal(ns, ns2) and torch.equal(s, s2)
        elif kind == 2:         # E-step twice (the first one's M-step never runs), read-outs before the M-step
            m._give_params(eng, q, full)
            eng.estep(xd)
            eng.estep(xd)
            r = eng.responsibilities().cpu().numpy()
            ns, x_bar, s, _h = m._pass(eng, xd, q, s, estep=False)
            st = check(ns, s, q)
            assert np.max(np.abs(r - st.r)) < 1e-9
        else:                   # read-outs after the M-step, then the next iteration carries on
            ns, x_bar, s, _h = m._pass(eng, xd, q, s, hint=full)
            st = check(ns, s, q)
            assert np.max(np.abs(eng.responsibilities().cpu().numpy() - st.r)) < 1e-9
            lb = eng.ln_rho().cpu().numpy()
            same = np.abs(st.ln_rho - lb) <= 1e-8 * np.maximum(1.0, np.abs(st.ln_rho))
            assert np.all(lb[~same] >= st.ln_rho[~same])
        check(ns, s, q)
        wk = eng.work()
        if wk["accumulated"] >= 0:
            cached = max(cached, wk["active"] - wk["accumulated"])
    assert cached > 0.1 * N, cached
