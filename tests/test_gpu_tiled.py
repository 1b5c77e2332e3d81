"""Row-tiled data pass (``_engine.TiledDataPass``): a sample matrix whose per-pair workspace would not fit the GPU goes
through the data pass in row tiles - resident tiles (a workspace per tile sharing the pass-local buffers,
``gmmvb_workspace_create_tile``: bounds are carried) or all tiles through ONE workspace; statistics, posterior and
read-outs must equal the untiled run's (the reference has no such limit: it keeps its [N, K] arrays on the host,
``_gaussianmixture.py:835-836``)."""
import os
import warnings

import numpy as np
import pytest
import torch

from conftest import rel_err
from oracle import gmm_vb_oracle as orc

pytestmark = pytest.mark.gpu


def _fit(x, K, D, tile, init_type, force, resident=True, max_itr=8):
    from bayesml_amd import gaussianmixture as gm
    old = {k: os.environ.get(k) for k in ("BAYESML_AMD_TILE_ROWS", "GMMVB_ESTEP_PRUNE", "BAYESML_AMD_TILE_RESIDENT")}
    try:
        os.environ.pop("BAYESML_AMD_TILE_ROWS", None)
        os.environ.pop("GMMVB_ESTEP_PRUNE", None)
        os.environ["BAYESML_AMD_TILE_RESIDENT"] = "1" if resident else "0"
        if tile:
            os.environ["BAYESML_AMD_TILE_ROWS"] = str(tile)
        if force:
            os.environ["GMMVB_ESTEP_PRUNE"] = "force"
        m = gm.LearnModel(K, D, seed=0, device="cuda:0", verbose=False)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            m.update_posterior(x, max_itr=max_itr, num_init=2, tolerance=0.0, init_type=init_type)
    finally:
        for k, v in old.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v
    return m


@pytest.mark.parametrize("resident", [True, False])
@pytest.mark.parametrize("init_type,force", [("subsampling", False), ("subsampling", True), ("random_responsibility", False)])
def test_tiled_run_equals_untiled(init_type, force, resident):
    from bayesml_amd._engine import DataPass, TiledDataPass
    K, D, N = 12, 64, 100_000
    x = orc.synth_gmm(K, D, N, np.float32)
    one = _fit(x, K, D, 0, init_type, force)
    til = _fit(x, K, D, 30016, init_type, force, resident)
    assert isinstance(one._engine, DataPass) and isinstance(til._engine, TiledDataPass)
    assert til._engine.n_tiles == 4 and til._engine.resident == resident
    assert len(til._engine.tiles) == (4 if resident else 1)
    if force:
        c = til._engine.pass_counts()
        if resident:
            assert c["estep_sweep"] >= 4, c           # every tile carries its own bounds from iteration to iteration
        else:
            assert c["estep_bound"] >= 4 and c["estep_sweep"] == 0, c       # one workspace: fresh bound passes
    for key in ("hn_alpha_vec", "hn_m_vecs", "hn_kappas", "hn_nus", "hn_w_mats"):
        assert rel_err(til.get_hn_params()[key], one.get_hn_params()[key]) < 1e-9, key
    assert abs(til.vl - one.vl) < 1e-10 * abs(one.vl)
    assert rel_err(til.ns, one.ns) < 1e-10 and rel_err(til.s_mats, one.s_mats) < 1e-9
    assert np.max(np.abs(til.r_vecs - one.r_vecs)) < 1e-9                    # read-outs across the tiles
    z1 = one.estimate_latent_vars(x[40_000:70_000])
    z2 = til.estimate_latent_vars(x[40_000:70_000])
    assert np.array_equal(z1, z2)
    # and the oracle's posterior (8 iterations, 2 restarts) for the subsampling start
    if init_type == "subsampling" and not force:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            ref = orc.update_posterior(x.astype(np.float64), orc.Prior.default(K, D),
                                       orc.Posterior.from_prior(orc.Prior.default(K, D)), np.random.default_rng(0),
                                       max_itr=8, num_init=2, tolerance=0.0)
        assert rel_err(til.hn_m_vecs, ref.posterior.m) < 1e-7 and rel_err(til.hn_w_mats, ref.posterior.w) < 1e-7


def test_tiled_readout_after_loaded_responsibilities():
    """Responsibilities loaded into an engine that already holds parameters are what the read-outs return (not an E-step
    under those parameters), and an M-step after an E-step re-runs that E-step - same behaviour as DataPass."""
    from bayesml_amd._engine import DataPass, TiledDataPass
    K, D, N = 5, 32, 20_000
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(3)
    x = torch.from_numpy(orc.synth_gmm(K, D, N, np.float32)).to(dev)
    r = torch.from_numpy(rng.dirichlet(np.ones(K), N)).to(dev)
    c = torch.zeros(K, dtype=torch.float64, device=dev)
    m = torch.from_numpy(rng.standard_normal((K, D))).to(dev)
    u = torch.eye(D, dtype=torch.float64, device=dev).repeat(K, 1, 1).contiguous()
    outs = []
    for eng in (DataPass(K, D, x.dtype, N, dev), TiledDataPass(K, D, x.dtype, N, dev, 6016),
                TiledDataPass(K, D, x.dtype, N, dev, 6016, resident=True)):
        eng.set_pivot(torch.zeros(D, dtype=torch.float64, device=dev))
        eng.prepare_rows(x)
        eng.set_params(c, m, u)
        st_e = eng.estep_mstep(x).clone()
        r_e = eng.responsibilities().clone()
        eng.load_responsibilities(r)
        r_back = eng.responsibilities()
        assert torch.equal(r_back, r)                                        # the loaded values, not an E-step
        assert torch.equal(eng.argmax(), torch.argmax(r, dim=1).to(torch.int32))
        st_l = eng.mstep(x).clone()
        assert float((st_l[:K] - r.sum(dim=0)).abs().max()) < 1e-9
        eng.estep(x)
        assert float((eng.responsibilities() - r_e).abs().max()) < 1e-12      # back to the E-step under the parameters
        outs.append((st_e, st_l, r_e))
        eng.close()
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            assert float((a - b).abs().max() / a.abs().max()) < 1e-12


def test_resident_tiles_f64_rows_forced_sparse():
    """f64 rows: the list M-step reads the centred copy, which resident tiles share - after another tile's turn it is rebuilt
    from the tile's own (regrouped) rows."""
    from bayesml_amd._engine import TiledDataPass
    K, D, N = 12, 64, 100_000
    x = orc.synth_gmm(K, D, N, np.float64)
    one = _fit(x, K, D, 0, "subsampling", True)
    til = _fit(x, K, D, 30016, "subsampling", True, True)
    assert isinstance(til._engine, TiledDataPass) and til._engine.resident
    c = til._engine.pass_counts()
    assert c["estep_sweep"] >= 4 and c["mstep_list"] >= 4, c
    for key in ("hn_alpha_vec", "hn_m_vecs", "hn_kappas", "hn_nus", "hn_w_mats"):
        assert rel_err(til.get_hn_params()[key], one.get_hn_params()[key]) < 1e-9, key
    assert abs(til.vl - one.vl) < 1e-10 * abs(one.vl)


def test_resident_tiles_under_the_default_policy():
    """Tiles large enough for the default policy to prune (K x tile rows >= 2^23): every tile runs bound passes and sweeps
    of its own carried bounds, and the fit equals the untiled one."""
    from bayesml_amd._engine import TiledDataPass
    K, D, N = 64, 64, 600_000
    x = orc.synth_gmm(K, D, N, np.float32)
    one = _fit(x, K, D, 0, "subsampling", False, max_itr=12)
    til = _fit(x, K, D, 150_016, "subsampling", False, True, max_itr=12)
    eng = til._engine
    assert isinstance(eng, TiledDataPass) and eng.resident and eng.n_tiles == 4
    c = eng.pass_counts()
    assert c["estep_sweep"] >= 8 and c["mstep_list"] >= 8, c
    wk = eng.work()                                    # read back after the pass, summed over the tiles
    assert 0 < wk["evaluated"] < 0.2 * N * K and 0 < wk["active"] < 3 * N, wk
    for key in ("hn_alpha_vec", "hn_m_vecs", "hn_kappas", "hn_nus", "hn_w_mats"):
        assert rel_err(til.get_hn_params()[key], one.get_hn_params()[key]) < 1e-9, key
    assert abs(til.vl - one.vl) < 1e-10 * abs(one.vl)
    assert np.max(np.abs(til.r_vecs[::97] - one.r_vecs[::97])) < 1e-9
    # one set of pass-local buffers: less than half of four stand-alone workspaces
    alone = one._engine.workspace_bytes
    assert eng.workspace_bytes < 0.7 * alone, (eng.workspace_bytes, alone)


def test_tile_group_contract():
    """gmmvb_workspace_create_tile: the pass-local buffers hold one tile's E-step output at a time."""
    from bayesml_amd._engine import DataPass, EngineError
    K, D, n1, n2 = 6, 48, 5000, 3000
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(5)
    x = torch.from_numpy(orc.synth_gmm(K, D, n1 + n2, np.float32)).to(dev)
    xa, xb = x[:n1], x[n1:]
    c = torch.zeros(K, dtype=torch.float64, device=dev)
    m = torch.from_numpy(rng.standard_normal((K, D))).to(dev)
    u = torch.eye(D, dtype=torch.float64, device=dev).repeat(K, 1, 1).contiguous()
    piv = torch.zeros(D, dtype=torch.float64, device=dev)
    a = DataPass(K, D, x.dtype, n1, dev)
    b = DataPass(K, D, x.dtype, n2, dev, tile_of=a)
    alone = DataPass(K, D, x.dtype, n1, dev)
    with pytest.raises(EngineError, match="GMMVB_EINVAL"):
        DataPass(K, D, x.dtype, n1 + 64, dev, tile_of=a)
    assert b.workspace_bytes < a.workspace_bytes - K * n2 * 8
    for eng, xt in ((a, xa), (b, xb), (alone, xa)):
        eng.set_pivot(piv)
        eng.prepare_rows(xt)
        eng.set_params(c, m, u)
    want = alone.estep_mstep(xa).clone()
    r_want = alone.responsibilities().clone()
    a.estep(xa)
    b.estep(xb)                                        # takes the buffers over
    with pytest.raises(EngineError, match="GMMVB_ESTATE"):
        a.mstep(xa)
    with pytest.raises(EngineError, match="GMMVB_ESTATE"):
        a.responsibilities()
    sb = b.mstep(xb).clone()
    got = a.estep_mstep(xa).clone()                    # (its centred copy was lost too: rebuilt from the rows)
    assert torch.equal(got, want)
    assert torch.equal(a.responsibilities(), r_want)
    with pytest.raises(EngineError, match="GMMVB_ESTATE"):
        b.responsibilities()
    assert torch.equal(b.estep_mstep(xb), sb)
    with pytest.raises(EngineError, match="GMMVB_EUNSUPPORTED"):
        a.enable_hmm()
    a.close()                                          # any order: the buffers live as long as a member does
    assert torch.equal(b.estep_mstep(xb), sb)
    b.close()
    alone.close()
