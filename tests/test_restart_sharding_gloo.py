"""world_size 2 and 3 (gloo, CPU): restarts of update_posterior spread over processes (comm = RestartShard, SURVEY.md
section 8f.2) give the single-process result - same winner, same posterior, same progress lines - and the reference's.
The data pass is the CPU stand-in of tests/fake_engine.py."""
import io
import os
import socket
import warnings
from contextlib import redirect_stdout

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import load_golden, rel_err


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, seed, kw, out_dir):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from fake_engine import cpu_factory
    from bayesml_amd import RestartShard
    from bayesml_amd import gaussianmixture as gm
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    x = load_golden("gmm_c1_sample.npz")["x"]
    m = gm.LearnModel(3, 2, seed=seed, comm=RestartShard())
    m._data_pass_factory = cpu_factory
    with redirect_stdout(io.StringIO()) as buf, warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        m.update_posterior(x, **kw)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), vl=m.vl, ns=m.ns, stdout=buf.getvalue(), warned=len(w) > 0,
             vl_q_z=m._vl_q_z, **m.get_hn_params())
    dist.destroy_process_group()


def _single(seed, kw):
    from fake_engine import cpu_factory
    from bayesml_amd import gaussianmixture as gm
    m = gm.LearnModel(3, 2, seed=seed)
    m._data_pass_factory = cpu_factory
    with redirect_stdout(io.StringIO()) as buf, warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m.update_posterior(load_golden("gmm_c1_sample.npz")["x"], **kw)
    return m, buf.getvalue()


@pytest.mark.parametrize("world,seed,kw,fixture", [
    (2, 0, dict(), "gmm_f3_c1_subsampling.npz"),
    (3, 5, dict(num_init=4, init_type="random_responsibility"), "gmm_f3_c1_random_resp.npz"),
    (2, 1, dict(num_init=2, max_itr=3, tolerance=0.0), "gmm_f3_c1_noconv.npz"),
])
def test_restarts_over_ranks_equal_single_process_and_reference(tmp_path, world, seed, kw, fixture):
    mp.spawn(_worker, args=(world, _free_port(), seed, kw, str(tmp_path)), nprocs=world, join=True)
    ranks = [dict(np.load(os.path.join(str(tmp_path), f"rank{r}.npz"))) for r in range(world)]
    one, text = _single(seed, kw)
    g = load_golden(fixture)
    for r, res in enumerate(ranks):
        for key in ("hn_alpha_vec", "hn_m_vecs", "hn_kappas", "hn_nus", "hn_w_mats"):
            assert np.array_equal(res[key], one.get_hn_params()[key]), (r, key)      # same arithmetic, same bits
            assert rel_err(res[key], g[key]) < 1e-7, (r, key)                        # and the reference's posterior
        assert float(res["vl"]) == one.vl and float(res["vl_q_z"]) == one._vl_q_z
        assert bool(res["warned"]) == bool(g["result_warning"])
    assert str(ranks[0]["stdout"]) == text                          # rank 0 prints the same lines, stars included
    assert all(str(res["stdout"]) == "" for res in ranks[1:])
    stars = [ln.endswith("*") for ln in text.split("\n") if ln.strip()]
    assert max(i for i, s in enumerate(stars) if s) == int(g["winner"])
