"""Row sharding through the REAL HIP engine: two processes share cuda:0 and exchange the statistics block
over gloo (RCCL needs one device per rank; the collective is the same torch.distributed.all_reduce call that
bench.py issues over "nccl").  The sharded posterior must equal the single-process one."""
import io
import os
import socket
import warnings
from contextlib import redirect_stdout

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import load_golden, mat_functionals, rel_err
from oracle import gmm_vb_oracle as orc

pytestmark = pytest.mark.gpu

K, D, N = 8, 32, 20000
KW = dict(num_init=2, max_itr=8, tolerance=0.0)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, cuts, out_dir):
    from bayesml_amd import RowShard
    from bayesml_amd import gaussianmixture as gm
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    x = orc.synth_gmm(K, D, N, np.float32)
    m = gm.LearnModel(K, D, seed=0, comm=RowShard(), device="cuda:0", verbose=False)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m.update_posterior(x[cuts[rank]:cuts[rank + 1]], **KW)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), vl=m.vl, ns=m.ns, **m.get_hn_params())
    dist.destroy_process_group()


def test_two_ranks_on_one_gpu_match_single_process(tmp_path):
    from bayesml_amd import gaussianmixture as gm
    cuts = [0, 7777, N]
    mp.spawn(_worker, args=(2, _free_port(), cuts, str(tmp_path)), nprocs=2, join=True)
    x = orc.synth_gmm(K, D, N, np.float32)
    one = gm.LearnModel(K, D, seed=0, verbose=False)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        one.update_posterior(x, **KW)
    for r in range(2):
        res = dict(np.load(os.path.join(str(tmp_path), f"rank{r}.npz")))
        for key in ("hn_alpha_vec", "hn_m_vecs", "hn_kappas", "hn_nus", "hn_w_mats"):
            assert rel_err(res[key], one.get_hn_params()[key]) < 1e-9, (r, key)
        assert rel_err(res["ns"], one.ns) < 1e-9
        assert abs(float(res["vl"]) - one.vl) < 1e-9 * abs(one.vl)


@pytest.mark.gpu
def test_c_abi_rccl_communicator_single_rank():
    """gmmvb_comm_* / gmmvb_allreduce_stats (include/gmmvb.h): librccl is resolved at run time, a one-rank communicator
    is created on the GPU and the in-place all-reduce of a statistics block leaves it unchanged.  (Two RCCL ranks need
    two GPUs; the N > 1 path is what bench.py --gpus N runs.)"""
    import torch
    from bayesml_amd._engine import RcclComm
    dev = torch.device("cuda", 0)
    comm = RcclComm(0, 1, dev, bootstrap=lambda b: b)
    t = torch.arange(64 * (2 + 128 + 128 * 128), dtype=torch.float64, device=dev)
    ref = t.clone()
    comm.all_reduce_(t)
    torch.cuda.synchronize()
    assert torch.equal(t, ref)
    comm.close()


def test_c_abi_comm_argument_errors():
    import ctypes
    from bayesml_amd import _engine
    lib = _engine.load_library()
    h = ctypes.c_void_p()
    buf = (ctypes.c_ubyte * 128)()
    assert lib.gmmvb_comm_create(buf, 2, 2, ctypes.byref(h)) == 1          # rank outside [0, n_ranks)
    assert lib.gmmvb_comm_create(None, 1, 0, ctypes.byref(h)) == 1
    assert lib.gmmvb_allreduce_stats(None, None, 0, None) == 1
    assert lib.gmmvb_comm_destroy(None) == 0


# ---- sharded AND sparse ----------------------------------------------------------------------------------------------
def _sparse_worker(rank, world, port, out_dir, divergent, tile_rows=0):
    """The reference fixture K=64, D=128, N=140000 split 35 % / 65 % over two ranks, pruning forced (a shard is below the
    default policy's size threshold).  divergent: rank 1 additionally ignores the drift hint, so it runs a fresh bound pass
    wherever rank 0 carries - results must not depend on it.  Otherwise both ranks decide from the job-wide counters that
    travel with the statistics block (gmmvb_policy_export / import) and must run the same kernels in every pass although
    their own shards' counters differ."""
    import json
    os.environ["GMMVB_ESTEP_PRUNE"] = "force"
    if divergent and rank == 1:
        os.environ["GMMVB_ESTEP_CARRY_OFF"] = "1"
    if tile_rows:           # every rank's shard through resident row tiles (a workspace per tile, gmmvb_workspace_create_tile)
        os.environ["BAYESML_AMD_TILE_ROWS"] = str(tile_rows)
        os.environ["BAYESML_AMD_TILE_RESIDENT"] = "1"
    from bayesml_amd import RowShard
    from bayesml_amd import gaussianmixture as gm
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = load_golden("gmm_f3_k64_d128_n140000_f32.npz")
    Kk, Dd, Nn = int(g["K"]), int(g["D"]), int(g["N"])
    x = orc.synth_gmm(int(g["K_data"]), Dd, Nn, np.float32, spread=float(g["spread"]))
    cut = int(0.35 * Nn) + 333
    lo, hi = (0, cut) if rank == 0 else (cut, Nn)
    m = gm.LearnModel(Kk, Dd, seed=int(g["seed"]), comm=RowShard(), device="cuda:0", verbose=False)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m.update_posterior(x[lo:hi], **json.loads(str(g["kw"])))
    counts = m._engine.pass_counts()
    counts["tiles"] = getattr(m._engine, "n_tiles", 1)
    np.savez(os.path.join(out_dir, f"sparse_rank{rank}.npz"), vl=m.vl, ns=m.ns, x_bar=m.x_bar_vecs,
             w_inv=m.hn_w_mats_inv, s=m.s_mats, counts=json.dumps(counts), **m.get_hn_params())
    dist.destroy_process_group()


@pytest.mark.parametrize("divergent", [True, False])
def test_two_sparse_ranks_match_the_reference(tmp_path, divergent):
    import json
    mp.spawn(_sparse_worker, args=(2, _free_port(), str(tmp_path), divergent), nprocs=2, join=True)
    g = load_golden("gmm_f3_k64_d128_n140000_f32.npz")
    res = [dict(np.load(os.path.join(str(tmp_path), f"sparse_rank{r}.npz"))) for r in range(2)]
    c0, c1 = (json.loads(str(r["counts"])) for r in res)
    assert c0["estep_bound"] >= 1 and c0["estep_sweep"] >= 3 and c0["mstep_list"] >= 3, c0
    if divergent:
        assert c1["estep_sweep"] == 0 and c1["estep_bound"] > c0["estep_bound"], (c0, c1)
    else:           # one policy for both ranks: the same kind of E-step in every pass
        for key in ("estep_dense", "estep_bound", "estep_fell_back_dense", "estep_sweep"):
            assert c0[key] == c1[key], (key, c0, c1)
    for r in res:
        for key in ("hn_alpha_vec", "hn_m_vecs", "hn_kappas", "hn_nus"):
            assert rel_err(r[key], g[key]) < 1e-6, key
        for key, got in (("hn_w_mats", r["hn_w_mats"]), ("hn_w_mats_inv", r["w_inv"])):
            for fn, val in mat_functionals(got).items():
                ref = g[f"{key}_{fn}"]
                if fn == "logabsdet":
                    assert np.max(np.abs(val - ref)) < 1e-6 * max(1.0, float(np.max(np.abs(ref)))), (key, fn)
                else:
                    assert rel_err(val, ref) < 1e-6, (key, fn)
        assert rel_err(r["ns"], g["ns"]) < 1e-6 and rel_err(r["x_bar"], g["x_bar_vecs"]) < 1e-6
        assert abs(float(r["vl"]) - float(g["final_vl"])) <= 1e-8 * abs(float(g["final_vl"]))
    for key in ("hn_m_vecs", "hn_w_mats", "ns"):          # both ranks hold the same posterior, bit for bit
        assert np.array_equal(res[0][key], res[1][key]), key


# ---- restart-level parallelism on the real engine ---------------------------------------------------------------------
def _restart_worker(rank, world, port, out_dir):
    import io
    from contextlib import redirect_stdout
    from bayesml_amd import RestartShard
    from bayesml_amd import gaussianmixture as gm
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    x = load_golden("gmm_c1_sample.npz")["x"]
    m = gm.LearnModel(3, 2, seed=0, comm=RestartShard(), device="cuda:0")
    with redirect_stdout(io.StringIO()) as buf, warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        m.update_posterior(x)
    np.savez(os.path.join(out_dir, f"restart_rank{rank}.npz"), vl=m.vl, ns=m.ns, stdout=buf.getvalue(), warned=len(w) > 0,
             **m.get_hn_params())
    dist.destroy_process_group()


def test_restarts_over_two_ranks_on_the_real_engine(tmp_path):
    """comm = RestartShard (reference loop ``_gaussianmixture.py:847-883``) with the HIP engine: restart i
    runs on rank i mod 2; winner, lower-bound trace and posterior are those of the single-process GPU run, bit for bit,
    and the reference's (fixture gmm_f3_c1_subsampling.npz)."""
    import io
    from contextlib import redirect_stdout
    from bayesml_amd import gaussianmixture as gm
    mp.spawn(_restart_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    g = load_golden("gmm_f3_c1_subsampling.npz")
    x = load_golden("gmm_c1_sample.npz")["x"]
    # (the single-process run at this size would take the one-launch small-problem path, whose sums run in another order:
    # bit-for-bit equality is a statement about the general engine, which is what the sharded ranks ran)
    old = os.environ.get("BAYESML_AMD_SMALL")
    os.environ["BAYESML_AMD_SMALL"] = "0"
    try:
        one = gm.LearnModel(3, 2, seed=0, device="cuda:0")
        with redirect_stdout(io.StringIO()) as buf, warnings.catch_warnings():
            warnings.simplefilter("ignore")
            one.update_posterior(x)
    finally:
        os.environ.pop("BAYESML_AMD_SMALL", None)
        if old is not None:
            os.environ["BAYESML_AMD_SMALL"] = old
    assert one._engine is not None
    text = buf.getvalue()
    ranks = [dict(np.load(os.path.join(str(tmp_path), f"restart_rank{r}.npz"))) for r in range(2)]
    for r, res in enumerate(ranks):
        for key in ("hn_alpha_vec", "hn_m_vecs", "hn_kappas", "hn_nus", "hn_w_mats"):
            assert np.array_equal(res[key], one.get_hn_params()[key]), (r, key)
            assert rel_err(res[key], g[key]) < 1e-7, (r, key)
        assert float(res["vl"]) == one.vl
        assert bool(res["warned"]) == bool(g["result_warning"])
    assert str(ranks[0]["stdout"]) == text and str(ranks[1]["stdout"]) == ""
    stars = [ln.endswith("*") for ln in text.split("\n") if ln.strip()]
    assert max(i for i, s in enumerate(stars) if s) == int(g["winner"])
    lines = [ln for ln in text.split("\n") if ln.strip()]
    tr = g["vl_trace"]
    for i, ln in enumerate(lines):
        vals = [float(seg.split("VL: ")[1].split(" ")[0].rstrip("*").replace("(converged)", "")) for seg in ln.split("\r") if seg]
        ref = tr[i][~np.isnan(tr[i])]
        assert len(vals) == len(ref) and np.allclose(vals, ref, rtol=1e-9, atol=0), i


def test_two_ranks_of_resident_tiles_match_the_reference(tmp_path):
    """Row shards AND row tiles: each of the two ranks runs its shard of the reference fixture through resident tiles of
    24000 rows (3 and 4 of them); the tiles of both ranks decide from the job-wide counters (the policy tail is summed over
    tiles, then over ranks) and every tile carries its own bounds."""
    import json
    mp.spawn(_sparse_worker, args=(2, _free_port(), str(tmp_path), False, 24000), nprocs=2, join=True)
    g = load_golden("gmm_f3_k64_d128_n140000_f32.npz")
    res = [dict(np.load(os.path.join(str(tmp_path), f"sparse_rank{r}.npz"))) for r in range(2)]
    c0, c1 = (json.loads(str(r["counts"])) for r in res)
    assert (c0["tiles"], c1["tiles"]) == (3, 4), (c0, c1)
    for c in (c0, c1):
        assert c["estep_bound"] >= c["tiles"] and c["estep_sweep"] >= 3 * c["tiles"] and c["mstep_list"] >= 3 * c["tiles"], c
    # one policy for all tiles of all ranks: the same kinds of pass per tile
    for key in ("estep_dense", "estep_bound", "estep_fell_back_dense", "estep_sweep"):
        assert c0[key] * c1["tiles"] == c1[key] * c0["tiles"], (key, c0, c1)
    for r in res:
        for key in ("hn_alpha_vec", "hn_m_vecs", "hn_kappas", "hn_nus"):
            assert rel_err(r[key], g[key]) < 1e-6, key
        assert rel_err(r["ns"], g["ns"]) < 1e-6 and rel_err(r["x_bar"], g["x_bar_vecs"]) < 1e-6
        assert abs(float(r["vl"]) - float(g["final_vl"])) <= 1e-8 * abs(float(g["final_vl"]))
    for key in ("hn_m_vecs", "hn_w_mats", "ns"):          # both ranks hold the same posterior, bit for bit
        assert np.array_equal(res[0][key], res[1][key]), key



@pytest.mark.gpu
@pytest.mark.parametrize("K,D", [(64, 128), (256, 64), (3, 5), (1, 1)])
def test_wire_block_kernels_match_the_index_map(K, D):
    """gmmvb_stats_pack / gmmvb_stats_unpack against the torch index map of _kside.stats_triangle (the CPU tests' path)."""
    from bayesml_amd import _kside
    g = torch.Generator().manual_seed(K * 1000 + D)
    head = torch.randn(K * (2 + D), dtype=torch.float64, generator=g)
    B = torch.randn(K, D, D, dtype=torch.float64, generator=g)
    B = B + B.transpose(1, 2)
    full = torch.cat([head, B.reshape(-1)])
    n = _kside.packed_stats_len(K, D)
    want = torch.zeros(n, dtype=torch.float64)
    _kside.stats_triangle(True, K, D, full, want)
    dev = torch.device("cuda", 0)
    got = torch.full((n,), float("nan"), dtype=torch.float64, device=dev)
    _kside.stats_triangle(True, K, D, full.to(dev), got)
    assert torch.equal(got.cpu(), want)
    back = torch.full_like(full, float("nan")).to(dev)
    _kside.stats_triangle(False, K, D, got, back)
    assert torch.equal(back.cpu(), full)
