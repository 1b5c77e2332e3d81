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

from conftest import rel_err
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
