"""The C-ABI library loads without a GPU and exports every symbol include/gmmvb.h declares."""
import os
import re

import pytest

from conftest import ROOT


def header_functions():
    text = open(os.path.join(ROOT, "include", "gmmvb.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b((?:gmmvb|hmmvb)_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_entry_points():
    names = header_functions()
    for must in ("gmmvb_workspace_create", "gmmvb_workspace_destroy", "gmmvb_set_params", "gmmvb_estep",
                 "gmmvb_mstep", "gmmvb_estep_mstep", "gmmvb_responsibilities", "gmmvb_argmax", "gmmvb_stats_len"):
        assert must in names


def test_library_exports_every_declared_symbol():
    from bayesml_amd import _engine
    if not os.path.exists(_engine.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = _engine.load_library()
    declared = header_functions()
    assert sorted(_engine.SYMBOLS) == declared, "ctypes table and header disagree"
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.gmmvb_abi_version() == 8
    assert lib.gmmvb_stats_len(64, 128) == 64 * (2 + 128 + 128 * 128)
    assert lib.gmmvb_stats_len(0, 4) == -1
    assert lib.gmmvb_stats_packed_len(64, 128) == 64 * (2 + 128 + 128 * 129 // 2)
    assert lib.gmmvb_stats_packed_len(3, 0) == -1


def test_argument_errors_without_a_gpu():
    """Pure argument validation returns error codes (no compute, no device needed)."""
    import ctypes
    from bayesml_amd import _engine
    lib = _engine.load_library()
    h = ctypes.c_void_p()
    assert lib.gmmvb_workspace_create(0, 4, 0, 10, ctypes.byref(h)) == 1          # GMMVB_EINVAL
    assert lib.gmmvb_workspace_create(4, 4, 0, 0, ctypes.byref(h)) == 1
    assert lib.gmmvb_workspace_create(4, 4, 7, 10, ctypes.byref(h)) == 1
    assert b"x_dtype" in lib.gmmvb_last_error()
    assert lib.gmmvb_workspace_destroy(None) == 0


def test_product_path_fails_loudly_without_gpu():
    import numpy as np
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from bayesml_amd import gaussianmixture as gm
    from bayesml_amd._engine import EngineUnavailableError
    m = gm.LearnModel(3, 2, seed=0)
    with pytest.raises(EngineUnavailableError):
        m.update_posterior(np.zeros((10, 2)))
    with pytest.raises(EngineUnavailableError):
        m.estimate_latent_vars(np.zeros((10, 2)))
