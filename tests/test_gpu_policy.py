"""The pass policy's unit costs (csrc/policy.h, include/gmmvb.h gmmvb_policy_table / gmmvb_policy_calibrate): literals scaled to
the workspace's shape, replaced by what the workspace's own first dense E-step, dense M-step and bound pass take on the
device.  Results never depend on the table - the fits below equal each other to rounding with and without calibration."""
import warnings

import numpy as np
import pytest
import torch

from conftest import rel_err
from oracle import gmm_vb_oracle as orc

pytestmark = pytest.mark.gpu


def _fit(K, D, N, calibrate):
    from bayesml_amd import gaussianmixture as gm
    x = orc.synth_gmm(K, D, N, np.float32)
    m = gm.LearnModel(K, D, seed=0, device=torch.device("cuda", 0), verbose=False)
    eng, xd = m._open(x)
    before = eng.policy_table()
    if not calibrate:
        eng.lib.gmmvb_policy_calibrate(eng._ws, 0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m.update_posterior(xd, max_itr=12, num_init=1, tolerance=0.0)
    return m, before, m._engine.policy_table(), m._engine.pass_counts()


@pytest.mark.parametrize("K,D,N", [(64, 128, 140_000), (256, 64, 36_000)])
def test_policy_table_is_calibrated_from_the_fits_own_passes(K, D, N):
    m1, before, after, counts = _fit(K, D, N, True)
    # the literals at this shape: the benchmark shape's numbers scaled by the tile-pair counts
    f = (D // 16) * (D // 16 + 1) / 2 / 36.0
    assert abs(before["literal_dense_e_ns"] - 0.269 * f) < 1e-12 and before["measured"] == 0 and before["calibrating"]
    assert 0.30 < before["prune_below"] < 0.55 and 0.45 < before["dense_again_above"] < 0.75 and 0.5 < before["list_m_below"] < 0.7
    # the dense E- and M-step of the restart's first passes were timed (N K >= 2^23) and lie within [1/2, 2] x the literal
    assert counts["estep_dense"] >= 1 and counts["mstep_dense"] >= 1
    # (a measurement outside that range - the process's first launch of a kernel pays its code upload - is discarded and
    # tried again on a later pass: at these sizes a fit has one or two dense passes, so at least one of the two is in)
    assert after["measured"] & 3, after
    assert 0.5 * after["literal_dense_e_ns"] <= after["dense_e_ns"] <= 2.0 * after["literal_dense_e_ns"]
    assert 0.5 * after["literal_dense_m_ns"] <= after["dense_m_ns"] <= 2.0 * after["literal_dense_m_ns"]
    assert 0.25 < after["prune_below"] < 0.7 and 0.4 < after["list_m_below"] < 0.8, after
    # switched off: the scaled literals stay, and the fit is the same to rounding (only kernel choices may differ)
    m0, _b, lit, _c = _fit(K, D, N, False)
    assert lit["measured"] == 0 and not lit["calibrating"] and lit["dense_e_ns"] == lit["literal_dense_e_ns"]
    for key in ("hn_m_vecs", "hn_w_mats", "hn_alpha_vec"):
        assert rel_err(m1.get_hn_params()[key], m0.get_hn_params()[key]) < 1e-9, key
