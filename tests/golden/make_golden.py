#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by running the REFERENCE itself.

Run in the build container only (it imports ``/root/reference``, which does not exist on
the GPU box and is never copied):

    MPLBACKEND=Agg python tests/golden/make_golden.py

Outputs (committed): ``gmm_*.npz`` + ``gmm_errors.json``.  A fixture is data only: inputs
(or the seed/recipe that regenerates them, with a checksum) and the reference's outputs.
Fixture families follow SURVEY.md section 8c (F1 single E+M step, F2 K-side step, F3 full
driver, F4 read-outs, F5 boundary errors).
"""
import contextlib
import hashlib
import io
import json
import os
import re
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
os.environ.setdefault("MPLBACKEND", "Agg")

from bayesml import gaussianmixture as ref_gm          # noqa: E402  (the reference)
from bayesml._exceptions import ResultWarning          # noqa: E402
from oracle.gmm_vb_oracle import synth_gmm             # noqa: E402  (own data recipe)


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def hn_state(m) -> dict:
    return dict(hn_alpha_vec=m.hn_alpha_vec.copy(), hn_m_vecs=m.hn_m_vecs.copy(),
                hn_kappas=m.hn_kappas.copy(), hn_nus=m.hn_nus.copy(),
                hn_w_mats=m.hn_w_mats.copy(), hn_w_mats_inv=m.hn_w_mats_inv.copy())


def feat_state(m) -> dict:
    return dict(e_ln_pi_vec=m._e_ln_pi_vec.copy(), e_ln_lambda_dets=m._e_ln_lambda_dets.copy(),
                e_lambda_mats=m._e_lambda_mats.copy(), ln_b_hn_w_nus=m._ln_b_hn_w_nus.copy())


def vl_terms(m) -> dict:
    return dict(vl=m.vl, vl_p_x=m._vl_p_x, vl_p_z=m._vl_p_z, vl_p_pi=m._vl_p_pi,
                vl_p_mu_lambda=m._vl_p_mu_lambda, vl_q_z=m._vl_q_z, vl_q_pi=m._vl_q_pi,
                vl_q_mu_lambda=m._vl_q_mu_lambda)


def quiet(fn, *a, **k):
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        out = fn(*a, **k)
    return out, buf.getvalue()


# ------------------------------------------------------------------ F1 + F2: single steps
def single_step(name, K, D, x, seed, store_x, row_keep=None, warm_iters=0):
    """State after _init_subsampling (optionally after `warm_iters` VB iterations so that
    the posterior is well conditioned), one _update_q_z, _calc_vl, then one K-side step."""
    m = ref_gm.LearnModel(K, D, seed=seed)
    N = x.shape[0]
    m._ln_rho = np.empty((N, K))
    m.r_vecs = np.empty((N, K))
    m.s_mats[:] = 0.0                      # reference leaves np.empty; pin the undefined value
    m.reset_hn_params()
    m._init_rho_r()
    m._init_subsampling(x)
    for _ in range(warm_iters):
        m._update_q_z(x)
        m._update_q_mu_lambda()
        m._update_q_pi()
    out = {}
    out.update({"in_" + k: v for k, v in hn_state(m).items()})
    out.update({"in_" + k: v for k, v in feat_state(m).items()})
    m._update_q_z(x)
    m._calc_vl()
    keep = slice(None) if row_keep is None else slice(0, row_keep)
    out.update(ln_rho=m._ln_rho[keep].copy(), r_vecs=m.r_vecs[keep].copy(), ns=m.ns.copy(),
               x_bar_vecs=m.x_bar_vecs.copy(), s_mats=m.s_mats.copy(),
               r_colsum=m.r_vecs.sum(axis=0), ln_rho_rowmax_sum=float(m._ln_rho.max(axis=1).sum()))
    out.update(vl_terms(m))
    # F2: the K-side step that follows
    m._update_q_mu_lambda()
    m._update_q_pi()
    out.update({"out_" + k: v for k, v in hn_state(m).items()})
    out.update({"out_" + k: v for k, v in feat_state(m).items()})
    out.update(K=K, D=D, N=N, seed=seed, x_sha256=sha(x), x_dtype=str(x.dtype), warm_iters=warm_iters)
    if store_x:
        out["x"] = x
    np.savez_compressed(os.path.join(HERE, name), **out)
    print("wrote", name, {k: getattr(v, "shape", v) for k, v in out.items() if k in ("x", "ln_rho")})


# ------------------------------------------------------------------ F3 + F4: full driver
LINE = re.compile(r"^(\d+)\. VL: (\S+)(?: t=(\d+) )?(\(converged\))?(\*)?$")


def parse_trace(text):
    """stdout protocol of the reference driver (_gaussianmixture.py:861,868,871,874,883)."""
    traces, winners, converged = [], [], []
    for line in text.split("\n"):
        if not line.strip():
            continue
        segs = [s for s in line.split("\r") if s]
        vals, star, conv = [], False, False
        for s in segs:
            mm = LINE.match(s)
            assert mm, repr(s)
            vals.append(float(mm.group(2)))
            conv = conv or bool(mm.group(4))
            star = star or bool(mm.group(5))
        traces.append(vals)
        winners.append(star)
        converged.append(conv)
    return traces, winners, converged


def full_driver_state(K, D, x, seed, **kw):
    """Run the reference's update_posterior on x.astype(float64) and collect what the driver fixtures hold."""
    m = ref_gm.LearnModel(K, D, seed=seed)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        _, text = quiet(m.update_posterior, x.astype(np.float64), **kw)
    warned = any(issubclass(i.category, ResultWarning) for i in w)
    traces, winners, converged = parse_trace(text)
    winner = max(i for i, s in enumerate(winners) if s)
    L = max(len(t) for t in traces)
    tr = np.full((len(traces), L), np.nan)
    for i, t in enumerate(traces):
        tr[i, :len(t)] = t
    out = dict(K=K, D=D, N=x.shape[0], seed=seed, x_sha256=sha(x), x_dtype=str(x.dtype),
               vl_trace=tr, winner=winner, converged=np.array(converged), result_warning=warned,
               final_vl=m.vl, ns=m.ns.copy(), x_bar_vecs=m.x_bar_vecs.copy(), s_mats=m.s_mats.copy(),
               r_colsum=m.r_vecs.sum(axis=0), r_head=m.r_vecs[:64].copy(),
               kw=json.dumps(kw))
    out.update(hn_state(m))
    out.update(feat_state(m))
    out["_model"] = m
    return out


def full_driver(name, K, D, x, seed, store_x, readouts=True, **kw):
    """The reference is fed x.astype(float64): for float32 rows its _init_subsampling would otherwise sum
    the subsample in float32 (`_subsample.sum(axis=0)`, _gaussianmixture.py:790), a 1e-7 perturbation of
    the start point that the VB transient amplifies to ~2e-6.  "Identical inputs" means identical values
    (SURVEY.md section 7); the engine stores the same float32 values and widens them on load."""
    out = full_driver_state(K, D, x, seed, **kw)
    m = out.pop("_model")
    winner, warned, traces = out["winner"], out["result_warning"], out["vl_trace"]
    if not readouts:
        out.pop("e_lambda_mats")          # = hn_nus * hn_w_mats; keep the big-D fixture small
        np.savez_compressed(os.path.join(HERE, name), **out)
        print("wrote", name, "winner", winner, "vl", m.vl, "warned", warned)
        return
    # F4 read-outs on the final posterior
    pi_sq, mu_sq, lam_sq = m.estimate_params("squared")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        pi_01, mu_01, lam_01 = m.estimate_params("0-1")
    out.update(est_sq_pi=pi_sq, est_sq_lambda=np.array(lam_sq), est_01_pi=pi_01, est_01_lambda=lam_01)
    # NOTE: update_posterior never calls calc_pred_dist (the last call is inside the last
    # restart's reset_hn_params, _gaussianmixture.py:848 -> :640), so p_* are STALE (prior-derived) here.
    out.update({"stale_" + k: np.array(v) for k, v in m.get_p_params().items()})
    out.update(stale_p_pi_vec=m.p_pi_vec.copy(), stale_pred_squared=m.make_prediction("squared"))
    m.calc_pred_dist()
    out.update({k: np.array(v) for k, v in m.get_p_params().items()})
    out.update(p_pi_vec=m.p_pi_vec.copy(), pred_squared=m.make_prediction("squared"),
               pred_01=m.make_prediction("0-1"))
    xs = x[:128].astype(np.float64)
    out.update(latent_01=m.estimate_latent_vars(xs, "0-1"), latent_sq=m.estimate_latent_vars(xs, "squared").copy())
    if store_x:
        out["x"] = x
    np.savez_compressed(os.path.join(HERE, name), **out)
    print("wrote", name, "winner", winner, "vl", m.vl, "warned", warned, "iters", traces.shape)


# ------------------------------------------------------------------ F5: boundary errors
def boundary_errors():
    cases = {
        "ctor_float_degree": lambda: ref_gm.LearnModel(3, 2.0),
        "ctor_zero_classes": lambda: ref_gm.LearnModel(0, 2),
        "ctor_bool_like_negative": lambda: ref_gm.LearnModel(3, -1),
        "h0_m_vecs_wrong_dim": lambda: ref_gm.LearnModel(3, 2, h0_m_vecs=np.zeros((3, 3))),
        "h0_w_mats_not_pd": lambda: ref_gm.LearnModel(3, 2, h0_w_mats=np.array([[1.0, 2.0], [2.0, 1.0]])),
        "h0_w_mats_not_sym": lambda: ref_gm.LearnModel(3, 2, h0_w_mats=np.array([[1.0, 0.5], [0.0, 1.0]])),
        "h0_nus_too_small": lambda: ref_gm.LearnModel(3, 2, h0_nus=1.0),
        "h0_alpha_nonpos": lambda: ref_gm.LearnModel(3, 2, h0_alpha_vec=np.array([1.0, 0.0, 1.0])),
        "h0_kappas_negative": lambda: ref_gm.LearnModel(3, 2, h0_kappas=-1.0),
        "x_wrong_last_dim": lambda: quiet(ref_gm.LearnModel(3, 2).update_posterior, np.zeros((10, 3))),
        "x_not_ndarray": lambda: quiet(ref_gm.LearnModel(3, 2).update_posterior, [[0.0, 1.0]]),
        "x_complex": lambda: quiet(ref_gm.LearnModel(3, 2).update_posterior, np.zeros((4, 2), dtype=complex)),
        "bad_init_type": lambda: quiet(ref_gm.LearnModel(3, 2, seed=0).update_posterior,
                                       np.random.default_rng(0).standard_normal((50, 2)), init_type="kmeans"),
        "bad_loss_estimate_params": lambda: ref_gm.LearnModel(3, 2).estimate_params("L1"),
        "bad_loss_make_prediction": lambda: ref_gm.LearnModel(3, 2).make_prediction("KL"),
        "bad_loss_latent": lambda: ref_gm.LearnModel(3, 2).estimate_latent_vars(np.zeros((4, 2)), "L1"),
        "pred_and_update_wrong_shape": lambda: quiet(ref_gm.LearnModel(3, 2).pred_and_update, np.zeros((1, 2))),
        "gen_pi_not_sum1": lambda: ref_gm.GenModel(3, 2, pi_vec=np.array([0.5, 0.4, 0.2])),
        "gen_sample_size_float": lambda: ref_gm.GenModel(3, 2).gen_sample(10.0),
        "visualize_d3": lambda: quiet(ref_gm.LearnModel(2, 3).visualize_posterior),
    }
    ok_cases = {
        "ctor_numpy_int": lambda: ref_gm.LearnModel(np.int64(3), np.int32(2)),
        "h0_scalar_broadcast": lambda: ref_gm.LearnModel(3, 2, h0_kappas=2.0, h0_nus=3, h0_w_mats=np.eye(2) * 2),
        "x_int_dtype": lambda: quiet(ref_gm.LearnModel(2, 2, seed=0).update_posterior,
                                     np.random.default_rng(0).integers(-5, 5, (40, 2)), num_init=1, max_itr=2),
        "x_3d_reshaped": lambda: quiet(ref_gm.LearnModel(2, 2, seed=0).update_posterior,
                                       np.random.default_rng(0).standard_normal((5, 8, 2)), num_init=1, max_itr=2),
    }
    res = {}
    for name, fn in {**cases, **ok_cases}.items():
        try:
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                fn()
            res[name] = None
        except Exception as e:      # noqa: BLE001
            res[name] = type(e).__name__
    with open(os.path.join(HERE, "gmm_errors.json"), "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)
    print("wrote gmm_errors.json", res)


def main():
    # config 1 data: the reference's own GenModel (per-sample Python loop, _gaussianmixture.py:241-264)
    gen = ref_gm.GenModel(3, 2, pi_vec=np.array([0.3, 0.3, 0.4]),
                          mu_vecs=np.array([[-4.0, 0.0], [4.0, 0.0], [0.0, 5.0]]),
                          lambda_mats=np.array([[[1.0, 0.3], [0.3, 2.0]], [[2.0, 0.0], [0.0, 0.5]], [[1.0, -0.4], [-0.4, 1.0]]]),
                          seed=123)
    x1, z1 = gen.gen_sample(1000)
    np.savez_compressed(os.path.join(HERE, "gmm_c1_sample.npz"), x=x1, z=z1)

    single_step("gmm_f1_c1_k3_d2_n1000.npz", 3, 2, x1, seed=0, store_x=True)
    x2 = synth_gmm(16, 32, 2048, np.float64)
    single_step("gmm_f1_k16_d32_n2048.npz", 16, 32, x2, seed=0, store_x=True)
    x3 = synth_gmm(8, 128, 32768, np.float32)
    single_step("gmm_f1_k4_d128_n32768_f32.npz", 4, 128, x3, seed=0, store_x=False, row_keep=256, warm_iters=2)
    x3b = synth_gmm(8, 64, 1024, np.float32)        # sqrt(N) < D: rank-deficient init, |ln_rho| ~ 1e7
    single_step("gmm_f1_k8_d64_n1024_f32_illcond.npz", 8, 64, x3b, seed=0, store_x=True)

    full_driver("gmm_f3_c1_subsampling.npz", 3, 2, x1, seed=0, store_x=False)
    full_driver("gmm_f3_c1_random_resp.npz", 3, 2, x1, seed=5, store_x=False, num_init=4,
                init_type="random_responsibility")
    full_driver("gmm_f3_c1_noconv.npz", 3, 2, x1, seed=1, store_x=False, num_init=2, max_itr=3, tolerance=0.0)
    x4 = synth_gmm(16, 32, 16384, np.float64)
    full_driver("gmm_f3_k16_d32_n16384.npz", 16, 32, x4, seed=0, store_x=False, num_init=1, max_itr=10,
                tolerance=0.0)
    full_driver("gmm_f3_k8_d128_n32768_f32.npz", 8, 128, x3, seed=0, store_x=False, readouts=False,
                num_init=1, max_itr=10, tolerance=0.0)
    x5 = synth_gmm(4, 1, 1, np.float64)[:1]            # N = 1 (pred_and_update path, :1148)
    full_driver("gmm_f3_n1.npz", 2, 1, x5, seed=0, store_x=True, num_init=2, max_itr=5,
                init_type="random_responsibility")
    boundary_errors()


if __name__ == "__main__":
    main()
