"""``update_posterior`` for small problems: every restart and every VB iteration in ONE kernel launch.

The reference's driver (``_gaussianmixture.py:846-896``) is ``for i in range(num_init)`` x ``for t in range(max_itr)``
around K-sized closed forms and two passes over x.  At the sizes its tutorials use (K = 3, D = 2, N = 1000) an iteration
is a few thousand flops; the general engine spends ~0.1 ms per iteration on launches and its one host synchronisation.
``gmmvb_small_fit`` (csrc/small.hip) runs restart r in workgroup r, convergence test included; this module is the host
side: the reference's random draws in the reference's order, one upload, one launch, one download, then the reference's
winner rule (``:873``) and progress lines replayed from the traces.

One deviation, as in ``RestartShard``: the reference keeps ``s_mats[k]`` of a component with ``ns[k] == 0`` from the
previous pass - across restarts too (``:729``).  Here the restarts run side by side, so for an EXACTLY empty component
``s_mats[k]`` (an attribute, not part of the posterior: it enters ``hn_w_mats_inv`` multiplied by ``ns[k] = 0``) is the
value from the same restart (zeros at its start).
"""
from __future__ import annotations

import ctypes
import os
import warnings

import numpy as np

from .._exceptions import ResultWarning

TERM_KEYS = ("p_x", "p_z", "p_pi", "p_mu_lambda", "q_z", "q_pi", "q_mu_lambda", "vl")
MAX_TRACE = 100_000        # iterations whose lower bounds the launch may have to keep


def applicable(model, n_rows, max_itr, num_init, init_type) -> bool:
    """The small-problem launch covers this call: one process, a shape inside the kernel's range, a known init_type."""
    if os.environ.get("BAYESML_AMD_SMALL", "1") == "0":
        return False
    if getattr(model._comm, "world", 1) != 1 or init_type not in ("subsampling", "random_responsibility"):
        return False
    if num_init < 1 or max_itr < 0 or max_itr > MAX_TRACE:
        return False
    if model._small_fit_impl is not None:
        return True
    if model._data_pass_factory is not None:            # a test injected its own data pass: the general driver is under test
        return False
    from .._engine import load_library
    return bool(load_library().gmmvb_small_supported(model.c_num_classes, model.c_degree, int(n_rows)))


def out_len(K, D, max_itr):
    return 2 + len(TERM_KEYS) + (max_itr + 1) + 7 * K + 2 * K * D + 3 * K * D * D


def _device_fit(model, xh, pivot, prior, init, n_restarts, init_type, max_itr, tolerance):
    """Upload, ``gmmvb_small_fit``, download.  Returns (out [R, out_len] host array, r [R, N, K] device tensor)."""
    import torch
    from .._engine import EngineUnavailableError, _check, _vp, load_library
    if not torch.cuda.is_available():
        raise EngineUnavailableError(
            f"bayesml_amd {type(model).__module__}.LearnModel needs an MI355X: the data pass has no CPU fallback")
    lib = load_library()
    K, D = model.c_num_classes, model.c_degree
    dev = torch.device("cuda", torch.cuda.current_device()) if model._device is None else torch.device(model._device)
    if isinstance(xh, torch.Tensor):
        xd = xh.to(dev).contiguous()
    else:
        xd = torch.from_numpy(xh).to(dev)
    n = xd.shape[0]
    small = torch.from_numpy(np.concatenate([pivot, prior, init.reshape(-1)])).to(dev)
    pv, pr, it = small[:D], small[D:D + prior.size], small[D + prior.size:]
    L = out_len(K, D, max_itr)
    out = torch.empty((n_restarts, L), dtype=torch.float64, device=dev)
    r = torch.empty((n_restarts, n, K), dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        st = _vp(torch.cuda.current_stream(dev).cuda_stream)
        _check(lib, lib.gmmvb_small_fit(K, D, 1 if xd.dtype == torch.float64 else 0, xd.data_ptr(), D, n, pv.data_ptr(),
                                        pr.data_ptr(), n_restarts, init_type, it.data_ptr(), int(max_itr),
                                        ctypes.c_double(float(tolerance)), out.data_ptr(), r.data_ptr(), st),
               "gmmvb_small_fit")
    return out.cpu().numpy(), r


def fit(model, x, max_itr, num_init, tolerance, init_type):
    """The whole of ``update_posterior`` for a small problem.  ``x``: validated ``[N, D]`` host array (f32 / f64) or
    device tensor."""
    import torch
    K, D = model.c_num_classes, model.c_degree
    if isinstance(x, torch.Tensor):
        if x.dtype not in (torch.float32, torch.float64):
            x = x.to(torch.float64)
        xh = x.detach().cpu().numpy()          # the sub-sample moments are formed on the host (a few hundred rows)
    else:
        xh = np.ascontiguousarray(x if x.dtype in (np.float32, np.float64) else x.astype(np.float64))
        x = xh
    n = xh.shape[0]
    x64 = xh.astype(np.float64, copy=False)
    pivot = x64[: min(n, 4096)].mean(axis=0)
    prior = np.concatenate([model.h0_alpha_vec, model.h0_m_vecs.reshape(-1), model.h0_kappas, model.h0_nus,
                            model.h0_w_mats_inv.reshape(-1), model._ln_b_h0_w_nus, [model._ln_c_h0_alpha]]).astype(np.float64)

    # ---- the restarts' random draws, in the reference's order (ref:786-796, 734-736)
    if init_type == "subsampling":
        size = int(np.sqrt(n))
        init = np.empty((num_init, K * D + K * D * D))
        eye = np.eye(D) * 1.0e-5
        for i in range(num_init):
            for k in range(K):
                draw = model.rng.choice(n, size=size, replace=False, shuffle=False)
                sub = x64[draw]
                m = sub.sum(axis=0) / size
                c = sub - m
                init[i, k * D:(k + 1) * D] = m
                init[i, K * D + k * D * D: K * D + (k + 1) * D * D] = (c.T @ c / size * model.h0_nus[k] + eye).reshape(-1)
        code = 0
    else:
        init = np.stack([model.rng.dirichlet(np.ones(K), n) for _ in range(num_init)])
        code = 1

    impl = model._small_fit_impl
    if impl is not None:
        out, r_all = impl(K, D, xh, pivot, prior, init, num_init, code, max_itr, tolerance)
    else:
        out, r_all = _device_fit(model, x, pivot, prior, init, num_init, code, max_itr, tolerance)

    # ---- the reference's winner rule and progress lines, replayed from the traces (ref:861-885)
    model._reset_small()
    # a workspace of an earlier, larger fit describes other rows: release it, so that the lazily fetched [N, K] attributes
    # (r_vecs from this launch, _ln_rho on demand from these rows) all describe THIS fit
    if model._engine is not None and model._data_pass_factory is None:
        model._engine.close()
    model._engine = model._x_dev = None
    # (a copy: the lazily made _ln_rho must describe the rows of THIS fit even if the caller changes its array afterwards;
    # at most 16384 x 8 values)
    model._small_x = np.array(x, copy=True) if isinstance(x, np.ndarray) else x.clone()
    model._small_ln_rho = None
    t0 = 2 + len(TERM_KEYS)
    best_vl, winner, never_converged = 0.0, 0, True
    for i in range(num_init):
        n_vl, conv = int(out[i, 0]), bool(out[i, 1])
        trace = out[i, t0:t0 + n_vl]
        model._say(f"\r{i}. VL: {trace[0]}")
        for t in range(n_vl - 1):
            model._say(f"\r{i}. VL: {trace[t + 1]} t={t} ")
        if conv:
            model._say("(converged)")
            never_converged = False
        vl = float(trace[-1])
        if i == 0 or vl > best_vl:
            model._say("*", end="\n")
            best_vl, winner = vl, i
        else:
            model._say("", end="\n")
    if never_converged:
        warnings.warn("Algorithm has not converged even once.", ResultWarning)

    o = out[winner, t0 + max_itr + 1:]
    cut = np.cumsum([0, K, K * D, K, K, K * D * D, K * D * D, K, K, K, K, K * D, K * D * D])
    alpha, m, kappa, nu, w_inv, w, elp, eld, lnb, ns, x_bar, s = (o[cut[j]:cut[j + 1]] for j in range(12))
    model.hn_alpha_vec[:] = alpha
    model.hn_m_vecs[:] = m.reshape(K, D)
    model.hn_kappas[:] = kappa
    model.hn_nus[:] = nu
    model.hn_w_mats[:] = w.reshape(K, D, D)
    model.hn_w_mats_inv[:] = w_inv.reshape(K, D, D)
    model._e_ln_pi_vec[:] = elp
    model._e_lambda_mats[:] = model.hn_nus[:, np.newaxis, np.newaxis] * model.hn_w_mats
    model._e_ln_lambda_dets[:] = eld
    model._ln_b_hn_w_nus[:] = lnb
    # the final pass of the reference (ref:895) repeats the winner's last data pass: its moments and responsibilities
    model.ns[:] = ns
    model.x_bar_vecs[:] = x_bar.reshape(K, D)
    model.s_mats[:] = s.reshape(K, D, D)
    last = out[num_init - 1, 2:2 + len(TERM_KEYS)]          # the reference leaves the LAST restart's terms (ref:882)
    for key, v in zip(TERM_KEYS, last):
        setattr(model, "vl" if key == "vl" else "_vl_" + key, float(v))
    model._small_r = r_all[winner]
    return model
