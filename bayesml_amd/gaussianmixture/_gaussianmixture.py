"""Gaussian mixture model with a Dirichlet x Normal-Wishart prior: ``GenModel`` / ``LearnModel``.

Drop-in for ``bayesml.gaussianmixture`` on the variational-Bayes posterior-update path
(reference: ``bayesml/gaussianmixture/_gaussianmixture.py``, cited below as ``ref:<lines>``).
Same constructor/method signatures, ``h0_*/hn_*/p_*`` dict keys and key order, exceptions,
stdout protocol and NumPy ``Generator`` consumption order.  What differs is *where* the work runs:

* every N-sized computation (E-step, sufficient statistics, the ``sum r ln r`` term of the lower
  bound, responsibilities / argmax read-out) is done by the gfx950 HIP kernels behind the C ABI of
  ``include/gmmvb.h``; there is no NumPy or PyTorch fallback for them;
* the K-sized closed-form updates run in fp64 torch on the same GPU (``bayesml_amd._kside``);
* host attributes (``hn_*`` ...) are float64 ndarrays refreshed when ``update_posterior`` returns.
"""
from __future__ import annotations

import warnings

import numpy as np
import torch

from .. import _check, _kside, base
from . import _small
from .._device import DeviceModel
from .._dist import SingleProcess
from .._exceptions import CriteriaError, DataFormatError, ParameterFormatError, ResultWarning

_PLOT_MSG = "if c_degree > 2, it is impossible to visualize the model by this function."


def _np(t: torch.Tensor) -> np.ndarray:
    return t.detach().to("cpu", torch.float64).numpy()


def _assign_niw(obj, prefix, D, alpha, m, kappa, nu, w, w_inv_attr=None):
    """Validated in-place assignment shared by the h_/h0_/hn_ setters (ref:526-557, 604-635, 111-141):
    ``arr[:] = val`` so scalars and single matrices broadcast and shapes stay fixed."""
    def name(s):
        return prefix + s
    if alpha is not None:
        _check.pos_floats(alpha, name("alpha_vec"), ParameterFormatError)
        getattr(obj, name("alpha_vec"))[:] = alpha
    if m is not None:
        _check.float_vecs(m, name("m_vecs"), ParameterFormatError)
        if m.shape[-1] != D:
            raise ParameterFormatError(f"{name('m_vecs')}.shape[-1] must coincide with self.c_degree: "
                                       f"{name('m_vecs')}.shape[-1] = {m.shape[-1]}, self.c_degree = {D}")
        getattr(obj, name("m_vecs"))[:] = m
    if kappa is not None:
        _check.pos_floats(kappa, name("kappas"), ParameterFormatError)
        getattr(obj, name("kappas"))[:] = kappa
    if nu is not None:
        _check.pos_floats(nu, name("nus"), ParameterFormatError)
        if np.any(nu <= D - 1):
            raise ParameterFormatError(f"All the values of {name('nus')} must be greater than self.c_degree - 1: "
                                       f"self.c_degree = {D}, {name('nus')} = {nu}")
        getattr(obj, name("nus"))[:] = nu
    if w is not None:
        _check.pos_def_sym_mats(w, name("w_mats"), ParameterFormatError)
        if w.shape[-1] != D:
            raise ParameterFormatError(
                f"{name('w_mats')}.shape[-1] and {name('w_mats')}.shape[-2] must coincide with self.c_degree: "
                f"{name('w_mats')}.shape[-1] and {name('w_mats')}.shape[-2] = {w.shape[-1]}, self.c_degree = {D}")
        getattr(obj, name("w_mats"))[:] = w
        if w_inv_attr is not None:
            getattr(obj, w_inv_attr)[:] = np.linalg.inv(getattr(obj, name("w_mats")))


def _niw_defaults(K, D):
    """Default hyper-parameters (ref:70-74, 438-442): alpha = 1/2, m = 0, kappa = 1, nu = D, W = I."""
    return (np.ones(K) / 2, np.zeros([K, D]), np.ones(K), np.ones(K) * D, np.tile(np.eye(D), [K, 1, 1]))


class GenModel(base.Generative):
    """Data-generating model and its prior (API shell of ref:18-367; plotting is out of scope).

    Parameters follow the reference: ``c_num_classes``, ``c_degree``, ``pi_vec``, ``mu_vecs``,
    ``lambda_mats``, ``h_alpha_vec``, ``h_m_vecs``, ``h_kappas``, ``h_nus``, ``h_w_mats``, ``seed``.
    """

    def __init__(self, c_num_classes, c_degree, pi_vec=None, mu_vecs=None, lambda_mats=None,
                 h_alpha_vec=None, h_m_vecs=None, h_kappas=None, h_nus=None, h_w_mats=None, seed=None):
        self.c_degree = _check.pos_int(c_degree, "c_degree", ParameterFormatError)
        self.c_num_classes = _check.pos_int(c_num_classes, "c_num_classes", ParameterFormatError)
        self.rng = np.random.default_rng(seed)
        K, D = self.c_num_classes, self.c_degree
        self.pi_vec = np.ones(K) / K
        self.mu_vecs = np.zeros([K, D])
        self.lambda_mats = np.tile(np.eye(D), [K, 1, 1])
        self.h_alpha_vec, self.h_m_vecs, self.h_kappas, self.h_nus, self.h_w_mats = _niw_defaults(K, D)
        self.set_params(pi_vec, mu_vecs, lambda_mats)
        self.set_h_params(h_alpha_vec, h_m_vecs, h_kappas, h_nus, h_w_mats)

    def get_constants(self):
        return {"c_num_classes": self.c_num_classes, "c_degree": self.c_degree}

    def set_h_params(self, h_alpha_vec=None, h_m_vecs=None, h_kappas=None, h_nus=None, h_w_mats=None):
        _assign_niw(self, "h_", self.c_degree, h_alpha_vec, h_m_vecs, h_kappas, h_nus, h_w_mats)
        return self

    def get_h_params(self):
        return {"h_alpha_vec": self.h_alpha_vec, "h_m_vecs": self.h_m_vecs, "h_kappas": self.h_kappas,
                "h_nus": self.h_nus, "h_w_mats": self.h_w_mats}

    def gen_params(self):
        """Draw (pi, Lambda_k, mu_k) from the prior with the reference's call order (ref:176-185)."""
        from scipy.stats import wishart
        self.pi_vec[:] = self.rng.dirichlet(self.h_alpha_vec)
        for k in range(self.c_num_classes):
            self.lambda_mats[k] = wishart.rvs(df=self.h_nus[k], scale=self.h_w_mats[k], random_state=self.rng)
            self.mu_vecs[k] = self.rng.multivariate_normal(
                mean=self.h_m_vecs[k], cov=np.linalg.inv(self.h_kappas[k] * self.lambda_mats[k]))
        return self

    def set_params(self, pi_vec=None, mu_vecs=None, lambda_mats=None):
        K, D = self.c_num_classes, self.c_degree
        if pi_vec is not None:
            _check.float_vec_sum_1(pi_vec, "pi_vec", ParameterFormatError)
            if pi_vec.shape[0] != K:
                raise ParameterFormatError("pi_vec.shape[0] must coincide with self.c_num_classes: "
                                           f"pi_vec.shape[0] = {pi_vec.shape[0]}, self.c_num_classes = {K}")
            self.pi_vec[:] = pi_vec
        if mu_vecs is not None:
            _check.float_vecs(mu_vecs, "mu_vecs", ParameterFormatError)
            if mu_vecs.shape[-1] != D:
                raise ParameterFormatError("mu_vecs.shape[-1] must coincide with self.c_degree: "
                                           f"mu_vecs.shape[-1] = {mu_vecs.shape[-1]}, self.c_degree = {D}")
            self.mu_vecs[:] = mu_vecs
        if lambda_mats is not None:
            _check.pos_def_sym_mats(lambda_mats, "lambda_mats", ParameterFormatError)
            if lambda_mats.shape[-1] != D:
                raise ParameterFormatError(
                    "lambda_mats.shape[-1] and lambda_mats.shape[-2] must coincide with self.c_degree:"
                    f"lambda_mats.shape[-1] and lambda_mats.shape[-2] = {lambda_mats.shape[-1]}, self.c_degree = {D}")
            self.lambda_mats[:] = lambda_mats
        return self

    def get_params(self):
        return {"pi_vec": self.pi_vec, "mu_vecs": self.mu_vecs, "lambda_mats": self.lambda_mats}

    def gen_sample(self, sample_size, *, device=None, dtype=torch.float64):
        """(x [n, D], one-hot z [n, K]); one ``choice`` + one ``multivariate_normal`` per row so the
        stream matches the reference for a given seed (ref:241-264).

        Extension: with ``device`` (e.g. ``"cuda"``) the sample is drawn ON that device by HIP kernels and returned as
        torch tensors ``(x [n, D] of ``dtype``, z [n] int64 class indices)`` - the reference's per-row Python loop takes
        minutes per million rows.  The stream is Philox4x64-10 keyed by a seed drawn from ``self.rng`` (reproducible for
        a given ``seed``, and on the host with ``numpy.random.Philox``), not the reference's."""
        _check.pos_int(sample_size, "sample_size", DataFormatError)
        if device is not None:
            return self._gen_sample_device(int(sample_size), torch.device(device), dtype)
        z = np.zeros([sample_size, self.c_num_classes], dtype=int)
        x = np.empty([sample_size, self.c_degree])
        cov = np.linalg.inv(self.lambda_mats)
        for n in range(sample_size):
            k = self.rng.choice(self.c_num_classes, p=self.pi_vec)
            z[n, k] = 1
            x[n] = self.rng.multivariate_normal(mean=self.mu_vecs[k], cov=cov[k])
        return x, z

    def _gen_sample_device(self, n, dev, dtype):
        """z ~ Categorical(pi_vec), x = mu_z + eps L_z^-1 with Lambda_z = L_z L_z^T: HIP kernels over the Philox stream
        of a seed drawn from ``self.rng`` (``bayesml_amd._sample``, ``gmmvb_sample_latent`` / ``gmmvb_sample_emissions``);
        the seed of the last device sample stays readable as ``self.device_sample_seed``."""
        from .. import _sample
        self.device_sample_seed = int(self.rng.integers(0, 2 ** 63 - 1))
        return _sample.mixture(self.pi_vec, self.mu_vecs, self.lambda_mats, n, self.device_sample_seed, dev, dtype)

    def save_sample(self, filename, sample_size):
        """``numpy.savez_compressed(filename, x=x, z=z)`` (ref:266-284)."""
        x, z = self.gen_sample(sample_size)
        np.savez_compressed(filename, x=x, z=z)

    def visualize_model(self, sample_size=100):
        """Prints the parameters like the reference; drawing is outside this package's scope."""
        if self.c_degree > 2:
            raise ParameterFormatError(_PLOT_MSG)
        print(f"pi_vec:\n {self.pi_vec}")
        print(f"mu_vecs:\n {self.mu_vecs}")
        print(f"lambda_mats:\n {self.lambda_mats}")
        raise NotImplementedError("plotting is out of scope for bayesml_amd (SURVEY.md section 2, row 2)")


class LearnModel(DeviceModel, base.Posterior, base.PredictiveMixin):
    """Variational posterior and predictive distribution of the Gaussian mixture.

    Positional parameters are the reference's (ref:421-431): ``c_num_classes, c_degree,
    h0_alpha_vec=None, h0_m_vecs=None, h0_kappas=None, h0_nus=None, h0_w_mats=None, seed=None``.

    Extensions (keyword-only, defaulting to the reference's behaviour):
      device   torch device of the MI355X to run on (default: the current one)
      comm     a ``bayesml_amd.RowShard``: ``x`` given to ``update_posterior`` is then this rank's
               block of rows of the global sample matrix (one process per GPU)
      verbose  False silences the per-iteration progress line

    A sample matrix whose per-pair workspace does not fit the GPU (K N of the order of 2e10) is processed in row tiles
    through one workspace (``_engine.TiledDataPass``); the reference keeps its [N, K] arrays on the host instead.
    """

    _row_tiling = True

    def __init__(self, c_num_classes, c_degree, h0_alpha_vec=None, h0_m_vecs=None, h0_kappas=None,
                 h0_nus=None, h0_w_mats=None, seed=None, *, device=None, comm=None, verbose=True):
        self.c_degree = _check.pos_int(c_degree, "c_degree", ParameterFormatError)
        self.c_num_classes = _check.pos_int(c_num_classes, "c_num_classes", ParameterFormatError)
        from .._engine import check_limits
        check_limits(self.c_degree, self.c_num_classes)
        self.rng = np.random.default_rng(seed)
        self._device = device
        self._comm = comm if comm is not None else SingleProcess()
        self._verbose = verbose
        self._data_pass_factory = None      # test seam only; the default is the HIP engine (no fallback)
        K, D = self.c_num_classes, self.c_degree

        self.h0_alpha_vec, self.h0_m_vecs, self.h0_kappas, self.h0_nus, self.h0_w_mats = _niw_defaults(K, D)
        self.h0_w_mats_inv = np.linalg.inv(self.h0_w_mats)
        self._ln_c_h0_alpha = 0.0
        self._ln_b_h0_w_nus = np.empty(K)

        self.hn_alpha_vec = np.empty([K])
        self.hn_m_vecs = np.empty([K, D])
        self.hn_kappas = np.empty([K])
        self.hn_nus = np.empty([K])
        self.hn_w_mats = np.empty([K, D, D])
        self.hn_w_mats_inv = np.empty([K, D, D])
        self._e_lambda_mats = np.empty([K, D, D])
        self._e_ln_lambda_dets = np.empty(K)
        self._ln_b_hn_w_nus = np.empty(K)
        self._e_ln_pi_vec = np.empty(K)

        # statistics of the last data pass (the reference leaves s_mats uninitialised; zeros here)
        self.x_bar_vecs = np.zeros([K, D])
        self.ns = np.zeros(K)
        self.s_mats = np.zeros([K, D, D])
        self._engine = None
        self._x_dev = None
        self._r_cache = None
        self._small_r = None                # device responsibilities of a small-problem fit (_small.py), fetched lazily
        self._small_x = None                # that fit's rows (<= 16384 x 8, a copy): _ln_rho re-runs the final E-step on them when asked
        self._small_ln_rho = None           # ... once
        self._small_fit_impl = None         # test seam of the small-problem launch (tests/fake_engine.py); None = gmmvb_small_fit

        self.vl = 0.0
        self._vl_p_x = self._vl_p_z = self._vl_p_pi = self._vl_p_mu_lambda = 0.0
        self._vl_q_z = self._vl_q_pi = self._vl_q_mu_lambda = 0.0

        self.p_pi_vec = np.empty([K])
        self.p_mu_vecs = np.empty([K, D])
        self.p_nus = np.empty([K])
        self.p_lambda_mats = np.empty([K, D, D])

        self.set_h0_params(h0_alpha_vec, h0_m_vecs, h0_kappas, h0_nus, h0_w_mats)

    # ------------------------------------------------------------------ parameter plumbing
    def get_constants(self):
        return {"c_num_classes": self.c_num_classes, "c_degree": self.c_degree}

    def set_h0_params(self, h0_alpha_vec=None, h0_m_vecs=None, h0_kappas=None, h0_nus=None, h0_w_mats=None):
        """Validated prior assignment, then prior features and ``reset_hn_params`` (ref:503-561)."""
        _assign_niw(self, "h0_", self.c_degree, h0_alpha_vec, h0_m_vecs, h0_kappas, h0_nus, h0_w_mats,
                    w_inv_attr="h0_w_mats_inv")
        p = self._prior_tensors("cpu")
        self._ln_c_h0_alpha = p.ln_c_alpha
        self._ln_b_h0_w_nus[:] = _np(p.ln_b_w_nu)
        self.reset_hn_params()
        return self

    def get_h0_params(self):
        return {"h0_alpha_vec": self.h0_alpha_vec, "h0_m_vecs": self.h0_m_vecs, "h0_kappas": self.h0_kappas,
                "h0_nus": self.h0_nus, "h0_w_mats": self.h0_w_mats}

    def set_hn_params(self, hn_alpha_vec=None, hn_m_vecs=None, hn_kappas=None, hn_nus=None, hn_w_mats=None):
        """Validated posterior assignment, derived features, predictive parameters (ref:581-641)."""
        _assign_niw(self, "hn_", self.c_degree, hn_alpha_vec, hn_m_vecs, hn_kappas, hn_nus, hn_w_mats,
                    w_inv_attr="hn_w_mats_inv")
        self._refresh_host_features()
        self.calc_pred_dist()
        return self

    def get_hn_params(self):
        return {"hn_alpha_vec": self.hn_alpha_vec, "hn_m_vecs": self.hn_m_vecs, "hn_kappas": self.hn_kappas,
                "hn_nus": self.hn_nus, "hn_w_mats": self.hn_w_mats}

    def _prior_tensors(self, device) -> _kside.PriorT:
        return _kside.prior_from_numpy(self.h0_alpha_vec, self.h0_m_vecs, self.h0_kappas, self.h0_nus,
                                       self.h0_w_mats, device)

    def _post_tensors(self, device) -> _kside.PostT:
        t = lambda a: torch.as_tensor(a, dtype=torch.float64, device=device).clone()   # noqa: E731
        return _kside.features(_kside.PostT(t(self.hn_alpha_vec), t(self.hn_m_vecs), t(self.hn_kappas),
                                            t(self.hn_nus), t(self.hn_w_mats_inv)))

    def _refresh_host_features(self):
        """_calc_q_pi_features + _calc_q_lambda_features on the host arrays (ref:738-739, 745-756)."""
        q = self._post_tensors("cpu")
        self._e_ln_pi_vec[:] = _np(q.e_ln_pi)
        self._e_lambda_mats[:] = self.hn_nus[:, np.newaxis, np.newaxis] * self.hn_w_mats
        self._e_ln_lambda_dets[:] = _np(q.e_ln_lambda_det)
        self._ln_b_hn_w_nus[:] = _np(q.ln_b_w_nu)

    def _reset_hn_from(self, q0: _kside.PostT):
        """reset_hn_params() for the restarts of update_posterior (ref:846): same host state, but the already
        validated h0_* arrays and their cached inverse are copied and the derived features come from the device
        (the validated setter re-inverts and re-factorises K D x D matrices on the host: 0.9 s per restart at
        K = 64, D = 128 on the GPU box's CPU, more than twenty sparse VB iterations)."""
        self.hn_alpha_vec[:] = self.h0_alpha_vec
        self.hn_m_vecs[:] = self.h0_m_vecs
        self.hn_kappas[:] = self.h0_kappas
        self.hn_nus[:] = self.h0_nus
        self.hn_w_mats[:] = self.h0_w_mats
        self.hn_w_mats_inv[:] = self.h0_w_mats_inv
        self._e_ln_pi_vec[:] = _np(q0.e_ln_pi)
        self._e_lambda_mats[:] = self.hn_nus[:, np.newaxis, np.newaxis] * self.hn_w_mats
        self._e_ln_lambda_dets[:] = _np(q0.e_ln_lambda_det)
        self._ln_b_hn_w_nus[:] = _np(q0.ln_b_w_nu)
        self.calc_pred_dist()

    def _store_posterior(self, q: _kside.PostT):
        """Device posterior -> host ``hn_*`` arrays and derived features."""
        self.hn_alpha_vec[:] = _np(q.alpha)
        self.hn_m_vecs[:] = _np(q.m)
        self.hn_kappas[:] = _np(q.kappa)
        self.hn_nus[:] = _np(q.nu)
        self.hn_w_mats[:] = _np(q.w)
        self.hn_w_mats_inv[:] = _np(q.w_inv)
        self._e_ln_pi_vec[:] = _np(q.e_ln_pi)
        self._e_lambda_mats[:] = self.hn_nus[:, np.newaxis, np.newaxis] * self.hn_w_mats
        self._e_ln_lambda_dets[:] = _np(q.e_ln_lambda_det)
        self._ln_b_hn_w_nus[:] = _np(q.ln_b_w_nu)

    # ------------------------------------------------------------------ device plumbing (see _device.py)
    def _stepper(self, eng, prior, xd) -> _kside.KStepper:
        """The K-side of one VB iteration as one unit (a replayed hipGraph on the GPU, see _kside.KStepper)."""
        want = bool(hasattr(eng, "wants_drift") and eng.wants_drift(xd.shape[0]))
        return _kside.KStepper(prior, eng.pivot, eng.stats_len, want)

    def _data_pass(self, eng, xd, ks, estep=True):
        """One data pass under the parameters the engine holds: statistics block into ``ks.stats``, summed over the
        row shards (the one collective per VB iteration)."""
        if estep:
            eng.estep_mstep(xd, out=ks.stats)
        else:
            eng.mstep(xd, out=ks.stats)
        comm = self._comm
        sharded = (getattr(comm, "world", 1) > 1 or getattr(comm, "always", False)) and not getattr(comm, "restart_parallel", False)
        if not sharded:
            return
        # row shards: ONE collective per iteration, of the block on the wire - [ns | h | a | upper triangles of B] (B is
        # symmetric: half of the full block's bytes) - with the engine's pass-policy counters riding behind it, so that every
        # rank's next E-step decides from the same job-wide numbers
        buf, _packed, tail = ks.wire()
        policy = estep and getattr(comm, "world", 1) > 1 and hasattr(eng, "policy_export")
        if policy:
            eng.policy_export(tail)
        else:
            tail.zero_()
        ks.pack()
        comm.all_reduce_(buf)
        ks.unpack()
        if policy:
            eng.policy_import(tail)

    def _give_params(self, eng, q, hint=None):
        """Hand a posterior's E-step parameters to the engine; ``hint`` = (gamma, delta, big_gamma, mean gamma) of the update
        that led to it from the parameters of the engine's last E-step (gmmvb_set_drift), or None."""
        if hint is not None:
            eng.set_drift(*hint)
        eng.set_params(q.c, q.m, q.u)

    def _pass(self, eng, xd, q, s_prev, estep=True, hint=None):
        """One stand-alone data pass (read-outs, tests): statistics -> all-reduce -> reference moments."""
        if estep:
            self._give_params(eng, q, hint)
            stats = eng.estep_mstep(xd)
        else:
            stats = eng.mstep(xd)
        self._comm.all_reduce_(stats)
        ns, h, a, B = eng.split_stats(stats)
        x_bar, s = _kside.moments_from_stats(ns, a, B, eng.pivot, s_prev)
        return ns, x_bar, s, h.sum()

    @staticmethod
    def _drift_hint(eng, xd, q_from, q):
        """(gamma, delta, big_gamma) of gmmvb_set_drift for the update q_from -> q, or None when the engine cannot use it."""
        if q_from is None or q is None or not hasattr(eng, "wants_drift") or not eng.wants_drift(xd.shape[0]):
            return None
        return _kside.drift(q_from, q)

    # ------------------------------------------------------------------ the hot path
    def update_posterior(self, x, max_itr=100, num_init=10, tolerance=1.0E-8, init_type="subsampling"):
        """Variational-Bayes update of ``hn_*`` from data (driver of ref:802-896).

        ``x``: ``(sample_size, c_degree)`` real ndarray (float32 stays float32 in HBM and is widened
        on load; integers are cast to float64), or a torch tensor already on the GPU.
        ``init_type``: ``'subsampling'`` or ``'random_responsibility'``.

        Per VB iteration the host does: one parameter hand-over, one data-pass call, (one all-reduce), one K-side
        step (a hipGraph replay) and ONE device-to-host copy of nine doubles (the lower bound's terms and the mean
        drift)."""
        x = self._check_rows(x)
        if _small.applicable(self, x.shape[0], max_itr, num_init, init_type):
            # every restart and iteration in one launch (csrc/small.hip); same draws, same winner rule, same progress lines
            return _small.fit(self, x, max_itr, num_init, tolerance, init_type)
        self._reset_small()
        eng, xd = self._open(x)
        K, D = self.c_num_classes, self.c_degree
        dev = xd.device
        n_global = self._comm.global_rows
        prior = self._prior_tensors(dev)
        ks = self._stepper(eng, prior, xd)
        ks.s_prev.copy_(torch.as_tensor(self.s_mats, dtype=torch.float64, device=dev))

        best_host = {k: np.array(v) for k, v in self.get_hn_params().items()}
        best_host["hn_w_mats_inv"] = np.array(self.hn_w_mats_inv)
        best_q, best_vl, best_i = None, 0.0, -1
        last_estep_under_q = False
        never_converged = True
        terms = None
        # restart-level parallelism (comm = RestartShard): restart i runs on rank i mod world, its progress line is
        # kept and printed in order afterwards
        par = getattr(self._comm, "restart_parallel", False)
        mine = {}                                # restart -> (vl, converged, progress text, posterior, terms)
        for i in range(num_init):
            skip = par and self._comm.owner(i) != self._comm.rank
            text = []
            say = (lambda t, end="": text.append(t + end)) if par else self._say
            q = _kside.post_from_prior(prior)
            if not skip:
                self._reset_hn_from(q)
            if hasattr(eng, "forget"):
                eng.forget()            # a restart's parameters are unrelated to the last pass: expect dense responsibilities
            if init_type == "subsampling":
                if skip:
                    self._subsample_moments(eng, xd, n_global, draw_only=True)
                    continue
                q = self._init_subsampling(eng, xd, q, n_global)
                ks.load(q)
                self._give_params(eng, q)
                self._data_pass(eng, xd, ks)
            elif init_type == "random_responsibility":
                r = self.rng.dirichlet(np.ones(K), n_global)
                if skip:
                    continue
                lo = self._comm.row_offset
                eng.load_responsibilities(torch.from_numpy(r[lo: lo + xd.shape[0]]).to(dev))
                ks.load(q)
                self._data_pass(eng, xd, ks, estep=False)
            else:
                raise ValueError(f"init_type={init_type} is unsupported. This function supports only "
                                 '"subsampling" and "random_responsibility"')
            # lower bound under the current posterior, the next posterior and its drift hint: one K-side step
            ks.step()
            terms, gmean = ks.read()
            vl = terms["vl"]
            carried = init_type == "subsampling"      # the engine's last E-step belongs to ks.q
            converged = False
            say(f"\r{i}. VL: {vl}")
            for t in range(max_itr):
                vl_before = vl
                self._give_params(eng, ks.q_next, ks.hint(gmean) if carried else None)
                ks.advance()
                carried = True
                self._data_pass(eng, xd, ks)
                ks.step()
                terms, gmean = ks.read()                     # the one host sync per iteration
                vl = terms["vl"]
                say(f"\r{i}. VL: {vl} t={t} ")
                with np.errstate(divide="ignore", invalid="ignore"):
                    if np.abs((vl - vl_before) / vl_before) < tolerance:
                        converged = True
                        say("(converged)")
                        break
            never_converged = never_converged and not converged
            last_estep_under_q = carried          # the workspace's E-step output belongs to this restart's final posterior
            if par:
                mine[i] = (vl, converged, "".join(text), ks.current(), terms)
                continue
            if i == 0 or vl > best_vl:
                self._say("*", end="\n")
                best_vl, best_q, best_i = vl, ks.current(), i
            else:
                self._say("", end="\n")
            self.vl = vl
        if par and num_init > 0:
            best_q, terms, never_converged = self._merge_restarts(mine, num_init, dev)
        if never_converged:
            warnings.warn("Algorithm has not converged even once.", ResultWarning)

        if best_q is not None:
            self._store_posterior(best_q)
            q = best_q
        else:                                   # num_init == 0: keep what the model held on entry
            for k, v in best_host.items():
                getattr(self, k)[:] = v
            self._refresh_host_features()
            q = self._post_tensors(dev)
        if terms is not None:
            for k, v in terms.items():
                setattr(self, "vl" if k == "vl" else "_vl_" + k, float(v))
        # ref:895 - final E+M pass so that r_vecs / ns / x_bar_vecs / s_mats match the kept posterior.  When the kept
        # posterior is the one the LAST data pass ran under (the last restart won) that pass is the final pass: its
        # moments are in the stepper and its E-step output in the workspace - nothing to repeat (at the benchmark shape a
        # repeated pass without a drift hint is a 45-ms bound pass).
        if best_q is not None and not par and best_i == num_init - 1 and last_estep_under_q:
            ns, x_bar, s = ks.ns, ks.x_bar, ks.s
        else:
            ns, x_bar, s, _h = self._pass(eng, xd, q, ks.s_prev)
        self.ns[:], self.x_bar_vecs[:], self.s_mats[:] = _np(ns), _np(x_bar), _np(s)
        return self

    def _merge_restarts(self, mine, num_init, dev):
        """RestartShard: gather every restart's (lower bound, converged, progress line), replay the reference's winner
        rule (ref:873: ``i == 0 or vl > best``) in restart order, print the lines, broadcast the winner's posterior."""
        comm = self._comm
        table = {}
        for part in comm.gather({i: (v[0], v[1], v[2]) for i, v in mine.items()}):
            table.update(part)
        winner, best_vl = 0, 0.0
        for i in range(num_init):
            vl, _conv, text = table[i]
            star = i == 0 or vl > best_vl
            if star:
                winner, best_vl = i, vl
            self._say(text + ("*" if star else ""), end="\n")
        self.vl = table[num_init - 1][0]                       # the reference leaves the LAST restart's bound (ref:882)
        owner = comm.owner(winner)
        # (ranks that do not own the winner receive into a posterior of the right shapes)
        q = mine[winner][3] if owner == comm.rank else _kside._clone_post(_kside.post_from_prior(self._prior_tensors(dev)))
        comm.broadcast_([getattr(q, f) for f in _kside._POST_FIELDS], owner)
        last = comm.gather(mine[num_init - 1][4] if (num_init - 1) in mine else None)
        terms = next(t for t in last if t is not None)
        return q, terms, not any(table[i][1] for i in range(num_init))

    def _init_subsampling(self, eng, xd, q, n_global):
        """ref:786-796 on the GPU (index draw on the host, see _device.DeviceModel._subsample_moments)."""
        size, a, B = self._subsample_moments(eng, xd, n_global)
        return _kside.subsample_moments_init(q, size, a, B, eng.pivot, _kside.features)

    # lazily fetched [N, K] arrays of the last data pass (the reference keeps them as attributes)
    def _reset_small(self):
        self._r_cache = None
        self._small_r = None
        self._small_x = None
        self._small_ln_rho = None

    @property
    def r_vecs(self):
        if self._small_r is not None:
            if self._r_cache is None:
                self._r_cache = self._small_r if isinstance(self._small_r, np.ndarray) else _np(self._small_r)
            return self._r_cache
        if self._engine is None:
            return None
        if self._r_cache is None:
            self._r_cache = _np(self._engine.responsibilities())
        return self._r_cache

    @property
    def _ln_rho(self):
        if self._small_r is not None:
            # a small-problem fit keeps no [N, K] ln rho (one launch, csrc/small.hip) and no engine: the reference's final pass
            # (ref:895) is an E-step under the final posterior - run exactly that on the fit's rows (general engine), leaving the
            # fit's own r_vecs in place
            if self._small_x is None:
                return None
            if getattr(self, "_small_ln_rho", None) is not None:      # (made once per fit, like _r_cache)
                return self._small_ln_rho
            keep = (self._small_r, self._small_x, self._r_cache)
            eng, xd = self._open(self._small_x)
            self._give_params(eng, self._post_tensors(xd.device))
            eng.estep(xd)
            out = _np(eng.ln_rho())
            eng.close()
            self._engine = self._x_dev = None
            self._small_r, self._small_x, self._r_cache = keep
            self._small_ln_rho = out
            return out
        return None if self._engine is None else _np(self._engine.ln_rho())

    # ------------------------------------------------------------------ read-outs
    def estimate_params(self, loss="squared"):
        """Point estimates of (pi, mu_k, Lambda_k) under ``loss`` in {"squared", "0-1", "KL"} (ref:898-961).
        "0-1" follows the reference's code, which divides by ``sum(alpha) - c_degree`` (ref:935)."""
        K, D = self.c_num_classes, self.c_degree
        if loss == "squared":
            return self.hn_alpha_vec / self.hn_alpha_vec.sum(), self.hn_m_vecs, self._e_lambda_mats
        if loss == "0-1":
            pi_hat = np.empty(K)
            if np.all(self.hn_alpha_vec > 1):
                pi_hat[:] = (self.hn_alpha_vec - 1) / (np.sum(self.hn_alpha_vec) - D)
            else:
                warnings.warn("MAP estimate of pi_vec doesn't exist for the current hn_alpha_vec.", ResultWarning)
                pi_hat[:] = np.nan
            lam = np.empty([K, D, D])
            for k in range(K):
                if self.hn_nus[k] >= D + 1:
                    lam[k] = (self.hn_nus[k] - D - 1) * self.hn_w_mats[k]
                else:
                    warnings.warn(f"MAP estimate of lambda_mat doesn't exist for the current hn_nus[{k}].",
                                  ResultWarning)
                    lam[k] = np.nan
            return pi_hat, self.hn_m_vecs, lam
        if loss == "KL":
            from scipy.stats import dirichlet, multivariate_t, wishart
            dof = self.hn_nus - D + 1
            mus = [multivariate_t(loc=self.hn_m_vecs[k], shape=self.hn_w_mats_inv[k] / self.hn_kappas[k] / dof[k],
                                  df=dof[k]) for k in range(K)]
            lams = [wishart(df=self.hn_nus[k], scale=self.hn_w_mats[k]) for k in range(K)]
            return dirichlet(self.hn_alpha_vec), mus, lams
        raise CriteriaError(f"loss={loss} is unsupported. "
                            "This function supports \"squared\", \"0-1\", and \"KL\".")

    def visualize_posterior(self):
        """Prints the posterior like the reference (ref:994-1007); drawing is out of scope."""
        if self.c_degree > 2:
            raise ParameterFormatError(_PLOT_MSG)
        for title, val in (("hn_alpha_vec:", self.hn_alpha_vec),
                           ("E[pi_vec]:", self.hn_alpha_vec / self.hn_alpha_vec.sum()),
                           ("hn_m_vecs:", self.hn_m_vecs), ("hn_kappas:", self.hn_kappas),
                           ("hn_nus:", self.hn_nus), ("hn_w_mats:", self.hn_w_mats),
                           ("E[lambda_mats]=", self._e_lambda_mats)):
            print(title)
            print(f"{val}")
        raise NotImplementedError("plotting is out of scope for bayesml_amd (SURVEY.md section 2, row 2)")

    def get_p_params(self):
        return {"p_mu_vecs": self.p_mu_vecs, "p_nus": self.p_nus, "p_lambda_mats": self.p_lambda_mats}

    def calc_pred_dist(self):
        """Student-t mixture parameters of the predictive distribution (ref:1064-1070).  As in the
        reference, ``update_posterior`` does not call this; call it before ``make_prediction``."""
        self.p_pi_vec[:] = self.hn_alpha_vec / self.hn_alpha_vec.sum()
        self.p_mu_vecs[:] = self.hn_m_vecs
        self.p_nus[:] = self.hn_nus - self.c_degree + 1
        self.p_lambda_mats[:] = (self.hn_kappas * self.p_nus / (self.hn_kappas + 1))[:, np.newaxis, np.newaxis] * self.hn_w_mats
        return self

    def make_prediction(self, loss="squared"):
        """Predicted next point: mixture mean ("squared") or the highest weighted mode ("0-1") (ref:1072-1102)."""
        if loss == "squared":
            return np.sum(self.p_pi_vec[:, np.newaxis] * self.p_mu_vecs, axis=0)
        if loss == "0-1":
            from scipy.stats import multivariate_t
            best, arg = -1.0, np.empty([self.c_degree])
            for k in range(self.c_num_classes):
                dens = multivariate_t.pdf(x=self.p_mu_vecs[k], loc=self.p_mu_vecs[k],
                                          shape=np.linalg.inv(self.p_lambda_mats[k]), df=self.p_nus[k])
                if dens * self.p_pi_vec[k] > best:
                    arg[:] = self.p_mu_vecs[k]
                    best = dens * self.p_pi_vec[k]
            return arg
        raise CriteriaError(f"loss={loss} is unsupported. "
                            "This function supports \"squared\" and \"0-1\".")

    def pred_and_update(self, x, loss="squared", max_itr=100, num_init=10, tolerance=1.0E-8,
                        init_type="random_responsibility"):
        """Predict one point, then fold it into the posterior with h0 <- hn (ref:1104-1155)."""
        _check.float_vec(x, "x", DataFormatError)
        if x.shape != (self.c_degree,):
            raise DataFormatError(f"x must be a 1-dimensional float array whose size is c_degree: {self.c_degree}.")
        self.calc_pred_dist()
        prediction = self.make_prediction(loss=loss)
        self.overwrite_h0_params()
        self.update_posterior(x[np.newaxis, :], max_itr=max_itr, num_init=num_init, tolerance=tolerance,
                              init_type=init_type)
        return prediction

    def estimate_latent_vars(self, x, loss="0-1"):
        """Responsibilities ("squared"/"KL") or one-hot MAP assignments ("0-1") of ``x`` under the
        current posterior (ref:1157-1196).  Like the reference's ``_update_q_z`` it also refreshes
        ``ns`` / ``x_bar_vecs`` / ``s_mats``."""
        eng, xd = self._open(x)
        q = self._post_tensors(xd.device)
        s_prev = torch.as_tensor(self.s_mats, dtype=torch.float64, device=xd.device)
        ns, x_bar, s, _h = self._pass(eng, xd, q, s_prev)
        self.ns[:], self.x_bar_vecs[:], self.s_mats[:] = _np(ns), _np(x_bar), _np(s)
        if loss in ("squared", "KL"):
            return self.r_vecs
        if loss == "0-1":
            z = eng.argmax().to("cpu").numpy()
            return np.eye(self.c_num_classes, dtype=int)[z]
        raise CriteriaError(f"loss={loss} is unsupported. "
                            "This function supports \"squared\", \"0-1\", and \"KL\".")

    def estimate_latent_vars_and_update(self, x, loss="0-1", max_itr=100, num_init=10, tolerance=1.0E-8,
                                        init_type="subsampling"):
        """``estimate_latent_vars`` followed by a sequential update with h0 <- hn (ref:1198-1245)."""
        z_hat = self.estimate_latent_vars(x, loss=loss)
        self.overwrite_h0_params()
        self.update_posterior(x, max_itr=max_itr, num_init=num_init, tolerance=tolerance, init_type=init_type)
        return z_hat
