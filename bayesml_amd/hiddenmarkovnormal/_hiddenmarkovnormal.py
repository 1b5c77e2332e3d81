"""Hidden Markov model with Gaussian emissions: ``GenModel`` / ``LearnModel``.

Drop-in for ``bayesml.hiddenmarkovnormal`` on the variational-Bayes posterior-update path (reference:
``bayesml/hiddenmarkovnormal/_hiddenmarkovnormal.py``, cited as ``ref:<lines>``).  Constructor
arguments after ``c_degree`` are keyword-only like the reference's; dict keys, exceptions, stdout protocol
and ``Generator`` consumption order are kept.  The N-sized work runs on the GPU behind the C ABI:

* emission term ``ln rho`` (ref:988-996): the GMM E-step kernel without ``E[ln pi]``;
* scaled forward-backward, ``gamma``, ``sum_t xi_t`` (ref:999-1018, 839): chunk-parallel f64 MFMA kernels
  (``csrc/hmm.h``); ``xi_mats`` ([T, K, K], 82 GB at T=1e7, K=32) is never formed, only its sum ``ms``;
* NIW statistics (ref:837-845): the GMM M-step kernel with ``gamma`` as the responsibilities.

``alpha_vecs``, ``beta_vecs``, ``gamma_vecs`` and ``xi_mats`` (``xi_rows(row0, n)`` for a row range) are formed on the GPU and
fetched on access; nothing T-sized is kept on the host.
"""
from __future__ import annotations

import warnings

import numpy as np
import torch

from .. import _check, _kside, base
from .._device import DeviceModel
from .._dist import SingleProcess
from .._exceptions import CriteriaError, DataFormatError, ParameterFormatError, ResultWarning

_PLOT_MSG = "if c_degree > 2, it is impossible to visualize the model by this function."


def _np(t):
    return t.detach().to("cpu", torch.float64).numpy()


def _assign_hmm(obj, pre, K, D, eta, zeta, m, kappa, nu, w):
    """Validated in-place assignment shared by the h_/h0_/hn_ setters (ref:151-186, 640-684, 742-786).
    Note the reference's ``nus`` test here is ``np.all(nus <= D - 1)`` (ref:170, 666, 768), weaker than the GMM's."""
    def nm(s):
        return pre + s
    if eta is not None:
        _check.pos_floats(eta, nm("eta_vec"), ParameterFormatError)
        getattr(obj, nm("eta_vec"))[:] = eta
    if zeta is not None:
        _check.pos_floats(zeta, nm("zeta_vecs"), ParameterFormatError)
        getattr(obj, nm("zeta_vecs"))[:] = zeta
    if m is not None:
        _check.float_vecs(m, nm("m_vecs"), ParameterFormatError)
        _check.shape_consistency(m.shape[-1], nm("m_vecs") + ".shape[-1]", D, "self.c_degree", ParameterFormatError)
        getattr(obj, nm("m_vecs"))[:] = m
    if kappa is not None:
        _check.pos_floats(kappa, nm("kappas"), ParameterFormatError)
        getattr(obj, nm("kappas"))[:] = kappa
    if nu is not None:
        _check.floats(nu, nm("nus"), ParameterFormatError)
        if np.all(nu <= D - 1):
            raise ParameterFormatError(f"All the values in {nm('nus')} must be greater than self.c_degree - 1: "
                                       f"self.c_degree = {D}, {nm('nus')} = {nu}")
        getattr(obj, nm("nus"))[:] = nu
    if w is not None:
        _check.pos_def_sym_mats(w, nm("w_mats"), ParameterFormatError)
        _check.shape_consistency(w.shape[-1], f"{nm('w_mats')}.shape[-1] and {nm('w_mats')}.shape[-2]", D,
                                 "self.c_degree", ParameterFormatError)
        getattr(obj, nm("w_mats"))[:] = w


def _defaults(K, D):
    """ref:533-539 / 93-98: eta = 1/2, zeta = 1/2, m = 0, kappa = 1, nu = D, W = I."""
    return (np.ones(K) / 2.0, np.ones([K, K]) / 2.0, np.zeros([K, D]), np.ones([K]), np.ones(K) * D,
            np.tile(np.eye(D), [K, 1, 1]))


class GenModel(base.Generative):
    """Data-generating HMM and its prior (API shell of ref:18-450; plotting out of scope)."""

    def __init__(self, c_num_classes, c_degree, *, pi_vec=None, a_mat=None, mu_vecs=None, lambda_mats=None,
                 h_eta_vec=None, h_zeta_vecs=None, h_m_vecs=None, h_kappas=None, h_nus=None, h_w_mats=None, seed=None):
        self.c_num_classes = _check.pos_int(c_num_classes, "c_num_classes", ParameterFormatError)
        self.c_degree = _check.pos_int(c_degree, "c_degree", ParameterFormatError)
        self.rng = np.random.default_rng(seed)
        K, D = self.c_num_classes, self.c_degree
        self.pi_vec = np.ones(K) / K
        self.a_mat = np.ones([K, K]) / K
        self.mu_vecs = np.zeros([K, D])
        self.lambda_mats = np.tile(np.eye(D), [K, 1, 1])
        (self.h_eta_vec, self.h_zeta_vecs, self.h_m_vecs, self.h_kappas, self.h_nus, self.h_w_mats) = _defaults(K, D)
        self.set_params(pi_vec, a_mat, mu_vecs, lambda_mats)
        self.set_h_params(h_eta_vec, h_zeta_vecs, h_m_vecs, h_kappas, h_nus, h_w_mats)

    def get_constants(self):
        return {"c_num_classes": self.c_num_classes, "c_degree": self.c_degree}

    def set_params(self, pi_vec=None, a_mat=None, mu_vecs=None, lambda_mats=None):
        K, D = self.c_num_classes, self.c_degree
        if pi_vec is not None:
            _check.float_vec_sum_1(pi_vec, "pi_vec", ParameterFormatError)
            _check.shape_consistency(pi_vec.shape[0], "pi_vec.shape[0]", K, "self.c_num_classes", ParameterFormatError)
            self.pi_vec[:] = pi_vec
        if a_mat is not None:
            _check.float_vecs_sum_1(a_mat, "a_mat", ParameterFormatError)
            _check.shape_consistency(a_mat.shape[-1], "a_mat.shape[-1]", K, "self.c_num_classes", ParameterFormatError)
            self.a_mat[:] = a_mat
        if mu_vecs is not None:
            _check.float_vecs(mu_vecs, "mu_vecs", ParameterFormatError)
            _check.shape_consistency(mu_vecs.shape[-1], "mu_vecs.shape[-1]", D, "self.c_degree", ParameterFormatError)
            self.mu_vecs[:] = mu_vecs
        if lambda_mats is not None:
            _check.pos_def_sym_mats(lambda_mats, "lambda_mats", ParameterFormatError)
            _check.shape_consistency(lambda_mats.shape[-1], "lambda_mats.shape[-1] and lambda_mats.shape[-2]", D,
                                     "self.c_degree", ParameterFormatError)
            self.lambda_mats[:] = lambda_mats
        return self

    def set_h_params(self, h_eta_vec=None, h_zeta_vecs=None, h_m_vecs=None, h_kappas=None, h_nus=None, h_w_mats=None):
        _assign_hmm(self, "h_", self.c_num_classes, self.c_degree, h_eta_vec, h_zeta_vecs, h_m_vecs, h_kappas, h_nus,
                    h_w_mats)
        return self

    def get_params(self):
        return {"pi_vec": self.pi_vec, "a_mat": self.a_mat, "mu_vecs": self.mu_vecs, "lambda_mats": self.lambda_mats}

    def get_h_params(self):
        return {"h_eta_vec": self.h_eta_vec, "h_zeta_vecs": self.h_zeta_vecs, "h_m_vecs": self.h_m_vecs,
                "h_kappas": self.h_kappas, "h_nus": self.h_nus, "h_w_mats": self.h_w_mats}

    def gen_params(self):
        """Prior draw with the reference's call order (ref:293-301)."""
        from scipy.stats import wishart
        self.pi_vec[:] = self.rng.dirichlet(self.h_eta_vec)
        for k in range(self.c_num_classes):
            self.a_mat[k] = self.rng.dirichlet(self.h_zeta_vecs[k])
        for k in range(self.c_num_classes):
            self.lambda_mats[k] = wishart.rvs(df=self.h_nus[k], scale=self.h_w_mats[k], random_state=self.rng)
            self.mu_vecs[k] = self.rng.multivariate_normal(
                mean=self.h_m_vecs[k], cov=np.linalg.inv(self.h_kappas[k] * self.lambda_mats[k]))
        return self

    def gen_sample(self, sample_length, *, device=None, dtype=torch.float64):
        """(x [T, D], one-hot z [T, K]) with the reference's per-step draws (ref:344-358).

        Extension: with ``device`` (e.g. ``"cuda"``) the sequence is drawn ON that device by HIP kernels - the Markov
        chain as a chunk-parallel composition of per-step state maps (``gmmvb_sample_chain``), the emissions in one pass
        (``gmmvb_sample_emissions``) - and returned as torch tensors ``(x [T, D] of ``dtype``, z [T] int64 state
        indices)``; the reference's loop takes 2.4 s per 2e4 steps.  The stream is Philox4x64-10 keyed by a seed drawn
        from ``self.rng`` (reproducible per ``seed``, and on the host with ``numpy.random.Philox``), not the reference's."""
        _check.pos_int(sample_length, "sample_length", DataFormatError)
        if device is not None:
            from .. import _sample
            self.device_sample_seed = int(self.rng.integers(0, 2 ** 63 - 1))
            return _sample.hidden_markov(self.pi_vec, self.a_mat, self.mu_vecs, self.lambda_mats, int(sample_length),
                                         self.device_sample_seed, device, dtype)
        K = self.c_num_classes
        z = np.zeros([sample_length, K], dtype=int)
        x = np.empty([sample_length, self.c_degree])
        cov = np.linalg.inv(self.lambda_mats)
        prev = None
        for t in range(sample_length):
            k = self.rng.choice(K, p=self.pi_vec if prev is None else self.a_mat[prev])
            z[t, k] = 1
            x[t] = self.rng.multivariate_normal(mean=self.mu_vecs[k], cov=cov[k])
            prev = k
        return x, z

    def save_sample(self, filename, sample_length):
        x, z = self.gen_sample(sample_length)
        np.savez_compressed(filename, x=x, z=z)

    def visualize_model(self, sample_length=200):
        if self.c_degree > 2:
            raise ParameterFormatError(_PLOT_MSG)
        for title, val in (("pi_vec", self.pi_vec), ("a_mat", self.a_mat), ("mu_vecs", self.mu_vecs),
                           ("lambda_mats", self.lambda_mats)):
            print(f"{title}:\n{val}")
        raise NotImplementedError("plotting is out of scope for bayesml_amd")


class LearnModel(DeviceModel, base.Posterior, base.PredictiveMixin):
    """Variational posterior and predictive distribution of the Gaussian-emission HMM.

    Reference signature (ref:513-525): ``c_num_classes, c_degree, *, h0_eta_vec=None, h0_zeta_vecs=None,
    h0_m_vecs=None, h0_kappas=None, h0_nus=None, h0_w_mats=None, seed=None``; extensions ``device`` and
    ``verbose`` as in ``gaussianmixture.LearnModel``.  The time axis does not shard: one GPU per sequence.
    """

    def __init__(self, c_num_classes, c_degree, *, h0_eta_vec=None, h0_zeta_vecs=None, h0_m_vecs=None,
                 h0_kappas=None, h0_nus=None, h0_w_mats=None, seed=None, device=None, verbose=True):
        self.c_degree = _check.pos_int(c_degree, "c_degree", ParameterFormatError)
        self.c_num_classes = _check.pos_int(c_num_classes, "c_num_classes", ParameterFormatError)
        from .._engine import check_limits
        check_limits(self.c_degree, self.c_num_classes, hmm=True)
        self.rng = np.random.default_rng(seed)
        self._device, self._comm, self._verbose = device, SingleProcess(), verbose
        self._data_pass_factory = None          # test seam only (see gaussianmixture.LearnModel)
        self._engine = self._x_dev = self._r_cache = None
        K, D = self.c_num_classes, self.c_degree
        (self.h0_eta_vec, self.h0_zeta_vecs, self.h0_m_vecs, self.h0_kappas, self.h0_nus, self.h0_w_mats) = _defaults(K, D)
        self.h0_w_mats_inv = np.linalg.inv(self.h0_w_mats)
        self._ln_c_h0_eta_vec = self._ln_c_h0_zeta_vecs_sum = 0.0
        self._ln_b_h0_w_nus = np.empty(K)
        self.hn_eta_vec = np.empty(K)
        self.hn_zeta_vecs = np.empty([K, K])
        self.hn_m_vecs = np.empty([K, D])
        self.hn_kappas = np.empty([K])
        self.hn_nus = np.empty(K)
        self.hn_w_mats = np.empty([K, D, D])
        self.hn_w_mats_inv = np.empty([K, D, D])
        self._length = 0
        self._e_lambda_mats = np.empty([K, D, D])
        self._e_ln_lambda_dets = np.empty(K)
        self._ln_b_hn_w_nus = np.empty(K)
        self._ln_pi_tilde_vec = np.empty(K)
        self._pi_tilde_vec = np.empty(K)
        self._ln_a_tilde_mat = np.empty([K, K])
        self._a_tilde_mat = np.empty([K, K])
        self._ln_c_hn_zeta_vecs_sum = 0.0
        self.x_bar_vecs = np.zeros([K, D])
        self.ns = np.zeros(K)
        self.ms = np.zeros([K, K])
        self.s_mats = np.zeros([K, D, D])
        self._gamma_first = np.full(K, 1.0 / K)
        self._gamma_last = np.full(K, 1.0 / K)
        self.vl = 0.0
        for t in ("p_x", "p_z", "p_pi", "p_a", "p_mu_lambda", "q_z", "q_pi", "q_a", "q_mu_lambda"):
            setattr(self, "_vl_" + t, 0.0)
        self.p_a_mat = np.ones([K, K]) / K
        self.p_mu_vecs = np.empty([K, D])
        self.p_nus = np.empty([K])
        self.p_lambda_mats = np.empty([K, D, D])
        self.p_lambda_mats_inv = np.empty([K, D, D])
        self.set_h0_params(h0_eta_vec, h0_zeta_vecs, h0_m_vecs, h0_kappas, h0_nus, h0_w_mats)

    # ------------------------------------------------------------------ parameter plumbing
    def get_constants(self):
        return {"c_num_classes": self.c_num_classes, "c_degree": self.c_degree}

    def set_h0_params(self, h0_eta_vec=None, h0_zeta_vecs=None, h0_m_vecs=None, h0_kappas=None, h0_nus=None,
                      h0_w_mats=None):
        """ref:614-691."""
        _assign_hmm(self, "h0_", self.c_num_classes, self.c_degree, h0_eta_vec, h0_zeta_vecs, h0_m_vecs, h0_kappas,
                    h0_nus, h0_w_mats)
        self.h0_w_mats_inv[:] = np.linalg.inv(self.h0_w_mats)
        p = self._prior_tensors("cpu")
        self._ln_c_h0_eta_vec, self._ln_c_h0_zeta_vecs_sum = p.ln_c_eta, p.ln_c_zeta_sum
        self._ln_b_h0_w_nus[:] = _np(p.ln_b_w_nu)
        self.reset_hn_params()
        return self

    def get_h0_params(self):
        return {"h0_eta_vec": self.h0_eta_vec, "h0_zeta_vecs": self.h0_zeta_vecs, "h0_m_vecs": self.h0_m_vecs,
                "h0_kappas": self.h0_kappas, "h0_nus": self.h0_nus, "h0_w_mats": self.h0_w_mats}

    def set_hn_params(self, hn_eta_vec=None, hn_zeta_vecs=None, hn_m_vecs=None, hn_kappas=None, hn_nus=None,
                      hn_w_mats=None):
        """ref:716-795."""
        _assign_hmm(self, "hn_", self.c_num_classes, self.c_degree, hn_eta_vec, hn_zeta_vecs, hn_m_vecs, hn_kappas,
                    hn_nus, hn_w_mats)
        self.hn_w_mats_inv[:] = np.linalg.inv(self.hn_w_mats)
        self._refresh_host_features(self._post_tensors("cpu"))
        self.calc_pred_dist()
        return self

    def get_hn_params(self):
        return {"hn_eta_vec": self.hn_eta_vec, "hn_zeta_vecs": self.hn_zeta_vecs, "hn_m_vecs": self.hn_m_vecs,
                "hn_kappas": self.hn_kappas, "hn_nus": self.hn_nus, "hn_w_mats": self.hn_w_mats}

    def _prior_tensors(self, device):
        return _kside.hmm_prior_from_numpy(self.h0_eta_vec, self.h0_zeta_vecs, self.h0_m_vecs, self.h0_kappas,
                                           self.h0_nus, self.h0_w_mats, device)

    def _post_tensors(self, device):
        t = lambda a: torch.as_tensor(a, dtype=torch.float64, device=device).clone()   # noqa: E731
        return _kside.hmm_features(_kside.HmmPostT(t(self.hn_eta_vec), t(self.hn_zeta_vecs), t(self.hn_m_vecs),
                                                   t(self.hn_kappas), t(self.hn_nus), t(self.hn_w_mats_inv)))

    def _refresh_host_features(self, q):
        self._ln_pi_tilde_vec[:], self._pi_tilde_vec[:] = _np(q.ln_pi_tilde), _np(q.pi_tilde)
        self._ln_a_tilde_mat[:], self._a_tilde_mat[:] = _np(q.ln_a_tilde), _np(q.a_tilde)
        self._ln_c_hn_zeta_vecs_sum = float(q.ln_c_zeta_sum)
        self._e_lambda_mats[:] = self.hn_nus[:, np.newaxis, np.newaxis] * self.hn_w_mats
        self._e_ln_lambda_dets[:] = _np(q.e_ln_lambda_det)
        self._ln_b_hn_w_nus[:] = _np(q.ln_b_w_nu)

    def _store_posterior(self, q):
        self.hn_eta_vec[:], self.hn_zeta_vecs[:] = _np(q.eta), _np(q.zeta)
        self.hn_m_vecs[:], self.hn_kappas[:], self.hn_nus[:] = _np(q.m), _np(q.kappa), _np(q.nu)
        self.hn_w_mats[:], self.hn_w_mats_inv[:] = _np(q.w), _np(q.w_inv)
        self._refresh_host_features(q)

    # ------------------------------------------------------------------ the data pass
    def _pass(self, eng, xd, q, s_prev):
        """_update_q_z (ref:1020-1026) on the GPU: emission -> forward-backward -> statistics."""
        eng.set_params(q.c, q.m, q.u)
        eng.estep(xd)
        ms, g0, gl, sum_ln_c = eng.forward_backward(q.pi_tilde, q.a_tilde)
        ns, h, a, B = eng.split_stats(eng.mstep(xd))
        x_bar, s = _kside.moments_from_stats(ns, a, B, eng.pivot, s_prev)
        # sum gamma ln rho (ref:905) from the moments, in the closed form the reference uses for E[ln p(x|z)] (ref:871-877):
        # the M-step then neither reads the ln rho array nor accumulates h (hmmvb_skip_h)
        sum_g_ln_rho = _kside.sum_gamma_ln_rho(q, ns, x_bar, s)
        return dict(ns=ns, ms=ms, x_bar=x_bar, s=s, g0=g0, gl=gl, sum_g_ln_rho=sum_g_ln_rho, sum_ln_c=sum_ln_c)

    def _whole_iteration_graph(self, xd) -> bool:
        """The shortest sequences (below 4096 steps there is no forgetting pass - no gate, no pinned copy, no host-side hold-off -;
        chunk-parallel kernels of up to 64 states) on the real engine: the data pass is a fixed launch sequence without
        host-side decisions and can be captured with the K-side."""
        return bool(xd.is_cuda and self._data_pass_factory is None and xd.shape[0] < 4096 and self.c_num_classes <= 64)

    @staticmethod
    def _stepper_pass(eng, xd, ks):
        """_update_q_z (ref:1020-1026) under the stepper's current posterior: emission -> forward-backward -> statistics,
        written into the stepper's buffers (the K-sized rest of the iteration is ``ks.step()``)."""
        q = ks.q
        eng.set_params(q.c, q.m, q.u)
        eng.estep(xd)
        eng.forward_backward(q.pi_tilde, q.a_tilde, out=ks.fb)
        eng.mstep(xd, out=ks.stats)

    def _random_pass(self, eng, xd, ks):
        """_init_random_responsibility (ref:941-950): gamma and ms from host Dirichlet draws, written where the data pass
        writes them (the stepper's statistics block and forward-backward summary); ln rho = 0 and cs = 1 keep their
        _init_fb_params values (ref:931-939), so both N-sized sums of the lower bound are 0."""
        K, T, dev = self.c_num_classes, xd.shape[0], xd.device
        if T == 1:
            gamma = self.rng.dirichlet(np.ones(K))[np.newaxis, :]
            ms = np.zeros([K, K])
        else:
            xi = self.rng.dirichlet(np.ones(K ** 2), T).reshape(T, K, K)
            xi[0] = 0.0
            gamma = xi.sum(axis=1)
            gamma[0] = xi[1].sum(axis=1)
            ms = xi.sum(axis=0)
        eng.load_responsibilities(torch.from_numpy(np.ascontiguousarray(gamma)).to(dev))
        eng.mstep(xd, out=ks.stats)
        ks.fb.copy_(torch.from_numpy(np.concatenate([ms.reshape(-1), gamma[0], gamma[-1], [0.0]])).to(dev))

    def _vl(self, prior, q, st):
        return _kside.hmm_lower_bound(prior, q, st["ns"], st["ms"], st["x_bar"], st["s"], st["g0"],
                                      st["sum_g_ln_rho"], st["sum_ln_c"])

    def update_posterior(self, x, max_itr=100, num_init=10, tolerance=1.0E-8, init_type="subsampling"):
        """Variational-Bayes update of ``hn_*`` from one observed sequence (driver of ref:1028-1134)."""
        eng, xd = self._open(x)
        eng.enable_hmm()
        eng.hmm_skip_h(True)
        eng.emission_target(True)        # the VB passes read the emission through the forward-backward recursions only
        self._length = xd.shape[0]
        dev = xd.device
        prior = self._prior_tensors(dev)
        s_prev = torch.as_tensor(self.s_mats, dtype=torch.float64, device=dev)
        keep = {k: np.array(v) for k, v in self.get_hn_params().items()}
        keep["hn_w_mats_inv"] = np.array(self.hn_w_mats_inv)
        best_q, best_vl, never_converged, terms, vl = None, 0.0, True, None, 0.0
        # the K-sized half of an iteration - moments, lower bound under q, q' - as one unit (a replayed hipGraph on the GPU,
        # _kside.HmmKStepper); the data pass writes its statistics and the forward-backward summary into the stepper's buffers
        ks = _kside.HmmKStepper(prior, eng.pivot, eng.stats_len)
        ks.s_prev.copy_(s_prev)
        whole_iteration = self._whole_iteration_graph(xd)

        def data_pass():
            self._stepper_pass(eng, xd, ks)

        for i in range(num_init):
            self.reset_hn_params()
            q = _kside.hmm_post_from_prior(prior)
            if init_type == "subsampling":
                size, a, B = self._subsample_moments(eng, xd, self._length)
                q = _kside.subsample_moments_init(q, size, a, B, eng.pivot, _kside.hmm_features)
                ks.load(q)
                ks.h_scale.fill_(1.0)
                data_pass()
            elif init_type == "random_responsibility":
                ks.load(q)
                ks.h_scale.fill_(0.0)          # (no emission in this pass: both N-sized sums of the lower bound are 0, ref:931-939)
                self._random_pass(eng, xd, ks)
            else:
                raise ValueError(f"init_type={init_type} is unsupported. This function supports only "
                                 '"subsampling" and "random_responsibility"')
            ks.step()                          # lower bound under q on this pass's moments, and q' from them
            terms = ks.read()
            vl = terms["vl"]
            self._say(f"\r{i}. VL: {vl}")
            ks.h_scale.fill_(1.0)
            for t in range(max_itr):
                vl_before = vl
                # q <- q' (_update_q_mu_lambda / _update_q_pi / _update_q_a, ref:1099-1101), _update_q_z (ref:1102), _calc_vl
                # (ref:1103) and the next q' - for short sequences all of it one replayed hipGraph (HmmKStepper.iterate)
                ks.iterate(data_pass, whole_iteration)
                terms = ks.read()
                vl = terms["vl"]
                self._say(f"\r{i}. VL: {vl} t={t} ")
                with np.errstate(divide="ignore", invalid="ignore"):
                    if np.abs((vl - vl_before) / vl_before) < tolerance:
                        never_converged = False
                        self._say("(converged)")
                        break
            if i == 0 or vl > best_vl:
                self._say("*", end="\n")
                best_vl, best_q = vl, ks.current()
            else:
                self._say("", end="\n")
            self.vl = vl
        s_prev = ks.s_prev.clone()
        if never_converged:
            warnings.warn("Algorithm has not converged even once.", ResultWarning)
        if best_q is not None:
            self._store_posterior(best_q)
            q = best_q
        else:
            for k, v in keep.items():
                getattr(self, k)[:] = v
            q = self._post_tensors(dev)
            self._refresh_host_features(q)
        if terms is not None:
            for k, v in terms.items():
                setattr(self, "vl" if k == "vl" else "_vl_" + k, float(v))
        st = self._pass(eng, xd, q, s_prev)                      # ref:1133
        self._store_stats(st)
        return self

    def _store_stats(self, st):
        self.ns[:], self.ms[:] = _np(st["ns"]), _np(st["ms"])
        self.x_bar_vecs[:], self.s_mats[:] = _np(st["x_bar"]), _np(st["s"])
        self._gamma_first[:], self._gamma_last[:] = _np(st["g0"]), _np(st["gl"])
        self._r_cache = None

    @property
    def gamma_vecs(self):
        if self._engine is None:
            return None
        if self._r_cache is None:
            self._r_cache = _np(self._engine.responsibilities())
        return self._r_cache

    @property
    def alpha_vecs(self):
        return None if self._engine is None else _np(self._engine.hmm_readout("alpha"))

    @property
    def beta_vecs(self):
        """The reference's scaled backward variables (gamma = alpha o beta, ref:1013-1014), formed on access."""
        return None if self._engine is None else _np(self._engine.hmm_readout("beta"))

    def xi_rows(self, row0=0, n=None):
        """``xi_mats[row0:row0 + n]`` of the last pass ([n, K, K]; xi_mats[0] = 0 as in the reference, ref:1068).  The
        whole array is T K^2 doubles (82 GB at T = 1e7, K = 32): it is never materialised, ask for the rows needed."""
        if self._engine is None:
            return None
        return _np(self._engine.hmm_readout("xi", row0, n, self._a_tilde_mat))

    @property
    def xi_mats(self):
        """The full [T, K, K] array, for sizes that fit the host (<= 2 GiB); use ``xi_rows`` otherwise."""
        if self._engine is None:
            return None
        T, K = self._engine.rows, self.c_num_classes
        if T * K * K * 8 > 2 ** 31:
            raise MemoryError(f"xi_mats would be {T * K * K * 8 / 2 ** 30:.1f} GiB; use xi_rows(row0, n)")
        return self.xi_rows(0, T)

    # ------------------------------------------------------------------ read-outs
    def estimate_params(self, loss="squared"):
        """(pi, A, mu_k, Lambda_k) estimates (ref:1136-1211).  "0-1" follows the reference's code: it divides
        by ``sum - c_degree`` and tests ``hn_eta_vec > 1`` for every row of A (ref:1166-1172)."""
        K, D = self.c_num_classes, self.c_degree
        if loss == "squared":
            return (self.hn_eta_vec / self.hn_eta_vec.sum(),
                    self.hn_zeta_vecs / self.hn_zeta_vecs.sum(axis=1, keepdims=True), self.hn_m_vecs, self._e_lambda_mats)
        if loss == "0-1":
            pi_hat = np.empty(K)
            ok = np.all(self.hn_eta_vec > 1)
            if ok:
                pi_hat[:] = (self.hn_eta_vec - 1) / (np.sum(self.hn_eta_vec) - D)
            else:
                warnings.warn("MAP estimate of pi_vec doesn't exist for the current hn_eta_vec.", ResultWarning)
                pi_hat[:] = np.nan
            a_hat = np.empty([K, K])
            for i in range(K):
                if ok:
                    a_hat[i] = (self.hn_zeta_vecs[i] - 1) / (np.sum(self.hn_zeta_vecs[i]) - D)
                else:
                    warnings.warn(f"MAP estimate of a_mat[{i}] doesn't exist for the current hn_zeta_vecs[{i}].",
                                  ResultWarning)
                    a_hat[i] = np.nan
            lam = np.empty([K, D, D])
            for k in range(K):
                if self.hn_nus[k] >= D + 1:
                    lam[k] = (self.hn_nus[k] - D - 1) * self.hn_w_mats[k]
                else:
                    warnings.warn(f"MAP estimate of lambda_mat doesn't exist for the current hn_nus[{k}].", ResultWarning)
                    lam[k] = np.nan
            return pi_hat, a_hat, self.hn_m_vecs, lam
        if loss == "KL":
            from scipy.stats import dirichlet, multivariate_t, wishart
            dof = self.hn_nus - D + 1
            return (dirichlet(self.hn_eta_vec), [dirichlet(self.hn_zeta_vecs[k]) for k in range(K)],
                    [multivariate_t(loc=self.hn_m_vecs[k], shape=self.hn_w_mats_inv[k] / self.hn_kappas[k] / dof[k],
                                    df=dof[k]) for k in range(K)],
                    [wishart(df=self.hn_nus[k], scale=self.hn_w_mats[k]) for k in range(K)])
        raise CriteriaError(f"loss={loss} is unsupported. "
                            "This function supports \"squared\", \"0-1\", and \"KL\".")

    def visualize_posterior(self):
        if self.c_degree > 2:
            raise ParameterFormatError(_PLOT_MSG)
        for title, val in (("hn_eta_vec:", self.hn_eta_vec), ("hn_zeta_vecs:", self.hn_zeta_vecs),
                           ("hn_m_vecs:", self.hn_m_vecs), ("hn_kappas:", self.hn_kappas), ("hn_nus:", self.hn_nus),
                           ("hn_w_mats:", self.hn_w_mats)):
            print(title)
            print(f"{val}")
        raise NotImplementedError("plotting is out of scope for bayesml_amd")

    def get_p_params(self):
        return {"p_a_mat": self.p_a_mat, "p_mu_vecs": self.p_mu_vecs, "p_nus": self.p_nus,
                "p_lambda_mats": self.p_lambda_mats}

    def calc_pred_dist(self):
        """ref:1332-1338."""
        self.p_a_mat[:] = self.hn_zeta_vecs / self.hn_zeta_vecs.sum(axis=1, keepdims=True)
        self.p_mu_vecs[:] = self.hn_m_vecs
        self.p_nus[:] = self.hn_nus - self.c_degree + 1
        self.p_lambda_mats[:] = (self.hn_kappas * self.p_nus / (self.hn_kappas + 1))[:, np.newaxis, np.newaxis] * self.hn_w_mats
        return self

    def make_prediction(self, loss="squared"):
        """Next-observation prediction from ``gamma_vecs[-1] @ p_a_mat`` (ref:1340-1370)."""
        w = self._gamma_last @ self.p_a_mat
        if loss == "squared":
            return np.sum(w[:, np.newaxis] * self.p_mu_vecs, axis=0)
        if loss == "0-1":
            from scipy.stats import multivariate_t
            best, arg = -1.0, np.empty([self.c_degree])
            for k in range(self.c_num_classes):
                dens = multivariate_t.pdf(x=self.p_mu_vecs[k], loc=self.p_mu_vecs[k],
                                          shape=np.linalg.inv(self.p_lambda_mats[k]), df=self.p_nus[k])
                if dens * w[k] > best:
                    arg[:] = self.p_mu_vecs[k]
                    best = dens * w[k]
            return arg
        raise CriteriaError(f"loss={loss} is unsupported. "
                            "This function supports \"squared\" and \"0-1\".")

    def pred_and_update(self, x, loss="squared", max_itr=100, num_init=10, tolerance=1.0E-8,
                        init_type="random_responsibility"):
        """ref:1372-1423."""
        _check.float_vec(x, "x", DataFormatError)
        if x.shape != (self.c_degree,):
            raise DataFormatError(f"x must be a 1-dimensional float array whose size is c_degree: {self.c_degree}.")
        self.calc_pred_dist()
        prediction = self.make_prediction(loss=loss)
        self.overwrite_h0_params()
        self.update_posterior(x[np.newaxis, :], max_itr=max_itr, num_init=num_init, tolerance=tolerance,
                              init_type=init_type)
        return prediction

    def estimate_latent_vars(self, x, loss="0-1", viterbi=True):
        """One-hot Viterbi path (``viterbi=True``, "0-1" only) or posterior marginals (ref:1425-1499)."""
        eng, xd = self._open(x)
        eng.enable_hmm()
        eng.hmm_skip_h(True)
        eng.emission_target(not viterbi)     # (the Viterbi pass reads the ln rho array)
        self._length = xd.shape[0]
        q = self._post_tensors(xd.device)
        K = self.c_num_classes
        if viterbi:
            if loss != "0-1":
                raise CriteriaError(f"loss=\"{loss}\" is unsupported. "
                                    "When viterbi == True, this function supports only \"0-1\".")
            eng.set_params(q.c, q.m, q.u)
            eng.estep(xd)
            z = eng.viterbi(q.ln_pi_tilde, q.ln_a_tilde).to("cpu").numpy()
            return np.eye(K, dtype=int)[z]
        s_prev = torch.as_tensor(self.s_mats, dtype=torch.float64, device=xd.device)
        self._store_stats(self._pass(eng, xd, q, s_prev))
        if loss in ("squared", "KL"):
            return self.gamma_vecs
        if loss == "0-1":
            return np.eye(K, dtype=int)[eng.argmax().to("cpu").numpy()]
        raise CriteriaError(f"loss=\"{loss}\" is unsupported. "
                            "When viterbi == False, This function supports \"squared\", \"0-1\", and \"KL\".")

    def estimate_latent_vars_and_update(self, x, loss="0-1", viterbi=True, max_itr=100, num_init=10,
                                        tolerance=1.0E-8, init_type="subsampling"):
        """ref:1501-1559 (the reference validates ``x`` as ONE c_degree-vector here)."""
        _check.float_vec(x, "x", DataFormatError)
        if x.shape != (self.c_degree,):
            raise DataFormatError(f"x must be a 1-dimensional float array whose size is c_degree: {self.c_degree}.")
        z_hat = self.estimate_latent_vars(x, loss=loss, viterbi=viterbi)
        self.overwrite_h0_params()
        self.update_posterior(x, max_itr=max_itr, num_init=num_init, tolerance=tolerance, init_type=init_type)
        return z_hat
