"""``multivariate_normal.GenModel`` / ``LearnModel``: drop-in for the posterior-update path of
``bayesml/multivariate_normal/_multivariatenormal.py`` (cited below as ``ref:<lines>``).

The conjugate update (ref:501-524) needs n, the sample mean and the scatter matrix of x - one pass over the data.
That pass is the GMM M-step kernel with a single component whose responsibilities are all one
(``gmmvb_load_responsibilities`` + ``gmmvb_mstep``): ns = n, a = sum (x - p), B = sum (x - p)(x - p)^T about a
pivot p, from which  x_bar = p + a/n  and  sum (x - x_bar)(x - x_bar)^T = B - a a^T / n.  Everything K-sized stays
on the host in NumPy exactly as in the reference.  No CPU fallback: without the library or a GPU
``update_posterior`` raises ``EngineUnavailableError``.
"""
from __future__ import annotations

import warnings

import numpy as np
import torch

from .. import _check, base
from .._exceptions import CriteriaError, DataFormatError, ParameterFormatError, ResultWarning

_PLOT_MSG = "if c_degree > 2, it is impossible to visualize the model by this function."


def _assign_nw(obj, prefix, D, m, kappa, nu, w):
    """Validated assignment shared by set_h_params / set_h0_params / set_hn_params (ref:92-123, 378-409, 441-472)."""
    if m is not None:
        _check.float_vec(m, prefix + "m_vec", ParameterFormatError)
        _check.shape_consistency(m.shape[0], prefix + "m_vec.shape[0]", D, "self.c_degree", ParameterFormatError)
        getattr(obj, prefix + "m_vec")[:] = m
    if kappa is not None:
        setattr(obj, prefix + "kappa", _check.pos_float(kappa, prefix + "kappa", ParameterFormatError))
    if nu is not None:
        setattr(obj, prefix + "nu", _check.pos_float(nu, prefix + "nu", ParameterFormatError))
        if nu <= D - 1:
            raise ParameterFormatError(f"{prefix}nu must be greater than self.c_degree - 1: "
                                       f"self.c_degree = {D}, {prefix}nu = {nu}")
    if w is not None:
        _check.pos_def_sym_mat(w, prefix + "w_mat", ParameterFormatError)
        _check.shape_consistency(w.shape[0], f"{prefix}w_mat.shape[0] and {prefix}w_mat.shape[1]", D, "self.c_degree",
                                 ParameterFormatError)
        getattr(obj, prefix + "w_mat")[:] = w


class GenModel(base.Generative):
    """Data-generating model and its Gauss-Wishart prior (ref:18-279; plotting is out of scope)."""

    def __init__(self, c_degree, mu_vec=None, lambda_mat=None, h_m_vec=None, h_kappa=1.0, h_nu=None, h_w_mat=None,
                 seed=None):
        self.c_degree = _check.pos_int(c_degree, "c_degree", ParameterFormatError)
        self.rng = np.random.default_rng(seed)
        D = self.c_degree
        self.mu_vec = np.zeros(D)
        self.lambda_mat = np.eye(D)
        self.h_m_vec = np.zeros(D)
        self.h_kappa = 1.0
        self.h_nu = float(D)
        self.h_w_mat = np.eye(D)
        self.set_params(mu_vec, lambda_mat)
        self.set_h_params(h_m_vec, h_kappa, h_nu, h_w_mat)

    def get_constants(self):
        return {"c_degree": self.c_degree}

    def set_h_params(self, h_m_vec=None, h_kappa=None, h_nu=None, h_w_mat=None):
        _assign_nw(self, "h_", self.c_degree, h_m_vec, h_kappa, h_nu, h_w_mat)
        return self

    def get_h_params(self):
        return {"h_m_vec": self.h_m_vec, "h_kappa": self.h_kappa, "h_nu": self.h_nu, "h_w_mat": self.h_w_mat}

    def gen_params(self):
        """Lambda ~ Wishart(h_nu, h_w_mat), then mu ~ N(h_m_vec, (h_kappa Lambda)^-1): the reference's call order
        on ``self.rng`` (ref:133-140)."""
        from scipy.stats import wishart
        self.lambda_mat[:] = wishart.rvs(df=self.h_nu, scale=self.h_w_mat, random_state=self.rng)
        self.mu_vec[:] = self.rng.multivariate_normal(mean=self.h_m_vec, cov=np.linalg.inv(self.h_kappa * self.lambda_mat))
        return self

    def set_params(self, mu_vec=None, lambda_mat=None):
        D = self.c_degree
        if mu_vec is not None:
            _check.float_vec(mu_vec, "mu_vec", ParameterFormatError)
            _check.shape_consistency(mu_vec.shape[0], "mu_vec.shape[0]", D, "self.c_degree", ParameterFormatError)
            self.mu_vec[:] = mu_vec
        if lambda_mat is not None:
            _check.pos_def_sym_mat(lambda_mat, "lambda_mat", ParameterFormatError)
            _check.shape_consistency(lambda_mat.shape[0], "lambda_mat.shape[0] and lambda_mat.shape[1]", D,
                                     "self.c_degree", ParameterFormatError)
            self.lambda_mat[:] = lambda_mat
        return self

    def get_params(self):
        return {"mu_vec": self.mu_vec, "lambda_mat": self.lambda_mat}

    def gen_sample(self, sample_size):
        """One ``multivariate_normal(size=sample_size)`` draw, like the reference (ref:190-192)."""
        _check.pos_int(sample_size, "sample_size", DataFormatError)
        return self.rng.multivariate_normal(mean=self.mu_vec, cov=np.linalg.inv(self.lambda_mat), size=sample_size)

    def save_sample(self, filename, sample_size):
        np.savez_compressed(filename, x=self.gen_sample(sample_size))

    def visualize_model(self, sample_size=100):
        if self.c_degree > 2:
            raise ParameterFormatError(_PLOT_MSG)
        print(f"mu:\n{self.mu_vec}")
        print(f"lambda_mat:\n{self.lambda_mat}")
        raise NotImplementedError("plotting is out of scope for bayesml_amd (SURVEY.md section 2)")


class LearnModel(base.Posterior, base.PredictiveMixin):
    """Posterior and predictive distribution (ref:281-800).  Positional parameters are the reference's:
    ``c_degree, h0_m_vec=None, h0_kappa=1.0, h0_nu=None, h0_w_mat=None``; keyword-only ``device`` selects the GPU."""

    def __init__(self, c_degree, h0_m_vec=None, h0_kappa=1.0, h0_nu=None, h0_w_mat=None, *, device=None):
        self.c_degree = _check.pos_int(c_degree, "c_degree", ParameterFormatError)
        from .._engine import check_limits
        check_limits(self.c_degree)
        D = self.c_degree
        self._device = device
        self._engine = None
        self._data_pass_factory = None       # test seam only (tests/fake_engine.py); the default is the HIP engine
        self.h0_m_vec = np.zeros(D)
        self.h0_kappa = 1.0
        self.h0_nu = float(D)
        self.h0_w_mat = np.eye(D)
        self.h0_w_mat_inv = np.eye(D)
        self.hn_m_vec = np.zeros(D)
        self.hn_kappa = 1.0
        self.hn_nu = float(D)
        self.hn_w_mat = np.eye(D)
        self.hn_w_mat_inv = np.eye(D)
        self.p_m_vec = np.zeros(D)
        self.p_nu = 1.0
        self.p_v_mat = np.eye(D) / 2.0
        self.p_v_mat_inv = np.eye(D) * 2.0
        self.set_h0_params(h0_m_vec, h0_kappa, h0_nu, h0_w_mat)

    def get_constants(self):
        return {"c_degree": self.c_degree}

    def set_h0_params(self, h0_m_vec=None, h0_kappa=None, h0_nu=None, h0_w_mat=None):
        _assign_nw(self, "h0_", self.c_degree, h0_m_vec, h0_kappa, h0_nu, h0_w_mat)
        self.h0_w_mat_inv = np.linalg.inv(self.h0_w_mat)
        self.reset_hn_params()
        return self

    def get_h0_params(self):
        return {"h0_m_vec": self.h0_m_vec, "h0_kappa": self.h0_kappa, "h0_nu": self.h0_nu, "h0_w_mat": self.h0_w_mat}

    def set_hn_params(self, hn_m_vec=None, hn_kappa=None, hn_nu=None, hn_w_mat=None):
        _assign_nw(self, "hn_", self.c_degree, hn_m_vec, hn_kappa, hn_nu, hn_w_mat)
        self.hn_w_mat_inv = np.linalg.inv(self.hn_w_mat)
        self.calc_pred_dist()
        return self

    def get_hn_params(self):
        return {"hn_m_vec": self.hn_m_vec, "hn_kappa": self.hn_kappa, "hn_nu": self.hn_nu, "hn_w_mat": self.hn_w_mat}

    # ------------------------------------------------------------------ the data pass
    def _moments(self, x):
        """(n, x_bar [D], scatter [D, D] = sum (x - x_bar)(x - x_bar)^T) of the rows of x, from ONE pass of the GMM
        M-step kernel with K = 1 and unit responsibilities (no E-step, nothing N-sized on the host)."""
        D = self.c_degree
        if isinstance(x, torch.Tensor):
            if not (x.dtype.is_floating_point and x.dim() >= 1):
                raise DataFormatError("x must be a numpy.ndarray whose ndim >= 1.")
        else:
            _check.float_vecs(x, "x", DataFormatError)
        if x.shape[-1] != D:
            raise DataFormatError(f"x.shape[-1] must be c_degree:{D}")
        x = x.reshape(-1, D)
        n = x.shape[0]
        if self._data_pass_factory is not None:
            eng = self._data_pass_factory(1, D, x)
            xd = eng.adopt(x)
        else:
            from .._engine import DataPass, EngineUnavailableError
            if not torch.cuda.is_available():
                raise EngineUnavailableError("bayesml_amd.multivariate_normal.LearnModel needs an MI355X: "
                                             "the data pass has no CPU fallback")
            dev = torch.device("cuda", torch.cuda.current_device()) if self._device is None else torch.device(self._device)
            if isinstance(x, torch.Tensor):
                xd = x.to(dev)
                if xd.dtype not in (torch.float32, torch.float64):
                    xd = xd.to(torch.float64)
            else:
                xh = np.ascontiguousarray(x if x.dtype in (np.float32, np.float64) else x.astype(np.float64))
                xd = torch.from_numpy(xh).to(dev)
            xd = xd.contiguous()
            eng = self._engine
            if (eng is None or eng.D != D or eng.x_dtype != xd.dtype or eng.max_rows < n or eng.device != dev
                    or getattr(eng, "_ws", None) is None):
                if eng is not None:
                    eng.close()
                eng = DataPass(1, D, xd.dtype, n, dev)
            self._engine = eng
        pivot = xd[: min(n, 4096)].to(torch.float64).mean(dim=0)
        eng.set_pivot(pivot)
        eng.load_responsibilities(torch.ones((n, 1), dtype=torch.float64, device=xd.device))
        ns, _h, a, B = eng.split_stats(eng.mstep(xd))
        abar = a[0] / ns[0]
        scatter = B[0] - ns[0] * abar[:, None] * abar[None, :]
        to_np = lambda t: t.detach().to("cpu", torch.float64).numpy()   # noqa: E731
        return n, to_np(pivot + abar), to_np(scatter)

    def update_posterior(self, x):
        """Conjugate Gauss-Wishart update (ref:501-524): the N-sized sums come from the GPU, the D-sized closed form
        is the reference's."""
        n, x_bar, scatter = self._moments(x)
        diff = x_bar - self.hn_m_vec
        self.hn_w_mat_inv[:] = (self.hn_w_mat_inv + scatter
                                + diff[:, np.newaxis] @ diff[np.newaxis, :] * self.hn_kappa * n / (self.hn_kappa + n))
        self.hn_m_vec[:] = (self.hn_kappa * self.hn_m_vec + n * x_bar) / (self.hn_kappa + n)
        self.hn_kappa += n
        self.hn_nu += n
        self.hn_w_mat[:] = np.linalg.inv(self.hn_w_mat_inv)
        return self

    # ------------------------------------------------------------------ read-outs
    def estimate_params(self, loss="squared", dict_out=False):
        """(mu_vec, lambda_mat) under "squared" / "0-1" (None when the MAP does not exist) or the frozen
        posterior distributions under "KL" (ref:538-592)."""
        D = self.c_degree
        if loss == "squared":
            est = (self.hn_m_vec, self.hn_nu * self.hn_w_mat)
        elif loss == "0-1":
            if self.hn_nu >= D + 1:
                est = (self.hn_m_vec, (self.hn_nu - D - 1) * self.hn_w_mat)
            else:
                warnings.warn("MAP estimate of lambda_mat doesn't exist for the current hn_nu.", ResultWarning)
                est = (self.hn_m_vec, None)
        elif loss == "KL":
            from scipy.stats import multivariate_t, wishart
            dof = self.hn_nu - D + 1
            return (multivariate_t(loc=self.hn_m_vec, shape=self.hn_w_mat_inv / self.hn_kappa / dof, df=dof),
                    wishart(df=self.hn_nu, scale=self.hn_w_mat))
        else:
            raise CriteriaError("Unsupported loss function! "
                                "This function supports \"squared\", \"0-1\", and \"KL\".")
        return {"mu_vec": est[0], "lambda_mat": est[1]} if dict_out else est

    def visualize_posterior(self):
        if self.c_degree > 2:
            raise ParameterFormatError(_PLOT_MSG)
        for title, val in (("hn_m_vec:", self.hn_m_vec), ("hn_kappa:", self.hn_kappa), ("hn_nu:", self.hn_nu),
                           ("hn_w_mat:", self.hn_w_mat), ("E[lambda_mat]=", self.hn_nu * self.hn_w_mat)):
            print(title)
            print(f"{val}")
        raise NotImplementedError("plotting is out of scope for bayesml_amd (SURVEY.md section 2)")

    def get_p_params(self):
        return {"p_m_vec": self.p_m_vec, "p_nu": self.p_nu, "p_v_mat": self.p_v_mat}

    def calc_pred_dist(self):
        """Student-t predictive parameters (ref:705-711)."""
        self.p_m_vec[:] = self.hn_m_vec
        self.p_nu = self.hn_nu - self.c_degree + 1
        self.p_v_mat[:] = self.hn_kappa * self.p_nu / (self.hn_kappa + 1) * self.hn_w_mat
        self.p_v_mat_inv[:] = (self.hn_kappa + 1) / self.hn_kappa / self.p_nu * self.hn_w_mat_inv
        return self

    def make_prediction(self, loss="squared"):
        if loss in ("squared", "0-1"):
            return self.p_m_vec
        if loss == "KL":
            from scipy.stats import multivariate_t
            return multivariate_t(loc=self.p_m_vec, shape=self.p_v_mat_inv, df=self.p_nu)
        raise CriteriaError("Unsupported loss function! "
                            "This function supports \"squared\", \"0-1\", and \"KL\".")

    def pred_and_update(self, x, loss="squared"):
        """Predict one point, then fold it into the posterior (ref:753-759)."""
        _check.float_vec(x, "x", DataFormatError)
        if x.shape != (self.c_degree,):
            raise DataFormatError(f"x must be a 1-dimensional float array whose size is c_degree: {self.c_degree}.")
        self.calc_pred_dist()
        prediction = self.make_prediction(loss=loss)
        self.update_posterior(x[np.newaxis, :])
        return prediction

    def fit(self, x):
        self.reset_hn_params()
        self.update_posterior(x)
        return self

    def predict(self):
        self.calc_pred_dist()
        return self.make_prediction(loss="squared")
