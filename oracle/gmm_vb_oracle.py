"""CPU oracle for the GMM variational-Bayes posterior update (TEST INFRASTRUCTURE ONLY).

This file is a NumPy/fp64 restatement of the algorithm implemented by
``bayesml.gaussianmixture.LearnModel`` in the reference (v0.3.1).  It is *not*
part of the product: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it, and only as the checker /
the timed CPU baseline.  The product path (``bayesml_amd``) never imports it and
fails loudly when the HIP extension is missing.

Pinning: the reference ships no test for this path, so
the oracle is pinned against outputs of the reference itself, generated in the
build container by ``tests/golden/make_golden.py`` (which imports
``/root/reference``) and committed as ``tests/golden/*.npz``.
``tests/test_oracle_golden.py`` checks every function below against them.

The arithmetic deliberately follows the reference's formulation (per-component
loops of ``(diff @ Lambda) * diff`` and ``(r_k * diff.T) @ diff``, two data
passes, fp64 everywhere), not the engine's, so that it is an independent check
of the engine's single-pass, whitened, mixed-precision formulation.

Every function cites the reference lines it restates as
``_gaussianmixture.py:<lines>`` (= ``bayesml/gaussianmixture/_gaussianmixture.py``).
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np
from scipy.special import digamma, gammaln, xlogy
from scipy.stats import dirichlet as _dirichlet

LN_2PI = float(np.log(2.0 * np.pi))


# --------------------------------------------------------------------------- state
@dataclass
class Prior:
    """h0_* hyper-parameters (defaults: _gaussianmixture.py:438-443)."""
    alpha: np.ndarray          # [K]
    m: np.ndarray              # [K, D]
    kappa: np.ndarray          # [K]
    nu: np.ndarray             # [K]
    w: np.ndarray              # [K, D, D]
    w_inv: np.ndarray = None   # [K, D, D]
    ln_c_alpha: float = 0.0    # _gaussianmixture.py:662
    ln_b_w_nu: np.ndarray = None  # _gaussianmixture.py:663-669

    @staticmethod
    def default(K: int, D: int) -> "Prior":
        p = Prior(alpha=np.full(K, 0.5), m=np.zeros((K, D)), kappa=np.ones(K),
                  nu=np.full(K, float(D)), w=np.tile(np.eye(D), (K, 1, 1)))
        return p.refresh()

    def refresh(self) -> "Prior":
        """_calc_prior_features, _gaussianmixture.py:661-669 (+ h0_w_mats_inv, :557)."""
        K, D = self.m.shape
        self.w_inv = np.linalg.inv(self.w)
        self.ln_c_alpha = float(gammaln(self.alpha.sum()) - gammaln(self.alpha).sum())
        self.ln_b_w_nu = ln_wishart_b_from_logdet_w(np.linalg.slogdet(self.w)[1], self.nu, D, sign=-1.0)
        return self


def ln_wishart_b_from_logdet_w(logdet: np.ndarray, nu: np.ndarray, D: int, sign: float) -> np.ndarray:
    """ln B(W, nu).  ``sign=-1`` with ``logdet = ln|W|`` is the prior form
    (_gaussianmixture.py:663-669); ``sign=+1`` with ``logdet = ln|W^-1|`` is the
    posterior form (_gaussianmixture.py:750-756).  Both are the same quantity."""
    lg = gammaln((nu[:, None] - np.arange(D)) / 2.0).sum(axis=1)
    return (sign * nu * logdet - nu * D * np.log(2.0) - D * (D - 1) / 2.0 * np.log(np.pi) - 2.0 * lg) / 2.0


@dataclass
class Posterior:
    """hn_* hyper-parameters and the derived expectations the E-step consumes."""
    alpha: np.ndarray
    m: np.ndarray
    kappa: np.ndarray
    nu: np.ndarray
    w: np.ndarray
    w_inv: np.ndarray
    e_ln_pi: np.ndarray = None        # _gaussianmixture.py:739
    e_lambda: np.ndarray = None       # _gaussianmixture.py:746
    e_ln_lambda_det: np.ndarray = None  # _gaussianmixture.py:747-749
    ln_b_w_nu: np.ndarray = None      # _gaussianmixture.py:750-756

    @staticmethod
    def from_prior(p: Prior) -> "Posterior":
        """reset_hn_params (base.py:260-267) -> set_hn_params (_gaussianmixture.py:581-641)."""
        q = Posterior(alpha=p.alpha.copy(), m=p.m.copy(), kappa=p.kappa.copy(), nu=p.nu.copy(),
                      w=p.w.copy(), w_inv=np.linalg.inv(p.w))
        q.refresh_pi()
        q.refresh_lambda()
        return q

    def refresh_pi(self) -> None:
        """_calc_q_pi_features, _gaussianmixture.py:738-739."""
        self.e_ln_pi = digamma(self.alpha) - digamma(self.alpha.sum())

    def refresh_lambda(self) -> None:
        """_calc_q_lambda_features, _gaussianmixture.py:745-756."""
        D = self.m.shape[1]
        self.e_lambda = self.nu[:, None, None] * self.w
        logdet_inv = np.linalg.slogdet(self.w_inv)[1]
        self.e_ln_lambda_det = (digamma((self.nu[:, None] - np.arange(D)) / 2.0).sum(axis=1)
                                + D * np.log(2.0) - logdet_inv)
        self.ln_b_w_nu = ln_wishart_b_from_logdet_w(logdet_inv, self.nu, D, sign=+1.0)

    def copy(self) -> "Posterior":
        return Posterior(*(None if a is None else np.array(a) for a in (
            self.alpha, self.m, self.kappa, self.nu, self.w, self.w_inv,
            self.e_ln_pi, self.e_lambda, self.e_ln_lambda_det, self.ln_b_w_nu)))


@dataclass
class Stats:
    """Output of one E+M data pass (attributes r_vecs/ns/x_bar_vecs/s_mats of the reference)."""
    ln_rho: np.ndarray   # [N, K]
    r: np.ndarray        # [N, K]
    ns: np.ndarray       # [K]
    x_bar: np.ndarray    # [K, D]
    s: np.ndarray        # [K, D, D]


# --------------------------------------------------------------------------- N-side
def e_step(x: np.ndarray, q: Posterior) -> tuple[np.ndarray, np.ndarray]:
    """_update_q_z without its trailing statistics call, _gaussianmixture.py:772-783.

    Returns (ln_rho, r).  The Mahalanobis term uses the reference's
    ``sum((diff @ Lambda_k) * diff, axis=1)`` form, component by component."""
    N, D = x.shape
    K = q.m.shape[0]
    ln_rho = np.empty((N, K))
    ln_rho[:] = q.e_ln_pi + (q.e_ln_lambda_det - D * LN_2PI - D / q.kappa) / 2.0
    for k in range(K):
        diff = x - q.m[k]
        ln_rho[:, k] -= np.sum((diff @ q.e_lambda[k]) * diff, axis=1) / 2.0
    r = np.exp(ln_rho - ln_rho.max(axis=1, keepdims=True))
    r /= r.sum(axis=1, keepdims=True)
    return ln_rho, r


def m_step_stats(x: np.ndarray, r: np.ndarray, s_prev: np.ndarray | None = None):
    """_calc_n_x_bar_s, _gaussianmixture.py:725-732.

    ``s_prev`` plays the role of the reference's persistent ``self.s_mats``: for a
    component with ``ns[k] == 0`` the reference leaves ``x_bar_vecs[k]`` as the raw
    (zero) sum and does not touch ``s_mats[k]`` (:729).  The reference's initial
    ``s_mats`` is ``np.empty`` (undefined, :466); the oracle defines it as zeros."""
    N, D = x.shape
    K = r.shape[1]
    ns = r.sum(axis=0)
    x_bar = r.T @ x
    s = np.zeros((K, D, D)) if s_prev is None else np.array(s_prev, dtype=float)
    for k in range(K):
        if ns[k] > 0:
            x_bar[k] /= ns[k]
            diff = x - x_bar[k]
            s[k] = ((r[:, k] * diff.T) @ diff) / ns[k]
    return ns, x_bar, s


def data_pass(x: np.ndarray, q: Posterior, s_prev: np.ndarray | None = None) -> Stats:
    """_update_q_z including the statistics, _gaussianmixture.py:772-784."""
    ln_rho, r = e_step(x, q)
    ns, x_bar, s = m_step_stats(x, r, s_prev)
    return Stats(ln_rho, r, ns, x_bar, s)


# --------------------------------------------------------------------------- K-side
def update_q_mu_lambda(p: Prior, q: Posterior, st: Stats) -> None:
    """_update_q_mu_lambda, _gaussianmixture.py:758-770 (in place on ``q``)."""
    q.kappa = p.kappa + st.ns
    q.m = (p.kappa[:, None] * p.m + st.ns[:, None] * st.x_bar) / q.kappa[:, None]
    q.nu = p.nu + st.ns
    dev = st.x_bar - p.m
    q.w_inv = (p.w_inv + st.ns[:, None, None] * st.s
               + (p.kappa * st.ns / q.kappa)[:, None, None] * (dev[:, :, None] @ dev[:, None, :]))
    q.w = np.linalg.inv(q.w_inv)
    q.refresh_lambda()


def update_q_pi(p: Prior, q: Posterior, st: Stats) -> None:
    """_update_q_pi, _gaussianmixture.py:741-743."""
    q.alpha = p.alpha + st.ns
    q.refresh_pi()


def lower_bound(p: Prior, q: Posterior, st: Stats) -> dict:
    """_calc_vl, _gaussianmixture.py:671-723.  Returns the seven terms and their sum."""
    K, D = q.m.shape
    dev = st.x_bar - q.m
    quad_x = (dev[:, None, :] @ q.e_lambda @ dev[:, :, None])[:, 0, 0]
    p_x = np.sum(st.ns * (q.e_ln_lambda_det - D / q.kappa
                          - (st.s * q.e_lambda).sum(axis=(1, 2)) - quad_x - D * LN_2PI)) / 2.0
    p_z = float((st.ns * q.e_ln_pi).sum())
    p_pi = p.ln_c_alpha + float(((p.alpha - 1.0) * q.e_ln_pi).sum())
    dm = q.m - p.m
    quad_m = (dm[:, None, :] @ q.e_lambda @ dm[:, :, None])[:, 0, 0]
    p_mu_lambda = np.sum(D * (np.log(p.kappa) - LN_2PI - p.kappa / q.kappa)
                         - p.kappa * quad_m + 2.0 * p.ln_b_w_nu
                         + (p.nu - D) * q.e_ln_lambda_det
                         - np.sum(p.w_inv * q.e_lambda, axis=(1, 2))) / 2.0
    q_z = -float(np.sum(xlogy(st.r, st.r)))
    q_pi = float(_dirichlet.entropy(q.alpha))
    q_mu_lambda = np.sum(D * (1.0 + LN_2PI - np.log(q.kappa)) - 2.0 * q.ln_b_w_nu
                         - (q.nu - D) * q.e_ln_lambda_det + q.nu * D) / 2.0
    terms = dict(p_x=float(p_x), p_z=p_z, p_pi=float(p_pi), p_mu_lambda=float(p_mu_lambda),
                 q_z=q_z, q_pi=q_pi, q_mu_lambda=float(q_mu_lambda))
    terms["vl"] = (terms["p_x"] + terms["p_z"] + terms["p_pi"] + terms["p_mu_lambda"]
                   + terms["q_z"] + terms["q_pi"] + terms["q_mu_lambda"])
    return terms


# --------------------------------------------------------------------------- restarts
def init_subsampling(x: np.ndarray, q: Posterior, rng: np.random.Generator) -> None:
    """_init_subsampling, _gaussianmixture.py:786-796.  Consumes ``rng`` exactly as the
    reference does: K calls of ``rng.choice(x, int(sqrt(N)), replace=False, axis=0, shuffle=False)``."""
    N, D = x.shape
    size = int(np.sqrt(N))
    for k in range(q.m.shape[0]):
        sub = rng.choice(x, size=size, replace=False, axis=0, shuffle=False)
        q.m[k] = sub.sum(axis=0) / size
        c = sub - q.m[k]
        q.w_inv[k] = c.T @ c / size * q.nu[k] + np.eye(D) * 1.0e-5
        q.w[k] = np.linalg.inv(q.w_inv[k])
    q.refresh_lambda()


def init_random_responsibility(x: np.ndarray, K: int, rng: np.random.Generator,
                               s_prev: np.ndarray | None = None) -> Stats:
    """_init_random_responsibility, _gaussianmixture.py:734-736."""
    r = rng.dirichlet(np.ones(K), x.shape[0])
    ns, x_bar, s = m_step_stats(x, r, s_prev)
    return Stats(np.zeros_like(r), r, ns, x_bar, s)


@dataclass
class RunResult:
    posterior: Posterior
    stats: Stats
    vl: float
    vl_trace: list = field(default_factory=list)    # per restart: [vl_init, vl_t0, vl_t1, ...]
    winner: int = -1
    converged_any: bool = False


def update_posterior(x: np.ndarray, p: Prior, q0: Posterior, rng: np.random.Generator,
                     max_itr: int = 100, num_init: int = 10, tolerance: float = 1.0e-8,
                     init_type: str = "subsampling") -> RunResult:
    """Driver, _gaussianmixture.py:802-896.

    ``q0`` is the posterior the model holds on entry; it is the value that is
    restored if ``num_init == 0`` (the reference's ``tmp_*`` snapshot, :838-844).
    ``x`` is used in the dtype it arrives in, like the reference (:829 discards the
    validator's cast); NumPy promotes float32 rows on the first subtraction."""
    x = x.reshape(-1, p.m.shape[1])
    K = p.m.shape[0]
    best = q0.copy()
    best_vl = 0.0
    never_converged = True
    traces, winner = [], -1
    s_carry = None                      # persistent self.s_mats across restarts
    for i in range(num_init):
        q = Posterior.from_prior(p)
        if init_type == "subsampling":
            init_subsampling(x, q, rng)
            st = data_pass(x, q, s_carry)
        elif init_type == "random_responsibility":
            st = init_random_responsibility(x, K, rng, s_carry)
        else:
            raise ValueError(f"init_type={init_type} is unsupported. This function supports only "
                             '"subsampling" and "random_responsibility"')
        vl = lower_bound(p, q, st)["vl"]
        trace = [vl]
        for _t in range(max_itr):
            vl_before = vl
            update_q_mu_lambda(p, q, st)
            update_q_pi(p, q, st)
            st = data_pass(x, q, st.s)
            vl = lower_bound(p, q, st)["vl"]
            trace.append(vl)
            with np.errstate(divide="ignore", invalid="ignore"):
                if np.abs((vl - vl_before) / vl_before) < tolerance:
                    never_converged = False
                    break
        s_carry = st.s
        traces.append(trace)
        if i == 0 or vl > best_vl:
            best_vl, best, winner = vl, q.copy(), i
    best.refresh_pi()
    best.refresh_lambda()
    st = data_pass(x, best, s_carry)      # :895 — leaves r/ns/x_bar/s consistent with the winner
    return RunResult(best, st, float(vl) if num_init else 0.0, traces, winner, not never_converged)


# --------------------------------------------------------------------------- read-outs
def predictive_params(q: Posterior) -> dict:
    """calc_pred_dist, _gaussianmixture.py:1064-1070."""
    D = q.m.shape[1]
    p_nus = q.nu - D + 1
    return dict(p_pi_vec=q.alpha / q.alpha.sum(), p_mu_vecs=q.m.copy(), p_nus=p_nus,
                p_lambda_mats=(q.kappa * p_nus / (q.kappa + 1))[:, None, None] * q.w)


def estimate_params(q: Posterior, loss: str = "squared"):
    """estimate_params for the array-valued losses, _gaussianmixture.py:930-947.
    Follows the code at :935 (``sum(alpha) - c_degree``), not the textbook ``- K``."""
    K, D = q.m.shape
    if loss == "squared":
        return q.alpha / q.alpha.sum(), q.m, q.e_lambda
    if loss == "0-1":
        pi_hat = np.full(K, np.nan)
        if np.all(q.alpha > 1):
            pi_hat = (q.alpha - 1) / (q.alpha.sum() - D)
        lam = np.full((K, D, D), np.nan)
        for k in range(K):
            if q.nu[k] >= D + 1:
                lam[k] = (q.nu[k] - D - 1) * q.w[k]
        return pi_hat, q.m, lam
    raise ValueError(loss)


def estimate_latent_vars(x: np.ndarray, q: Posterior, loss: str = "0-1") -> np.ndarray:
    """estimate_latent_vars, _gaussianmixture.py:1178-1193."""
    _, r = e_step(x.reshape(-1, q.m.shape[1]), q)
    if loss in ("squared", "KL"):
        return r
    if loss == "0-1":
        return np.eye(q.m.shape[0], dtype=int)[np.argmax(r, axis=1)]
    raise ValueError(loss)


# --------------------------------------------------------------------------- workloads
def synth_gmm(K: int, D: int, N: int, dtype=np.float64, seed: int = 20250711,
              chunk: int = 1 << 20, spread: float = 2.0, weights_alpha: float | None = None,
              scale_range: tuple | None = None) -> np.ndarray:
    """The benchmark's synthetic sample matrix:
    ``mu = 2*standard_normal((K,D))``, ``z = integers(0,K,N)``,
    ``x = mu[z] + standard_normal((N,D))`` drawn in fixed 2**20-row chunks.
    ``spread`` scales the means (2 = the benchmark; 0.3 = heavily overlapping clusters, the hard workload).
    Off the benchmark's recipe (round 5; the defaults draw nothing extra, so the streams of the older fixtures stand):
    ``weights_alpha``: mixing weights ~ Dirichlet(weights_alpha) instead of equal ones (0.3: a few big clusters, many tiny);
    ``scale_range = (lo, hi)``: every cluster gets its own per-feature standard deviations, log-uniform in [lo, hi]
    (anisotropic, axis-aligned covariances) instead of the identity."""
    rng = np.random.default_rng(seed)
    mu = spread * rng.standard_normal((K, D))
    w = rng.dirichlet(np.full(K, float(weights_alpha))) if weights_alpha is not None else None
    sc = (np.exp(rng.uniform(np.log(scale_range[0]), np.log(scale_range[1]), (K, D)))
          if scale_range is not None else None)
    x = np.empty((N, D), dtype=dtype)
    for lo in range(0, N, chunk):
        hi = min(N, lo + chunk)
        z = rng.integers(0, K, hi - lo) if w is None else rng.choice(K, size=hi - lo, p=w)
        e = rng.standard_normal((hi - lo, D))
        x[lo:hi] = (mu[z] + (e if sc is None else e * sc[z])).astype(dtype)
    return x
