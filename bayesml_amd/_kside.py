"""K-sized closed-form updates of the Dirichlet x Normal-Wishart posterior, in fp64 torch.

Device-agnostic (runs where its input tensors live: the GPU inside ``update_posterior``, the CPU
in the unit tests).  These are the K-sized steps of the reference that bracket the data pass:

  moments_from_stats      _calc_n_x_bar_s's per-component finishing   _gaussianmixture.py:729-732
  update_q                _update_q_mu_lambda + _update_q_pi          _gaussianmixture.py:758-770, 741-743
  features                _calc_q_pi_features, _calc_q_lambda_features  :738-739, :745-756
  lower_bound             _calc_vl                                    :671-723
  subsample_moments_init  _init_subsampling                           :786-796

Unlike the reference, ``W = inv(W^-1)`` is obtained from ONE Cholesky factorisation
W^-1 = G G^T per component, which also yields ln det W^-1 = 2 sum ln diag G and the whitening
factor U = sqrt(nu) G^-1 (U^T U = nu W = E[Lambda]) that the E-step kernel consumes.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np

import torch

LN_2PI = math.log(2.0 * math.pi)
LN_2 = math.log(2.0)
LN_PI = math.log(math.pi)


@dataclass
class PriorT:
    alpha: torch.Tensor     # [K]
    m: torch.Tensor         # [K, D]
    kappa: torch.Tensor     # [K]
    nu: torch.Tensor        # [K]
    w_inv: torch.Tensor     # [K, D, D]
    ln_c_alpha: float       # _gaussianmixture.py:662
    ln_b_w_nu: torch.Tensor  # [K]  :663-669


@dataclass
class PostT:
    alpha: torch.Tensor
    m: torch.Tensor
    kappa: torch.Tensor
    nu: torch.Tensor
    w_inv: torch.Tensor
    # derived (features())
    w: torch.Tensor = None
    u: torch.Tensor = None               # [K, D, D] lower triangular, u^T u = nu * w
    e_ln_pi: torch.Tensor = None
    e_ln_lambda_det: torch.Tensor = None
    ln_b_w_nu: torch.Tensor = None
    c: torch.Tensor = None               # E-step constant per component

    def clone(self) -> "PostT":
        return PostT(*(None if t is None else t.clone() for t in (
            self.alpha, self.m, self.kappa, self.nu, self.w_inv, self.w, self.u, self.e_ln_pi,
            self.e_ln_lambda_det, self.ln_b_w_nu, self.c)))


def _half_lgamma_sum(nu: torch.Tensor, D: int) -> torch.Tensor:
    d = torch.arange(D, dtype=nu.dtype, device=nu.device)
    return torch.lgamma((nu[:, None] - d) / 2.0).sum(dim=1)


def _half_digamma_sum(nu: torch.Tensor, D: int) -> torch.Tensor:
    d = torch.arange(D, dtype=nu.dtype, device=nu.device)
    return torch.digamma((nu[:, None] - d) / 2.0).sum(dim=1)


def ln_wishart_b(logdet_w_inv: torch.Tensor, nu: torch.Tensor, D: int) -> torch.Tensor:
    """ln B(W, nu) from ln det W^-1 (_gaussianmixture.py:750-756; the prior form :663-669 is the
    same expression with ln det W = -ln det W^-1)."""
    return (nu * logdet_w_inv - nu * D * LN_2 - D * (D - 1) / 2.0 * LN_PI - 2.0 * _half_lgamma_sum(nu, D)) / 2.0


def features(q: PostT) -> PostT:
    """Refresh every derived quantity of ``q`` from (alpha, m, kappa, nu, w_inv)."""
    q.e_ln_pi = torch.digamma(q.alpha) - torch.digamma(q.alpha.sum())
    niw_features(q)
    q.c = q.e_ln_pi + q.c
    return q


def niw_features(q):
    """Normal-Wishart part shared by the GMM and the HMM: w, u, E[ln det Lambda], ln B, and the emission
    constant c = (E[ln det Lambda] - D ln 2pi - D/kappa)/2 (the HMM's _calc_rho constant,
    _hiddenmarkovnormal.py:989-993; the GMM adds E[ln pi] to it)."""
    K, D = q.m.shape
    if q.w_inv.is_cuda and D <= 128:
        # the library's LDS-resident factorisation (csrc/kside.hip): plain kernels, capturable in a hipGraph
        from ._engine import kside_factor
        g, g_inv, logdet = kside_factor(q.w_inv)
    else:
        g, g_inv, logdet = _factor_checked(q.w_inv)
    q.w = g_inv.transpose(1, 2) @ g_inv
    q.w = 0.5 * (q.w + q.w.transpose(1, 2))
    q.u = torch.sqrt(q.nu)[:, None, None] * g_inv
    q.u_inv = g / torch.sqrt(q.nu)[:, None, None]          # u^-1 (lower triangular), for drift()
    q.e_ln_lambda_det = _half_digamma_sum(q.nu, D) + D * LN_2 - logdet
    q.ln_b_w_nu = ln_wishart_b(logdet, q.nu, D)
    q.c = (q.e_ln_lambda_det - D * LN_2PI - D / q.kappa) / 2.0
    return q


def _factor_checked(w_inv):
    """(G, G^-1, ln det) of a batch of SPD matrices W^-1 = G G^T where the library's kernel does not reach (CPU tensors,
    the GPU past 128 features): the framework's batched Cholesky and triangular solve, VERIFIED - the residuals of
    G G^T - W^-1 and G^-1 G - I are formed next to them and one flag is read back - and redone with numpy on the host for
    the matrices that fail.  tools/probe_torch_linalg.py found the routines sound at every order from 129 to 260 with 24 and
    64 matrices, but at order 65 they are not (see spd_inverse), and nothing says which other (order, batch) pairs share
    that fate.  NaN / non-SPD inputs are not failures of the routine: they propagate as before."""
    K, D, _ = w_inv.shape
    g, info = torch.linalg.cholesky_ex(w_inv)                 # NaNs propagate instead of raising, like inv()
    eye = torch.eye(D, dtype=w_inv.dtype, device=w_inv.device).expand(K, D, D)
    g_inv = torch.linalg.solve_triangular(g, eye, upper=False)
    scale = w_inv.abs().amax(dim=(1, 2)).clamp_min(torch.finfo(w_inv.dtype).tiny)
    off = ((g @ g.transpose(1, 2) - w_inv).abs().amax(dim=(1, 2)) / scale > 1e-10) | ((g_inv @ g - eye).abs().amax(dim=(1, 2)) > 1e-7)
    off = off & (info == 0) & torch.isfinite(w_inv).all(dim=2).all(dim=1)
    if bool(off.any()):
        host = w_inv.detach().cpu().numpy()
        for k in torch.nonzero(off).flatten().tolist():
            try:
                gk = np.linalg.cholesky(host[k])
            except np.linalg.LinAlgError:
                continue
            g[k] = torch.as_tensor(gk, dtype=w_inv.dtype, device=w_inv.device)
            g_inv[k] = torch.as_tensor(np.linalg.inv(gk), dtype=w_inv.dtype, device=w_inv.device).tril()
    logdet = 2.0 * torch.log(torch.diagonal(g, dim1=1, dim2=2)).sum(dim=1)
    return g, g_inv, logdet


def spd_inverse(w, device):
    """(W^-1 symmetrised, ln det W^-1) of a batch of symmetric positive definite matrices ``w`` (array-like ``[K, D, D]``)
    as float64 tensors on ``device``.

    On the GPU up to D = 128 this is the library's own LDS-resident factorisation (W = G G^T, W^-1 = G^-T G^-1,
    csrc/kside.hip), elsewhere numpy on the host - never the framework's batched GPU inverse: on this image
    ``torch.linalg.inv`` and ``torch.linalg.solve_triangular`` return wrong entries (an O(1) error in the last diagonal
    element of some matrices, different from run to run) for batches of 24 or more 65 x 65 float64 matrices
    (tools/probe_torch_linalg.py maps the orders 2..260; found by tests/fuzz_sparse.py in round 6, where it made fits
    with c_degree = 65 irreproducible)."""
    dev = torch.device(device)
    D = int(np.shape(w)[-1])
    if dev.type == "cuda" and D <= 128:
        from ._engine import kside_factor
        _g, g_inv, logdet_w = kside_factor(torch.as_tensor(w, dtype=torch.float64, device=dev).clone())
        w_inv = g_inv.transpose(1, 2) @ g_inv
        return 0.5 * (w_inv + w_inv.transpose(1, 2)), -logdet_w
    host = np.asarray(w.detach().cpu().numpy() if isinstance(w, torch.Tensor) else w, dtype=np.float64)
    w_inv = np.linalg.inv(host)
    w_inv = 0.5 * (w_inv + np.swapaxes(w_inv, 1, 2))
    t = lambda a: torch.as_tensor(a, dtype=torch.float64, device=dev)             # noqa: E731
    return t(w_inv), t(-np.linalg.slogdet(host)[1])


def prior_from_numpy(alpha, m, kappa, nu, w, device) -> PriorT:
    t = lambda a: torch.as_tensor(a, dtype=torch.float64, device=device).clone()   # noqa: E731
    alpha, m, kappa, nu = t(alpha), t(m), t(kappa), t(nu)
    D = m.shape[1]
    w_inv, logdet_w_inv = spd_inverse(w, device)
    ln_c = float(torch.lgamma(alpha.sum()) - torch.lgamma(alpha).sum())
    return PriorT(alpha, m, kappa, nu, w_inv, ln_c, ln_wishart_b(logdet_w_inv, nu, D))


def post_from_prior(p: PriorT) -> PostT:
    """reset_hn_params (reference base.py:260-267)."""
    return features(PostT(p.alpha.clone(), p.m.clone(), p.kappa.clone(), p.nu.clone(), p.w_inv.clone()))


def _norm2_upper(a, squarings: int):
    """Upper bound of the spectral norm of every matrix of the batch ``a``: ||A||_2^2 = lambda_max(G), G = A^T A, is
    bounded from above by ||G^(2^s)||_F^(1/2^s): s repeated squarings with the Frobenius norms divided out (their
    logarithms summed with weights 2^-i), at most D^(1/2^(s+1)) above the true norm (1.01 at D = 128, s = 8)."""
    g = a.transpose(1, 2) @ a
    tiny = torch.finfo(g.dtype).tiny
    f = torch.linalg.matrix_norm(g).clamp_min(tiny)
    log_lmax = torch.log(f)
    g = g / f[:, None, None]
    w = 0.5
    for _ in range(squarings):
        g = g @ g
        f = torch.linalg.matrix_norm(g).clamp_min(tiny)
        log_lmax = log_lmax + w * torch.log(f)
        g = g / f[:, None, None]
        w *= 0.5
    return torch.exp(0.5 * log_lmax)


def drift(q_old, q_new, squarings: int = 8, squarings_big: int = 6):
    """(gamma, delta, big_gamma) of gmmvb_set_drift for the parameter update q_old -> q_new: per component
    gamma <= sigma_min(u_new u_old^-1), big_gamma >= sigma_max(u_new u_old^-1), delta >= ||u_new (m_new - m_old)||, so that
    gamma ||u_old (x - m_old)|| - delta <= ||u_new (x - m_new)|| <= big_gamma ||u_old (x - m_old)|| + delta for every x.

    sigma_min(u_new u_old^-1) = 1 / ||A||_2 with A = u_old u_new^-1 (u^-1 = G / sqrt(nu) is kept by niw_features, so
    no triangular solve); sigma_max = ||u_new u_old^-1||_2.  Both norms by _norm2_upper (rigorous, a per cent or so
    loose).  (Bounds through A = alpha I + E were tried: the triangle inequality costs 15 % in the first iterations,
    where the singular values of A spread from 0.9 to 1.4.)  big_gamma only sets how far below the carried best value
    the E-step has to look, so it gets fewer squarings (4 % loose)."""
    if q_old.u.is_cuda and q_old.m.shape[1] <= 128:
        from ._engine import kside_drift             # the library's one-launch kernel (csrc/kside.hip)
        return kside_drift(q_old, q_new, squarings, squarings_big)
    gamma = (1.0 - 1e-9) / _norm2_upper(q_old.u @ q_new.u_inv, squarings)
    b = q_new.u @ q_old.u_inv
    big = (1.0 + 1e-9) * _norm2_upper(b, squarings_big)
    # u_new u_old^-1 = I + E: 1 -/+ ||E|| bounds its extreme singular values, tightly once the components hardly move
    e = _norm2_upper(b - torch.eye(b.shape[-1], dtype=b.dtype, device=b.device), squarings) * (1.0 + 1e-9)
    gamma = torch.maximum(gamma, (1.0 - e) * (1.0 - 1e-9))
    big = torch.minimum(big, (1.0 + e) * (1.0 + 1e-9))
    d = (q_new.u @ (q_new.m - q_old.m)[:, :, None])[:, :, 0]
    delta = torch.linalg.vector_norm(d, dim=1) * (1.0 + 1e-9)
    # anything non-finite: no information (gamma = 0 makes every carried bound the trivial one, big_gamma = inf leaves
    # no lower bound of the best value: the row is evaluated in full)
    bad = ~(torch.isfinite(gamma) & torch.isfinite(delta) & torch.isfinite(big))
    gamma = torch.where(bad, torch.zeros_like(gamma), gamma)
    delta = torch.where(bad, torch.zeros_like(delta), delta)
    big = torch.where(bad, torch.full_like(big, float("inf")), big)
    return gamma, delta, big


def moments_from_stats(ns, a, B, pivot, s_prev):
    """Engine statistics (about ``pivot``) -> the reference's (x_bar_vecs, s_mats).

    For ns[k] == 0 the reference leaves x_bar_vecs[k] as the raw zero sum and does not touch
    s_mats[k] (_gaussianmixture.py:729); ``s_prev`` carries that previous value (zeros initially)."""
    pos = ns > 0
    safe = torch.where(pos, ns, torch.ones_like(ns))
    abar = a / safe[:, None]
    x_bar = torch.where(pos[:, None], pivot[None, :] + abar, torch.zeros_like(abar))
    s = B / safe[:, None, None] - abar[:, :, None] * abar[:, None, :]
    s = torch.where(pos[:, None, None], s, s_prev)
    return x_bar, s


def update_q(p: PriorT, ns, x_bar, s) -> PostT:
    """_update_q_mu_lambda (:758-770) + _update_q_pi (:741-743), then features()."""
    kappa = p.kappa + ns
    m = (p.kappa[:, None] * p.m + ns[:, None] * x_bar) / kappa[:, None]
    nu = p.nu + ns
    dev = x_bar - p.m
    w_inv = (p.w_inv + ns[:, None, None] * s
             + (p.kappa * ns / kappa)[:, None, None] * (dev[:, :, None] * dev[:, None, :]))
    return features(PostT(p.alpha + ns, m, kappa, nu, w_inv))


def dirichlet_entropy(alpha: torch.Tensor) -> torch.Tensor:
    """Differential entropy of Dirichlet(alpha) — what scipy.stats.dirichlet.entropy returns
    (used at _gaussianmixture.py:707)."""
    a0 = alpha.sum()
    K = alpha.numel()
    ln_b = torch.lgamma(alpha).sum() - torch.lgamma(a0)
    return ln_b + (a0 - K) * torch.digamma(a0) - ((alpha - 1.0) * torch.digamma(alpha)).sum()


def lower_bound(p: PriorT, q: PostT, ns, x_bar, s, sum_r_ln_r) -> dict:
    """_calc_vl (_gaussianmixture.py:671-723).  ``sum_r_ln_r`` = sum_nk r ln r from the M-step kernel.
    Returns 0-dim tensors (no host sync here)."""
    K, D = q.m.shape
    e_lambda = q.nu[:, None, None] * q.w

    def quad(v):
        return torch.einsum("ki,kij,kj->k", v, e_lambda, v)

    p_x = (ns * (q.e_ln_lambda_det - D / q.kappa - (s * e_lambda).sum(dim=(1, 2)) - quad(x_bar - q.m)
                 - D * LN_2PI)).sum() / 2.0
    p_z = (ns * q.e_ln_pi).sum()
    p_pi = p.ln_c_alpha + ((p.alpha - 1.0) * q.e_ln_pi).sum()
    p_mu_lambda = (D * (torch.log(p.kappa) - LN_2PI - p.kappa / q.kappa) - p.kappa * quad(q.m - p.m)
                   + 2.0 * p.ln_b_w_nu + (p.nu - D) * q.e_ln_lambda_det
                   - (p.w_inv * e_lambda).sum(dim=(1, 2))).sum() / 2.0
    q_z = -sum_r_ln_r
    q_pi = dirichlet_entropy(q.alpha)
    q_mu_lambda = (D * (1.0 + LN_2PI - torch.log(q.kappa)) - 2.0 * q.ln_b_w_nu
                   - (q.nu - D) * q.e_ln_lambda_det + q.nu * D).sum() / 2.0
    terms = dict(p_x=p_x, p_z=p_z, p_pi=p_pi, p_mu_lambda=p_mu_lambda, q_z=q_z, q_pi=q_pi, q_mu_lambda=q_mu_lambda)
    terms["vl"] = p_x + p_z + p_pi + p_mu_lambda + q_z + q_pi + q_mu_lambda
    return terms


def subsample_moments_init(q, cnt, a, B, pivot, refresh=None):
    """_init_subsampling (_gaussianmixture.py:786-796) from per-component raw moments of the drawn
    rows about ``pivot``: cnt (scalar subsample size), a [K, D] = sum (x - pivot), B [K, D, D]."""
    D = q.m.shape[1]
    abar = a / cnt
    q.m = pivot[None, :] + abar
    cov = B / cnt - abar[:, :, None] * abar[:, None, :]
    eye = torch.eye(D, dtype=cov.dtype, device=cov.device)
    q.w_inv = cov * q.nu[:, None, None] + eye * 1.0e-5
    return (refresh or features)(q)


# ----------------------------------------------------------------------------------------------- HMM
@dataclass
class HmmPriorT:
    eta: torch.Tensor        # [K]
    zeta: torch.Tensor       # [K, K]
    m: torch.Tensor
    kappa: torch.Tensor
    nu: torch.Tensor
    w_inv: torch.Tensor
    ln_c_eta: float          # _hiddenmarkovnormal.py:851
    ln_c_zeta_sum: float     # :852
    ln_b_w_nu: torch.Tensor  # :853-859


@dataclass
class HmmPostT:
    eta: torch.Tensor
    zeta: torch.Tensor
    m: torch.Tensor
    kappa: torch.Tensor
    nu: torch.Tensor
    w_inv: torch.Tensor
    w: torch.Tensor = None
    u: torch.Tensor = None
    e_ln_lambda_det: torch.Tensor = None
    ln_b_w_nu: torch.Tensor = None
    c: torch.Tensor = None              # emission constant
    ln_pi_tilde: torch.Tensor = None    # :862
    pi_tilde: torch.Tensor = None       # :863
    ln_a_tilde: torch.Tensor = None     # :866
    a_tilde: torch.Tensor = None        # :867  (ONE global max)
    ln_c_zeta_sum: torch.Tensor = None  # :868

    def clone(self) -> "HmmPostT":
        return hmm_features(HmmPostT(*(t.clone() for t in (self.eta, self.zeta, self.m, self.kappa, self.nu, self.w_inv))))


def _ln_c_rows(z: torch.Tensor) -> torch.Tensor:
    return (torch.lgamma(z.sum(dim=-1)) - torch.lgamma(z).sum(dim=-1)).sum()


def hmm_features(q: HmmPostT) -> HmmPostT:
    """_calc_q_pi_features :861-863, _calc_q_a_features :865-868, _calc_q_lambda_features :870-881."""
    q.ln_pi_tilde = torch.digamma(q.eta) - torch.digamma(q.eta.sum())
    q.pi_tilde = torch.exp(q.ln_pi_tilde - q.ln_pi_tilde.max())
    q.ln_a_tilde = torch.digamma(q.zeta) - torch.digamma(q.zeta.sum(dim=1, keepdim=True))
    q.a_tilde = torch.exp(q.ln_a_tilde - q.ln_a_tilde.max())
    q.ln_c_zeta_sum = _ln_c_rows(q.zeta)
    return niw_features(q)


def hmm_prior_from_numpy(eta, zeta, m, kappa, nu, w, device) -> HmmPriorT:
    t = lambda a: torch.as_tensor(a, dtype=torch.float64, device=device).clone()   # noqa: E731
    eta, zeta, m, kappa, nu = t(eta), t(zeta), t(m), t(kappa), t(nu)
    D = m.shape[1]
    w_inv, logdet_w_inv = spd_inverse(w, device)
    return HmmPriorT(eta, zeta, m, kappa, nu, w_inv, float(torch.lgamma(eta.sum()) - torch.lgamma(eta).sum()),
                     float(_ln_c_rows(zeta)), ln_wishart_b(logdet_w_inv, nu, D))


def hmm_post_from_prior(p: HmmPriorT) -> HmmPostT:
    return hmm_features(HmmPostT(p.eta.clone(), p.zeta.clone(), p.m.clone(), p.kappa.clone(), p.nu.clone(),
                                 p.w_inv.clone()))


def hmm_update_q(p: HmmPriorT, ns, ms, x_bar, s) -> HmmPostT:
    """_update_q_mu_lambda :966-978, _update_q_pi :980-982 (eta += ns: the code, not the docstring's
    gamma_1), _update_q_a :984-986."""
    kappa = p.kappa + ns
    m = (p.kappa[:, None] * p.m + ns[:, None] * x_bar) / kappa[:, None]
    dev = x_bar - p.m
    w_inv = (p.w_inv + ns[:, None, None] * s
             + (p.kappa * ns / kappa)[:, None, None] * (dev[:, :, None] * dev[:, None, :]))
    return hmm_features(HmmPostT(p.eta + ns, p.zeta + ms, m, kappa, p.nu + ns, w_inv))


def sum_gamma_ln_rho(q, ns, x_bar, s) -> torch.Tensor:
    """sum_tk gamma_tk ln rho_tk (first term of -E[ln q(Z)], _hiddenmarkovnormal.py:905) from the pass's moments instead of
    the N x K arrays: ln rho_tk = c_k - (x_t - m_k)^T nu_k W_k (x_t - m_k) / 2 (ref:989-996, ``q`` = the parameters the
    emission was evaluated under), so the gamma-weighted sum over t is
    ns_k (c_k - ((s_k o nu_k W_k).sum() + (x_bar_k - m_k)^T nu_k W_k (x_bar_k - m_k)) / 2) - the expression the reference
    uses for E[ln p(x|z)] (ref:871-877).  Used when the emission went straight into the forward-backward buffers and no
    ln rho array exists (``DataPass.emission_target``)."""
    e_lambda = q.nu[:, None, None] * q.w
    dev = x_bar - q.m
    quad = torch.einsum("ki,kij,kj->k", dev, e_lambda, dev)
    per = q.c - 0.5 * ((s * e_lambda).sum(dim=(1, 2)) + quad)
    return torch.where(ns > 0, ns * per, torch.zeros_like(ns)).sum()


def hmm_lower_bound(p: HmmPriorT, q: HmmPostT, ns, ms, x_bar, s, gamma0, sum_gamma_ln_rho, sum_ln_c) -> dict:
    """_calc_vl, _hiddenmarkovnormal.py:883-943 (nine terms)."""
    K, D = q.m.shape
    e_lambda = q.nu[:, None, None] * q.w

    def quad(v):
        return torch.einsum("ki,kij,kj->k", v, e_lambda, v)

    p_x = (ns * (q.e_ln_lambda_det - D / q.kappa - (s * e_lambda).sum(dim=(1, 2)) - quad(x_bar - q.m)
                 - D * LN_2PI)).sum() / 2.0
    p_z = (gamma0 * q.ln_pi_tilde).sum() + (ms * q.ln_a_tilde).sum()
    p_pi = p.ln_c_eta + ((p.eta - 1.0) * q.ln_pi_tilde).sum()
    p_a = p.ln_c_zeta_sum + ((p.zeta - 1.0) * q.ln_a_tilde).sum()
    p_mu_lambda = (D * (torch.log(p.kappa) - LN_2PI - p.kappa / q.kappa) - p.kappa * quad(q.m - p.m)
                   + 2.0 * p.ln_b_w_nu + (p.nu - D) * q.e_ln_lambda_det
                   - (p.w_inv * e_lambda).sum(dim=(1, 2))).sum() / 2.0
    q_z = (-sum_gamma_ln_rho - (ms * (q.ln_a_tilde - q.ln_a_tilde.max())).sum()
           - (gamma0 * (q.ln_pi_tilde - q.ln_pi_tilde.max())).sum() + sum_ln_c)
    q_pi = dirichlet_entropy(q.eta)
    q_a = -q.ln_c_zeta_sum - ((q.zeta - 1.0) * q.ln_a_tilde).sum()
    q_mu_lambda = (D * (1.0 + LN_2PI - torch.log(q.kappa)) - 2.0 * q.ln_b_w_nu
                   - (q.nu - D) * q.e_ln_lambda_det + q.nu * D).sum() / 2.0
    t = dict(p_x=p_x, p_z=p_z, p_pi=p_pi, p_a=p_a, p_mu_lambda=p_mu_lambda, q_z=q_z, q_pi=q_pi, q_a=q_a,
             q_mu_lambda=q_mu_lambda)
    t["vl"] = p_x + p_z + p_pi + p_a + p_mu_lambda + q_z + q_pi + q_a + q_mu_lambda
    return t


HMM_TERM_KEYS = ("p_x", "p_z", "p_pi", "p_a", "p_mu_lambda", "q_z", "q_pi", "q_a", "q_mu_lambda", "vl")
_HMM_POST_FIELDS = ("eta", "zeta", "m", "kappa", "nu", "w_inv", "w", "u", "u_inv", "e_ln_lambda_det", "ln_b_w_nu", "c",
                    "ln_pi_tilde", "pi_tilde", "ln_a_tilde", "a_tilde", "ln_c_zeta_sum")


class HmmKStepper:
    """The K-sized half of one VB iteration of hiddenmarkovnormal.LearnModel as ONE unit on the GPU (round 5; the mixture's
    twin is KStepper below):

        statistics block + forward-backward summary -> moments (x_bar, S)                  ref :837-845
                                                     -> sum gamma ln rho in closed form       ref :905 via :871-877
                                                     -> lower bound under q (ten terms)       ref :883-943
                                                     -> q' = closed-form update + features    ref :966-986, :861-881

    About a hundred small torch kernels (the factorisation is the library's own, ``kside_factor``: capturable), captured
    once in a hipGraph on the second call and replayed: at config 5 the launches were 0.6 ms of a 9.8-ms iteration, below
    1e5 steps most of it.  The data pass writes ``stats`` (gmmvb_mstep) and ``fb`` (hmmvb_forward_backward) in place; ``q``
    and ``q_next`` are fixed buffer sets; ``scal`` is the iteration's one device-to-host copy.  Eager on the CPU (host-logic
    tests) and with BAYESML_AMD_KSIDE_GRAPH=0."""

    def __init__(self, prior: HmmPriorT, pivot: torch.Tensor, stats_len: int):
        import os
        K, D = prior.m.shape
        dev = prior.m.device
        self.prior, self.pivot, self.K, self.D = prior, pivot, K, D
        self.stats = torch.zeros(stats_len, dtype=torch.float64, device=dev)
        self.fb = torch.zeros(K * K + 2 * K + 1, dtype=torch.float64, device=dev)      # [ms | gamma_0 | gamma_last | sum ln c]
        self.h_scale = torch.ones((), dtype=torch.float64, device=dev)     # 0: the pass had no emission (random-responsibility start)
        self.s_prev = torch.zeros(K, D, D, dtype=torch.float64, device=dev)
        self.q = self._buffers(hmm_post_from_prior(prior))
        self.q_next = self._buffers(self.q)
        self.ns = torch.zeros(K, dtype=torch.float64, device=dev)
        self.x_bar = torch.zeros(K, D, dtype=torch.float64, device=dev)
        self.s = torch.zeros(K, D, D, dtype=torch.float64, device=dev)
        self.scal = torch.zeros(len(HMM_TERM_KEYS), dtype=torch.float64, device=dev)
        self._graph = None
        self._calls = 0
        self._use_graph = dev.type == "cuda" and D <= 128 and os.environ.get("BAYESML_AMD_KSIDE_GRAPH", "1") != "0"
        # The Normal-Wishart half of the step is the mixture's (ref:966-978 = _gaussianmixture.py:758-770; E[ln p(x|z)], the
        # mu/Lambda terms of the lower bound likewise): on the GPU it runs in the library's one-launch-per-component kernel
        # (gmmvb_kside_step, csrc/kside.hip) on views of the HMM posterior's own buffers - the mixture's Dirichlet fields are
        # dummies whose results are ignored - and the K- and K x K-sized Dirichlet part (eta, zeta) in one launch of its own
        # (hmmvb_kside_dirichlet): five kernels instead of ~100 torch ones, which were most of a short sequence's iteration.
        self._fused = dev.type == "cuda" and D <= 128 and K <= 256 and os.environ.get("BAYESML_AMD_KSIDE_FUSED", "1") != "0"
        if self._fused:
            from types import SimpleNamespace
            from . import _engine
            self._eng = _engine
            z = lambda *shape: torch.zeros(*shape, dtype=torch.float64, device=dev)   # noqa: E731
            self._prior_alpha = torch.ones(K, dtype=torch.float64, device=dev)
            self._prior_v = _engine.prior_view(SimpleNamespace(alpha=self._prior_alpha, m=prior.m, kappa=prior.kappa, nu=prior.nu,
                                                               w_inv=prior.w_inv, ln_b_w_nu=prior.ln_b_w_nu, ln_c_alpha=0.0))

            def mix_view(q):
                extra = SimpleNamespace(alpha=torch.ones(K, dtype=torch.float64, device=dev), e_ln_pi=z(K), c=z(K))
                ns_ = SimpleNamespace(alpha=extra.alpha, m=q.m, kappa=q.kappa, nu=q.nu, w_inv=q.w_inv, w=q.w, u=q.u, u_inv=q.u_inv,
                                      e_ln_pi=extra.e_ln_pi, e_ln_lambda_det=q.e_ln_lambda_det, ln_b_w_nu=q.ln_b_w_nu, c=extra.c)
                return ns_

            self._mix = {id(self.q): mix_view(self.q), id(self.q_next): mix_view(self.q_next)}
            self._scal_g = z(9)
            self._ln_c0 = (float(prior.ln_c_eta), float(prior.ln_c_zeta_sum))      # (host floats once: no sync per step)
            self._scratch = z(13 * K)
            self._dummy = z(K)

    @staticmethod
    def _buffers(src):
        q = HmmPostT(*(getattr(src, f).clone() for f in _HMM_POST_FIELDS[:6]))
        for f in _HMM_POST_FIELDS[6:]:
            setattr(q, f, torch.as_tensor(getattr(src, f)).clone())
        return q

    @staticmethod
    def _copy(dst, src):
        for f in _HMM_POST_FIELDS:
            getattr(dst, f).copy_(getattr(src, f))

    def load(self, q: HmmPostT):
        self._copy(self.q, q)

    def fb_views(self):
        K = self.K
        return self.fb[:K * K].view(K, K), self.fb[K * K:K * K + K], self.fb[K * K + K:K * K + 2 * K], self.fb[K * K + 2 * K]

    def _body_fused(self):
        """The step with the Normal-Wishart half in gmmvb_kside_step (see __init__)."""
        e, q, qn, p = self._eng, self.q, self.q_next, self.prior
        mq, mqn = self._mix[id(q)], self._mix[id(qn)]
        e.kside_step(self.K, self.D, self._prior_v, e.post_view(mq), e.post_view(mqn), self.stats, self.pivot, self.s_prev, self.ns,
                     self.x_bar, self.s, False, self._dummy, self._dummy, self._dummy, self._scal_g, self._scratch)
        # (the kernel has written ns, x_bar, s, s_prev and q_next's m, kappa, nu, w_inv, w, u, u_inv, E[ln det Lambda], ln B -
        # and the mixture's c' = E[ln pi'] + the emission constant: the HMM's is the second term)
        torch.sub(mqn.c, mqn.e_ln_pi, out=qn.c)
        # the Dirichlet half - the lower bound's eta / zeta terms under q, eta', zeta' and their features - in one launch
        e.hmm_kside_dirichlet(self.K, p, self._ln_c0[0], self._ln_c0[1], q, qn, self.fb, self.ns, self._scal_g, self.h_scale, self.scal)

    def _body(self):
        if self._fused:
            return self._body_fused()
        K, D = self.K, self.D
        st = self.stats
        ns = st[:K]
        a = st[2 * K:2 * K + K * D].view(K, D)
        B = st[2 * K + K * D:].view(K, D, D)
        ms, g0, _gl, sum_ln_c = self.fb_views()
        x_bar, s = moments_from_stats(ns, a, B, self.pivot, self.s_prev)
        sg = sum_gamma_ln_rho(self.q, ns, x_bar, s) * self.h_scale
        terms = hmm_lower_bound(self.prior, self.q, ns, ms, x_bar, s, g0, sg, sum_ln_c)
        qn = hmm_update_q(self.prior, ns, ms, x_bar, s)
        self._copy(self.q_next, qn)
        self.ns.copy_(ns)
        self.x_bar.copy_(x_bar)
        self.s.copy_(s)
        self.s_prev.copy_(s)
        self.scal.copy_(torch.stack([terms[k].reshape(()) for k in HMM_TERM_KEYS]))

    def step(self):
        """Run the K-side on ``stats`` / ``fb``.  Results: ns, x_bar, s, q_next, scal."""
        self._calls += 1
        if self._use_graph and self._graph is None and self._calls >= 2:
            try:                                    # (the first call ran eagerly: library / BLAS warm-up)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self._body()
                self._graph = g
            except Exception as e:                  # noqa: BLE001 - capture is an optimisation, never a requirement
                import warnings
                warnings.warn(f"bayesml_amd: HMM K-side hipGraph capture failed ({type(e).__name__}: {e}); running eagerly")
                self._use_graph = False
                torch.cuda.synchronize()
        if self._graph is not None:
            self._graph.replay()
        else:
            self._body()

    def read(self) -> dict:
        """The lower bound's terms as host floats - the iteration's one host sync."""
        return dict(zip(HMM_TERM_KEYS, self.scal.tolist()))

    def advance(self):
        self._copy(self.q, self.q_next)

    def iterate(self, data_pass, whole: bool):
        """One VB iteration: q <- q', the data pass under q (``data_pass()``: the C-ABI launches that fill ``stats`` / ``fb``),
        the K-side.  ``whole`` (short sequences on the GPU): all of it - ~30 data-pass launches and ~100 K-sized ones - is
        captured in ONE hipGraph on the third call and replayed afterwards; the data pass of a short sequence has no host-side
        decision in it (no forgetting pass below 2^15 steps, no pruning with an HMM), so the captured launch sequence IS the
        iteration.  At T = 1e4 an iteration is launch-bound: 1.1 ms with the K-side graph alone."""
        g = getattr(self, "_it_graph", None)
        if whole and self._use_graph and g is None:
            self._it_calls = getattr(self, "_it_calls", 0) + 1
            if self._it_calls >= 3:                 # (two eager iterations first: library warm-up, the K-side's own graph)
                try:
                    torch.cuda.synchronize()
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g):
                        self.advance()
                        data_pass()
                        self._body()
                except Exception as e:              # noqa: BLE001 - capture is an optimisation, never a requirement
                    import warnings
                    warnings.warn(f"bayesml_amd: whole-iteration hipGraph capture failed ({type(e).__name__}: {e}); running eagerly")
                    g = False
                    torch.cuda.synchronize()
                self._it_graph = g
        if g:
            g.replay()
            return
        self.advance()
        data_pass()
        self.step()

    def current(self) -> HmmPostT:
        return self._buffers(self.q)

    def moments(self) -> dict:
        ms, g0, gl, sum_ln_c = self.fb_views()
        return dict(ns=self.ns.clone(), ms=ms.clone(), x_bar=self.x_bar.clone(), s=self.s.clone(), g0=g0.clone(), gl=gl.clone())


# ----------------------------------------------------------------------------------------------- one K-side step
_POST_FIELDS = ("alpha", "m", "kappa", "nu", "w_inv", "w", "u", "u_inv", "e_ln_pi", "e_ln_lambda_det", "ln_b_w_nu", "c")
TERM_KEYS = ("p_x", "p_z", "p_pi", "p_mu_lambda", "q_z", "q_pi", "q_mu_lambda", "vl")


def _copy_post(dst: PostT, src: PostT):
    for f in _POST_FIELDS:
        getattr(dst, f).copy_(getattr(src, f))


def _clone_post(src: PostT) -> PostT:
    q = PostT(src.alpha.clone(), src.m.clone(), src.kappa.clone(), src.nu.clone(), src.w_inv.clone())
    for f in _POST_FIELDS[5:]:
        setattr(q, f, getattr(src, f).clone())
    return q


def packed_stats_len(K: int, D: int) -> int:
    """gmmvb_stats_packed_len: [ns K | h K | a K D | upper triangles of B, K D (D + 1) / 2]."""
    return K * (2 + D) + K * D * (D + 1) // 2


def stats_triangle(pack: bool, K: int, D: int, src: torch.Tensor, dst: torch.Tensor):
    """The statistics block <-> the block on the wire (gmmvb_stats_pack / gmmvb_stats_unpack): the C-ABI kernels on a
    GPU, the same index map in torch on the CPU (host-logic and gloo tests)."""
    if src.is_cuda:
        from . import _engine
        _engine.stats_triangle(pack, K, D, src, dst)
        return
    head = K * (2 + D)
    iu = torch.triu_indices(D, D)
    if pack:
        dst[:head] = src[:head]
        dst[head:] = src[head:].view(K, D, D)[:, iu[0], iu[1]].reshape(-1)
    else:
        dst[:head] = src[:head]
        B = dst[head:].view(K, D, D)
        tri = src[head:].view(K, -1)
        B[:, iu[0], iu[1]] = tri
        B[:, iu[1], iu[0]] = tri


class KStepper:
    """Everything K-sized that one VB iteration does between two data passes, as ONE unit on the GPU:

        statistics block -> moments (x_bar, S)            _calc_n_x_bar_s's finishing      ref :729-732
                         -> lower bound under q            _calc_vl                         ref :671-723
                         -> q' = closed-form update        _update_q_mu_lambda/_update_q_pi ref :741-770
                         -> drift hint (gamma, delta) of q -> q' for the engine (gmmvb_set_drift)
                         -> [p_x .. vl | min_k (gamma_k - delta_k / 30)] in one small vector (the iteration's single
                            device-to-host copy)

    On a GPU (D <= 128) all of it is ONE C-ABI call, gmmvb_kside_step (csrc/kside.hip): one workgroup per component
    assembles, factorises and inverts W'^-1 in LDS and forms the lower bound's traces on the way; a second launch gives
    the drift hint, a third adds the per-component terms.  ``q`` (the posterior whose parameters the engine holds) and
    ``q_next`` are two fixed buffer sets that ``advance()`` swaps.  The functions of this module remain the executable
    specification: they run eagerly on the CPU (host-logic tests), and on a GPU - captured once in a hipGraph and
    replayed - with BAYESML_AMD_KSIDE_FUSED=0 (BAYESML_AMD_KSIDE_GRAPH=0: eagerly); tests/test_gpu_kside.py holds the
    kernel to them."""

    def __init__(self, prior: PriorT, pivot: torch.Tensor, stats_len: int, want_drift: bool):
        import os
        K, D = prior.m.shape
        dev = prior.m.device
        self.prior, self.pivot, self.K, self.D = prior, pivot, K, D
        self.want_drift = want_drift
        # the data pass writes `stats`; behind it, in the same allocation, the engine's policy counters of a row-sharded
        # job (include/gmmvb.h: GMMVB_POLICY_LEN): one all-reduce of `stats_and_tail` carries both
        self.stats_and_tail = torch.zeros(stats_len + 16, dtype=torch.float64, device=dev)
        self.stats = self.stats_and_tail[:stats_len]
        self.tail = self.stats_and_tail[stats_len:]
        # the block on the wire of a row-sharded job (made on first use): [ns | h | a | upper triangles of B | policy tail] -
        # B is symmetric, the lower triangles need not travel (include/gmmvb.h, gmmvb_stats_pack)
        self._wire = None
        self.s_prev = torch.zeros(K, D, D, dtype=torch.float64, device=dev)
        self.q = _clone_post(post_from_prior(prior))
        self.q_next = _clone_post(self.q)
        self.ns = torch.zeros(K, dtype=torch.float64, device=dev)
        self.x_bar = torch.zeros(K, D, dtype=torch.float64, device=dev)
        self.s = torch.zeros(K, D, D, dtype=torch.float64, device=dev)
        self.gamma = torch.zeros(K, dtype=torch.float64, device=dev)
        self.delta = torch.zeros(K, dtype=torch.float64, device=dev)
        self.big_gamma = torch.zeros(K, dtype=torch.float64, device=dev)
        self.scal = torch.zeros(len(TERM_KEYS) + 1, dtype=torch.float64, device=dev)
        self._graph = None
        self._calls = 0
        # (beyond D = 128 the factorisations go through torch.linalg - MAGMA / rocSOLVER -, which cannot be captured)
        self._use_graph = dev.type == "cuda" and D <= 128 and os.environ.get("BAYESML_AMD_KSIDE_GRAPH", "1") != "0"
        self._fused = dev.type == "cuda" and D <= 128 and os.environ.get("BAYESML_AMD_KSIDE_FUSED", "1") != "0"
        if self._fused:
            from . import _engine
            self._eng = _engine
            self._prior_v = _engine.prior_view(prior)
            self._scratch = torch.zeros(13 * K, dtype=torch.float64, device=dev)

    def wire(self):
        """(whole wire buffer, its packed statistics part, its policy tail) of a row-sharded job."""
        if self._wire is None:
            n = packed_stats_len(self.K, self.D)
            buf = torch.zeros(n + 16, dtype=torch.float64, device=self.stats.device)
            self._wire = (buf, buf[:n], buf[n:])
        return self._wire

    def pack(self):
        """statistics block -> wire buffer (the tail is written there directly by the engine's policy_export)."""
        stats_triangle(True, self.K, self.D, self.stats, self.wire()[1])

    def unpack(self):
        """summed wire buffer -> statistics block, B mirrored from its summed upper triangle."""
        stats_triangle(False, self.K, self.D, self.wire()[1], self.stats)

    def load(self, q: PostT):
        """Make ``q`` (with its features) the current posterior."""
        _copy_post(self.q, q)

    def _body(self):
        K, D = self.K, self.D
        st = self.stats
        ns, h = st[:K], st[K:2 * K]
        a = st[2 * K:2 * K + K * D].view(K, D)
        B = st[2 * K + K * D:].view(K, D, D)
        x_bar, s = moments_from_stats(ns, a, B, self.pivot, self.s_prev)
        terms = lower_bound(self.prior, self.q, ns, x_bar, s, h.sum())
        qn = update_q(self.prior, ns, x_bar, s)
        if self.want_drift:
            gamma, delta, big = drift(self.q, qn)
            self.gamma.copy_(gamma)
            self.delta.copy_(delta)
            self.big_gamma.copy_(big)
            # one pessimistic scalar for the engine's choice of carrying scheme: the slowest component decides how
            # fast a row's shared rest bound erodes (delta in whitened units; 30 ~ a typical distance to the rest)
            gmean = (gamma - delta / 30.0).min()
        else:
            gmean = torch.zeros((), dtype=torch.float64, device=st.device)
        _copy_post(self.q_next, qn)
        self.ns.copy_(ns)
        self.x_bar.copy_(x_bar)
        self.s.copy_(s)
        self.s_prev.copy_(s)
        self.scal.copy_(torch.stack([terms[k].reshape(()) for k in TERM_KEYS] + [gmean.reshape(())]))

    def step(self):
        """Run the K-side on ``self.stats`` (already all-reduced).  Results: ns, x_bar, s, q_next, gamma, delta,
        big_gamma, scal."""
        self._calls += 1
        if self._fused:
            e = self._eng
            e.kside_step(self.K, self.D, self._prior_v, e.post_view(self.q), e.post_view(self.q_next), self.stats, self.pivot,
                         self.s_prev, self.ns, self.x_bar, self.s, self.want_drift, self.gamma, self.delta, self.big_gamma,
                         self.scal, self._scratch)
            return
        if self._use_graph and self._graph is None and self._calls >= 2:
            try:                                    # the first call ran eagerly (library / BLAS warm-up)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self._body()
                self._graph = g
            except Exception as e:                  # noqa: BLE001 - capture is an optimisation, never a requirement
                import warnings
                warnings.warn(f"bayesml_amd: K-side hipGraph capture failed ({type(e).__name__}: {e}); running eagerly")
                self._use_graph = False
                torch.cuda.synchronize()
        if self._graph is not None:
            self._graph.replay()
        else:
            self._body()

    def read(self):
        """(dict of the lower bound's terms as host floats, the drift summary min_k (gamma_k - delta_k / 30)) - the
        iteration's one host sync."""
        v = self.scal.tolist()
        return dict(zip(TERM_KEYS, v[:-1])), v[-1]

    def hint(self, gmean):
        """The drift hint of q -> q_next for DataPass.set_drift, or None when the engine cannot use one."""
        # (the engine reads a summary <= 0 as "unknown"; a summary that IS that bad - delta > 30 gamma for some component -
        # must say "bound afresh" instead, which any value below 0.5 does)
        return (self.gamma, self.delta, self.big_gamma, max(float(gmean), 1e-6)) if self.want_drift else None

    def advance(self):
        """q <- q_next (after the engine has been given q_next's parameters)."""
        if self._fused:
            self.q, self.q_next = self.q_next, self.q          # two fixed buffer sets: no copy
        else:
            _copy_post(self.q, self.q_next)

    def current(self) -> PostT:
        return _clone_post(self.q)
