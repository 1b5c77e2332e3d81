r"""Gaussian mixture model with a Dirichlet x Normal-Wishart prior (variational Bayes).

Model (K = c_num_classes, D = c_degree), as in the reference's package docstring
(``bayesml/gaussianmixture/__init__.py``):

    pi ~ Dir(alpha0),  Lambda_k ~ Wishart(W0_k, nu0_k),  mu_k | Lambda_k ~ N(m0_k, (kappa0_k Lambda_k)^-1)
    z_n ~ Cat(pi),     x_n | z_n = k ~ N(mu_k, Lambda_k^-1)

The variational posterior q(z) q(pi) prod_k q(mu_k, Lambda_k) keeps the same families with
hyper-parameters ``hn_alpha_vec, hn_m_vecs, hn_kappas, hn_nus, hn_w_mats``.
"""
from ._gaussianmixture import GenModel, LearnModel

__all__ = ["GenModel", "LearnModel"]
