r"""Hidden Markov model with Gaussian emissions and a Dirichlet / Dirichlet-rows / Normal-Wishart prior.

    pi ~ Dir(eta0),  a_k ~ Dir(zeta0_k),  Lambda_k ~ Wishart(W0_k, nu0_k),  mu_k | Lambda_k ~ N(m0_k, (kappa0_k Lambda_k)^-1)
    z_1 ~ Cat(pi),   z_t | z_{t-1} = i ~ Cat(a_i),   x_t | z_t = k ~ N(mu_k, Lambda_k^-1)

(`bayesml/hiddenmarkovnormal/__init__.py` of the reference; note its code updates eta with ns, not gamma_1.)
"""
from ._hiddenmarkovnormal import GenModel, LearnModel

__all__ = ["GenModel", "LearnModel"]
