"""Multivariate normal model with a Gauss-Wishart prior (``bayesml.multivariate_normal``), posterior-update path.

SURVEY.md section 8f.4: the exact conjugate update is the K = 1, r = 1 case of the GMM M-step kernel, so
``LearnModel.update_posterior`` runs its one pass over x on the GPU through the same C ABI (``gmmvb_mstep``)."""
from ._multivariatenormal import GenModel, LearnModel

__all__ = ["GenModel", "LearnModel"]
