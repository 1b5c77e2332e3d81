"""bayesml_amd — MI355X-native engine for the variational-Bayes GMM posterior update of BayesML.

Only the path named by BASELINE.json's ``north_star`` lives here: ``gaussianmixture.GenModel`` /
``gaussianmixture.LearnModel`` (and ``hiddenmarkovnormal`` for config 5) with the reference's API, backed by hand-written gfx950 HIP kernels
(``csrc/``) behind the C ABI of ``include/gmmvb.h``.  The rest of BayesML is out of scope.
"""
from . import gaussianmixture, hiddenmarkovnormal, multivariate_normal
from ._dist import RestartShard, RowShard
from ._exceptions import (CriteriaError, DataFormatError, ParameterFormatError, ParameterFormatWarning,
                          ResultWarning)

__all__ = ["gaussianmixture", "hiddenmarkovnormal", "multivariate_normal", "RowShard", "RestartShard", "ParameterFormatError", "DataFormatError", "CriteriaError",
           "ResultWarning", "ParameterFormatWarning"]
