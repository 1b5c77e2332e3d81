"""Abstract model contract shared by the model packages.

Restates ``bayesml/base.py`` (reference): ``Generative`` (:8-137), ``Posterior`` (:139-290) and
``PredictiveMixin`` (:292-357).  The pickle round-trips are positional over ``dict.values()``
(reference base.py:191,251), so the key ORDER of the get_* dicts is part of the API.
"""
import pickle
from abc import ABCMeta, abstractmethod

from ._exceptions import ParameterFormatError

_LOAD_MSG = (" must be a pickled python dictionary obtained by ``GenModel.save_h_params()``, "
             "``LearnModel.save_h0_params()`` or ``LearnModel.save_hn_params()``.")


def _dump(obj, filename):
    with open(filename, "wb") as f:
        pickle.dump(obj, f)


def _load_dict(filename, message):
    with open(filename, "rb") as f:
        obj = pickle.load(f)
    if type(obj) is not dict:
        raise ParameterFormatError(filename + message)
    return obj


class Generative(metaclass=ABCMeta):
    @abstractmethod
    def set_h_params(self): ...

    @abstractmethod
    def get_h_params(self): ...

    @abstractmethod
    def gen_params(self): ...

    @abstractmethod
    def set_params(self): ...

    @abstractmethod
    def get_params(self): ...

    @abstractmethod
    def gen_sample(self): ...

    @abstractmethod
    def save_sample(self): ...

    @abstractmethod
    def visualize_model(self): ...

    def save_h_params(self, filename):
        """Pickle ``get_h_params()`` (reference base.py:17-35)."""
        _dump(self.get_h_params(), filename)

    def load_h_params(self, filename):
        """Positional ``set_h_params(*dict.values())`` (reference base.py:37-67)."""
        self.set_h_params(*_load_dict(filename, _LOAD_MSG).values())
        return self

    def save_params(self, filename):
        """Pickle ``get_params()`` (reference base.py:81-99)."""
        _dump(self.get_params(), filename)

    def load_params(self, filename):
        """Positional ``set_params(*dict.values())`` (reference base.py:101-125)."""
        self.set_params(*_load_dict(
            filename, " must be a pickled python dictionary obtained by ``GenModel.save_params()``").values())
        return self


class Posterior(metaclass=ABCMeta):
    @abstractmethod
    def set_h0_params(self): ...

    @abstractmethod
    def get_h0_params(self): ...

    @abstractmethod
    def set_hn_params(self): ...

    @abstractmethod
    def get_hn_params(self): ...

    @abstractmethod
    def update_posterior(self): ...

    @abstractmethod
    def estimate_params(self): ...

    @abstractmethod
    def visualize_posterior(self): ...

    def save_h0_params(self, filename):
        """reference base.py:148-166."""
        _dump(self.get_h0_params(), filename)

    def load_h0_params(self, filename):
        """reference base.py:168-198."""
        self.set_h0_params(*_load_dict(filename, _LOAD_MSG).values())
        return self

    def save_hn_params(self, filename):
        """reference base.py:208-226."""
        _dump(self.get_hn_params(), filename)

    def load_hn_params(self, filename):
        """reference base.py:228-258."""
        self.set_hn_params(*_load_dict(filename, _LOAD_MSG).values())
        return self

    def reset_hn_params(self):
        """hn_* <- h0_* through the validated setter (reference base.py:260-267)."""
        self.set_hn_params(*self.get_h0_params().values())
        return self

    def overwrite_h0_params(self):
        """h0_* <- hn_* through the validated setter (reference base.py:269-276)."""
        self.set_h0_params(*self.get_hn_params().values())
        return self


class PredictiveMixin(metaclass=ABCMeta):
    @abstractmethod
    def get_p_params(self): ...

    @abstractmethod
    def calc_pred_dist(self): ...

    @abstractmethod
    def make_prediction(self): ...

    @abstractmethod
    def pred_and_update(self): ...
