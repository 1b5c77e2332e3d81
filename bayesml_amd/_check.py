"""Input validators of the boundary (the subset the GMM/HMM path uses).

Behavioural restatement of ``bayesml/_check.py`` (reference file:line in each docstring): each
validator returns the (possibly float-cast) value or raises ``exc(name + message)``.  Written
around two predicates (`_is_int`, `_is_real`) instead of the reference's copy-per-function style.
"""
import numpy as np

_EPSILON = np.sqrt(np.finfo(np.float64).eps)


def _is_int(v):
    return np.issubdtype(type(v), np.integer)


def _is_real(v):
    return _is_int(v) or np.issubdtype(type(v), np.floating)


def _arr_kind(v):
    """'i' / 'f' for integer / floating ndarrays, None for anything else (lists, complex, ...)."""
    if type(v) is not np.ndarray:
        return None
    if np.issubdtype(v.dtype, np.integer):
        return "i"
    if np.issubdtype(v.dtype, np.floating):
        return "f"
    return None


def pos_int(val, name, exc):
    """_check.py:28-32 — Python/NumPy integers > 0 only (floats such as 2.0 are rejected)."""
    if _is_int(val) and val > 0:
        return val
    raise exc(name + " must be int. Its value must be positive (not including 0).")


def pos_float(val, name, exc):
    """_check.py:19-26 — positive real scalar (integers are cast to float)."""
    if _is_real(val) and val > 0.0:
        return float(val) if _is_int(val) else val
    raise exc(name + " must be positive (not including 0.0).")


def floats(val, name, exc):
    """_check.py:163-173 — real scalar or real ndarray (ints are cast); no sign condition."""
    if _is_real(val):
        return float(val) if _is_int(val) else val
    kind = _arr_kind(val)
    if kind is not None:
        return val.astype(float) if kind == "i" else val
    raise exc(name + " must be float or a numpy.ndarray.")


def pos_floats(val, name, exc):
    """_check.py:175-185 — positive real scalar or positive real ndarray (ints are cast)."""
    if _is_real(val) and val > 0.0:
        return float(val) if _is_int(val) else val
    kind = _arr_kind(val)
    if kind is not None and np.all(val > 0):
        return val.astype(float) if kind == "i" else val
    raise exc(name + " must be float or a numpy.ndarray. Its values must be positive (not including 0)")


def float_vec(val, name, exc):
    """_check.py:187-193 — 1-dimensional real ndarray."""
    kind = _arr_kind(val)
    if kind is not None and val.ndim == 1:
        return val.astype(float) if kind == "i" else val
    raise exc(name + " must be a 1-dimensional numpy.ndarray.")


def float_vecs(val, name, exc):
    """_check.py:203-209 — real ndarray with ndim >= 1."""
    kind = _arr_kind(val)
    if kind is not None and val.ndim >= 1:
        return val.astype(float) if kind == "i" else val
    raise exc(name + " must be a numpy.ndarray whose ndim >= 1.")


def float_vec_sum_1(val, name, exc):
    """_check.py:219-225 — 1-dimensional real ndarray summing to 1 within sqrt(eps)."""
    kind = _arr_kind(val)
    if kind is not None and val.ndim == 1 and abs(val.sum() - 1.0) <= _EPSILON:
        return val.astype(float) if kind == "i" else val
    raise exc(name + " must be a 1-dimensional numpy.ndarray, and the sum of its elements must equal to 1.")


def float_vecs_sum_1(val, name, exc):
    """_check.py:227-233 — real ndarray whose last axis sums to 1 within sqrt(eps)."""
    kind = _arr_kind(val)
    if kind is not None and val.ndim >= 1 and np.all(np.abs(np.sum(val, axis=-1) - 1.0) <= _EPSILON):
        return val.astype(float) if kind == "i" else val
    raise exc(name + " must be a numpy.ndarray whose ndim >= 1, and the sum along the last dimension must equal to 1.")


def pos_def_sym_mats(val, name, exc):
    """_check.py:140-154 — stack of symmetric (np.allclose) positive-definite (batched Cholesky) matrices."""
    ok = (type(val) is np.ndarray and val.ndim >= 2 and val.shape[-1] == val.shape[-2]
          and np.allclose(val, np.swapaxes(val, -1, -2)))
    if not ok:
        raise exc(name + " must be a symmetric 2-dimensional numpy.ndarray.")
    try:
        np.linalg.cholesky(val)
    except np.linalg.LinAlgError:
        raise exc(name + " must be a positive definite symmetric 2-dimensional numpy.ndarray.") from None
    return val


def pos_def_sym_mat(val, name, exc):
    """_check.py:124-138 — one symmetric (np.allclose) positive-definite (Cholesky) matrix."""
    ok = type(val) is np.ndarray and val.ndim == 2 and val.shape[0] == val.shape[1] and np.allclose(val, val.T)
    if not ok:
        raise exc(name + " must be a symmetric 2-dimensional numpy.ndarray.")
    try:
        np.linalg.cholesky(val)
    except np.linalg.LinAlgError:
        raise exc(name + " must be a positive definite symmetric 2-dimensional numpy.ndarray.") from None
    return val


def shape_consistency(val, val_name, correct, correct_name, exc):
    """_check.py:268-272."""
    if val != correct:
        raise exc(f"{val_name} must coincide with {correct_name}: {val_name} = {val}, {correct_name} = {correct}")
