"""Device-side data generation for ``GenModel.gen_sample(..., device=...)``.

The reference draws every row in a Python loop (``_gaussianmixture.py:241-264``: one ``choice`` and one
``multivariate_normal`` per row; ``_hiddenmarkovnormal.py:344-358`` the same along a Markov chain), minutes per million
rows.  Here the latent classes and the emissions ``mu_z + eps L_z^-1`` (``Lambda_z = L_z L_z^T``) are drawn by the HIP
kernels of ``csrc/sample.hip`` behind the C ABI (``gmmvb_sample_*``), on a counter-based stream that
``numpy.random.Philox(key=[seed, stream])`` reproduces on the host value by value (``include/gmmvb.h``).  Same
distributions, reproducible per seed - but not the reference's random stream (the host path of ``gen_sample`` keeps that).
The K-sized preparation (cumulative distributions, Cholesky factors) is host NumPy like every other K-sized read-out;
PyTorch owns the device memory.  No CPU fallback: without the library and a GPU this raises ``EngineUnavailableError``.
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from . import _engine


def emission_factors(lambda_mats) -> np.ndarray:
    """A [K, D, D] lower triangular with A_k^T A_k = Lambda_k^-1 (A_k = L_k^-1, Lambda_k = L_k L_k^T), so that
    mu_k + eps A_k ~ N(mu_k, Lambda_k^-1) for eps ~ N(0, I)."""
    return np.linalg.inv(np.linalg.cholesky(np.asarray(lambda_mats, dtype=np.float64)))


def _device(device) -> torch.device:
    dev = torch.device(device)
    if dev.type != "cuda" or not torch.cuda.is_available():
        raise _engine.EngineUnavailableError(
            f"gen_sample(device={device!r}): the device sampler is a HIP kernel and needs an MI355X; "
            "call gen_sample without `device` for the reference's host stream")
    return torch.device("cuda", torch.cuda.current_device() if dev.index is None else dev.index)


def _dtype_code(dtype) -> int:
    if dtype == torch.float32:
        return _engine.GMMVB_F32
    if dtype == torch.float64:
        return _engine.GMMVB_F64
    raise ValueError("dtype must be torch.float32 or torch.float64")


def _up(a, dev):
    return torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64), device=dev)


def _stream(dev):
    return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def draw_emissions(z: torch.Tensor, mu_vecs, lambda_mats, seed: int, dtype, row0: int = 0, grouped: bool = True,
                   factors=None) -> torch.Tensor:
    """x [n, D] of ``dtype``: row i ~ N(mu[z_i], Lambda[z_i]^-1) with the normals of global row ``row0 + i``
    (``grouped``: the kernel visits the rows class by class - same values, the factors stay in L2)."""
    lib, dev = _engine.load_library(), z.device
    mu = np.asarray(mu_vecs, dtype=np.float64)
    K, D = mu.shape
    x = torch.empty((z.shape[0], D), dtype=dtype, device=dev)
    if factors is not None:
        a_d = _up(factors, dev)
    elif D <= _engine.MAX_MFMA_DEGREE:
        # the library's LDS-resident factorisation (gmmvb_kside_factor: Lambda = G G^T, G^-1), one workgroup per class; on a
        # 256-core host the same K small LAPACK calls cost 0.7 s at K 64, D 128 (profiles/r6_sampler.json)
        a_d = _engine.kside_factor(_up(lambda_mats, dev))[1]
    else:
        a_d = _up(emission_factors(lambda_mats), dev)
    mu_d = _up(mu, dev)
    nbytes = max(0, lib.gmmvb_sample_emissions_work_bytes(K, z.shape[0])) if grouped else 0
    work = torch.empty(nbytes, dtype=torch.uint8, device=dev) if nbytes else None
    with torch.cuda.device(dev):
        _engine._check(lib, lib.gmmvb_sample_emissions(K, D, z.data_ptr(), mu_d.data_ptr(), a_d.data_ptr(), int(seed), int(row0),
                                                       z.shape[0], _dtype_code(dtype), x.data_ptr(), D,
                                                       work.data_ptr() if nbytes else None, nbytes, _stream(dev)),
                       "gmmvb_sample_emissions")
    return x


def mixture(pi_vec, mu_vecs, lambda_mats, n: int, seed: int, device, dtype, row0: int = 0):
    """(x [n, D], z [n] int64) of the mixture: rows ``row0 .. row0 + n`` of the sample that ``seed`` defines."""
    lib, dev = _engine.load_library(), _device(device)
    pi = np.asarray(pi_vec, dtype=np.float64)
    z = torch.empty(n, dtype=torch.int64, device=dev)
    cdf = _up(np.cumsum(pi), dev)
    with torch.cuda.device(dev):
        _engine._check(lib, lib.gmmvb_sample_latent(pi.shape[0], cdf.data_ptr(), int(seed), int(row0), n, z.data_ptr(), _stream(dev)),
                       "gmmvb_sample_latent")
    return draw_emissions(z, mu_vecs, lambda_mats, seed, dtype, row0), z


def markov_chain(pi_vec, a_mat, length: int, seed: int, device) -> torch.Tensor:
    """z [length] int64 with z_0 ~ pi, z_t ~ a_mat[z_{t-1}] (reference ``_hiddenmarkovnormal.py:349-357``), without a
    sequential pass over the sequence: chunk maps on the K states, composed on the device (``gmmvb_sample_chain``)."""
    lib, dev = _engine.load_library(), _device(device)
    pi, a = np.asarray(pi_vec, dtype=np.float64), np.asarray(a_mat, dtype=np.float64)
    K = pi.shape[0]
    z = torch.empty(length, dtype=torch.int64, device=dev)
    nbytes = lib.gmmvb_sample_chain_work_bytes(K, length)
    work = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    cp, ca = _up(np.cumsum(pi), dev), _up(np.cumsum(a, axis=1), dev)
    with torch.cuda.device(dev):
        _engine._check(lib, lib.gmmvb_sample_chain(K, cp.data_ptr(), ca.data_ptr(), int(seed), length, z.data_ptr(), work.data_ptr(),
                                                   nbytes, _stream(dev)), "gmmvb_sample_chain")
    return z


def hidden_markov(pi_vec, a_mat, mu_vecs, lambda_mats, length: int, seed: int, device, dtype):
    """(x [T, D], z [T] int64) of the HMM."""
    z = markov_chain(pi_vec, a_mat, length, seed, device)
    return draw_emissions(z, mu_vecs, lambda_mats, seed, dtype), z
