"""TEST INFRASTRUCTURE - host restatement of the device samplers' stream (csrc/sample.hip, include/gmmvb.h "device-side
data generation").  Only tests/, __graft_entry__.smoke() and bench.py's checker legs may import this.

The distributions are the reference's ``GenModel.gen_sample`` (bayesml/gaussianmixture/_gaussianmixture.py:241-264: z ~
Categorical(pi_vec), x ~ N(mu_z, Lambda_z^-1); bayesml/hiddenmarkovnormal/_hiddenmarkovnormal.py:344-358: z_0 ~ pi_vec,
z_t ~ a_mat[z_{t-1}], the same emission).  The reference's own random stream (PCG64 through ``Generator.choice`` /
``multivariate_normal``) cannot be drawn in parallel; the device sampler is defined on a counter-based stream instead, and
that stream's generator is not restated here but TAKEN from NumPy: ``numpy.random.Philox(key=[seed, stream])`` (Philox4x64-10,
Random123), whose ``random_raw`` output is the device's block sequence.  Pinned by tests/test_samplers.py: this module against
a pure-Python Philox4x64-10 with the Random123 known-answer vectors, the device against this module value by value.
"""
from __future__ import annotations

import numpy as np

_TWO_M53 = 2.0 ** -53


def raw_blocks(seed: int, stream: int, first_block: int, n_blocks: int) -> np.ndarray:
    """[n_blocks, 4] uint64: blocks first_block .. of the stream.  NumPy's Philox increments its counter before it
    generates, so a generator constructed with counter = L produces block L (the device's counter value L + 1) first."""
    ctr = np.array([first_block & (2 ** 64 - 1), first_block >> 64, 0, 0], dtype=np.uint64)
    bg = np.random.Philox(key=np.array([seed, stream], dtype=np.uint64), counter=ctr)
    return bg.random_raw(4 * n_blocks).reshape(n_blocks, 4)


def latent_uniforms(seed: int, row0: int, n: int) -> np.ndarray:
    """u_t = (raw_t >> 11) 2^-53 for t in [row0, row0 + n): stream 0, one 64-bit output per row."""
    b0 = row0 // 4
    nb = (row0 + n + 3) // 4 - b0
    raw = raw_blocks(seed, 0, b0, nb).reshape(-1)[row0 - 4 * b0: row0 - 4 * b0 + n]
    return (raw >> np.uint64(11)).astype(np.float64) * _TWO_M53


def inverse_cdf(cdf: np.ndarray, u: np.ndarray) -> np.ndarray:
    """Number of entries of cdf[:-1] that are <= u."""
    return np.searchsorted(cdf[:-1], u, side="right").astype(np.int64)


def normals(seed: int, row0: int, n: int, D: int) -> np.ndarray:
    """eps [n, D]: stream 1, row r owns blocks r ceil(D/4) ..; Box-Muller on (r0, r1) and (r2, r3) of a block."""
    nb = (D + 3) // 4
    raw = raw_blocks(seed, 1, row0 * nb, n * nb)
    f = (raw >> np.uint64(11)).astype(np.float64) * _TWO_M53
    out = np.empty((n * nb, 4))
    for h in (0, 1):
        rad = np.sqrt(-2.0 * np.log(1.0 - f[:, 2 * h]))
        ang = 2.0 * np.pi * f[:, 2 * h + 1]
        out[:, 2 * h] = rad * np.cos(ang)
        out[:, 2 * h + 1] = rad * np.sin(ang)
    return out.reshape(n, 4 * nb)[:, :D]


def emission_factors(lambda_mats: np.ndarray) -> np.ndarray:
    """A [K, D, D] lower triangular, A_k = L_k^-1 with Lambda_k = L_k L_k^T, so that mu + eps A ~ N(mu, Lambda^-1)."""
    return np.stack([np.linalg.inv(np.linalg.cholesky(l)) for l in np.asarray(lambda_mats, dtype=np.float64)])


def emissions(z: np.ndarray, mu: np.ndarray, a: np.ndarray, seed: int, row0: int = 0) -> np.ndarray:
    eps = normals(seed, row0, z.shape[0], mu.shape[1])
    return mu[z] + np.einsum("ni,nij->nj", eps, a[z])


def mixture_latent(pi_vec: np.ndarray, seed: int, row0: int, n: int) -> np.ndarray:
    return inverse_cdf(np.cumsum(pi_vec), latent_uniforms(seed, row0, n))


def markov_chain(pi_vec: np.ndarray, a_mat: np.ndarray, seed: int, length: int) -> np.ndarray:
    """The plain sequential recursion (reference ``_hiddenmarkovnormal.py:349-357`` with inverse-CDF draws)."""
    u = latent_uniforms(seed, 0, length)
    cp, ca = np.cumsum(pi_vec), np.cumsum(a_mat, axis=1)
    z = np.empty(length, dtype=np.int64)
    s = int(np.searchsorted(cp[:-1], u[0], side="right"))
    z[0] = s
    for t in range(1, length):
        s = int(np.searchsorted(ca[s, :-1], u[t], side="right"))
        z[t] = s
    return z


# ---- an independent statement of the generator, used only to pin NumPy's against the published known answers ----------

_M0, _M1 = 0xD2E7470EE14C6C93, 0xCA5A826395121157
_W0, _W1 = 0x9E3779B97F4A7C15, 0xBB67AE8584CAA73B
_MASK = (1 << 64) - 1


def philox4x64_10(ctr, key):
    """Salmon et al., "Parallel random numbers: as easy as 1, 2, 3" (SC'11), Philox-4x64 with ten rounds."""
    c, k = list(ctr), list(key)
    for _ in range(10):
        p0, p1 = _M0 * c[0], _M1 * c[2]
        c = [(p1 >> 64) ^ c[1] ^ k[0], p1 & _MASK, (p0 >> 64) ^ c[3] ^ k[1], p0 & _MASK]
        k = [(k[0] + _W0) & _MASK, (k[1] + _W1) & _MASK]
    return c
