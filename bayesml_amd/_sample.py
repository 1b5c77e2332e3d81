"""Device-side data generation for ``GenModel.gen_sample(..., device=...)``.

The reference draws every row in a Python loop (``_gaussianmixture.py:241-264``: one ``choice`` and one
``multivariate_normal`` per row; ``_hiddenmarkovnormal.py:344-358`` the same along a Markov chain), minutes per million
rows.  Here the latent classes are drawn in one batched pass on the device and the emissions as ``mu_z + eps L_z^-T``
(``Lambda_z = L_z L_z^T``).  Same distribution, reproducible per seed - but not the reference's random stream (the host
path of ``gen_sample`` keeps that).  PyTorch is plumbing: nothing here is on the posterior-update path.
"""
from __future__ import annotations

import torch


def emission_factors(lambda_mats: torch.Tensor) -> torch.Tensor:
    """a [K, D, D] with a_k^T a_k = Lambda_k^-1, so that mu_k + eps a_k ~ N(mu_k, Lambda_k^-1) for eps ~ N(0, I)."""
    chol = torch.linalg.cholesky(lambda_mats)
    eye = torch.eye(lambda_mats.shape[-1], dtype=lambda_mats.dtype, device=lambda_mats.device).expand_as(chol)
    return torch.linalg.solve_triangular(chol, eye, upper=False)


def draw_emissions(z: torch.Tensor, mu: torch.Tensor, a: torch.Tensor, gen: torch.Generator, dtype, chunk: int = 1 << 22):
    """x [n, D] of ``dtype``: row i ~ N(mu[z_i], Lambda[z_i]^-1), drawn in f64 in chunks of rows grouped by class."""
    n, (K, D) = z.shape[0], mu.shape
    x = torch.empty((n, D), dtype=dtype, device=z.device)
    for lo in range(0, n, chunk):
        hi = min(n, lo + chunk)
        zc = z[lo:hi]
        eps = torch.randn(hi - lo, D, dtype=torch.float64, device=z.device, generator=gen)
        order = torch.argsort(zc, stable=True)
        counts = torch.bincount(zc, minlength=K).tolist()
        out = torch.empty_like(eps)
        start = 0
        for k, c in enumerate(counts):
            if c:
                idx = order[start:start + c]
                out[idx] = mu[k] + eps[idx] @ a[k]
                start += c
        x[lo:hi] = out.to(dtype)
    return x


def markov_chain(pi: torch.Tensor, a_mat: torch.Tensor, length: int, gen: torch.Generator, chunk: int = 2048) -> torch.Tensor:
    """z [length] int64 with z_0 ~ pi, z_t ~ a_mat[z_{t-1}] (reference ``_hiddenmarkovnormal.py:349-357``), without a
    sequential pass over the sequence: with one uniform u_t per step, step t is the map i -> F_i^-1(u_t) (inverse CDF of
    row i) on the K states, and maps compose.  The sequence is cut into chunks; pass 1 carries ALL K start states through
    every chunk at once (chunk-parallel, `chunk` small launches), the chunks' end maps are chained on the host (T / chunk
    integers), pass 2 replays every chunk from its now known start state."""
    dev = pi.device
    K = pi.shape[0]
    u = torch.rand(length, dtype=torch.float64, device=dev, generator=gen)
    cdf_pi = torch.cumsum(pi, 0)[:-1].contiguous()                       # [K-1]
    cdf_a = torch.cumsum(a_mat, 1)[:, :-1].contiguous()                  # [K, K-1]
    first = int((u[0] >= cdf_pi).sum()) if K > 1 else 0
    L = max(1, min(int(chunk), length))
    C = (length + L - 1) // L
    pad = torch.zeros(C * L, dtype=torch.float64, device=dev)
    pad[:length] = u
    U = pad.view(C, L)
    if K == 1:
        return torch.zeros(length, dtype=torch.int64, device=dev)
    # pass 1: every start state through every chunk.  The gathered CDF rows are a [chunks, K, K - 1] temporary per step
    # (2.5 GB at K = 256, T = 1e7 with all chunks at once): blocks of chunks keep it under 256 MB.
    blk = max(1, min(C, (256 << 20) // (8 * K * K)))
    st = torch.arange(K, device=dev).expand(C, K).contiguous()           # [C, K]
    for c0 in range(0, C, blk):
        sb, ub = st[c0:c0 + blk], U[c0:c0 + blk]
        for t in range(L):
            sb = (ub[:, t, None, None] >= cdf_a[sb]).sum(-1)
            if t == 0 and c0 == 0:
                sb[0, :] = first                                          # the sequence's first step draws from pi
        st[c0:c0 + blk] = sb
    end = st.cpu()
    starts = torch.zeros(C, dtype=torch.int64)
    s = 0
    for c in range(C - 1):
        s = int(end[c, s])
        starts[c + 1] = s
    # pass 2: replay from the known start states
    cur = starts.to(dev)
    Z = torch.empty((C, L), dtype=torch.int64, device=dev)
    for t in range(L):
        cur = (U[:, t, None] >= cdf_a[cur]).sum(-1)
        if t == 0:
            cur[0] = first
        Z[:, t] = cur
    return Z.view(-1)[:length].contiguous()
