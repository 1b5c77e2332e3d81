"""CPU stand-in for ``bayesml_amd._engine.DataPass`` (TEST INFRASTRUCTURE ONLY).

Implements the C ABI's semantics (include/gmmvb.h) in fp64 torch on the CPU so that the HOST logic of
the drop-in (restart driver, RNG consumption order, stdout protocol, row sharding + all-reduce) can be
tested in the GPU-less build container and under gloo.  It is injected through the private
``LearnModel._data_pass_factory`` seam by tests only; the product path never constructs it and fails
loudly without the HIP extension.
"""
import torch


class CpuDataPass:
    def __init__(self, K, D, x):
        self.K, self.D = K, D
        self.device = torch.device("cpu")
        self.pivot = torch.zeros(D, dtype=torch.float64)
        self.stats_len = K * (2 + D + D * D)
        self._direct = None
        self.launch_info = "cpu stand-in"

    def adopt(self, x):
        t = x if isinstance(x, torch.Tensor) else torch.from_numpy(x.copy())
        if t.dtype not in (torch.float32, torch.float64):
            t = t.to(torch.float64)
        return t.contiguous()

    def set_pivot(self, p):
        self.pivot = torch.as_tensor(p, dtype=torch.float64).clone()

    def prepare_rows(self, x):
        pass

    def set_params(self, c, m, u):
        self.c, self.m, self.u = c.clone(), m.clone(), u.clone()

    def close(self):
        pass

    def estep(self, x):
        x = x.to(torch.float64)
        y = torch.einsum("kji,nki->nkj", self.u, x[:, None, :] - self.m[None])     # y = u (x - m)
        self._ln_rho = self.c[None, :] - 0.5 * (y * y).sum(dim=2)
        self._lse = torch.logsumexp(self._ln_rho, dim=1)
        self._direct = None
        self.rows = x.shape[0]

    def load_responsibilities(self, r):
        self._direct = torch.as_tensor(r, dtype=torch.float64).clone()
        self.rows = r.shape[0]

    def responsibilities(self, row0=0, n=None):
        n = self.rows - row0 if n is None else n
        if self._direct is not None:
            return self._direct[row0:row0 + n]
        return torch.exp(self._ln_rho[row0:row0 + n] - self._lse[row0:row0 + n, None])

    def ln_rho(self, row0=0, n=None):
        n = self.rows - row0 if n is None else n
        return self._ln_rho[row0:row0 + n]

    def argmax(self, row0=0, n=None):
        n = self.rows - row0 if n is None else n
        src = self._direct if self._direct is not None else self._ln_rho
        return torch.argmax(src[row0:row0 + n], dim=1).to(torch.int32)

    def mstep(self, x):
        K, D = self.K, self.D
        r = self.responsibilities()
        xp = x.to(torch.float64) - self.pivot
        stats = torch.empty(self.stats_len, dtype=torch.float64)
        stats[:K] = r.sum(dim=0)
        stats[K:2 * K] = torch.special.xlogy(r, r).sum(dim=0)
        stats[2 * K:2 * K + K * D] = (r.T @ xp).reshape(-1)
        stats[2 * K + K * D:] = torch.einsum("nk,ni,nj->kij", r, xp, xp).reshape(-1)
        return stats

    def estep_mstep(self, x):
        self.estep(x)
        return self.mstep(x)

    def split_stats(self, stats):
        K, D = self.K, self.D
        return (stats[:K], stats[K:2 * K], stats[2 * K:2 * K + K * D].view(K, D),
                stats[2 * K + K * D:].view(K, D, D))


def cpu_factory(K, D, x):
    return CpuDataPass(K, D, x)
