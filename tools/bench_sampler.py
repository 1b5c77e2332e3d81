"""Times GenModel.gen_sample(n, device="cuda") (csrc/sample.hip) at the benchmark configurations' sizes:
mixture K 64, D 128, N 1e7 (f32 rows) and HMM K 32, D 16, T 1e7.  One JSON line; HIP events on the launch stream."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesml_amd import _sample  # noqa: E402


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    return best


def main():
    rng = np.random.default_rng(0)
    out = {}
    for name, K, D, n in (("gmm_c3", 64, 128, 10_000_000), ("gmm_c2", 16, 32, 1_000_000), ("gmm_c4_shard", 256, 64, 12_500_000)):
        pi = rng.dirichlet(np.ones(K) * 2)
        mu = 2 * rng.standard_normal((K, D))
        lam = np.stack([np.eye(D) * (0.5 + rng.random())] * 1 * K)
        ms = timed(lambda: _sample.mixture(pi, mu, lam, n, 1, "cuda:0", torch.float32))
        t0 = time.perf_counter()
        fac = _sample.emission_factors(lam)
        host_ms = (time.perf_counter() - t0) * 1e3
        _x, z = _sample.mixture(pi, mu, lam, n, 1, "cuda:0", torch.float32)
        ms_k = timed(lambda: _sample.draw_emissions(z, mu, lam, 1, torch.float32, factors=fac))
        ms_plain = timed(lambda: _sample.draw_emissions(z, mu, lam, 1, torch.float32, factors=fac, grouped=False), reps=1)
        del _x, z
        out[name] = {"K": K, "D": D, "rows": n, "ms": round(ms, 2), "rows_per_s": round(n / ms * 1e3, 1),
                     "host_factor_ms": round(host_ms, 2), "emission_kernels_ms": round(ms_k, 2),
                     "emission_kernel_ungrouped_ms": round(ms_plain, 2),
                     "output_GB_per_s": round(n * D * 4 / ms_k / 1e6, 1)}
    K, D, T = 32, 16, 10_000_000
    pi, a = rng.dirichlet(np.ones(K)), rng.dirichlet(np.ones(K) * 0.7, K)
    mu = 2 * rng.standard_normal((K, D))
    lam = np.stack([np.eye(D)] * K)
    ms_chain = timed(lambda: _sample.markov_chain(pi, a, T, 1, "cuda:0"))
    ms = timed(lambda: _sample.hidden_markov(pi, a, mu, lam, T, 1, "cuda:0", torch.float32))
    out["hmm_c5"] = {"K": K, "D": D, "steps": T, "ms": round(ms, 2), "chain_ms": round(ms_chain, 2),
                     "steps_per_s": round(T / ms * 1e3, 1)}
    out["note"] = ("ms: the whole call - K-sized preparation (cumulative sums on the host, Cholesky factors by gmmvb_kside_factor), "
                   "output allocation, class grouping and kernels; host_factor_ms: what the factors would cost in NumPy; "
                   "reference gen_sample: a Python loop, about 1e4 rows per second")
    print(json.dumps(out))


if __name__ == "__main__":
    main()
