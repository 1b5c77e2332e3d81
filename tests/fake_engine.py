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
        self._hmm_gamma = None
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

    # ---- HMM (semantics of hmmvb_* in include/gmmvb.h), plain sequential recursions
    def enable_hmm(self):
        self._hmm_gamma = None

    emission_fused = False

    def hmm_skip_h(self, skip=True):        # (the fake computes h anyway; the host ignores it)
        pass

    def emission_target(self, fused):       # (the fake keeps its ln rho array: never in effect)
        return False

    def forward_backward(self, pi_tilde, a_tilde, out=None):
        ln_rho = self._ln_rho
        T, K = ln_rho.shape
        mx = ln_rho.max(dim=1).values
        rho = torch.exp(ln_rho - mx[:, None])
        alpha = torch.empty_like(rho)
        cs = torch.empty(T, dtype=torch.float64)
        a = rho[0] * pi_tilde
        cs[0] = a.sum()
        alpha[0] = a / cs[0]
        for t in range(1, T):
            a = rho[t] * (alpha[t - 1] @ a_tilde)
            cs[t] = a.sum()
            alpha[t] = a / cs[t]
        beta = torch.ones_like(rho)
        for t in range(T - 2, -1, -1):
            beta[t] = a_tilde @ (rho[t + 1] * beta[t + 1]) / cs[t + 1]
        gamma = alpha * beta
        ms = torch.zeros(K, K, dtype=torch.float64)
        for t in range(1, T):
            ms += alpha[t - 1][:, None] * rho[t][None, :] * a_tilde * beta[t][None, :] / cs[t]
        self._hmm_gamma, self._alpha = gamma, alpha
        self._direct = gamma
        res = (ms, gamma[0].clone(), gamma[-1].clone(), (torch.log(cs) + mx).sum())
        if out is None:
            return res
        out.copy_(torch.cat([res[0].reshape(-1), res[1], res[2], res[3].reshape(1)]))          # (the engine's out= contract)
        return out[:K * K].view(K, K), out[K * K:K * K + K], out[K * K + K:K * K + 2 * K], out[K * K + 2 * K]

    def viterbi(self, ln_pi_tilde, ln_a_tilde):
        ln_rho = self._ln_rho
        T, K = ln_rho.shape
        omega = ln_rho[0] + ln_pi_tilde
        phi = torch.zeros(T, K, dtype=torch.int64)
        for t in range(1, T):
            cand = ln_a_tilde + omega[:, None]
            best, arg = cand.max(dim=0)
            phi[t] = arg
            omega = ln_rho[t] + best
        z = torch.zeros(T, dtype=torch.int32)
        k = int(torch.argmax(omega))
        z[-1] = k
        for t in range(T - 2, -1, -1):
            k = int(phi[t + 1, k])
            z[t] = k
        return z

    def hmm_debug(self, what, row0=0, n=None):
        return self._alpha

    def mstep(self, x, out=None):
        K, D = self.K, self.D
        r = self.responsibilities()
        xp = x.to(torch.float64) - self.pivot
        stats = torch.empty(self.stats_len, dtype=torch.float64) if out is None else out
        stats[:K] = r.sum(dim=0)
        if getattr(self, "_hmm_gamma", None) is not None and self._direct is self._hmm_gamma:
            stats[K:2 * K] = (r * self._ln_rho).sum(dim=0)          # HMM mode: h = sum gamma ln rho
            stats[2 * K:2 * K + K * D] = (r.T @ xp).reshape(-1)
            stats[2 * K + K * D:] = torch.einsum("nk,ni,nj->kij", r, xp, xp).reshape(-1)
            return stats
        stats[K:2 * K] = torch.special.xlogy(r, r).sum(dim=0)
        stats[2 * K:2 * K + K * D] = (r.T @ xp).reshape(-1)
        stats[2 * K + K * D:] = torch.einsum("nk,ni,nj->kij", r, xp, xp).reshape(-1)
        return stats

    def estep_mstep(self, x, out=None):
        self.estep(x)
        return self.mstep(x, out)

    # ---- one pass policy for all ranks (semantics of gmmvb_set_shard / gmmvb_policy_export / gmmvb_policy_import):
    # the counters ride behind the statistics block through the iteration's ONE all-reduce; `policy_log` keeps what came
    # back so that a test can check every rank saw the same job-wide sums at every iteration
    def set_shard(self, global_rows, n_ranks):
        self._shard = (int(global_rows), int(n_ranks))
        self.policy_log = []

    def policy_export(self, tail):
        r = self.responsibilities()
        tail.zero_()
        tail[0] = float((r >= 2.0 ** -80).sum())       # active pairs of this rank's rows
        tail[1] = float(r.numel())                      # pairs evaluated exactly (the stand-in is dense)
        tail[9] = float(r.shape[0])                     # rows
        tail[10] = 1.0                                  # ranks
        tail[11] = 1.0                                  # ranks whose pass counted its pairs

    def policy_import(self, tail):
        self.policy_log.append(tail.clone())

    def split_stats(self, stats):
        K, D = self.K, self.D
        return (stats[:K], stats[K:2 * K], stats[2 * K:2 * K + K * D].view(K, D),
                stats[2 * K + K * D:].view(K, D, D))


def cpu_factory(K, D, x):
    return CpuDataPass(K, D, x)


# ---------------------------------------------------------------------------------------------------------------------
def cpu_small_fit(K, D, x, pivot, prior, init, n_restarts, init_type, max_itr, tolerance):
    """CPU stand-in for ``gmmvb_small_fit`` (csrc/small.hip) with the contract of include/gmmvb.h: every restart from its
    initial state to convergence, results in the ABI's ``out`` layout.  TEST INFRASTRUCTURE: built on the oracle's
    restatement of the reference, injected through ``LearnModel._small_fit_impl`` so that the host side of the
    small-problem path (draw order, winner rule, progress lines, attribute hand-over) is tested without a GPU."""
    import numpy as np
    from oracle import gmm_vb_oracle as orc
    x64 = np.asarray(x, dtype=np.float64)
    n = x64.shape[0]
    cut = np.cumsum([0, K, K * D, K, K, K * D * D, K])
    alpha0, m0, kappa0, nu0, w0_inv, _lnb = (prior[cut[j]:cut[j + 1]] for j in range(6))
    w0_inv = w0_inv.reshape(K, D, D)
    p = orc.Prior(alpha=alpha0.copy(), m=m0.reshape(K, D).copy(), kappa=kappa0.copy(), nu=nu0.copy(),
                  w=np.linalg.inv(w0_inv)).refresh()
    L = 2 + 8 + (max_itr + 1) + 7 * K + 2 * K * D + 3 * K * D * D
    out = np.zeros((n_restarts, L))
    r_all = np.zeros((n_restarts, n, K))
    keys = ("p_x", "p_z", "p_pi", "p_mu_lambda", "q_z", "q_pi", "q_mu_lambda", "vl")
    for i in range(n_restarts):
        q = orc.Posterior.from_prior(p)
        if init_type == 0:
            q.m = init[i, :K * D].reshape(K, D).copy()
            q.w_inv = init[i, K * D:].reshape(K, D, D).copy()
            q.w = np.linalg.inv(q.w_inv)
            q.refresh_lambda()
            st = orc.data_pass(x64, q)
        else:
            r = init[i].reshape(n, K)
            ns, x_bar, s = orc.m_step_stats(x64, r)
            st = orc.Stats(np.zeros_like(r), r, ns, x_bar, s)
        terms = orc.lower_bound(p, q, st)
        trace, conv = [terms["vl"]], False
        for _t in range(max_itr):
            before = trace[-1]
            orc.update_q_mu_lambda(p, q, st)
            orc.update_q_pi(p, q, st)
            st = orc.data_pass(x64, q, st.s)
            terms = orc.lower_bound(p, q, st)
            trace.append(terms["vl"])
            with np.errstate(divide="ignore", invalid="ignore"):
                if np.abs((terms["vl"] - before) / before) < tolerance:
                    conv = True
                    break
        out[i, 0], out[i, 1] = len(trace), float(conv)
        out[i, 2:10] = [terms[k] for k in keys]
        out[i, 10:10 + len(trace)] = trace
        out[i, 10 + max_itr + 1:] = np.concatenate([q.alpha, q.m.reshape(-1), q.kappa, q.nu, q.w_inv.reshape(-1), q.w.reshape(-1),
                                                    q.e_ln_pi, q.e_ln_lambda_det, q.ln_b_w_nu, st.ns, st.x_bar.reshape(-1),
                                                    st.s.reshape(-1)])
        r_all[i] = st.r
    return out, r_all
