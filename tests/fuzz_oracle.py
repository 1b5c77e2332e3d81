#!/usr/bin/env python3
"""[test utility, run by hand on a GPU box] Random small fits through the public classes against the CPU oracle:
python tests/fuzz_oracle.py [--cases 60] [--seed 1] [--models gmm,hmm,mvn[,hmm_long,hmm_states]]

Every case draws a model (Gaussian mixture, hidden Markov normal, single Gaussian), a shape (any c_degree up to 260, any
K the engine takes, row counts around the kernels' granules), a storage dtype, a prior (the defaults or random
hyper-parameters), the initialisation and the restart count, runs update_posterior on the GPU and the oracle's driver
on the host with the same seed, and compares the posterior hyper-parameters, the lower bound and the responsibilities.
A fit amplifies rounding differences, so a line carries the differences; a case is flagged above 1e-6 relative
(north_star asks 1e-5 on hyper-parameters; 1e-4 when the sample has fewer rows than four times c_degree).  The oracle is the checker here, as in tests/."""
import argparse
import json
import os
import sys
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GMMVB_DEBUG", "1")


NAN = {"dev": False, "ref": False}        # did the last case's device / oracle results hold a NaN


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    NAN["dev"] |= bool(np.isnan(a).any())
    NAN["ref"] |= bool(np.isnan(b).any())
    if np.isnan(b).any():                 # the oracle (like the reference) produced NaN: compare where it did not
        ok = ~np.isnan(b)
        if not ok.any():
            return 0.0
        a, b = a[ok], b[ok]
    return float(np.max(np.abs(a - b)) / max(1e-300, float(np.max(np.abs(b)))))


def random_niw(rng, K, D, default):
    if default:
        return dict(m=np.zeros((K, D)), kappa=np.ones(K), nu=np.full(K, float(D)), w=np.tile(np.eye(D), (K, 1, 1)))
    a = rng.standard_normal((K, D, D)) * (0.3 / np.sqrt(D))
    w = np.eye(D)[None] * rng.uniform(0.5, 2.0, (K, 1, 1)) + a @ a.transpose(0, 2, 1)
    return dict(m=rng.standard_normal((K, D)) * 0.5, kappa=rng.uniform(0.2, 3.0, K), nu=D - 1 + rng.uniform(0.1, 4.0, K),
                w=0.5 * (w + w.transpose(0, 2, 1)))


def draw_shape(rng, hmm):
    D = int(rng.choice([1, 2, 3, 7, 8, 15, 16, 17, 31, 32, 33, 48, 49, 63, 64, 65, 66, 80, 97, 127, 128, 129, 144, 160, 200, 241, 256, 260]))
    K = int(rng.choice([1, 2, 3, 4, 7, 8, 15, 16, 17, 24, 31, 32, 33, 63, 64, 65, 100] if not hmm else [1, 2, 3, 4, 7, 8, 15, 16, 17, 31, 32, 33, 48, 64]))
    N = int(rng.choice([1, 5, 63, 64, 65, 127, 128, 129, 255, 256, 257, 511, 513, 1000, 2049, 4097, 8191]))
    while K * N * D * D > 6e8 or K * N * D > 4e7:           # keep the host oracle in seconds
        N = max(1, N // 2)
    return K, D, N


def gmm_case(rng, dev):
    import torch
    from oracle import gmm_vb_oracle as orc
    from bayesml_amd import gaussianmixture as gm
    K, D, N = draw_shape(rng, False)
    default = bool(rng.random() < 0.5)
    c = dict(model="gmm", K=K, D=D, N=N, dtype=str(rng.choice(["float32", "float64"])), default_prior=default,
             init=str(rng.choice(["subsampling", "subsampling", "random_responsibility"])), num_init=int(rng.integers(1, 4)),
             iters=int(rng.integers(1, 8)), seed=int(rng.integers(0, 1000)), spread=float(rng.choice([2.0, 1.0, 0.5])),
             K_data=int(max(1, min(K, rng.choice([K, max(1, K // 2), 3])))))
    x = orc.synth_gmm(c["K_data"], D, N, np.dtype(c["dtype"]), seed=c["seed"], spread=c["spread"])
    pr = random_niw(rng, K, D, default)
    alpha = np.full(K, 0.5) if default else rng.uniform(0.2, 3.0, K)
    m = gm.LearnModel(K, D, h0_alpha_vec=alpha, h0_m_vecs=pr["m"], h0_kappas=pr["kappa"], h0_nus=pr["nu"], h0_w_mats=pr["w"],
                      seed=c["seed"], device=dev, verbose=False)
    kw = dict(max_itr=c["iters"], num_init=c["num_init"], tolerance=0.0, init_type=c["init"])
    m.update_posterior(x, **kw)
    p = orc.Prior(alpha=alpha.copy(), m=pr["m"].copy(), kappa=pr["kappa"].copy(), nu=pr["nu"].copy(), w=pr["w"].copy()).refresh()
    ref = orc.update_posterior(x.astype(np.float64), p, orc.Posterior.from_prior(p), np.random.default_rng(c["seed"]), **kw)
    hn = m.get_hn_params()
    q = ref.posterior
    d = dict(alpha=rel(hn["hn_alpha_vec"], q.alpha), m=rel(hn["hn_m_vecs"], q.m), kappa=rel(hn["hn_kappas"], q.kappa),
             nu=rel(hn["hn_nus"], q.nu), w=rel(hn["hn_w_mats"], q.w), vl=rel(m.vl, ref.vl),
             r=float(np.nanmax(np.abs(m.r_vecs - ref.stats.r))), ns=rel(m.ns, ref.stats.ns))
    info = getattr(m._engine, "launch_info", "small_fit") if m._engine is not None else "small_fit"
    # read-outs under the posterior the MODEL holds (no trajectory in between): latent variables of fresh rows, both
    # losses (ref:1178-1193), and the predictive parameters (ref:1064-1070)
    if not any(np.isnan(v).any() for v in hn.values()):
        own = orc.Posterior(alpha=hn["hn_alpha_vec"].copy(), m=hn["hn_m_vecs"].copy(), kappa=hn["hn_kappas"].copy(),
                            nu=hn["hn_nus"].copy(), w=hn["hn_w_mats"].copy(), w_inv=np.array(m.hn_w_mats_inv))
        own.refresh_pi()
        own.refresh_lambda()
        xs = orc.synth_gmm(c["K_data"], D, int(rng.choice([1, 7, 64, 300])), np.dtype(c["dtype"]), seed=c["seed"] + 1, spread=c["spread"])
        want = orc.estimate_latent_vars(xs.astype(np.float64), own, "squared")
        got = m.estimate_latent_vars(xs, loss="squared")
        d["latent_sq"] = float(np.max(np.abs(got - want)))
        top2 = np.sort(want, axis=1)[:, -2:] if K > 1 else np.stack([np.zeros(len(want)), np.ones(len(want))], 1)
        clear = top2[:, 1] - top2[:, 0] > 1e-6                # (a tie within rounding may go either way)
        z = m.estimate_latent_vars(xs, loss="0-1")
        d["latent_01"] = float(np.mean(z.argmax(axis=1)[clear] != want.argmax(axis=1)[clear])) if clear.any() else 0.0
        m.calc_pred_dist()
        pp, wp = m.get_p_params(), orc.predictive_params(own)
        d["pred"] = max(rel(pp[k], wp[k]) for k in pp)
    if m._engine is not None:
        m._engine.close()
    return c, d, str(info)[:70]


def hmm_case(rng, dev, long=False):
    from oracle import hmm_vb_oracle as orc
    from bayesml_amd import hiddenmarkovnormal as hm
    K, D, N = draw_shape(rng, True)
    N = max(N, 2)
    if long == "states":      # more states than the MFMA kernels' 64 / 128: hmm_wide.h and the generic recursions
        K = int(rng.choice([65, 100, 128, 129, 200, 256]))
        D = int(rng.choice([1, 2, 8, 16, 17, 33]))
        N = int(rng.choice([64, 513, 2049, 8191, 33_000]))
        long = N > 20_000
    elif long:        # sequences past 2^15 / 2^16 steps: the chunked forward-backward, forgetting and Viterbi-coalescence paths
        K = int(rng.choice([1, 2, 3, 5, 8, 16, 17, 32, 33, 64]))
        D = int(rng.choice([1, 2, 3, 8, 15, 16, 17, 32]))
        N = int(rng.choice([32_768, 40_000, 65_536, 70_001, 131_077]))
        if K > 32:
            N = min(N, 40_000)
    default = bool(rng.random() < 0.5)
    c = dict(model="hmm", K=K, D=D, N=N, dtype=str(rng.choice(["float32", "float64"])), default_prior=default,
             init=str(rng.choice(["subsampling", "subsampling", "random_responsibility"])), num_init=int(rng.integers(1, 3)),
             iters=int(rng.integers(1, 6)), seed=int(rng.integers(0, 1000)), stay=float(rng.choice([0.9, 0.5, 0.99])))
    if long:                                            # (the host oracle walks every step in Python)
        c["num_init"], c["iters"] = 1, min(c["iters"], 3)
    x = orc.synth_hmm(max(1, min(K, 8)), D, N, np.dtype(c["dtype"]), seed=c["seed"], stay=c["stay"])[0]
    pr = random_niw(rng, K, D, default)
    eta = np.full(K, 0.5) if default else rng.uniform(0.2, 3.0, K)
    zeta = np.full((K, K), 0.5) if default else rng.uniform(0.2, 3.0, (K, K))
    m = hm.LearnModel(K, D, h0_eta_vec=eta, h0_zeta_vecs=zeta, h0_m_vecs=pr["m"], h0_kappas=pr["kappa"], h0_nus=pr["nu"],
                      h0_w_mats=pr["w"], seed=c["seed"], device=dev, verbose=False)
    kw = dict(max_itr=c["iters"], num_init=c["num_init"], tolerance=0.0, init_type=c["init"])
    m.update_posterior(x, **kw)
    p = orc.HmmPrior(eta=eta.copy(), zeta=zeta.copy(), m=pr["m"].copy(), kappa=pr["kappa"].copy(), nu=pr["nu"].copy(),
                     w=pr["w"].copy()).refresh()
    ref = orc.update_posterior(x.astype(np.float64), p, orc.HmmPosterior.from_prior(p), np.random.default_rng(c["seed"]), **kw)
    hn = m.get_hn_params()
    q = ref.posterior
    d = dict(eta=rel(hn["hn_eta_vec"], q.eta), zeta=rel(hn["hn_zeta_vecs"], q.zeta), m=rel(hn["hn_m_vecs"], q.m),
             kappa=rel(hn["hn_kappas"], q.kappa), nu=rel(hn["hn_nus"], q.nu), w=rel(hn["hn_w_mats"], q.w), vl=rel(m.vl, ref.vl))
    info = str(getattr(m._engine, "launch_info", ""))[:70]
    # read-outs under the posterior the MODEL holds: Viterbi path and marginals of a fresh sequence (ref:1425-1499)
    if not any(np.isnan(v).any() for v in hn.values()):
        own = orc.HmmPosterior(hn["hn_eta_vec"].copy(), hn["hn_zeta_vecs"].copy(), hn["hn_m_vecs"].copy(), hn["hn_kappas"].copy(),
                               hn["hn_nus"].copy(), hn["hn_w_mats"].copy(), np.array(m.hn_w_mats_inv)).refresh()
        xs = orc.synth_hmm(max(1, min(K, 8)), D, int(rng.choice([2, 65, 300, 1500] if not long else [66_000, 70_001])), np.dtype(c["dtype"]),
                           seed=c["seed"] + 1, stay=c["stay"])[0]
        with np.errstate(all="ignore"):
            st = orc.data_pass(xs.astype(np.float64), own)
            path = orc.viterbi(xs.astype(np.float64), own)
            # the reference's rho = exp(ln rho) (ref:993) underflows - to denormals with a few bits, to exact zeros, to a NaN
            # pass - where a step's best ln rho lies below about -700: its marginals then say nothing (the engine scales
            # every row by its maximum; measured: gamma 0.199 where the oracle has 0.0 at K = 200, step 112 of 300)
            sound = float(orc.emission_ln_rho(xs.astype(np.float64), own).max(axis=1).min()) > -650.0
            if float(orc.emission_ln_rho(x.astype(np.float64), own).max(axis=1).min()) < -650.0:
                NAN["ref"] = True            # (the fit's own passes were in that regime too)
        if sound and not np.isnan(st.gamma).any():
            d["gamma"] = float(np.max(np.abs(m.estimate_latent_vars(xs, loss="squared", viterbi=False) - st.gamma)))
        got = m.estimate_latent_vars(xs, loss="0-1", viterbi=True)
        d["viterbi"] = float(np.mean(got.argmax(axis=1) != path.argmax(axis=1)))
    if m._engine is not None:
        m._engine.close()
    return c, d, info


def mvn_case(rng, dev):
    from oracle import mvn_oracle as orc
    from bayesml_amd import multivariate_normal as mvn
    _K, D, N = draw_shape(rng, False)
    N = min(max(N, 1) * int(rng.choice([1, 7, 40])), 200_000)
    c = dict(model="mvn", D=D, N=N, dtype=str(rng.choice(["float32", "float64"])), seed=int(rng.integers(0, 1000)))
    r = np.random.default_rng(c["seed"])
    x = (r.standard_normal((N, D)) * r.uniform(0.3, 2.0, D) + r.standard_normal(D)).astype(c["dtype"])
    pr = random_niw(rng, 1, D, bool(rng.random() < 0.5))
    m = mvn.LearnModel(D, h0_m_vec=pr["m"][0], h0_kappa=float(pr["kappa"][0]), h0_nu=float(pr["nu"][0]), h0_w_mat=pr["w"][0], device=dev)
    m.update_posterior(x)
    want = orc.update(pr["m"][0], float(pr["kappa"][0]), float(pr["nu"][0]), np.linalg.inv(pr["w"][0]), x)
    hn = m.get_hn_params()
    d = dict(m=rel(hn["hn_m_vec"], want[0]), kappa=rel(hn["hn_kappa"], want[1]), nu=rel(hn["hn_nu"], want[2]), w=rel(hn["hn_w_mat"], want[3]))
    return c, d, ""


def run(cases, seed, seconds=1e9, models=("gmm", "hmm", "mvn"), emit=print):
    """`cases` random cases from `seed`, the models in turn; returns (cases run, flagged lines, cases in which the oracle
    itself produced NaN).  tests/test_gpu_fuzz.py runs a few."""
    import torch
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(seed)
    fns = {"gmm": gmm_case, "hmm": hmm_case, "mvn": mvn_case, "hmm_long": lambda r, d: hmm_case(r, d, long=True),
           "hmm_states": lambda r, d: hmm_case(r, d, long="states")}
    t0, flagged, n, oracle_nan = time.time(), [], 0, 0
    for i in range(cases):
        if time.time() - t0 > seconds:
            break
        name = models[i % len(models)]
        n += 1
        NAN["dev"] = NAN["ref"] = False
        try:
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                c, d, info = fns[name](rng, dev)
        except Exception as e:                                         # noqa: BLE001  (the case is the finding)
            flagged.append(dict(model=name, draw=i, error=repr(e)[:500]))
            emit(json.dumps(flagged[-1]))
            continue
        soft = ("r", "latent_sq", "latent_01", "gamma", "viterbi")
        worst = max(v for k, v in d.items() if k not in soft)
        # an oracle NaN is the reference's own behaviour on that input (e.g. exp(ln rho) underflowing for every state of a
        # step, _hiddenmarkovnormal.py:993): reported, not flagged; a device NaN where the oracle has numbers is
        # (fewer rows than a few times c_degree: rank-deficient scatter matrices, |ln rho| of 1e4 and more - rounding
        # differences of the two formulations reach the responsibilities)
        thin = c["N"] < 4 * c["D"]
        tol, tol_r = (1e-4, 1e-2) if thin else (1e-6, 1e-5)
        if c["N"] < 16:       # a handful of rows: the subsampling initialisation inverts a rank-one scatter plus 1e-5 I
            tol = tol_r = 10.0   # (ref:792-794; condition 1e10 and more) - the trajectory is rounding noise on both sides
        bad = (not NAN["ref"]) and (not np.isfinite(worst) or worst > tol or d.get("r", 0.0) > tol_r or NAN["dev"])
        # the read-outs start from the same posterior on both sides: tight whatever the sample was like (a Viterbi path may
        # part at an exact tie of two states; more than one step in a hundred is not that)
        bad = bad or d.get("latent_sq", 0.0) > 1e-7 or d.get("latent_01", 0.0) > 0.0 or d.get("gamma", 0.0) > 1e-7 or d.get("viterbi", 0.0) > 0.01
        oracle_nan += NAN["ref"]
        line = dict(case=c, diff={k: float(f"{v:.1e}") for k, v in d.items()}, info=info, oracle_nan=NAN["ref"],
                    device_nan=NAN["dev"], flag=bool(bad))
        if bad:
            flagged.append(line)
        emit(json.dumps(line))
    emit(json.dumps(dict(cases=n, flagged=len(flagged), oracle_nan=oracle_nan, seconds=round(time.time() - t0, 1))))
    return n, flagged, oracle_nan


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=60)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--seconds", type=float, default=1e9)
    ap.add_argument("--models", default="gmm,hmm,mvn")
    a = ap.parse_args()
    run(a.cases, a.seed, a.seconds, tuple(a.models.split(",")), emit=lambda t: print(t, flush=True))
    return 0


if __name__ == "__main__":
    sys.exit(main())
