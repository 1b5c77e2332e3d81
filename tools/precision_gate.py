#!/usr/bin/env python3
"""The evidence behind DESIGN.md section 1 ("why the arithmetic is fp64"): NumPy emulation of the data pass in reduced
precision against the oracle (test infrastructure; this tool is a measurement, not product code).

BASELINE.md section 3's parity gate: posterior hyper-parameters within 1e-5 relative of the reference after 10 VB
iterations on the first N_ref = 2e4 rows of the benchmark recipe (K=64, D=128, x stored f32).  north_star suggests f32
arithmetic for that configuration; this script runs the engine's formulation (Cholesky-whitened E-step
ln rho = c - |U (x - m)|^2 / 2, pivoted one-pass moments in the M-step) with each half in f32 or f64 and prints the largest
relative error of hn_w_mats / hn_m_vecs / hn_w_mats_inv against the oracle's all-f64 run of the reference's formulation.

    python tools/precision_gate.py [--rows 20000] [--classes 64] [--degree 128] [--iters 10] [--json out.json]

f32 E-step: x - m, U and the product y = U (x - m) in f32 with f32 accumulation (what an f32 MFMA does), |y|^2 summed in
f32, soft-max in f64 on the f32 ln rho.  f32 M-step: the moments of each chunk of 1024 rows accumulated in f32 (operands
r, x - pivot rounded to f32), chunks added in f64 ("fp32 chunks, fp64 across chunks").  The K-sized updates are the
oracle's f64 ones in every variant.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import gmm_vb_oracle as orc      # noqa: E402

LN_2PI = float(np.log(2.0 * np.pi))


def whitened_params(q):
    """(c, U) of the engine's formulation from the oracle's posterior: w_inv = G G^T, U = sqrt(nu) G^-1."""
    K, D = q.m.shape
    g = np.linalg.cholesky(q.w_inv)
    eye = np.eye(D)
    u = np.stack([np.sqrt(q.nu[k]) * np.linalg.solve(g[k], eye) for k in range(K)])
    c = q.e_ln_pi + (q.e_ln_lambda_det - D * LN_2PI - D / q.kappa) / 2.0
    return c, u


def e_step(x32, q, dt, center64=False):
    """ln rho (returned as f64) with the contraction in dtype `dt`; center64: x - m formed in f64, then rounded to dt."""
    c, u = whitened_params(q)
    N, K = x32.shape[0], q.m.shape[0]
    ln_rho = np.empty((N, K))
    xs = x32.astype(dt)
    for k in range(K):
        if center64:
            diff = (x32.astype(np.float64) - q.m[k]).astype(dt)
        else:
            diff = xs - q.m[k].astype(dt)                # rounded to dt before the product
        y = diff @ u[k].astype(dt).T                     # accumulation in dt (sgemm / dgemm)
        ln_rho[:, k] = (c[k].astype(dt) - dt(0.5) * np.sum(y * y, axis=1, dtype=dt)).astype(np.float64)
    r = np.exp(ln_rho - ln_rho.max(axis=1, keepdims=True))
    r /= r.sum(axis=1, keepdims=True)
    return ln_rho, r


def m_step(x32, r, pivot, dt, s_prev, chunk=1024):
    """One-pass pivoted moments with the per-chunk sums in dtype `dt`, chunks added in f64 -> (ns, x_bar, s)."""
    N, D = x32.shape
    K = r.shape[1]
    ns = np.zeros(K)
    a = np.zeros((K, D))
    B = np.zeros((K, D, D))
    xp_all = x32.astype(np.float64) - pivot
    for lo in range(0, N, chunk):
        xp = xp_all[lo:lo + chunk].astype(dt)
        rc = r[lo:lo + chunk].astype(dt)
        ns += rc.sum(axis=0, dtype=dt).astype(np.float64)
        a += (rc.T @ xp).astype(np.float64)
        for k in range(K):
            B[k] += ((rc[:, k] * xp.T) @ xp).astype(np.float64)
    s = np.array(s_prev)
    x_bar = np.zeros((K, D))
    for k in range(K):
        if ns[k] > 0:
            ab = a[k] / ns[k]
            x_bar[k] = pivot + ab
            s[k] = B[k] / ns[k] - np.outer(ab, ab)
    return ns, x_bar, s


def run(x32, K, D, iters, e_dt, m_dt, center64=False):
    x64 = x32.astype(np.float64)
    p = orc.Prior.default(K, D)
    q = orc.Posterior.from_prior(p)
    orc.init_subsampling(x64, q, np.random.default_rng(0))
    pivot = x64[:4096].mean(axis=0)
    s_prev = np.zeros((K, D, D))

    def data_pass(s_prev):
        ln_rho, r = e_step(x32, q, e_dt, center64)
        ns, x_bar, s = m_step(x32, r, pivot, m_dt, s_prev)
        return orc.Stats(ln_rho, r, ns, x_bar, s)

    st = data_pass(s_prev)
    for _ in range(iters):
        orc.update_q_mu_lambda(p, q, st)
        orc.update_q_pi(p, q, st)
        st = data_pass(st.s)
    return q


def reference(x32, K, D, iters):
    x64 = x32.astype(np.float64)
    p = orc.Prior.default(K, D)
    q = orc.Posterior.from_prior(p)
    orc.init_subsampling(x64, q, np.random.default_rng(0))
    st = orc.data_pass(x64, q)
    for _ in range(iters):
        orc.update_q_mu_lambda(p, q, st)
        orc.update_q_pi(p, q, st)
        st = orc.data_pass(x64, q, st.s)
    return q


def rel(a, b):
    return float(np.max(np.abs(a - b)) / np.max(np.abs(b)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=20_000)
    ap.add_argument("--classes", type=int, default=64)
    ap.add_argument("--degree", type=int, default=128)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    K, D, N = args.classes, args.degree, args.rows
    x32 = orc.synth_gmm(K, D, N, np.float32)
    t0 = time.perf_counter()
    ref = reference(x32, K, D, args.iters)
    rows = []
    print(f"K={K} D={D} N={N} iterations={args.iters}; gate: 1e-5 relative (BASELINE.md section 3)")
    print("| E-step arithmetic | M-step arithmetic | hn_w_mats | hn_m_vecs | hn_w_mats_inv | gate |")
    print("|---|---|---|---|---|---|")
    for e_dt, m_dt, c64 in ((np.float32, np.float32, False), (np.float32, np.float64, False), (np.float32, np.float64, True),
                            (np.float64, np.float32, False), (np.float64, np.float64, False)):
        q = run(x32, K, D, args.iters, e_dt, m_dt, c64)
        errs = dict(hn_w_mats=rel(q.w, ref.w), hn_m_vecs=rel(q.m, ref.m), hn_w_mats_inv=rel(q.w_inv, ref.w_inv))
        ok = max(errs.values()) < 1e-5
        rows.append(dict(estep=np.dtype(e_dt).name, mstep=np.dtype(m_dt).name, centred_in_f64=c64, passed=ok, **errs))
        print(f"| {np.dtype(e_dt).name} (whitened{', x - m formed in f64' if c64 else ''}) | {np.dtype(m_dt).name}{' chunks, f64 across chunks' if m_dt is np.float32 else ''} | "
              f"{errs['hn_w_mats']:.1e} | {errs['hn_m_vecs']:.1e} | {errs['hn_w_mats_inv']:.1e} | {'pass' if ok else 'FAIL'} |", flush=True)
    out = dict(K=K, D=D, rows=N, iterations=args.iters, tolerance=1e-5, variants=rows, seconds=time.perf_counter() - t0)
    if args.json:
        with open(args.json, "w") as f:
            json.dump(out, f, indent=1)
    return out


if __name__ == "__main__":
    main()
