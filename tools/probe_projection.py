"""How many (row, component) pairs does the stateless table of csrc/project.h leave undecided, and why?  Runs the benchmark's
loop at a reduced N, and at chosen iterations restates the bound in f64 torch on a sample of rows with (a) the exact
lambda_min(U_k^T U_k), (b) its Gershgorin lower bound, (c) without that term, (d) the plain triangle bound on centre
distances - next to what the kernel counted (gmmvb_last_work out[7]).  Developer tool; prints one JSON line per probe."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def probe(w, x, rows=4096, state={}):
    q = w.ks.q
    K, D = w.K, w.D
    dev = x.device
    idx = state.setdefault('idx', torch.randperm(x.shape[0], device=dev)[:rows])
    xs = x[idx].double()
    m, u, c = q.m, q.u, q.c
    y = torch.einsum("kij,nkj->nki", u, xs[:, None, :] - m[None, :, :])
    lnrho = c[None, :] - 0.5 * (y * y).sum(-1)                                        # exact [rows, K]
    best, j = lnrho.max(dim=1)
    thr = best - 80 * np.log(2)
    lam = torch.einsum("kri,krj->kij", u, u)
    ev = torch.linalg.eigvalsh(lam)
    lmin = ev[:, 0]
    gersh = (torch.diagonal(lam, dim1=1, dim2=2) - (lam.abs().sum(-1) - torch.diagonal(lam, dim1=1, dim2=2).abs())).min(dim=1).values.clamp(min=0)
    dm = m[j][:, None, :] - m[None, :, :]                                                 # [rows, K, D] m_j - m_k
    v = torch.einsum("kri,nki->nkr", u, dm)
    s2 = (v * v).sum(-1)
    g = torch.einsum("kri,nkr->nki", u, v)
    dx = xs - m[j]
    p = (g * dx[:, None, :]).sum(-1)
    e = (dx * dx).sum(-1)
    out = {"true_relevant_others": float(((lnrho >= thr[:, None]).sum(1) - 1).double().mean())}
    for name, h in (("exact_lmin", lmin), ("gershgorin", gersh), ("no_e_term", torch.zeros_like(lmin))):
        ub = c[None, :] - 0.5 * (s2 + 2 * p + h[None, :] * e[:, None])
        left = (ub >= thr[:, None])
        left[torch.arange(rows), j] = False
        out["left_" + name] = float(left.sum(1).double().mean())
        out["slack_" + name + "_median"] = float((ub - lnrho)[left].median()) if left.any() else 0.0
    # (d) the triangle inequality on centre distances: || U_k (x - m_k) || >= s_jk - sqrt(lmax_k / lmin_j) d_j
    dj = torch.sqrt((2 * (c[j] - best)).clamp(min=0))
    sig = torch.sqrt(ev[None, :, -1] / ev[j, 0][:, None])
    tri = (torch.sqrt(s2) - sig * dj[:, None]).clamp(min=0)
    left = (c[None, :] - 0.5 * tri * tri) >= thr[:, None]
    left[torch.arange(rows), j] = False
    out["left_triangle"] = float(left.sum(1).double().mean())
    # the same with a STALE reference: the component that dominated the row at the first probe (the kernel's reference is
    # the group the row was sorted into at the last regrouping)
    j0 = state.setdefault("j0", j.clone())
    out["rows_whose_dominant_component_changed"] = float((j0 != j).double().mean())
    dm0 = m[j0][:, None, :] - m[None, :, :]
    v0 = torch.einsum("kri,nki->nkr", u, dm0)
    g0 = torch.einsum("kri,nkr->nki", u, v0)
    dx0 = xs - m[j0]
    ub0 = c[None, :] - 0.5 * ((v0 * v0).sum(-1) + 2 * (g0 * dx0[:, None, :]).sum(-1))
    left0 = ub0 >= thr[:, None]
    left0[torch.arange(rows), j] = False
    out["left_no_e_term_stale_reference"] = float(left0.sum(1).double().mean())
    out["left_no_e_term_stale_reference_rows_unchanged"] = float(left0[j0 == j].sum(1).double().mean()) if (j0 == j).any() else None
    out["lmin_range"] = [float(lmin.min()), float(lmin.median()), float(ev[:, -1].max())]
    out["gersh_range"] = [float(gersh.min()), float(gersh.median())]
    return out


def main():
    K, D, n = int(os.environ.get("PROBE_K", 64)), int(os.environ.get("PROBE_D", 128)), int(os.environ.get("PROBE_N", 1_000_000))
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    x = bench.device_rows(K, D, n, torch.float32, dev, bench.SEED + 1, 2.0)
    w = bench.Workload(K, D, x, dev, None)
    for it in range(1, 31):
        w.step()
        if it in (6, 10, 15, 20, 25, 30):
            wk = w.eng.work()
            rec = {"iteration": it, "kernel": w.eng.launch_info.split(" ")[0], "table_left_per_row": wk["table_left"] / n,
                   "proof_pairs_per_row": wk["proof_pairs"] / n, "active_per_row": wk["active"] / n,
                   "settled_rows": wk["settled_rows"] / n}
            rec.update(probe(w, x))
            print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
