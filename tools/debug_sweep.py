import os, sys, warnings
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import gmm_vb_oracle as orc
from bayesml_amd import _kside
from bayesml_amd import gaussianmixture as gm
from test_gpu_sparse_parity import env, VARIANTS, _oracle_post
K, D, N = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (24, 64, 24000)
x = orc.synth_gmm(K, D, N, np.float32); x64 = x.astype(np.float64)
dev = torch.device("cuda", 0)
with env(VARIANTS["force"]):
    m = gm.LearnModel(K, D, seed=0, device=dev, verbose=False)
    eng, xd = m._open(x)
prior = m._prior_tensors(dev)
q = m._init_subsampling(eng, xd, _kside.post_from_prior(prior), N)
s = torch.zeros(K, D, D, dtype=torch.float64, device=dev)
ns, x_bar, s, _h = m._pass(eng, xd, q, s)
prev_arg = eng.argmax().cpu().numpy()
for it in range(5):
    q_new = _kside.update_q(prior, ns, x_bar, s)
    hint = m._drift_hint(eng, xd, q, q_new)
    q = q_new
    ns, x_bar, s, _h = m._pass(eng, xd, q, s, hint=(*hint, float((hint[0] - hint[1] / 30.0).min())))
    lb = eng.ln_rho().cpu().numpy(); rb = eng.responsibilities().cpu().numpy()
    st = orc.data_pass(x64, _oracle_post(q)); la = st.ln_rho
    same = np.abs(la - lb) <= 1e-8 * np.maximum(1.0, np.abs(la))
    bad = (~same) & (lb < la)
    rows, ks = np.nonzero(bad)
    print(it, eng.launch_info.split(" ")[0], "gmean %.3f" % float((hint[0] - hint[1] / 30.0).min()), "bad", bad.sum(), "rows", np.unique(rows).size,
          "max|dr|", np.abs(rb - st.r).max(), "ns err", float(np.abs(ns.cpu().numpy() - st.ns).max()), eng.pass_counts())
    for r, k in list(zip(rows, ks))[:6]:
        print("   rec", eng.debug_record(int(r)))
        print("   row", r, "k", k, "la", la[r, k], "lb", lb[r, k], "best", la[r].max(), "argbest", la[r].argmax(), "prev_arg", prev_arg[r], "r", st.r[r, k])
    prev_arg = eng.argmax().cpu().numpy()
