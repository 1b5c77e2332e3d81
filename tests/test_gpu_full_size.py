"""Size-independent properties at BASELINE.json's FULL sizes (the oracle cannot run there):
  C3  GMM K=64, D=128, N=1e7 (f32 rows):  sum_k r_nk = 1 per row (=> sum ns = N), exact symmetry of B,
      run-to-run determinism, linearity over row shards (two halves add up to the whole to 1e-12),
      the centred second moments are shift-consistent (trace identity against a direct torch reduction).
  C5  HMM K=32, D=16, T=1e7:  sum ns = T, sum ms = T - 1 (every xi_t sums to one), rows of gamma sum to one.
  C3 / C4 through the SPARSE path the benchmark times (bound pass -> sweeps of carried bounds -> settled rows and their
      int8 proof round -> cache of single-component rows -> regrouped rows -> list M-step), driven exactly like
      update_posterior's loop: after >= 10 hinted iterations the statistics block of the last (carried) pass must equal,
      to rounding, what the DENSE kernels give for the same parameters on the same rows; sum ns = N; the counters show
      that every one of those mechanisms really ran.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _gmm_state(K, D, dev, seed=3):
    from bayesml_amd import _kside
    gen = torch.Generator(device=dev).manual_seed(seed)
    mu = 2.0 * torch.randn(K, D, dtype=torch.float64, device=dev, generator=gen)
    p = _kside.prior_from_numpy(np.full(K, 0.5), np.zeros((K, D)), np.ones(K), np.full(K, float(D)),
                                np.tile(np.eye(D), (K, 1, 1)), dev)
    q = _kside.post_from_prior(p)
    q.m = mu + 0.05 * torch.randn(K, D, dtype=torch.float64, device=dev, generator=gen)
    q.nu = q.nu + 50.0
    q.kappa = q.kappa + 50.0
    return mu, _kside.features(q)


def test_gmm_c3_full_size_properties():
    from bayesml_amd._engine import DataPass
    K, D, N = 64, 128, 10_000_000
    dev = torch.device("cuda", 0)
    mu, q = _gmm_state(K, D, dev)
    gen = torch.Generator(device=dev).manual_seed(11)
    x = torch.empty((N, D), dtype=torch.float32, device=dev)
    for lo in range(0, N, 1 << 20):
        hi = min(N, lo + (1 << 20))
        z = torch.randint(0, K, (hi - lo,), device=dev, generator=gen)
        x[lo:hi] = (mu[z] + torch.randn(hi - lo, D, dtype=torch.float64, device=dev, generator=gen)).to(torch.float32)
    piv = x[:4096].to(torch.float64).mean(dim=0)
    eng = DataPass(K, D, x.dtype, N, dev)
    eng.set_pivot(piv)
    eng.prepare_rows(x)
    eng.set_params(q.c, q.m, q.u)
    full = eng.estep_mstep(x)
    ns, h, a, B = eng.split_stats(full)
    assert abs(float(ns.sum()) - N) < 1e-5                      # responsibilities sum to one on every row
    assert torch.equal(B, B.transpose(1, 2))                    # exact symmetry
    assert float(h.sum()) <= 0.0
    assert torch.equal(full, eng.estep_mstep(x))                # deterministic
    # trace identity: sum_k tr B_k = sum_n |x_n - pivot|^2 (because sum_k r_nk = 1), against torch in f64 chunks
    tr = 0.0
    for lo in range(0, N, 1 << 21):
        d = x[lo:lo + (1 << 21)].to(torch.float64) - piv
        tr += float((d * d).sum())
    assert abs(float(torch.diagonal(B, dim1=1, dim2=2).sum()) - tr) < 1e-9 * tr
    r_head = eng.responsibilities(0, 4096)
    assert float((r_head.sum(dim=1) - 1.0).abs().max()) < 1e-12
    # linearity over row shards: what the multi-GPU all-reduce relies on
    half = N // 2 + 12345
    acc = torch.zeros_like(full)
    for lo, hi in ((0, half), (half, N)):
        part = DataPass(K, D, x.dtype, hi - lo, dev)
        part.set_pivot(piv)
        part.prepare_rows(x[lo:hi])
        part.set_params(q.c, q.m, q.u)
        acc += part.estep_mstep(x[lo:hi])
        part.close()
    rel = float((acc - full).abs().max() / full.abs().max())
    assert rel < 1e-12, rel
    eng.close()


def test_hmm_c5_full_size_properties():
    from bayesml_amd import _kside
    from bayesml_amd._engine import DataPass
    K, D, T = 32, 16, 10_000_000
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev).manual_seed(5)
    mu = 3.0 * torch.randn(K, D, dtype=torch.float64, device=dev, generator=gen)
    jump = torch.rand(T, device=dev, generator=gen) >= 0.9
    jump[0] = True
    idx = torch.arange(T, device=dev)
    last = torch.cummax(torch.where(jump, idx, torch.zeros_like(idx)), dim=0).values
    z = torch.randint(0, K, (T,), device=dev, generator=gen)[last]
    x = (mu[z] + torch.randn(T, D, dtype=torch.float64, device=dev, generator=gen)).to(torch.float32)
    p = _kside.hmm_prior_from_numpy(np.full(K, 0.5), np.full((K, K), 0.5), np.zeros((K, D)), np.ones(K),
                                    np.full(K, float(D)), np.tile(np.eye(D), (K, 1, 1)), dev)
    q = _kside.hmm_post_from_prior(p)
    q.m = mu + 0.05
    q.nu = q.nu + 20.0
    q.kappa = q.kappa + 20.0
    q.zeta = q.zeta + 5.0 * torch.eye(K, dtype=torch.float64, device=dev)
    q = _kside.hmm_features(q)
    eng = DataPass(K, D, x.dtype, T, dev)
    eng.set_pivot(x[:4096].to(torch.float64).mean(dim=0))
    eng.prepare_rows(x)
    eng.enable_hmm()
    eng.set_params(q.c, q.m, q.u)
    eng.estep(x)
    ms, g0, gl, lnc = eng.forward_backward(q.pi_tilde, q.a_tilde)
    ns, h, a, B = eng.split_stats(eng.mstep(x))
    assert abs(float(ns.sum()) - T) < 1e-5
    assert abs(float(ms.sum()) - (T - 1)) < 1e-5
    assert abs(float(g0.sum()) - 1.0) < 1e-12 and abs(float(gl.sum()) - 1.0) < 1e-12
    assert np.isfinite(float(lnc)) and float(h.sum()) < 0.0
    g = eng.responsibilities(T - 4096, 4096)
    assert float((g.sum(dim=1) - 1.0).abs().max()) < 1e-12
    # transition counts are consistent with state occupancies: row sums of ms = ns - gamma_last, column sums = ns - gamma_0
    assert float((ms.sum(dim=1) - (ns - gl)).abs().max()) < 1e-6
    assert float((ms.sum(dim=0) - (ns - g0)).abs().max()) < 1e-6
    eng.close()


def _sparse_vs_dense(K, D, N, iters, tol, tile_rows=0):
    """bench.py's workload at full size: `iters` VB iterations as update_posterior runs them, then the last pass's
    statistics against a dense data pass on the same parameters.  tile_rows: through resident row tiles of that size."""
    import bench
    from bayesml_amd._engine import DataPass, TiledDataPass
    dev = torch.device("cuda", 0)
    x = bench.device_rows(K, D, N, torch.float32, dev, bench.SEED + 1, 2.0)
    with bench.env_vars(BAYESML_AMD_TILE_ROWS=str(tile_rows) if tile_rows else None, BAYESML_AMD_TILE_RESIDENT="1"):
        w = bench.Workload(K, D, x, dev, None)
    if tile_rows:
        assert isinstance(w.eng, TiledDataPass) and w.eng.resident and w.eng.n_tiles == (N + tile_rows - 1) // tile_rows
    seen = dict(settled=0.0, proof=0.0, cached=0.0)
    for _ in range(iters):
        w.step()
        wk = w.eng.work()
        seen["settled"] = max(seen["settled"], wk["settled_rows"])
        seen["proof"] = max(seen["proof"], wk["proof_pairs"])
        if wk["accumulated"] >= 0:
            seen["cached"] = max(seen["cached"], wk["active"] - wk["accumulated"])
    counts = w.eng.pass_counts()
    tiles = getattr(w.eng, "n_tiles", 1)
    assert "estep_sweep" in w.eng.launch_info.split("|")[0], w.eng.launch_info      # the compared pass lived on carried bounds
    assert counts["estep_bound"] >= tiles and counts["estep_sweep"] >= 4 * tiles and counts["mstep_list"] >= 5 * tiles, counts
    assert counts["estep_gather"] >= 8 * tiles and w.eng.regroup_count >= tiles, counts
    assert seen["cached"] > 0.3 * N and seen["settled"] > 0.1 * N and seen["proof"] > 0, seen
    wk = w.eng.work()
    assert wk["evaluated"] < 0.1 * N * K and 0.99 * N <= wk["active"] < 3.0 * N, wk
    sparse = w.ks.stats.clone()
    q = w.q
    ns = sparse[:K]
    assert abs(float(ns.sum()) - N) < 1e-6
    with bench.env_vars(GMMVB_ESTEP_PRUNE="0", GMMVB_MSTEP_SPARSE="0"):       # switches are read at workspace creation
        ref = DataPass(K, D, x.dtype, N, dev)
    ref.set_pivot(w.eng.pivot)
    ref.prepare_rows(w.xd)
    ref.set_params(q.c, q.m, q.u)
    dense = ref.estep_mstep(w.xd)
    assert ref.pass_counts()["estep_dense"] == 1 and ref.pass_counts()["mstep_dense"] == 1
    for name, a_, b_ in zip(("ns", "h", "a", "B"), w.eng.split_stats(sparse), ref.split_stats(dense)):
        rel = float((a_ - b_).abs().max() / b_.abs().max())
        assert rel < tol, (name, rel)
    # hard assignments of the sparse pass (settled rows included, in the caller's row order) against the dense pass's
    zs = w.eng.argmax(0, 1 << 16).cpu()
    zd = ref.argmax(0, 1 << 16).cpu()
    assert torch.equal(zs, zd)
    rs = w.eng.responsibilities(N - 4096, 4096)
    rd = ref.responsibilities(N - 4096, 4096)
    assert float((rs - rd).abs().max()) < 1e-9
    ref.close()
    w.close()


def test_gmm_c3_full_size_sparse():
    _sparse_vs_dense(64, 128, 10_000_000, iters=12, tol=1e-12)


def test_gmm_c4_shard_full_size_sparse():
    """One GPU's shard of config 4: K=256, D=64, 1.25e7 rows (N = 1e8 over 8 GPUs)."""
    _sparse_vs_dense(256, 64, 12_500_000, iters=14, tol=1e-12)


def test_gmm_c4_shape_through_resident_tiles():
    """Config 4's shape through four resident row tiles (a workspace per tile sharing the pass-local buffers,
    gmmvb_workspace_create_tile): every tile runs its own bound pass, regrouping, sweeps, proof rounds and cache, loses the
    shared buffers to the next tile after every M-step - and the summed statistics equal the dense kernels'."""
    _sparse_vs_dense(256, 64, 5_000_000, iters=14, tol=1e-12, tile_rows=1_250_048)

