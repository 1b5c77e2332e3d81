"""ctypes binding of ``libgmmvb.so`` (C ABI in ``include/gmmvb.h``) over PyTorch-ROCm tensors.

PyTorch is plumbing here: it owns device memory and the HIP stream; every N-sized computation
is done by the hand-written gfx950 kernels behind the C ABI.  There is NO CPU fallback: if the
shared library or a GPU is missing, construction raises ``EngineUnavailableError``.
"""
from __future__ import annotations

import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# (BAYESML_AMD_LIB: a developer switch to load a variant build of the same ABI, e.g. for an A/B measurement on one box)
LIB_PATH = os.environ.get("BAYESML_AMD_LIB") or os.path.join(_HERE, "csrc", "libgmmvb.so")

GMMVB_F32, GMMVB_F64 = 0, 1
POLICY_LEN = 16           # include/gmmvb.h: GMMVB_POLICY_LEN
_STATUS = {0: "GMMVB_OK", 1: "GMMVB_EINVAL", 2: "GMMVB_EUNSUPPORTED", 3: "GMMVB_EHIP", 4: "GMMVB_ENOMEM",
           5: "GMMVB_ESTATE"}

# every symbol include/gmmvb.h declares: name -> (restype, argtypes)
_vp, _i64, _int = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
SYMBOLS = {
    "gmmvb_abi_version": (_int, []),
    "gmmvb_last_error": (ctypes.c_char_p, []),
    "gmmvb_stats_len": (_i64, [_int, _int]),
    "gmmvb_workspace_create": (_int, [_int, _int, _int, _i64, ctypes.POINTER(_vp)]),
    "gmmvb_workspace_create_tile": (_int, [_vp, _i64, ctypes.POINTER(_vp)]),
    "gmmvb_workspace_destroy": (_int, [_vp]),
    "gmmvb_workspace_bytes": (_i64, [_vp]),
    "gmmvb_set_pivot": (_int, [_vp, _vp, _vp]),
    "gmmvb_set_params": (_int, [_vp, _vp, _vp, _vp, _vp]),
    "gmmvb_set_drift": (_int, [_vp, _vp, _vp, _vp, ctypes.c_double, _vp]),
    "gmmvb_forget": (_int, [_vp]),
    "gmmvb_regroup_count": (_i64, [_vp]),
    "gmmvb_debug_record": (_int, [_vp, _i64, ctypes.POINTER(ctypes.c_double)]),
    "gmmvb_debug_proof": (_int, [_vp, _int, _i64, _vp, _vp, _vp]),
    "gmmvb_wants_drift": (_int, [_vp, _i64]),
    "gmmvb_prepare_rows": (_int, [_vp, _vp, _i64, _i64, _vp]),
    "gmmvb_estep": (_int, [_vp, _vp, _i64, _i64, _vp]),
    "gmmvb_load_responsibilities": (_int, [_vp, _vp, _i64, _vp]),
    "gmmvb_mstep": (_int, [_vp, _vp, _i64, _i64, _vp, _vp]),
    "gmmvb_estep_mstep": (_int, [_vp, _vp, _i64, _i64, _vp, _vp]),
    "gmmvb_responsibilities": (_int, [_vp, _i64, _i64, _vp, _vp]),
    "gmmvb_ln_rho": (_int, [_vp, _i64, _i64, _vp, _vp]),
    "gmmvb_argmax": (_int, [_vp, _i64, _i64, _vp, _vp]),
    "gmmvb_last_launch_info": (ctypes.c_char_p, [_vp]),
    "gmmvb_pass_counts": (_int, [_vp, ctypes.POINTER(_i64)]),
    "gmmvb_kside_factor": (_int, [_int, _int, _vp, _vp, _vp, _vp, _vp]),
    "gmmvb_kside_step": (_int, [_int, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _int, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gmmvb_kside_drift": (_int, [_int, _int, _vp, _vp, _vp, _vp, _vp, _vp, _int, _int, _vp, _vp, _vp, _vp, _vp]),
    "gmmvb_last_sparsity": (ctypes.c_int, [_vp, _vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]),
    "gmmvb_last_work": (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_double)]),
    "gmmvb_comm_unique_id": (_int, [_vp]),
    "gmmvb_comm_create": (_int, [_vp, _int, _int, ctypes.POINTER(_vp)]),
    "gmmvb_comm_destroy": (_int, [_vp]),
    "gmmvb_allreduce_stats": (_int, [_vp, _vp, _i64, _vp]),
    "gmmvb_stats_packed_len": (_i64, [_int, _int]),
    "gmmvb_stats_pack": (_int, [_int, _int, _vp, _vp, _vp]),
    "gmmvb_stats_unpack": (_int, [_int, _int, _vp, _vp, _vp]),
    "gmmvb_set_shard": (_int, [_vp, _i64, _int]),
    "gmmvb_policy_export": (_int, [_vp, _vp, _vp]),
    "gmmvb_policy_import": (_int, [_vp, _vp, _vp]),
    "gmmvb_small_supported": (_int, [_int, _int, _i64]),
    "gmmvb_small_out_len": (_i64, [_int, _int, _int]),
    "gmmvb_small_fit": (_int, [_int, _int, _int, _vp, _i64, _i64, _vp, _vp, _int, _int, _vp, _int, ctypes.c_double, _vp, _vp, _vp]),
    "hmmvb_kside_dirichlet": (_int, [_int, _vp, _vp, ctypes.c_double, ctypes.c_double] + [_vp] * 18),
    "hmmvb_out_len": (_i64, [_int]),
    "hmmvb_enable": (_int, [_vp]),
    "hmmvb_forward_backward": (_int, [_vp, _i64, _vp, _vp, _vp, _vp]),
    "hmmvb_debug_readout": (_int, [_vp, _int, _i64, _i64, _vp, _vp]),
    "hmmvb_readout": (_int, [_vp, _int, _i64, _i64, _vp, _vp, _vp]),
    "hmmvb_emission_target": (_int, [_vp, _int, _vp]),
    "hmmvb_skip_h": (_int, [_vp, _int]),
    "hmmvb_last_boundary_pass": (_int, [_vp]),
    "hmmvb_last_viterbi_pass": (_int, [_vp]),
    "hmmvb_viterbi": (_int, [_vp, _i64, _vp, _vp, _vp, _vp]),
    "gmmvb_profile_enable": (_int, [_vp, _int]),
    "gmmvb_profile_last_ms": (_int, [_vp, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float)]),
    "gmmvb_profile_spans": (_int, [_vp, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(_int)]),
    "gmmvb_profile_span_name": (ctypes.c_char_p, [_int]),
    "gmmvb_policy_calibrate": (_int, [_vp, _int]),
    "gmmvb_policy_table": (_int, [_vp, ctypes.POINTER(ctypes.c_double)]),
    "gmmvb_sample_latent": (_int, [_int, _vp, ctypes.c_uint64, _i64, _i64, _vp, _vp]),
    "gmmvb_sample_chain_work_bytes": (_i64, [_int, _i64]),
    "gmmvb_sample_chain": (_int, [_int, _vp, _vp, ctypes.c_uint64, _i64, _vp, _vp, _i64, _vp]),
    "gmmvb_sample_emissions_work_bytes": (_i64, [_int, _i64]),
    "gmmvb_sample_emissions": (_int, [_int, _int, _vp, _vp, _vp, ctypes.c_uint64, _i64, _i64, _int, _vp, _i64, _vp, _i64, _vp]),
}


class EngineUnavailableError(RuntimeError):
    """The HIP extension or the GPU is missing; the product path has no fallback."""


class EngineError(RuntimeError):
    """A C-ABI call returned a non-zero status."""


class EngineLimitError(ValueError):
    """The model's shape is outside what this version of the engine supports (the reference has no such limit)."""


MAX_MFMA_DEGREE = 128     # up to here the data pass runs on the MFMA kernels; beyond, on the plain f64 kernels of csrc/generic.h
MAX_HMM_FAST_CLASSES = 64 # up to here the HMM recursions run chunk-parallel on MFMA; beyond, sequentially (csrc/hmm_generic.h)
# What the correctness kernels can still LAUNCH (derived from their shapes, so that a too large model is refused at
# construction instead of failing with GMMVB_EHIP inside update_posterior after the data is on the GPU):
#   hmm_generic.h: hmm_xi_generic_kernel has grid.y = (ceil(K / 16))^2 <= 65535  ->  K <= 16 * 255;
#                  hmm_seq_shape(K).lds_bytes = (3 K + 1056) * 8 (+ 4 KB static in Viterbi) <= 160 KB  ->  K <= 6304
#   generic.h:     generic_rows(D) = 1 still keeps one centred row, D * 8 bytes, in LDS (<= 150 KB)  ->  D <= 19200
MAX_HMM_CLASSES = 16 * 255
MAX_DEGREE = 19200


def check_limits(c_degree: int, c_num_classes: int = 1, hmm: bool = False):
    """Raised at model construction, so that an unsupported shape does not surface as an EngineError from inside
    update_posterior after the sample matrix has already been copied to the GPU.  (c_degree has no limit: above 128 the
    engine switches to its generic f64 kernels, and an HMM with more than 64 states to sequential recursions - same
    results, far slower - up to the sizes those kernels can launch.)"""
    if c_degree > MAX_DEGREE:
        raise EngineLimitError(f"bayesml_amd supports c_degree <= {MAX_DEGREE} in this version (got {c_degree}); "
                               "bayesml itself has no such limit")
    if hmm and c_num_classes > MAX_HMM_CLASSES:
        raise EngineLimitError(f"bayesml_amd.hiddenmarkovnormal supports c_num_classes <= {MAX_HMM_CLASSES} in this "
                               f"version (got {c_num_classes}); bayesml itself has no such limit")


_lib = None


def load_library(path: str = LIB_PATH) -> ctypes.CDLL:
    """dlopen the in-tree library and declare every prototype (works without a GPU)."""
    global _lib
    if _lib is not None and path == LIB_PATH:
        return _lib
    if not os.path.exists(path):
        raise EngineUnavailableError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C bayesml_amd/csrc`). bayesml_amd has no CPU fallback for the data pass.")
    lib = ctypes.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    if path == LIB_PATH:
        _lib = lib
    return lib


def _check(lib, rc, what):
    if rc != 0:
        msg = lib.gmmvb_last_error()
        raise EngineError(f"{what}: {_STATUS.get(rc, rc)}: {msg.decode() if msg else ''}")


def _f64(t: torch.Tensor, shape, device) -> torch.Tensor:
    t = torch.as_tensor(t, dtype=torch.float64, device=device).contiguous()
    if tuple(t.shape) != tuple(shape):
        raise ValueError(f"expected shape {tuple(shape)}, got {tuple(t.shape)}")
    return t


class RcclComm:
    """The C ABI's row-shard communicator (gmmvb_comm_*): RCCL all-reduce of the statistics block, enqueued on the
    current torch stream.  ``bootstrap`` hands rank 0's 128-byte id to the other ranks; by default it is a
    ``torch.distributed`` broadcast over the default process group (any backend)."""

    def __init__(self, rank: int, world: int, device, bootstrap=None, group=None):
        """``rank`` / ``world`` are the ranks of ``group`` (default: the whole job); the id travels over that group."""
        self.lib = load_library()
        self.device = torch.device(device)
        buf = (ctypes.c_ubyte * 128)()
        if rank == 0:
            _check(self.lib, self.lib.gmmvb_comm_unique_id(buf), "gmmvb_comm_unique_id")
        if bootstrap is None:
            import torch.distributed as dist
            t = torch.tensor(list(buf), dtype=torch.uint8)
            if dist.get_backend(group) == "nccl":
                t = t.to(self.device)
            # (broadcast takes the GLOBAL rank of the source: rank 0 of the group)
            dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            raw = bytes(t.cpu().tolist())
        else:
            raw = bootstrap(bytes(buf))
        buf = (ctypes.c_ubyte * 128).from_buffer_copy(raw)
        handle = _vp()
        with torch.cuda.device(self.device):
            _check(self.lib, self.lib.gmmvb_comm_create(buf, int(world), int(rank), ctypes.byref(handle)), "gmmvb_comm_create")
        self._comm = handle

    def all_reduce_(self, t: torch.Tensor) -> torch.Tensor:
        if t.dtype != torch.float64 or not t.is_contiguous() or t.device != self.device:
            raise ValueError("the statistics block must be a contiguous float64 tensor on the communicator's device")
        with torch.cuda.device(self.device):
            st = _vp(torch.cuda.current_stream(self.device).cuda_stream)
            _check(self.lib, self.lib.gmmvb_allreduce_stats(self._comm, t.data_ptr(), t.numel(), st), "gmmvb_allreduce_stats")
        return t

    def close(self):
        if getattr(self, "_comm", None):
            torch.cuda.synchronize(self.device)
            self.lib.gmmvb_comm_destroy(self._comm)
            self._comm = None

    def __del__(self):
        # Never destroy the communicator from a finaliser at interpreter exit: by then torch may have torn down its process
        # group or the HIP runtime, the order differs between ranks, and a hang inside ncclCommDestroy cannot be caught.
        # Leaking it at exit is safe; RowShard.close() / RcclComm.close() is the orderly way.
        import sys
        if sys is None or sys.is_finalizing():
            return
        try:
            self.close()
        except Exception:      # noqa: BLE001
            pass


def kside_factor(w_inv: torch.Tensor):
    """(G, G^-1, ln det) of a batch of SPD matrices on the GPU (gmmvb_kside_factor): W^-1 = G G^T, lower triangular.
    Plain kernels on the current stream - no MAGMA / rocSOLVER handles, so the K-side can be captured in a hipGraph."""
    lib = load_library()
    K, D, _ = w_inv.shape
    w_inv = w_inv.contiguous()
    g = torch.empty_like(w_inv)
    g_inv = torch.empty_like(w_inv)
    logdet = torch.empty(K, dtype=torch.float64, device=w_inv.device)
    with torch.cuda.device(w_inv.device):
        st = _vp(torch.cuda.current_stream(w_inv.device).cuda_stream)
        _check(lib, lib.gmmvb_kside_factor(K, D, w_inv.data_ptr(), g.data_ptr(), g_inv.data_ptr(), logdet.data_ptr(), st),
               "gmmvb_kside_factor")
    return g, g_inv, logdet


class _PriorView(ctypes.Structure):
    _fields_ = [(n, _vp) for n in ("alpha", "m", "kappa", "nu", "w_inv", "ln_b_w_nu")] + [("ln_c_alpha", ctypes.c_double)]


class _PostView(ctypes.Structure):
    FIELDS = ("alpha", "m", "kappa", "nu", "w_inv", "w", "u", "u_inv", "e_ln_pi", "e_ln_lambda_det", "ln_b_w_nu", "c")
    _fields_ = [(n, _vp) for n in FIELDS]


def post_view(q) -> _PostView:
    return _PostView(*(getattr(q, f).data_ptr() for f in _PostView.FIELDS))


def prior_view(p) -> _PriorView:
    return _PriorView(p.alpha.data_ptr(), p.m.data_ptr(), p.kappa.data_ptr(), p.nu.data_ptr(), p.w_inv.data_ptr(),
                      p.ln_b_w_nu.data_ptr(), float(p.ln_c_alpha))


def kside_step(K, D, prior_v, q_v, qn_v, stats, pivot, s_prev, ns, x_bar, s, want_drift, gamma, delta, big, scal, scratch):
    """gmmvb_kside_step on the current stream of ``stats``' device (every tensor contiguous float64 there)."""
    lib = load_library()
    dev = stats.device
    with torch.cuda.device(dev):
        st = _vp(torch.cuda.current_stream(dev).cuda_stream)
        _check(lib, lib.gmmvb_kside_step(K, D, ctypes.byref(prior_v), ctypes.byref(q_v), ctypes.byref(qn_v), stats.data_ptr(),
                                         pivot.data_ptr(), s_prev.data_ptr(), ns.data_ptr(), x_bar.data_ptr(), s.data_ptr(),
                                         int(want_drift), gamma.data_ptr(), delta.data_ptr(), big.data_ptr(), scal.data_ptr(),
                                         scratch.data_ptr(), st), "gmmvb_kside_step")


def hmm_kside_dirichlet(K, prior, ln_c_eta0, ln_c_zeta0, q, qn, fb, ns, scal_nw, h_scale, scal):
    """hmmvb_kside_dirichlet on the current stream: the eta / zeta half of the HMM's K-side (contiguous float64 device tensors)."""
    lib = load_library()
    dev = fb.device
    with torch.cuda.device(dev):
        st = _vp(torch.cuda.current_stream(dev).cuda_stream)
        _check(lib, lib.hmmvb_kside_dirichlet(
            K, prior.eta.data_ptr(), prior.zeta.data_ptr(), ln_c_eta0, ln_c_zeta0, q.eta.data_ptr(),
            q.zeta.data_ptr(), q.ln_pi_tilde.data_ptr(), q.ln_a_tilde.data_ptr(), q.ln_c_zeta_sum.data_ptr(), fb.data_ptr(),
            ns.data_ptr(), scal_nw.data_ptr(), h_scale.data_ptr(), qn.eta.data_ptr(), qn.zeta.data_ptr(), qn.ln_pi_tilde.data_ptr(),
            qn.pi_tilde.data_ptr(), qn.ln_a_tilde.data_ptr(), qn.a_tilde.data_ptr(), qn.ln_c_zeta_sum.data_ptr(), scal.data_ptr(),
            st), "hmmvb_kside_dirichlet")


def stats_triangle(pack: bool, K: int, D: int, src: torch.Tensor, dst: torch.Tensor):
    """gmmvb_stats_pack / gmmvb_stats_unpack on the current stream of ``src``' device: the statistics block <-> the block
    on the wire, [ns | h | a | upper triangles of B] (contiguous float64 device tensors that do not overlap)."""
    lib = load_library()
    dev = src.device
    with torch.cuda.device(dev):
        st = _vp(torch.cuda.current_stream(dev).cuda_stream)
        if pack:
            _check(lib, lib.gmmvb_stats_pack(K, D, src.data_ptr(), dst.data_ptr(), st), "gmmvb_stats_pack")
        else:
            _check(lib, lib.gmmvb_stats_unpack(K, D, src.data_ptr(), dst.data_ptr(), st), "gmmvb_stats_unpack")


def kside_drift(q_old, q_new, squarings: int, squarings_big: int):
    """(gamma, delta, big_gamma) of the update q_old -> q_new on the GPU in one launch (gmmvb_kside_drift)."""
    lib = load_library()
    K, D = q_old.m.shape
    dev = q_old.m.device
    t = [x.contiguous() for x in (q_old.u, q_old.u_inv, q_old.m, q_new.u, q_new.u_inv, q_new.m)]
    out = torch.empty(4, K, dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        st = _vp(torch.cuda.current_stream(dev).cuda_stream)
        _check(lib, lib.gmmvb_kside_drift(K, D, *(x.data_ptr() for x in t), int(squarings), int(squarings_big),
                                          out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), out[3].data_ptr(), st),
               "gmmvb_kside_drift")
    # u_new u_old^-1 = I + E: 1 -/+ ||E|| bounds its extreme singular values, tightly once the components hardly move
    e = out[3] * (1.0 + 1e-9)
    gamma = torch.maximum(out[0], (1.0 - e) * (1.0 - 1e-9))
    big = torch.minimum(out[2], (1.0 + e) * (1.0 + 1e-9))
    return gamma, out[1], big


class DataPass:
    """One workspace = one (K, D, x dtype, max rows) problem on one GPU.

    ``estep``/``mstep`` enqueue on the current torch stream of ``device`` and return device
    tensors; nothing here synchronises with the host.
    """

    def __init__(self, K: int, D: int, x_dtype: torch.dtype, max_rows: int, device=None, tile_of: "DataPass" = None):
        """``tile_of``: another DataPass of the same (K, D, dtype) whose pass-local buffers this one shares
        (gmmvb_workspace_create_tile: a further tile of a row-tiled job)."""
        self.lib = load_library()
        if not torch.cuda.is_available():
            raise EngineUnavailableError("no ROCm GPU visible to PyTorch; the GMM-VB data pass runs on MI355X only")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if x_dtype not in (torch.float32, torch.float64):
            raise ValueError("x dtype must be float32 or float64")
        self.K, self.D, self.x_dtype, self.max_rows = int(K), int(D), x_dtype, int(max_rows)
        self.stats_len = int(self.lib.gmmvb_stats_len(self.K, self.D))
        handle = _vp()
        with torch.cuda.device(self.device):
            if tile_of is not None:
                if (tile_of.K, tile_of.D, tile_of.x_dtype, tile_of.device) != (self.K, self.D, x_dtype, self.device):
                    raise ValueError("a further tile has the shape, dtype and device of the group's first workspace")
                rc = self.lib.gmmvb_workspace_create_tile(tile_of._ws, self.max_rows, ctypes.byref(handle))
            else:
                rc = self.lib.gmmvb_workspace_create(self.K, self.D, GMMVB_F64 if x_dtype == torch.float64 else GMMVB_F32,
                                                     self.max_rows, ctypes.byref(handle))
        _check(self.lib, rc, "gmmvb_workspace_create_tile" if tile_of is not None else "gmmvb_workspace_create")
        self._ws = handle
        self.emission_fused = False       # (HMM) see emission_target
        self._keep = []      # tensors whose memory an enqueued kernel may still read

    # -- lifetime
    def close(self):
        if getattr(self, "_ws", None):
            torch.cuda.synchronize(self.device)
            self.lib.gmmvb_workspace_destroy(self._ws)
            self._ws = None

    def __del__(self):
        try:
            self.close()
        except Exception:      # noqa: BLE001  (interpreter shutdown)
            pass

    # -- helpers
    def _stream(self):
        return _vp(torch.cuda.current_stream(self.device).cuda_stream)

    def _x(self, x: torch.Tensor):
        if x.device != self.device or x.dtype != self.x_dtype or x.dim() != 2 or x.shape[1] != self.D:
            raise ValueError(f"x must be a [{'*'}, {self.D}] {self.x_dtype} tensor on {self.device}")
        if x.stride(1) != 1:
            x = x.contiguous()
        return x, int(x.stride(0)) if x.shape[0] > 1 else self.D

    @property
    def workspace_bytes(self) -> int:
        return int(self.lib.gmmvb_workspace_bytes(self._ws))

    @property
    def launch_info(self) -> str:
        s = self.lib.gmmvb_last_launch_info(self._ws)
        return s.decode() if s else ""

    PASS_NAMES = ("estep_dense", "estep_bound", "estep_carried", "estep_fell_back_dense", "estep_sweep",
                  "mstep_dense", "mstep_list", "estep_gather")

    def pass_counts(self) -> dict:
        """Launch counts since the workspace was created (gmmvb_pass_counts)."""
        out = (_i64 * 8)()
        _check(self.lib, self.lib.gmmvb_pass_counts(self._ws, out), "gmmvb_pass_counts")
        return dict(zip(self.PASS_NAMES, (int(v) for v in out)))

    def sparsity(self):
        """(active pairs, exactly evaluated pairs) of the last E-step; see gmmvb_last_sparsity."""
        a, e = ctypes.c_double(), ctypes.c_double()
        with torch.cuda.device(self.device):
            _check(self.lib, self.lib.gmmvb_last_sparsity(self._ws, self._stream(), ctypes.byref(a), ctypes.byref(e)),
                   "gmmvb_last_sparsity")
        return float(a.value), float(e.value)

    def work(self) -> dict:
        """Pairs of the last E-step (gmmvb_last_work): active, evaluated exactly, accumulated by the list M-step, and the
        rows the E-step did not evaluate at all (settled), pairs of the int8 proof round, and the share of the per-pair bound
        array the sweep went through (sweep_share; -1: the last E-step was not a lazy sweep), or - after a projected sweep
        (csrc/project.h) - the pairs its stateless table did not clear (table_left; -1: the last E-step was none)."""
        out = (ctypes.c_double * 8)()
        _check(self.lib, self.lib.gmmvb_last_work(self._ws, out), "gmmvb_last_work")
        tiles = (self.rows + 255) // 256 if getattr(self, "rows", 0) else 0
        share = float(out[6]) / (tiles * self.K) if out[6] >= 0 and tiles else -1.0
        return dict(active=float(out[0]), evaluated=float(out[1]), accumulated=float(out[2]), settled_rows=float(out[3]),
                    early_exits=float(out[4]), proof_pairs=float(out[5]), sweep_share=share, table_left=float(out[7]))

    def policy_table(self) -> dict:
        """Unit costs (ns per pair) and thresholds of the pass policy in force, and which of them this workspace has measured
        on its own passes (gmmvb_policy_table; csrc/policy.h)."""
        out = (ctypes.c_double * 16)()
        _check(self.lib, self.lib.gmmvb_policy_table(self._ws, out), "gmmvb_policy_table")
        keys = ("dense_e_ns", "dense_m_ns", "bound_ns", "exact_ns", "proof_ns", "list_m_ns", "literal_dense_e_ns",
                "literal_dense_m_ns", "literal_bound_ns", "prune_below", "dense_again_above", "list_m_below")
        d = {k: float(out[i]) for i, k in enumerate(keys)}
        d.update(measured=int(out[12]), discarded=int(out[13]), calibrating=bool(out[14]))
        return d

    def profile(self, on: bool = True):
        _check(self.lib, self.lib.gmmvb_profile_enable(self._ws, int(on)), "gmmvb_profile_enable")

    def last_kernel_ms(self):
        """(estep_mfma_f64 ms, mstep_mfma_f64 ms) of the last launches, from HIP events on the launch stream."""
        e, m = ctypes.c_float(), ctypes.c_float()
        _check(self.lib, self.lib.gmmvb_profile_last_ms(self._ws, ctypes.byref(e), ctypes.byref(m)),
               "gmmvb_profile_last_ms")
        return float(e.value), float(m.value)

    def kernel_spans(self) -> dict:
        """{group: (ms, launch groups)} of the last estep + mstep (gmmvb_profile_spans; needs profile(True))."""
        ms, cnt = (ctypes.c_float * 8)(), (_int * 8)()
        _check(self.lib, self.lib.gmmvb_profile_spans(self._ws, ms, cnt), "gmmvb_profile_spans")
        out = {}
        for i in range(8):
            name = self.lib.gmmvb_profile_span_name(i).decode()
            if name and cnt[i]:
                out[name] = (float(ms[i]), int(cnt[i]))
        return out

    # -- C ABI
    def set_pivot(self, pivot):
        p = _f64(pivot, (self.D,), self.device)
        with torch.cuda.device(self.device):
            _check(self.lib, self.lib.gmmvb_set_pivot(self._ws, p.data_ptr(), self._stream()), "gmmvb_set_pivot")
        self.pivot = p

    def prepare_rows(self, x: torch.Tensor):
        """Build the centred f64 copy the M-step streams (once per sample matrix, after set_pivot)."""
        x, ldx = self._x(x)
        with torch.cuda.device(self.device):
            _check(self.lib, self.lib.gmmvb_prepare_rows(self._ws, x.data_ptr(), ldx, x.shape[0], self._stream()),
                   "gmmvb_prepare_rows")

    def set_params(self, c, m, u):
        c = _f64(c, (self.K,), self.device)
        m = _f64(m, (self.K, self.D), self.device)
        u = _f64(u, (self.K, self.D, self.D), self.device)
        with torch.cuda.device(self.device):
            _check(self.lib, self.lib.gmmvb_set_params(self._ws, c.data_ptr(), m.data_ptr(), u.data_ptr(),
                                                       self._stream()), "gmmvb_set_params")
        self._keep = [c, m, u]

    def set_shard(self, global_rows: int, n_ranks: int):
        """This workspace holds one shard of a row-sharded job: pass-policy decisions use job-wide numbers only
        (gmmvb_set_shard), fed by policy_export / all-reduce / policy_import around the iteration's collective."""
        _check(self.lib, self.lib.gmmvb_set_shard(self._ws, int(global_rows), int(n_ranks)), "gmmvb_set_shard")

    def policy_export(self, tail: torch.Tensor):
        with torch.cuda.device(self.device):
            _check(self.lib, self.lib.gmmvb_policy_export(self._ws, tail.data_ptr(), self._stream()), "gmmvb_policy_export")

    def policy_import(self, tail: torch.Tensor):
        with torch.cuda.device(self.device):
            _check(self.lib, self.lib.gmmvb_policy_import(self._ws, tail.data_ptr(), self._stream()), "gmmvb_policy_import")

    def wants_drift(self, n_rows: int) -> bool:
        return bool(self.lib.gmmvb_wants_drift(self._ws, int(n_rows)))

    def set_drift(self, gamma, delta, big_gamma, typical_gamma: float = -1.0):
        """Hint for the pruned E-step (gmmvb_set_drift): call before the set_params of the updated parameters.
        ``typical_gamma``: mean of gamma as a host float if it is at hand without an extra sync, else <= 0."""
        g = _f64(gamma, (self.K,), self.device)
        d = _f64(delta, (self.K,), self.device)
        G = _f64(big_gamma, (self.K,), self.device)
        with torch.cuda.device(self.device):
            _check(self.lib, self.lib.gmmvb_set_drift(self._ws, g.data_ptr(), d.data_ptr(), G.data_ptr(),
                                                      float(typical_gamma), self._stream()), "gmmvb_set_drift")

    @property
    def regroup_count(self) -> int:
        return int(self.lib.gmmvb_regroup_count(self._ws))

    def debug_record(self, row: int) -> dict:
        out = (ctypes.c_double * 26)()
        _check(self.lib, self.lib.gmmvb_debug_record(self._ws, int(row), out), "gmmvb_debug_record")
        v = list(out)
        return dict(k=[int(x) for x in v[:8]], d=[round(x, 3) for x in v[8:16]], B=v[16], exact=int(v[17]), sel=int(v[18]),
                    flags=int(v[19]), khat=int(v[20]), lse=v[21], masks=[int(x) for x in v[22:26]])

    def debug_proof(self, k: int, n_rows: int):
        """(upper f32, lower f64) bounds of ln rho_nk for every prepared row from the proof round's int8 kernel."""
        ub = torch.empty(n_rows, dtype=torch.float32, device=self.device)
        lb = torch.empty(n_rows, dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            _check(self.lib, self.lib.gmmvb_debug_proof(self._ws, int(k), int(n_rows), ub.data_ptr(), lb.data_ptr(),
                                                        self._stream()), "gmmvb_debug_proof")
        return ub, lb

    def forget(self):
        """The next parameters are unrelated to the last E-step's (a new restart): see gmmvb_forget."""
        _check(self.lib, self.lib.gmmvb_forget(self._ws), "gmmvb_forget")

    def estep(self, x: torch.Tensor):
        x, ldx = self._x(x)
        with torch.cuda.device(self.device):
            _check(self.lib, self.lib.gmmvb_estep(self._ws, x.data_ptr(), ldx, x.shape[0], self._stream()),
                   "gmmvb_estep")
        self.rows = x.shape[0]

    def load_responsibilities(self, r: torch.Tensor):
        r = torch.as_tensor(r, dtype=torch.float64, device=self.device).contiguous()
        if r.dim() != 2 or r.shape[1] != self.K:
            raise ValueError("r must be [n, K]")
        with torch.cuda.device(self.device):
            _check(self.lib, self.lib.gmmvb_load_responsibilities(self._ws, r.data_ptr(), r.shape[0], self._stream()),
                   "gmmvb_load_responsibilities")
        self.rows = r.shape[0]
        self._keep.append(r)

    def _stats_out(self, out):
        if out is None:
            return torch.empty(self.stats_len, dtype=torch.float64, device=self.device)
        if out.dtype != torch.float64 or out.numel() != self.stats_len or not out.is_contiguous() or out.device != self.device:
            raise ValueError("out must be a contiguous float64 tensor of stats_len elements on the engine's device")
        return out

    def mstep(self, x: torch.Tensor, out: torch.Tensor = None) -> torch.Tensor:
        x, ldx = self._x(x)
        stats = self._stats_out(out)
        with torch.cuda.device(self.device):
            _check(self.lib, self.lib.gmmvb_mstep(self._ws, x.data_ptr(), ldx, x.shape[0], stats.data_ptr(),
                                                  self._stream()), "gmmvb_mstep")
        return stats

    def estep_mstep(self, x: torch.Tensor, out: torch.Tensor = None) -> torch.Tensor:
        x, ldx = self._x(x)
        stats = self._stats_out(out)
        with torch.cuda.device(self.device):
            _check(self.lib, self.lib.gmmvb_estep_mstep(self._ws, x.data_ptr(), ldx, x.shape[0], stats.data_ptr(),
                                                        self._stream()), "gmmvb_estep_mstep")
        self.rows = x.shape[0]
        return stats

    # -- HMM
    def enable_hmm(self):
        with torch.cuda.device(self.device):
            _check(self.lib, self.lib.hmmvb_enable(self._ws), "hmmvb_enable")

    def hmm_skip_h(self, skip: bool = True):
        """hmmvb_skip_h: the statistics of the HMM passes carry h = 0 (the caller takes sum gamma ln rho from the moments,
        ``_kside.sum_gamma_ln_rho``) and the M-step does not read the ln rho array."""
        _check(self.lib, self.lib.hmmvb_skip_h(self._ws, 1 if skip else 0), "hmmvb_skip_h")

    def last_boundary_pass(self) -> int:
        """hmmvb_last_boundary_pass: -1 chunk products, 0 the forgetting pass stood, 1 it ran and the products path behind it."""
        return int(self.lib.hmmvb_last_boundary_pass(self._ws))

    def last_viterbi_pass(self) -> int:
        """hmmvb_last_viterbi_pass: -1 chunk matrices / sequential, 0 the coalescence pass stood, 1 chunk matrices behind it."""
        return int(self.lib.hmmvb_last_viterbi_pass(self._ws))

    def emission_target(self, fused: bool) -> bool:
        """hmmvb_emission_target: ``fused`` asks the following ``estep`` calls to write rho' straight into the
        forward-backward buffers (no ln rho array: ``viterbi`` / ``ln_rho`` need a pass with ``fused=False``, and the
        statistics' h block is 0).  Returns whether the library does so for this shape."""
        eff = ctypes.c_int(0)
        _check(self.lib, self.lib.hmmvb_emission_target(self._ws, 1 if fused else 0, ctypes.byref(eff)),
               "hmmvb_emission_target")
        self.emission_fused = bool(eff.value)
        return self.emission_fused

    def forward_backward(self, pi_tilde, a_tilde, out: torch.Tensor = None):
        """(ms [K, K], gamma_0 [K], gamma_last [K], sum ln c) of the pass over the rows of the last estep;
        leaves gamma as the workspace's responsibilities (mstep / responsibilities / argmax use it).
        ``out``: a float64 device buffer of hmmvb_out_len(K) elements to write them into (views of it are returned)."""
        K = self.K
        pi = _f64(pi_tilde, (K,), self.device)
        a = _f64(a_tilde, (K, K), self.device)
        n_out = int(self.lib.hmmvb_out_len(K))
        if out is None:
            out = torch.empty(n_out, dtype=torch.float64, device=self.device)
        elif out.dtype != torch.float64 or out.numel() != n_out or not out.is_contiguous() or out.device != self.device:
            raise ValueError("out must be a contiguous float64 tensor of hmmvb_out_len(K) elements on the engine's device")
        with torch.cuda.device(self.device):
            _check(self.lib, self.lib.hmmvb_forward_backward(self._ws, self.rows, pi.data_ptr(), a.data_ptr(),
                                                             out.data_ptr(), self._stream()), "hmmvb_forward_backward")
        self._keep = self._keep[-3:] + [pi, a]
        return out[:K * K].view(K, K), out[K * K:K * K + K], out[K * K + K:K * K + 2 * K], out[K * K + 2 * K]

    def viterbi(self, ln_pi_tilde, ln_a_tilde) -> torch.Tensor:
        """Most probable state path (int32 [rows]) for the emission ln rho of the last estep."""
        K = self.K
        pi = _f64(ln_pi_tilde, (K,), self.device)
        a = _f64(ln_a_tilde, (K, K), self.device)
        z = torch.empty(self.rows, dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            _check(self.lib, self.lib.hmmvb_viterbi(self._ws, self.rows, pi.data_ptr(), a.data_ptr(), z.data_ptr(),
                                                    self._stream()), "hmmvb_viterbi")
        torch.cuda.current_stream(self.device).synchronize()     # pi / a must outlive the kernels
        return z

    def hmm_readout(self, what: str, row0=0, n=None, a_tilde=None) -> torch.Tensor:
        """alpha / beta [n, K] or xi [n, K, K] of the last forward_backward for rows [row0, row0 + n) (hmmvb_readout)."""
        code = {"alpha": 0, "beta": 1, "xi": 3}[what]
        n = self.rows - row0 if n is None else n
        K = self.K
        out = torch.empty((n, K, K) if code == 3 else (n, K), dtype=torch.float64, device=self.device)
        a = _f64(a_tilde, (K, K), self.device) if code == 3 else None
        with torch.cuda.device(self.device):
            _check(self.lib, self.lib.hmmvb_readout(self._ws, code, row0, n, a.data_ptr() if a is not None else None,
                                                    out.data_ptr(), self._stream()), "hmmvb_readout")
        if a is not None:
            torch.cuda.current_stream(self.device).synchronize()          # a must outlive the kernel
        return out

    def hmm_debug(self, what, row0=0, n=None):
        n = self.rows - row0 if n is None else n
        Kp = 16 * ((self.K + 15) // 16)
        out = torch.empty((n, Kp) if what == 0 else (n,), dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            _check(self.lib, self.lib.hmmvb_debug_readout(self._ws, what, row0, n, out.data_ptr(), self._stream()),
                   "hmmvb_debug_readout")
        if what == 0:          # lane order -> natural state order
            pos = [(s & ~15) + 4 * ((s & 15) & 3) + ((s & 15) >> 2) for s in range(self.K)]
            out = out[:, pos]
        return out

    def split_stats(self, stats: torch.Tensor):
        """[ns | h | a | B] views of a statistics block."""
        K, D = self.K, self.D
        return (stats[:K], stats[K:2 * K], stats[2 * K:2 * K + K * D].view(K, D),
                stats[2 * K + K * D:].view(K, D, D))

    def _readout(self, fn, name, row0, n, dtype=torch.float64, cols=None):
        n = self.rows - row0 if n is None else n
        out = torch.empty((n, self.K) if cols is None else (n,), dtype=dtype, device=self.device)
        with torch.cuda.device(self.device):
            _check(self.lib, fn(self._ws, row0, n, out.data_ptr(), self._stream()), name)
        return out

    def responsibilities(self, row0=0, n=None) -> torch.Tensor:
        return self._readout(self.lib.gmmvb_responsibilities, "gmmvb_responsibilities", row0, n)

    def ln_rho(self, row0=0, n=None) -> torch.Tensor:
        return self._readout(self.lib.gmmvb_ln_rho, "gmmvb_ln_rho", row0, n)

    def argmax(self, row0=0, n=None) -> torch.Tensor:
        return self._readout(self.lib.gmmvb_argmax, "gmmvb_argmax", row0, n, dtype=torch.int32, cols=1)


class TiledDataPass:
    """The same surface as ``DataPass`` for a sample matrix whose per-pair workspace does not fit the GPU (a workspace keeps
    5.5 KB per row at K = 256, D = 64: N = 1e8 would need 550 GB next to the 25.6 GB matrix - the reference has no such
    coupling, it allocates its [N, K] arrays on the host, ``_gaussianmixture.py:835-836``).

    The rows are cut into tiles of ``tile_rows`` rows that go through the data pass one after the other; the statistics
    blocks add up (they are linear in the rows, like over row shards).  Two forms:

    * ``resident=True`` (round 4): one workspace PER TILE (``gmmvb_workspace_create_tile``).  What a tile carries from one VB
      iteration to the next - f32 bounds, records, digit planes, settled rows and their cache, row order, policy counters
      (1.7 KB per row) - stays in its workspace; the pass-local two thirds (f64 ln rho, lists, centred copy, slabs) exist
      once, sized for one tile.  Every tile sweeps its carried bounds exactly like a row shard of a multi-GPU job does.
    * ``resident=False``: ONE workspace for all tiles, for matrices where even that does not fit.  Nothing per-pair survives
      from one tile to the next, so a tile's E-step is the dense kernel while the responsibilities are dense and a fresh
      int8 bound pass afterwards (the tiles share the pass policy through the job-wide counters of gmmvb_set_shard).

    Read-outs re-run the E-step of the tiles they touch."""

    def __init__(self, K, D, x_dtype, n_rows, device, tile_rows, resident=False):
        self.K, self.D, self.x_dtype, self.max_rows = int(K), int(D), x_dtype, int(n_rows)
        self.tile_rows = int(min(tile_rows, n_rows))
        self.n_tiles = (self.max_rows + self.tile_rows - 1) // self.tile_rows
        self.resident = bool(resident)
        self.inner = DataPass(K, D, x_dtype, self.tile_rows, device)
        self.tiles = [self.inner]
        self.device, self.lib, self.stats_len = self.inner.device, self.inner.lib, self.inner.stats_len
        try:
            if self.resident:
                for t in range(1, self.n_tiles):
                    lo, hi = self._tile(t)
                    self.tiles.append(DataPass(K, D, x_dtype, hi - lo, device, tile_of=self.inner))
            self._tmp = torch.zeros(self.stats_len, dtype=torch.float64, device=self.device)
            self._tail_tmp = torch.zeros(POLICY_LEN, dtype=torch.float64, device=self.device)
            self._tail_acc = torch.zeros(POLICY_LEN, dtype=torch.float64, device=self.device)
        except (RuntimeError, EngineError):       # (out of memory next to the workspaces that just fitted)
            self.close()
            raise
        self._tail_in = None           # job-wide counters of the previous pass (summed over tiles, and over ranks by the caller)
        self._ranks = 1
        self._global_rows = self.max_rows
        self._x = self._r = None
        self._params = None
        self._held = None              # tile whose E-step output the pass-local buffers currently hold
        self._src = None               # what the read-outs describe: 'loaded' responsibilities or the 'estep' under the parameters
        self._infos, self._work, self._spars, self._ms, self._spans = [], None, None, (0.0, 0.0), {}
        self._lazy = 0                 # resident tiles whose counters of the last pass have not been read back yet
        self.rows = 0
        # one workspace for all tiles: its counters describe another tile every time, so the tiles decide together from
        # job-wide sums.  Resident tiles each have their own lagged counters (and decide for themselves) unless the job is
        # also sharded over ranks (set_shard below).
        self._joint_policy = not self.resident
        if self._joint_policy:
            self.inner.set_shard(self._global_rows, self.n_tiles)

    # -- plumbing shared with DataPass
    _ws = property(lambda self: self.inner._ws)
    PASS_NAMES = DataPass.PASS_NAMES

    def close(self):
        for t in reversed(getattr(self, "tiles", [])):
            t.close()

    def _tile(self, t):
        lo = t * self.tile_rows
        return lo, min(self.max_rows, lo + self.tile_rows)

    def _eng(self, t):
        return self.tiles[t] if self.resident else self.inner

    @property
    def workspace_bytes(self):
        return sum(t.workspace_bytes for t in self.tiles)

    @property
    def launch_info(self):
        kind = "resident tiles" if self.resident else "tiles through one workspace"
        return f"{self.n_tiles} {kind} x {self.tile_rows} rows: " + (self._infos[0] if self._infos else "")

    @property
    def regroup_count(self):
        return sum(t.regroup_count for t in self.tiles)

    def pass_counts(self):
        out = dict.fromkeys(self.PASS_NAMES, 0)
        for t in self.tiles:
            for k, v in t.pass_counts().items():
                out[k] += v
        return out

    # counters and event times of the last data pass, summed over its tiles.  One workspace: read back tile by tile inside
    # the pass (a host synchronisation per tile).  Resident tiles keep their own counters and events: read here, on demand,
    # after the pass (the driver synchronises once per VB iteration anyway).
    def _gather(self):
        if self._lazy:
            acc = self._new_acc()
            for t in range(self._lazy):
                self._add_tile(acc, self.tiles[t], *self._tile(t))
            self._finish(acc)
            self._lazy = 0

    @staticmethod
    def _new_acc():
        return dict(active=0.0, evaluated=0.0, accumulated=0.0, settled_rows=0.0, early_exits=0.0, proof_pairs=0.0,
                    counted=True, e_ms=0.0, m_ms=0.0, spans={})

    def _add_tile(self, acc, eng, lo, hi):
        eng.rows = hi - lo
        a, e = eng.sparsity()
        wk = eng.work()
        acc["counted"] = acc["counted"] and a >= 0
        acc["active"] += max(a, 0.0)
        acc["evaluated"] += e
        for key in ("accumulated", "settled_rows", "early_exits", "proof_pairs"):
            acc[key] += max(wk[key], 0.0)
        if getattr(self, "_prof", False):
            k = eng.last_kernel_ms()
            acc["e_ms"] += k[0]
            acc["m_ms"] += k[1]
            for g, (ms, cnt) in eng.kernel_spans().items():
                o = acc["spans"].get(g, (0.0, 0))
                acc["spans"][g] = (o[0] + ms, o[1] + cnt)

    def _finish(self, acc):
        counted = acc["counted"]
        self._spars = (acc["active"] if counted else -1.0, acc["evaluated"])
        self._work = dict(active=acc["active"] if counted else -1.0, evaluated=acc["evaluated"],
                          accumulated=acc["accumulated"] if counted else -1.0, settled_rows=acc["settled_rows"],
                          early_exits=acc["early_exits"], proof_pairs=acc["proof_pairs"], sweep_share=-1.0)
        self._ms, self._spans = (acc["e_ms"], acc["m_ms"]), acc["spans"]

    def last_kernel_ms(self):
        self._gather()
        return self._ms

    def kernel_spans(self):
        self._gather()
        return dict(self._spans)

    def sparsity(self):
        self._gather()
        return self._spars if self._spars is not None else (-1.0, 0.0)

    def policy_table(self):
        return self.inner.policy_table()

    def work(self):
        self._gather()
        return self._work if self._work is not None else dict(active=-1.0, evaluated=0.0, accumulated=-1.0, settled_rows=0.0,
                                                              early_exits=0.0, proof_pairs=0.0, sweep_share=-1.0)

    def split_stats(self, stats):
        return self.inner.split_stats(stats)

    # -- state
    def set_pivot(self, pivot):
        for t in self.tiles:
            t.set_pivot(pivot)
        self.pivot = self.inner.pivot

    def prepare_rows(self, x):
        self._x, self._held = x, None
        if self.resident:                      # digit planes and row order belong to the tile's workspace: made once
            if x.shape[0] > self.max_rows:
                raise ValueError("more rows than the tiled workspace was created for")
            for t in range((x.shape[0] + self.tile_rows - 1) // self.tile_rows):
                lo, hi = t * self.tile_rows, min(x.shape[0], (t + 1) * self.tile_rows)
                self.tiles[t].prepare_rows(x[lo:hi])
        # (one workspace: tiles are prepared when they are processed)

    def set_params(self, c, m, u):
        for t in self.tiles:
            t.set_params(c, m, u)
        self._params = (c, m, u)
        self._held = None

    def wants_drift(self, n_rows):
        # one workspace: nothing to carry, a tile's bounds do not survive the other tiles
        return self.resident and self.inner.wants_drift(min(int(n_rows), self.tile_rows))

    def set_drift(self, *a, **k):
        if self.resident:
            for t in self.tiles:
                t.set_drift(*a, **k)

    def forget(self):
        for t in self.tiles:
            t.forget()
        self._tail_in = None

    def set_shard(self, global_rows, n_ranks):
        self._global_rows, self._ranks = int(global_rows), int(n_ranks)
        self._joint_policy = (not self.resident) or self._ranks > 1
        for t in self.tiles:
            t.set_shard(self._global_rows, self._ranks * self.n_tiles if self._joint_policy else 1)

    def policy_export(self, tail):
        tail.copy_(self._tail_acc)

    def policy_import(self, tail):
        self._tail_in = tail.clone()

    def load_responsibilities(self, r):
        self._r = torch.as_tensor(r, dtype=torch.float64, device=self.device)
        self.rows = self._r.shape[0]
        self._held = None
        self._src = "loaded"

    # -- the data pass
    def _run(self, x, out, estep):
        if x.shape[0] != self.max_rows and x.shape[0] > self.max_rows:
            raise ValueError("more rows than the tiled workspace was created for")
        stats = self.inner._stats_out(out)
        stats.zero_()
        if estep:
            self._r = None
            self._src = "estep"
        self._tail_acc.zero_()
        self._infos, self._lazy = [], 0
        acc = self._new_acc()
        n = x.shape[0]
        n_t = (n + self.tile_rows - 1) // self.tile_rows
        for t in range(n_t):
            lo, hi = t * self.tile_rows, min(n, (t + 1) * self.tile_rows)
            xt = x[lo:hi]
            eng = self._eng(t)
            if not self.resident:
                eng.prepare_rows(xt)
            if estep:
                if self._joint_policy and self._tail_in is not None:
                    eng.policy_import(self._tail_in)
                eng.estep_mstep(xt, out=self._tmp)
                if self._joint_policy:
                    eng.policy_export(self._tail_tmp)
                    self._tail_acc += self._tail_tmp
            else:
                eng.load_responsibilities(self._r[lo:hi])
                eng.mstep(xt, out=self._tmp)
            stats += self._tmp
            self._infos.append(eng.launch_info)
            if estep and not self.resident:
                self._add_tile(acc, eng, lo, hi)
        if estep:
            if self._tail_in is None or self._ranks == 1:
                self._tail_in = self._tail_acc.clone()      # a single process: the tiles' sums are the job's
            if self.resident:
                self._lazy = n_t
            else:
                self._finish(acc)
            self._held = n_t - 1
            self.rows = n
        return stats

    def profile(self, on=True):
        self._prof = bool(on)
        for t in self.tiles:
            t.profile(on)

    def estep_mstep(self, x, out=None):
        self._x = x
        return self._run(x, out, True)

    def mstep(self, x, out=None):
        self._x = x
        if self._src == "loaded":
            return self._run(x, out, False)
        return self._run(x, out, True)             # statistics of the E-step under the parameters in force

    def estep(self, x):
        self._x, self.rows, self._held = x, x.shape[0], None
        self._r, self._src = None, "estep"

    # -- read-outs: rows [row0, row0 + n) in the caller's order, tile by tile
    def _readout(self, what, row0, n, dtype, cols):
        n = self.rows - row0 if n is None else n
        out = torch.empty((n, self.K) if cols else (n,), dtype=dtype, device=self.device)
        pos = row0
        while pos < row0 + n:
            t = pos // self.tile_rows
            lo, hi = self._tile(t)
            hi = min(hi, self.rows)
            eng = self._eng(t)
            if self._held != t:
                xt = self._x[lo:hi]
                if not self.resident:
                    eng.prepare_rows(xt)
                if self._src == "loaded":
                    eng.load_responsibilities(self._r[lo:hi])
                else:
                    eng.estep(xt)
                self._held = t
            take = min(hi, row0 + n) - pos
            out[pos - row0: pos - row0 + take] = getattr(eng, what)(pos - lo, take)
            pos += take
        return out

    def responsibilities(self, row0=0, n=None):
        return self._readout("responsibilities", row0, n, torch.float64, True)

    def ln_rho(self, row0=0, n=None):
        return self._readout("ln_rho", row0, n, torch.float64, True)

    def argmax(self, row0=0, n=None):
        return self._readout("argmax", row0, n, torch.int32, False)


def open_data_pass(K, D, x_dtype, n_rows, device, tile_rows=None, resident=None):
    """A DataPass for all rows if its workspace fits the GPU - leaving room for the K-sized state and the caller's
    temporaries -, else a TiledDataPass: resident tiles (one workspace per tile, carried bounds) of the largest size that
    fits, halving down to 1/64 of the rows; if none does, tiles through ONE workspace.  ``tile_rows`` (or
    BAYESML_AMD_TILE_ROWS in the environment) forces tiles of that many rows, ``resident`` (BAYESML_AMD_TILE_RESIDENT=0/1,
    default 1) their form."""
    forced = tile_rows or int(os.environ.get("BAYESML_AMD_TILE_ROWS", "0"))
    if resident is None:
        resident = os.environ.get("BAYESML_AMD_TILE_RESIDENT", "1") != "0"
    if forced and forced < n_rows:
        return TiledDataPass(K, D, x_dtype, n_rows, device, forced, resident=resident)
    dev = torch.device(device)

    def fits(make):
        try:
            eng = make()
        except EngineError as e:
            if "GMMVB_ENOMEM" not in str(e):
                raise
            return None
        except RuntimeError as e:              # torch's own allocations next to a workspace that just fitted
            if "out of memory" not in str(e).lower():
                raise
            return None
        free, total = torch.cuda.mem_get_info(dev)
        if free < max(4 << 30, total // 16):          # no room left to work in
            eng.close()
            torch.cuda.empty_cache()
            return None
        return eng

    eng = fits(lambda: DataPass(K, D, x_dtype, n_rows, dev))
    sizes, rows = [], (n_rows + 1) // 2
    while rows >= 1 << 16:
        sizes.append((rows + 63) // 64 * 64)
        rows = (rows + 1) // 2
    for form, cand in ((True, sizes[:6]), (False, sizes)) if resident else ((False, sizes),):
        for r in cand:
            if eng is not None:
                break
            torch.cuda.empty_cache()
            eng = fits(lambda: TiledDataPass(K, D, x_dtype, n_rows, dev, r, resident=form))
    if eng is None:
        raise EngineError(f"no workspace fits the GPU even for tiles of {sizes[-1] if sizes else n_rows} rows (K={K}, D={D})")
    return eng
