"""Device plumbing shared by the model classes: moving the sample matrix to the GPU once, creating
the C-ABI workspace, the pivot, and the restart initialisation that needs rows of x.

PyTorch is used for memory and streams only; the N-sized arithmetic is in ``csrc/`` behind the C ABI.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _check
from ._exceptions import DataFormatError


class DeviceModel:
    """Mixin for LearnModel classes.  Expects ``c_degree``, ``c_num_classes``, ``rng``, ``_device``,
    ``_comm``, ``_verbose``, ``_engine``, ``_x_dev``, ``_r_cache``, ``_data_pass_factory`` on the instance."""

    def _check_rows(self, x):
        """Validate ``x`` like the reference (gaussianmixture ref:829-834, hiddenmarkovnormal ref:1055-1061) -> [N, D]."""
        D = self.c_degree
        if isinstance(x, torch.Tensor):
            if not (x.dtype.is_floating_point and x.dim() >= 1):
                raise DataFormatError("x must be a numpy.ndarray whose ndim >= 1.")
        else:
            _check.float_vecs(x, "x", DataFormatError)
        if x.shape[-1] != D:
            raise DataFormatError(f"x.shape[-1] must be self.c_degree: x.shape[-1]={x.shape[-1]}, self.c_degree={D}")
        return x.reshape(-1, D)

    def _open(self, x):
        """Validate ``x``, move it to the GPU once, (re)create the workspace, set the pivot, build the centred copy."""
        D, K = self.c_degree, self.c_num_classes
        x = self._check_rows(x)
        if self._data_pass_factory is not None:
            eng = self._data_pass_factory(K, D, x)
            xd = eng.adopt(x)
        else:
            from ._engine import DataPass, EngineUnavailableError, open_data_pass
            if not torch.cuda.is_available():
                raise EngineUnavailableError(
                    f"bayesml_amd {type(self).__module__}.LearnModel needs an MI355X: the data pass has no CPU fallback")
            dev = torch.device("cuda", torch.cuda.current_device()) if self._device is None else torch.device(self._device)
            if isinstance(x, torch.Tensor):
                xd = x.to(dev)
                if xd.dtype not in (torch.float32, torch.float64):
                    xd = xd.to(torch.float64)
            else:
                xh = np.ascontiguousarray(x if x.dtype in (np.float32, np.float64) else x.astype(np.float64))
                xd = torch.from_numpy(xh).to(dev)
            xd = xd.contiguous()
            eng = self._engine
            if (eng is None or eng.K != K or eng.D != D or eng.x_dtype != xd.dtype or eng.max_rows < xd.shape[0]
                    or eng.device != dev or getattr(eng, "_ws", None) is None):
                if eng is not None:
                    eng.close()
                # (a matrix whose per-pair workspace does not fit the GPU goes through it in row tiles; the HMM's time axis
                # does not tile)
                tiling = getattr(self, "_row_tiling", False)
                eng = open_data_pass(K, D, xd.dtype, xd.shape[0], dev) if tiling else DataPass(K, D, xd.dtype, xd.shape[0], dev)
        self._engine, self._x_dev, self._r_cache = eng, xd, None
        self._small_r = None               # (responsibilities of an earlier small-problem fit)
        self._comm.bind_rows(xd.shape[0], xd.device)
        if hasattr(eng, "set_shard") and not getattr(self._comm, "restart_parallel", False):
            eng.set_shard(self._comm.global_rows, self._comm.world)      # row shards decide their pass policy together
        # expansion point of the second moments: mean of the leading rows (any fixed point near the data works)
        head = xd[: min(xd.shape[0], 4096)].to(torch.float64)
        cnt = torch.tensor([float(head.shape[0])], dtype=torch.float64, device=xd.device)
        acc = torch.cat([head.sum(dim=0), cnt])
        self._comm.all_reduce_(acc)
        eng.set_pivot(acc[:-1] / acc[-1])
        eng.prepare_rows(xd)
        return eng, xd

    def _say(self, text, end=""):
        if self._verbose and self._comm.rank == 0:
            print(text, end=end)

    def _subsample_moments(self, eng, xd, n_global, draw_only=False):
        """Raw moments about the pivot of the K sub-samples of the 'subsampling' restart (gaussianmixture
        ref:786-796 / hiddenmarkovnormal ref:952-964).  The host draws the row INDICES with the model's
        Generator — ``rng.choice(x, size, replace=False, axis=0, shuffle=False)`` consumes the stream exactly
        like ``rng.choice(N, size, replace=False, shuffle=False)`` — and the GPU gathers the rows it owns."""
        K, D = self.c_num_classes, self.c_degree
        size = int(np.sqrt(n_global))
        dev = xd.device
        ab = torch.zeros(K, D + D * D, dtype=torch.float64, device=dev)
        for k in range(K):
            draw = self.rng.choice(n_global, size=size, replace=False, shuffle=False)
            if draw_only:              # another process runs this restart: only keep the Generator's stream in step
                continue
            idx = torch.from_numpy(draw).to(dev)
            rows = xd.index_select(0, self._comm.local_indices(idx)).to(torch.float64) - eng.pivot
            ab[k, :D] = rows.sum(dim=0)
            ab[k, D:] = (rows.T @ rows).reshape(-1)
        self._comm.all_reduce_(ab)
        return float(size), ab[:, :D], ab[:, D:].reshape(K, D, D)
