"""Multi-GPU use of the path: independent units sharded over ranks, one gather at the end.

The reference fits and predicts its per-mode models in a serial Python loop
(``/root/reference/gpras/gpr.py:272-274, 336-339``); the models share ``x`` and nothing else
(SURVEY.md section 8e).  Here every rank (one process per GPU, ``torch.distributed``; backend "nccl" is
RCCL on ROCm, "gloo" on CPU for tests) owns the units ``u`` with ``u % world == rank`` (round-robin balances
uneven optimiser iteration counts), runs them without any communication, and ONE ``all_gather`` collects
the results: the fitted parameters after ``fit`` (a few floats + Z per unit) and the (N*, K) mean / variance
after ``predict``.  A single large fit does not shard (replicas only).
"""

from __future__ import annotations

from typing import Any

import numpy as np

from .gpr import GPRAS
from .optimizers import OPTIMIZERS


def shard_units(n_units: int, rank: int, world: int) -> list[int]:
    """Units owned by ``rank``: round-robin."""
    return [u for u in range(n_units) if u % world == rank]


def _dist():
    import torch.distributed as dist

    if not dist.is_available() or not dist.is_initialized():
        raise RuntimeError("torch.distributed is not initialised (launch with torch.distributed.run)")
    return dist


def _all_gather_array(arr: np.ndarray) -> list[np.ndarray]:
    """One collective: gather equally-shaped float64 arrays from every rank (RCCL for nccl, gloo on CPU)."""
    import torch

    dist = _dist()
    backend = dist.get_backend()
    device = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    mine = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float64)).to(device)
    out = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    return [t.cpu().numpy() for t in out]


class ShardedGPRAS(GPRAS):
    """``GPRAS`` whose per-unit loops run only over this rank's units.

    After ``fit`` every rank holds the parameters of all units (gathered once), so ``to_file`` works on any
    rank and ``predict`` can be called sharded (default) or locally.
    """

    def __init__(self, kernel, device: int | None = None) -> None:
        dist = _dist()
        self.rank = dist.get_rank()
        self.world = dist.get_world_size()
        if device is None:
            device = 0
            if dist.get_backend() == "nccl":
                import torch

                device = torch.cuda.current_device()
        super().__init__(kernel, device=device)

    def _pack_params(self, units: list[int], width: int) -> np.ndarray:
        """Rows of the unconstrained [w_variance, w_lengthscales..., w_noise, Z.ravel()] of the owned units
        (bit-exact hand-over: no transform round trip), padded to a common count."""
        n_max = (len(self.models) + self.world - 1) // self.world
        out = np.full((n_max, width), np.nan)
        for row, u in enumerate(units):
            m = self.models[u]
            z = np.zeros(0) if m.Z is None else m.Z.ravel()
            out[row] = np.concatenate([[m.w_var], m.w_len, [m.w_noise], z])
        return out

    def fit(
        self, x, y, n_inducing, inducing_initializer="kmeans", optimization_method="two-stage", ard: bool = False, lockstep: bool | None = None,
        **opt_kwargs: Any,
    ) -> None:
        self.x = x.astype(np.float64)
        self.y = y.astype(np.float64)
        OPTIMIZERS[optimization_method]  # KeyError before any device work, as the reference
        self._init_models(self.x, self.y, n_inducing, inducing_initializer, ard)
        mine = shard_units(len(self.models), self.rank, self.world)
        # this rank's modes: in lock step on batched evaluations (as GPRAS.fit), no communication
        self._run_optimizers([self.models[u] for u in mine], optimization_method, lockstep, opt_kwargs)
        # the single collective of fit: everyone learns everyone's parameters
        n_len = self.engine.n_len
        zsize = 0 if self.models[0].Z is None else self.models[0].Z.size
        width = 2 + n_len + zsize
        gathered = _all_gather_array(self._pack_params(mine, width))
        for r, block in enumerate(gathered):
            for row, u in enumerate(shard_units(len(self.models), r, self.world)):
                if r == self.rank:
                    continue
                vals = block[row]
                m = self.models[u]
                m.w_var, m.w_len, m.w_noise = float(vals[0]), vals[1 : 1 + n_len].copy(), float(vals[1 + n_len])
                if zsize:
                    m.Z = vals[2 + n_len :].reshape(m.Z.shape)

    def predict(self, x, sharded: bool = True):
        if not sharded:
            return super().predict(x)
        x = x.astype(np.float64)
        k = len(self.models)
        mine = shard_units(k, self.rank, self.world)
        n_max = (k + self.world - 1) // self.world
        local = np.zeros((2, n_max, x.shape[0]))
        batched = self._predict_batched(x, mine) if mine else None  # exact models: one batched launch sequence
        for row, u in enumerate(mine):
            if batched is not None:
                local[0, row], local[1, row] = batched[0][:, u], batched[1][:, u]
                continue
            mean, var = self.models[u].predict_y(x)
            local[0, row] = mean[:, 0]
            local[1, row] = var[:, 0]
        gathered = _all_gather_array(local)  # the single collective of predict
        means = np.empty((x.shape[0], k))
        variances = np.empty((x.shape[0], k))
        for r, block in enumerate(gathered):
            for row, u in enumerate(shard_units(k, r, self.world)):
                means[:, u] = block[0, row]
                variances[:, u] = block[1, row]
        return means, variances
