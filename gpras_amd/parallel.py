"""Multi-GPU use of the path: independent units sharded over ranks, one gather at the end.

The reference fits and predicts its per-mode models in a serial Python loop
(``/root/reference/gpras/gpr.py:272-274, 336-339``); the models share ``x`` and nothing else
(SURVEY.md section 8e).  Here every rank (one process per GPU, launched by ``torch.distributed.run``) owns the units ``u``
with ``u % world == rank`` (round-robin balances uneven optimiser iteration counts), runs them without any communication,
and ONE collective gathers the results: the fitted parameters after ``fit`` (a few floats + Z per unit) and the (N*, K)
mean / variance after ``predict``.  On GPUs the collective is RCCL behind the C ABI (``gprx_comm_*``, ``gpras_amd.comm``),
device-resident for the predictions; ``torch.distributed`` only carries the 128-byte RCCL id at start-up.  With a CPU
process group ("gloo": the tests of this host logic) the same gather goes through ``torch.distributed``.
A single large fit does not shard (replicas only).
"""

from __future__ import annotations

from typing import Any

import numpy as np

from .gpr import GPRAS
from .optimizers import OPTIMIZERS


def shard_units(n_units: int, rank: int, world: int) -> list[int]:
    """Units owned by ``rank``: round-robin."""
    return [u for u in range(n_units) if u % world == rank]


def _torch_group():
    """The initialised torch.distributed module, or None (torch is only imported if the caller already did)."""
    import sys

    tdist = sys.modules.get("torch.distributed")
    if tdist is not None and tdist.is_available() and tdist.is_initialized():
        return tdist
    return None


def _dist():
    dist = _torch_group()
    if dist is None:
        raise RuntimeError("torch.distributed is not initialised (launch with torch.distributed.run)")
    return dist


def rank_and_world() -> tuple[int, int, str]:
    """(rank, world, launcher): from the torch.distributed process group when the caller initialised one ("torch"), else from
    the launcher's environment -- RANK / WORLD_SIZE as set by ``torch.distributed.run`` or any other per-GPU launcher -- with NO
    torch in the process ("env": the collective then runs over gprx_comm_*, bootstrapped through files)."""
    import os

    dist = _torch_group()
    if dist is not None:
        return dist.get_rank(), dist.get_world_size(), "torch"
    if "RANK" in os.environ and "WORLD_SIZE" in os.environ:
        return int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), "env"
    raise RuntimeError("no torch.distributed process group and no RANK / WORLD_SIZE in the environment: launch one process per GPU")


_COMM = None  # this process's gprx communicator (RCCL), created on first use (process group with backend "nccl", or torch-free launch)


def communicator(device: int | None = None):
    """The RCCL communicator behind the C ABI (``gprx_comm_*``).  With a torch.distributed process group the 128-byte id is
    broadcast through it (``None`` for CPU groups -- gloo: tests of the host logic); in a torch-free launch it travels through
    files (``gpras_amd.comm.file_rendezvous``), every rank first confirming that it can load RCCL."""
    global _COMM
    rank, world, launcher = rank_and_world()
    if launcher == "env":
        if _COMM is None:
            import os

            from .comm import Communicator, default_id_prefix

            dev = int(os.environ.get("LOCAL_RANK", "0")) if device is None else device
            _COMM = Communicator.bootstrap(dev, rank, world, id_file=default_id_prefix())
        return _COMM
    dist = _dist()
    if dist.get_backend() != "nccl":
        return None
    if _COMM is None:
        import torch

        from .comm import Communicator

        dev = torch.cuda.current_device() if device is None else device
        try:
            comm = Communicator.bootstrap(dev, dist.get_rank(), dist.get_world_size())
        except Exception as exc:  # noqa: BLE001
            import warnings

            warnings.warn(f"gprx communicator unavailable ({exc}); the gather goes through torch.distributed", stacklevel=2)
            comm = None
        # every rank takes the same path: agree on the outcome through the process group
        ok = torch.tensor([1 if comm is not None else 0], device=torch.device("cuda", dev))
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0 and comm is not None:
            comm.close()
            comm = None
        _COMM = comm if comm is not None else False
    return _COMM or None


def _all_gather_array(arr: np.ndarray) -> list[np.ndarray]:
    """One collective: equally-shaped float64 arrays from every rank -- ``gprx_comm_all_gather_host`` (RCCL through the C
    ABI) on GPUs, gloo on CPU (tests)."""
    comm = communicator()
    if comm is not None:
        return comm.all_gather(arr)
    import torch

    dist = _dist()  # (CPU process group: the gloo tests of the host logic)
    device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
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
        self.rank, self.world, launcher = rank_and_world()
        if device is None:
            device = 0
            if launcher == "env":
                import os

                device = int(os.environ.get("LOCAL_RANK", "0"))
            elif _dist().get_backend() == "nccl":
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

    def predict(self, x, sharded: bool = True, root: int | None = None):
        """Sharded predict: every rank predicts its units, ONE collective gathers the (N*, K) mean / variance.
        ``root=None``: every rank returns the full arrays (all-gather); ``root=r``: only rank r does (gather to the root, all
        of its inbound xGMI links in parallel), the others return ``None``.  On GPUs the predictions stay in device memory
        from ``gprx_predict_dev`` to the collective (``gprx_comm_all_gather`` / ``gprx_comm_gather``): no host bounce."""
        if not sharded:
            return super().predict(x)
        x = np.ascontiguousarray(x, dtype=np.float64)
        k = len(self.models)
        mine = shard_units(k, self.rank, self.world)
        n_max = (k + self.world - 1) // self.world
        ns = x.shape[0]
        comm = communicator(self.device)
        exact = all(m.Z is None for m in self.models)
        if comm is not None and exact and x.shape[1] <= 64 and hasattr(self.engine, "predict_dev"):
            return self._predict_device_resident(x, mine, n_max, comm, root)
        local = np.zeros((2, n_max, ns))
        batched = self._predict_batched(x, mine) if mine else None  # exact models: one batched launch sequence
        for row, u in enumerate(mine):
            if batched is not None:
                local[0, row], local[1, row] = batched[0][:, u], batched[1][:, u]
                continue
            mean, var = self.models[u].predict_y(x)
            local[0, row] = mean[:, 0]
            local[1, row] = var[:, 0]
        gathered = _all_gather_array(local)  # the single collective of predict
        return self._unpack_predictions(gathered, ns) if root is None or root == self.rank else None

    def _unpack_predictions(self, gathered, ns: int):
        k = len(self.models)
        means = np.empty((ns, k))
        variances = np.empty((ns, k))
        for r, block in enumerate(gathered):
            for row, u in enumerate(shard_units(k, r, self.world)):
                means[:, u] = block[0, row]
                variances[:, u] = block[1, row]
        return means, variances

    def _predict_device_resident(self, x, mine, n_max: int, comm, root):
        from ._lib import DeviceBuffer, check, load, ptr

        eng, ns = self.engine, x.shape[0]
        block = 2 * n_max * ns
        bufs: list = []

        def alloc(buf):
            bufs.append(buf)
            return buf

        try:
            dxs = alloc(DeviceBuffer.from_array(x, self.device))
            local = alloc(DeviceBuffer(8 * block, self.device))
            if len(mine) < n_max:  # the padding row of a ragged shard: defined values
                zeros = np.zeros(ns)
                for stat in range(2):
                    for row in range(len(mine), n_max):
                        check(load().gprx_memcpy_h2d(self.device, local.at((stat * n_max + row) * ns), ptr(zeros), zeros.nbytes))
            chunk = eng.max_cells(want_grad=False)
            for lo in range(0, len(mine), chunk):
                part = mine[lo : lo + chunk]
                _, ok = eng.factorize_batch([self.models[u].unit for u in part], np.stack([self.models[u].theta() for u in part]), 0)
                if not ok.all():
                    raise RuntimeError(f"kernel matrix not positive definite for mode(s) {[part[i] for i in np.flatnonzero(~ok)]}")
                for slot, u in enumerate(part):
                    row = lo + slot
                    eng.select_slot(slot)
                    eng.predict_dev(dxs, ns, local.at(row * ns), local.at((n_max + row) * ns), include_noise=True, wait=False)
            eng.synchronize()  # the predictions are complete before the communicator's stream reads them
            gather_to_all = root is None
            recv = alloc(DeviceBuffer(8 * block * self.world, self.device)) if gather_to_all or root == self.rank else None
            if gather_to_all:
                comm.all_gather_dev(local, recv, block)
            else:
                comm.gather_dev(local, recv, block, root)
            comm.synchronize()
            out = None
            if recv is not None:
                gathered = recv.to_array((self.world, 2, n_max, ns))
                out = self._unpack_predictions([gathered[r] for r in range(self.world)], ns)
            return out
        finally:  # (also when a factorisation raises: no device buffer outlives the call)
            for buf in bufs:
                buf.free()
