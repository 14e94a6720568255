"""The one collective of the path behind the C ABI: ``gprx_comm_*`` (RCCL over xGMI, loaded by ``libgprx.so`` at run time).

The reference has no multi-device code; its per-unit loops (``/root/reference/gpras/gpr.py:272-274, 336-339``) share
nothing but ``x``.  One process per GPU owns the units ``u % world == rank``; a ``Communicator``
gathers the device-resident results once at the end -- no host bounce, no dependency on ``torch`` for the data path.
The 128-byte RCCL id is the only out-of-band datum: ``Communicator.bootstrap`` moves it through whatever rendezvous the
launcher provides (``torch.distributed`` when it is initialised, else a file on a shared path).
"""

from __future__ import annotations

import ctypes as C
import os
import time

import numpy as np

from . import _lib
from ._lib import DeviceBuffer, as_f64, ptr


def _check(rc: int, comm=None) -> None:
    if rc == _lib.GPRX_OK:
        return
    msg = _lib.load().gprx_comm_last_error(comm)
    text = msg.decode() if msg else ""
    if rc == _lib.GPRX_EINVAL:
        raise ValueError(text)
    if rc == _lib.GPRX_ENOMEM:
        raise MemoryError(text)
    raise RuntimeError(f"libgprx communicator error {rc}: {text}")


def new_unique_id() -> bytes:
    """Rank 0: ``ncclGetUniqueId`` through ``gprx_comm_unique_id``."""
    buf = (C.c_ubyte * _lib.UNIQUE_ID_BYTES)()
    _check(_lib.load().gprx_comm_unique_id(buf))
    return bytes(buf)


class Communicator:
    def __init__(self, device: int, rank: int, world: int, unique_id: bytes):
        if len(unique_id) != _lib.UNIQUE_ID_BYTES:
            raise ValueError(f"the RCCL id has {_lib.UNIQUE_ID_BYTES} bytes")
        self._lib = _lib.load()
        self.device, self.rank, self.world = int(device), int(rank), int(world)
        self._c = C.c_void_p()
        idbuf = (C.c_ubyte * _lib.UNIQUE_ID_BYTES).from_buffer_copy(unique_id)
        _check(self._lib.gprx_comm_init(self.device, self.rank, self.world, idbuf, C.byref(self._c)))

    @classmethod
    def bootstrap(cls, device: int, rank: int | None = None, world: int | None = None, id_file: str | None = None, timeout_s: float = 120.0):
        """Create the communicator of this rank; the id travels through ``torch.distributed`` if a process group exists
        (``broadcast_object_list`` from rank 0), else through ``id_file`` (rank 0 writes it atomically, the others poll)."""
        rank = int(os.environ.get("RANK", "0")) if rank is None else rank
        world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else world
        dist = None
        try:
            import torch.distributed as tdist

            if tdist.is_available() and tdist.is_initialized():
                dist = tdist
        except ImportError:
            pass
        if dist is not None:
            box = [None]
            if rank == 0:
                try:
                    box = [new_unique_id()]
                except Exception as exc:  # noqa: BLE001  (the other ranks wait in the broadcast: they must hear about it)
                    box = [exc]
            dist.broadcast_object_list(box, src=0)
            if isinstance(box[0], Exception):
                raise RuntimeError(f"rank 0 could not create the RCCL id: {box[0]}")
            return cls(device, rank, world, box[0])
        if world == 1:
            return cls(device, 0, 1, new_unique_id())
        if id_file is None:
            raise RuntimeError("no torch.distributed process group and no id_file: the RCCL id cannot reach the other ranks")
        if rank == 0:
            tmp = f"{id_file}.tmp{os.getpid()}"
            with open(tmp, "wb") as f:
                f.write(new_unique_id())
            os.replace(tmp, id_file)
        t0 = time.time()
        while not os.path.exists(id_file):
            if time.time() - t0 > timeout_s:
                raise TimeoutError(f"rank {rank}: the RCCL id file {id_file} did not appear")
            time.sleep(0.05)
        with open(id_file, "rb") as f:
            return cls(device, rank, world, f.read())

    # -- lifetime --------------------------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_c", None) is not None and self._c.value:
            self._lib.gprx_comm_destroy(self._c)
            self._c = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- collectives (device buffers: gpras_amd._lib.DeviceBuffer or raw device pointers) ---------------------------
    @staticmethod
    def _p(buf):
        return buf.ptr if isinstance(buf, DeviceBuffer) else buf

    def all_gather_dev(self, send, recv, count: int) -> None:
        """``recv (world * count)`` <- every rank's ``send (count)`` doubles, rank-major; asynchronous (``synchronize``)."""
        _check(self._lib.gprx_comm_all_gather(self._c, self._p(send), self._p(recv), int(count)), self._c)

    def gather_dev(self, send, recv, count: int, root: int = 0) -> None:
        """To ``root`` only (grouped send / receive: all inbound links of the root at once); ``recv`` may be None elsewhere."""
        _check(self._lib.gprx_comm_gather(self._c, self._p(send), None if recv is None else self._p(recv), int(count), int(root)), self._c)

    def all_reduce_max_dev(self, buf, count: int) -> None:
        _check(self._lib.gprx_comm_all_reduce_max(self._c, self._p(buf), int(count)), self._c)

    def synchronize(self) -> None:
        _check(self._lib.gprx_comm_synchronize(self._c), self._c)

    def barrier(self) -> None:
        _check(self._lib.gprx_comm_barrier(self._c), self._c)

    # -- small host arrays (fitted parameters, timings) ----------------------------------------------------------------
    def all_gather(self, arr) -> list[np.ndarray]:
        """Equally-shaped float64 arrays from every rank (staged through the device; synchronous)."""
        arr = as_f64(arr)
        out = np.empty((self.world,) + arr.shape)
        _check(self._lib.gprx_comm_all_gather_host(self._c, ptr(arr), ptr(out), arr.size), self._c)
        return [out[r] for r in range(self.world)]

    def max(self, value: float) -> float:
        """Maximum of a scalar over the ranks (a benchmark's slowest rank)."""
        return float(max(v[0] for v in self.all_gather(np.array([float(value)]))))
