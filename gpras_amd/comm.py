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


def file_rendezvous(prefix: str, rank: int, world: int, status: str, make_id, timeout_s: float = 120.0, max_age_s: float = 900.0) -> bytes:
    """Torch-free rendezvous over files on a path every rank sees (one node: /tmp): returns the id that rank 0 created.

    1. every rank publishes ``status`` ("ok" or an error text) as ``<prefix>.ready.<rank>`` -- written to a temporary name and
       renamed, so a reader never sees half a file;
    2. every rank waits for all ``world`` status files: if ANY rank is not "ok", every rank raises the same RuntimeError and
       nobody enters the collective initialisation (a rank that cannot load RCCL must not leave the others blocked in
       ncclCommInitRank);
    3. rank 0 calls ``make_id()`` and publishes ``<prefix>.id``; the others poll for it.
    Files carry the wall-clock time of their writing and are ignored when older than ``max_age_s`` (leftovers of an earlier
    launch that reused the prefix)."""

    def publish(path: str, payload: bytes) -> None:
        tmp = f"{path}.tmp{os.getpid()}"
        with open(tmp, "wb") as f:
            f.write(repr(time.time()).encode() + b"\n" + payload)
        os.replace(tmp, path)

    def read_fresh(path: str):
        try:
            with open(path, "rb") as f:
                stamp, _, payload = f.read().partition(b"\n")
            return payload if time.time() - float(stamp) <= max_age_s else None
        except (OSError, ValueError):
            return None

    t0 = time.time()

    def wait_for(path: str, what: str) -> bytes:
        while True:
            got = read_fresh(path)
            if got is not None:
                return got
            if time.time() - t0 > timeout_s:
                raise TimeoutError(f"rank {rank}: {what} ({path}) did not appear within {timeout_s:.0f} s")
            time.sleep(0.02)

    publish(f"{prefix}.ready.{rank}", status.encode())
    states = [wait_for(f"{prefix}.ready.{r}", f"status of rank {r}").decode() for r in range(world)]
    bad = {r: st for r, st in enumerate(states) if st != "ok"}
    if bad:
        raise RuntimeError(f"the communicator cannot be created on every rank: {bad}")
    if rank == 0:
        publish(f"{prefix}.id", make_id())
    return wait_for(f"{prefix}.id", "the RCCL id of rank 0")


def report(prefix: str, stage: str, rank: int, world: int, ok: bool, timeout_s: float = 120.0) -> list:
    """Every rank reports ``ok`` for ``stage`` through files; returns what each rank said: True, False, or None for a rank that
    did not report within ``timeout_s`` (dead, or never started)."""
    path = f"{prefix}.{stage}."
    tmp = f"{path}{rank}.tmp{os.getpid()}"
    with open(tmp, "wb") as f:
        f.write(b"1" if ok else b"0")
    os.replace(tmp, f"{path}{rank}")
    t0, said = time.time(), []
    for r in range(world):
        while not os.path.exists(f"{path}{r}") and time.time() - t0 <= timeout_s:
            time.sleep(0.02)
        try:
            with open(f"{path}{r}", "rb") as f:
                said.append(f.read() == b"1")
        except OSError:
            said.append(None)
    return said


def agree(prefix: str, stage: str, rank: int, world: int, ok: bool, timeout_s: float = 120.0) -> bool:
    """True only if all ranks said yes for ``stage`` (a rank that never reports counts as no after ``timeout_s``).  Used where ranks
    must take the SAME branch without a working collective (fallback decisions)."""
    return all(v is True for v in report(prefix, stage, rank, world, ok, timeout_s))


class FileExchange:
    """Barrier / gather / maximum of a few bytes per rank through files on a path every rank of ONE node sees -- the last resort
    of ``bench.py`` when the RCCL communicator cannot be created on every rank (its data path has no collective: the ranks only
    meet at the timing barriers and for the final gather of a few kilobytes).  Never used while ``gprx_comm_*`` works; the bench line
    says which one ran.  Round n of a rank writes ``<prefix>.fx.<n>.<rank>`` (temporary name + rename) and reads every rank's file
    of that round; a rank's file of round n - 2 is removed when it enters round n (every reader has passed it by then)."""

    def __init__(self, prefix: str, rank: int, world: int, timeout_s: float = 600.0):
        self.prefix, self.rank, self.world, self.timeout_s = f"{prefix}.fx", int(rank), int(world), float(timeout_s)
        self.round = 0

    def _exchange(self, payload: bytes) -> list:
        self.round += 1
        base = f"{self.prefix}.{self.round}."
        tmp = f"{base}{self.rank}.tmp{os.getpid()}"
        with open(tmp, "wb") as f:
            f.write(payload)
        os.replace(tmp, f"{base}{self.rank}")
        if self.round > 2:
            try:
                os.remove(f"{self.prefix}.{self.round - 2}.{self.rank}")
            except OSError:
                pass
        out, t0 = [], time.time()
        for r in range(self.world):
            path = f"{base}{r}"
            while not os.path.exists(path):
                if time.time() - t0 > self.timeout_s:
                    raise TimeoutError(f"rank {self.rank}: rank {r} did not reach exchange {self.round} within {self.timeout_s:.0f} s")
                time.sleep(0.0005)
            with open(path, "rb") as f:
                out.append(f.read())
        return out

    def barrier(self) -> None:
        self._exchange(b"")

    def all_gather(self, arr) -> list:
        a = np.ascontiguousarray(arr, dtype=np.float64)
        return [np.frombuffer(b, dtype=np.float64).reshape(a.shape).copy() for b in self._exchange(a.tobytes())]

    def max(self, value: float) -> float:
        return float(max(np.frombuffer(b, dtype=np.float64)[0] for b in self._exchange(np.float64(value).tobytes())))

    def close(self) -> None:
        """Last meeting: after it every rank says it has read the final round (``<prefix>.done.<rank>``); rank 0 waits for all of
        those and removes what is left -- no rank deletes a file another rank may still have to read."""

        def rm(path: str) -> None:
            try:
                os.remove(path)
            except OSError:
                pass

        self.barrier()
        rm(f"{self.prefix}.{self.round - 1}.{self.rank}")
        with open(f"{self.prefix}.done.{self.rank}", "wb"):
            pass
        if self.rank == 0:
            t0 = time.time()
            for r in range(self.world):
                while not os.path.exists(f"{self.prefix}.done.{r}") and time.time() - t0 <= self.timeout_s:
                    time.sleep(0.001)
            for r in range(self.world):
                rm(f"{self.prefix}.{self.round}.{r}")
                rm(f"{self.prefix}.done.{r}")


def default_id_prefix() -> str:
    """A rendezvous prefix every rank of ONE launch computes identically and no other launch shares: explicit ``GPRX_ID_FILE``, else
    /tmp + the launcher's port + its run id + the launcher's pid (the ranks are siblings: children of one ``torch.distributed.run``
    agent or of ``bench.py``'s own spawner)."""
    explicit = os.environ.get("GPRX_ID_FILE")
    if explicit:
        return explicit
    import tempfile

    tag = f"{os.environ.get('MASTER_PORT', '0')}_{os.environ.get('TORCHELASTIC_RUN_ID', 'none')}_{os.getppid()}"
    return os.path.join(tempfile.gettempdir(), f"gprx_rccl_{tag}")


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
        # (torch is never imported HERE: a process group can only exist if the caller imported torch itself -- importing it now would
        # map torch's own HIP / HSA / RCCL tree into a rank that is meant to run on one runtime)
        import sys

        dist = sys.modules.get("torch.distributed")
        if dist is not None and not (dist.is_available() and dist.is_initialized()):
            dist = None
        if dist is not None:
            rc = _lib.load().gprx_comm_runtime_check(int(device))
            states = [None] * world
            dist.all_gather_object(states, "ok" if rc == _lib.GPRX_OK else f"rank {rank}: libgprx error {rc}")
            if any(st != "ok" for st in states):
                raise RuntimeError(f"the communicator cannot be created on every rank: {[st for st in states if st != 'ok']}")
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
        # torch-free: readiness of every rank agreed through files before anyone enters ncclCommInitRank
        rc = _lib.load().gprx_comm_runtime_check(int(device))
        status = "ok"
        if rc != _lib.GPRX_OK:
            msg = _lib.load().gprx_comm_last_error(None)
            status = f"rank {rank}: {msg.decode() if msg else 'libgprx error ' + str(rc)}"
        uid = file_rendezvous(id_file, rank, world, status, new_unique_id, timeout_s=timeout_s)
        comm = cls(device, rank, world, uid)  # (collective: when it returns, every rank has read the id)
        for leftover in [f"{id_file}.ready.{rank}"] + ([f"{id_file}.id"] if rank == 0 else []):
            try:
                os.remove(leftover)
            except OSError:
                pass
        return comm

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
