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


_TAGS: dict = {}  # prefix -> the launch tag this process agreed on (file_rendezvous); "" when it never got that far


def agreed_tag(prefix: str) -> str:
    """The per-launch tag ``file_rendezvous`` agreed on for ``prefix`` in this process ("" if it did not get that far).  ``report``
    and ``FileExchange`` stamp their files with it, so a leftover of another launch under the same prefix is never read as this
    launch's."""
    return _TAGS.get(prefix, "")


def _publish(path: str, payload: bytes) -> None:
    """Atomic: temporary name + rename, so a reader never sees half a file.  First line = wall-clock time of the writing."""
    tmp = f"{path}.tmp{os.getpid()}"
    with open(tmp, "wb") as f:
        f.write(repr(time.time()).encode() + b"\n" + payload)
    os.replace(tmp, path)


def _read_fresh(path: str, max_age_s: float):
    try:
        with open(path, "rb") as f:
            stamp, _, payload = f.read().partition(b"\n")
        return payload if time.time() - float(stamp) <= max_age_s else None
    except (OSError, ValueError):
        return None


def _rm(path: str) -> None:
    try:
        os.remove(path)
    except OSError:
        pass


def file_rendezvous(prefix: str, rank: int, world: int, status: str, make_id, timeout_s: float = 120.0, max_age_s: float = 900.0) -> bytes:
    """Torch-free rendezvous over files on a path every rank sees (one node: /tmp): returns the id that rank 0 created.

    1. every rank publishes ``<prefix>.ready.<rank>``: a stamp of its own (pid + time in ns: no other process ever writes the same)
       and ``status`` ("ok" or an error text);
    2. every rank reads all ``world`` ready files, hashes the stamps into a TAG and publishes it as ``<prefix>.ack.<rank>``; it goes
       on only when every rank's ack carries the same tag, and re-reads the ready files while they do not -- so a leftover ready
       file of an earlier launch under the same prefix (its rank not started yet, or not there at all) can delay the agreement or
       make it time out, but never be taken for this launch's: all ranks that leave this step have seen the same, current files;
    3. if ANY status is not "ok", every rank raises the same RuntimeError and nobody enters the collective initialisation (a rank
       that cannot load RCCL must not leave the others blocked in ncclCommInitRank);
    4. rank 0 -- which removed any old ``<prefix>.id`` before step 1 -- calls ``make_id()`` and publishes tag + id; the others
       accept only an id file that carries the agreed tag.
    Files carry the wall-clock time of their writing and are ignored when older than ``max_age_s``."""
    import hashlib

    t0 = time.time()

    def expired() -> bool:
        return time.time() - t0 > timeout_s

    if rank == 0:
        _rm(f"{prefix}.id")
    _TAGS[prefix] = ""
    mine = f"{os.getpid()}.{time.time_ns()}"
    _publish(f"{prefix}.ready.{rank}", mine.encode() + b"\n" + status.encode())
    tag = states = None
    published = None
    while True:
        got = [_read_fresh(f"{prefix}.ready.{r}", max_age_s) for r in range(world)]
        missing = [r for r, g in enumerate(got) if g is None]
        if not missing:
            pairs = [g.partition(b"\n") for g in got]
            tag = hashlib.sha1(b"|".join(p[0] for p in pairs)).hexdigest()[:20]
            states = [p[2].decode() for p in pairs]
            if tag != published:
                _publish(f"{prefix}.ack.{rank}", tag.encode())
                published = tag
            acks = [_read_fresh(f"{prefix}.ack.{r}", max_age_s) for r in range(world)]
            if all(a is not None and a.decode() == tag for a in acks):
                break
        if expired():
            what = f"status of rank {missing[0]} ({prefix}.ready.{missing[0]}) did not appear" if missing else "the ranks did not agree on one set of ready files"
            raise TimeoutError(f"rank {rank}: {what} within {timeout_s:.0f} s")
        time.sleep(0.02)
    _TAGS[prefix] = tag
    bad = {r: st for r, st in enumerate(states) if st != "ok"}
    if bad:
        raise RuntimeError(f"the communicator cannot be created on every rank: {bad}")
    if rank == 0:
        _publish(f"{prefix}.id", tag.encode() + b"\n" + make_id())
    while True:
        got = _read_fresh(f"{prefix}.id", max_age_s)
        if got is not None:
            its_tag, _, uid = got.partition(b"\n")
            if its_tag.decode() == tag:
                return uid
        if expired():
            raise TimeoutError(f"rank {rank}: the RCCL id of rank 0 ({prefix}.id) did not appear within {timeout_s:.0f} s")
        time.sleep(0.02)


def rendezvous_cleanup(prefix: str, rank: int, success: bool) -> None:
    """Remove this rank's rendezvous files.  After a SUCCESSFUL collective initialisation every rank has read everything: ready, ack
    and (rank 0) the id go.  After a failure the ready file stays -- another rank may still have to read the status in it, and it is
    stamped with this launch's pid + time, so no later launch can take it for its own -- while the ack and the id, which nobody may
    act on any more, go.  The ack stays after a failure too: a slower rank may still be between publishing its own ack and
    reading the others'; with this rank's ack gone it would time out after ``timeout_s`` with "the ranks did not agree" instead of raising
    the shared error that names the failing rank.  The ack carries the launch tag, so a later launch cannot mistake it."""
    if rank == 0:
        _rm(f"{prefix}.id")
    if success:
        _rm(f"{prefix}.ack.{rank}")
        _rm(f"{prefix}.ready.{rank}")


def report(prefix: str, stage: str, rank: int, world: int, ok: bool, timeout_s: float = 120.0, tag: str | None = None) -> list:
    """Every rank reports ``ok`` for ``stage`` through files; returns what each rank said: True, False, or None for a rank that
    did not report within ``timeout_s`` (dead, or never started).  Reports carry the launch tag (``agreed_tag(prefix)`` unless given):
    a file with another tag -- a leftover, or a rank that never agreed -- counts as not reported."""
    tag = agreed_tag(prefix) if tag is None else tag
    path = f"{prefix}.{stage}."
    tmp = f"{path}{rank}.tmp{os.getpid()}"
    with open(tmp, "wb") as f:
        f.write(tag.encode() + (b":1" if ok else b":0"))
    os.replace(tmp, f"{path}{rank}")
    t0, said = time.time(), []
    for r in range(world):
        val = None
        while True:
            try:
                with open(f"{path}{r}", "rb") as f:
                    its_tag, _, v = f.read().rpartition(b":")
                if its_tag.decode() == tag:
                    val = v == b"1"
                    break
            except OSError:
                pass
            if time.time() - t0 > timeout_s:
                break
            time.sleep(0.02)
        said.append(val)
    return said


def report_cleanup(prefix: str, stage: str, rank: int) -> None:
    _rm(f"{prefix}.{stage}.{rank}")


def agree(prefix: str, stage: str, rank: int, world: int, ok: bool, timeout_s: float = 120.0, tag: str | None = None) -> bool:
    """True only if all ranks said yes for ``stage`` (a rank that never reports counts as no after ``timeout_s``).  Used where ranks
    must take the SAME branch without a working collective (fallback decisions)."""
    return all(v is True for v in report(prefix, stage, rank, world, ok, timeout_s, tag))


class FileExchange:
    """Barrier / gather / maximum of a few bytes per rank through files on a path every rank of ONE node sees -- an explicit opt-in
    of ``bench.py`` (GPRX_BENCH_FILE_EXCHANGE=1) for when the RCCL communicator cannot be created on every rank (its data path has no
    collective: the ranks only meet at the timing barriers and for the final gather of a few kilobytes).  Never used while
    ``gprx_comm_*`` works; the bench line says which one ran.  Round n of a rank writes ``<prefix>.fx.<n>.<rank>`` (temporary name +
    rename; first line = the launch tag, and a file with another tag is waited out like a missing one) and reads every rank's file
    of that round; a rank's file of round n - 2 is removed when it enters round n (every reader has passed it by then)."""

    def __init__(self, prefix: str, rank: int, world: int, timeout_s: float = 600.0, tag: str | None = None):
        self.tag = (agreed_tag(prefix) if tag is None else tag).encode()
        self.prefix, self.rank, self.world, self.timeout_s = f"{prefix}.fx", int(rank), int(world), float(timeout_s)
        self.round = 0

    def _exchange(self, payload: bytes) -> list:
        self.round += 1
        base = f"{self.prefix}.{self.round}."
        tmp = f"{base}{self.rank}.tmp{os.getpid()}"
        with open(tmp, "wb") as f:
            f.write(self.tag + b"\n" + payload)
        os.replace(tmp, f"{base}{self.rank}")
        if self.round > 2:
            _rm(f"{self.prefix}.{self.round - 2}.{self.rank}")
        out, t0 = [], time.time()
        for r in range(self.world):
            path = f"{base}{r}"
            while True:
                try:
                    with open(path, "rb") as f:
                        its_tag, _, body = f.read().partition(b"\n")
                    if its_tag == self.tag:
                        out.append(body)
                        break
                except OSError:
                    pass
                if time.time() - t0 > self.timeout_s:
                    raise TimeoutError(f"rank {self.rank}: rank {r} did not reach exchange {self.round} within {self.timeout_s:.0f} s")
                time.sleep(0.0005)
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
        self.barrier()
        _rm(f"{self.prefix}.{self.round - 1}.{self.rank}")
        with open(f"{self.prefix}.done.{self.rank}", "wb") as f:
            f.write(self.tag)
        if self.rank == 0:
            t0 = time.time()
            for r in range(self.world):
                while time.time() - t0 <= self.timeout_s:
                    try:
                        with open(f"{self.prefix}.done.{r}", "rb") as f:
                            if f.read() == self.tag:
                                break
                    except OSError:
                        pass
                    time.sleep(0.001)
            for r in range(self.world):
                _rm(f"{self.prefix}.{self.round}.{r}")
                _rm(f"{self.prefix}.done.{r}")


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
        done = False
        try:
            uid = file_rendezvous(id_file, rank, world, status, new_unique_id, timeout_s=timeout_s)
            comm = cls(device, rank, world, uid)  # (collective: when it returns, every rank has read the id)
            done = True
        finally:
            rendezvous_cleanup(id_file, rank, done)
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

    def rank_and_world_seen_by_rccl(self) -> tuple:
        """(rank, world) as RCCL reports them for this communicator (``ncclCommUserRank`` / ``ncclCommCount`` behind ``gprx_comm_rank``)."""
        r, w = C.c_int(), C.c_int()
        _check(self._lib.gprx_comm_rank(self._c, C.byref(r), C.byref(w)), self._c)
        return int(r.value), int(w.value)

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
