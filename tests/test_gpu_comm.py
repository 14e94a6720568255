"""The collective behind the C ABI (``gprx_comm_*``, RCCL loaded by libgprx.so at run time).
A one-GPU box can only form a world of one (RCCL refuses two ranks on one device): that still exercises the dlopen, the
communicator, every entry point and the device-resident buffers.  The two-rank test runs where two devices are visible
(it is skipped on a one-GPU box); the host logic of the sharding is covered on CPU by tests/test_parallel.py (gloo)."""

import ctypes as C
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

from gpras_amd import _lib
from gpras_amd._lib import DeviceBuffer

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def device_count(lib):
    n = C.c_int()
    assert lib.gprx_device_count(C.byref(n)) == 0
    return n.value


def test_world_of_one_every_entry_point(lib):
    from gpras_amd.comm import Communicator

    try:
        comm = Communicator.bootstrap(0, rank=0, world=1)
    except RuntimeError:
        print("LOADED:", sorted({ln.split()[-1] for ln in open("/proc/self/maps") if any(k in ln for k in ("hsa", "amdhip", "rccl"))}))
        raise
    try:
        a = np.arange(12, dtype=np.float64).reshape(3, 4) * 1.5
        (got,) = comm.all_gather(a)
        assert np.array_equal(got, a)
        assert comm.max(3.25) == 3.25
        send, recv = DeviceBuffer.from_array(a), DeviceBuffer(8 * 12)
        comm.all_gather_dev(send, recv, 12)
        comm.synchronize()
        assert np.array_equal(recv.to_array((3, 4)), a)
        recv2 = DeviceBuffer(8 * 12)
        comm.gather_dev(send, recv2, 12, root=0)
        comm.synchronize()
        assert np.array_equal(recv2.to_array((3, 4)), a)
        comm.barrier()
        rank, world = C.c_int(), C.c_int()
        assert lib.gprx_comm_rank(comm._c, C.byref(rank), C.byref(world)) == 0 and (rank.value, world.value) == (0, 1)
        with pytest.raises(ValueError):
            comm.gather_dev(send, recv2, 12, root=3)
    finally:
        comm.close()


WORKER = r"""
import os, sys
import numpy as np
sys.path.insert(0, {root!r})
from gpras_amd.comm import Communicator
from gpras_amd._lib import DeviceBuffer
rank, world, id_file = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
comm = Communicator.bootstrap(rank, rank=rank, world=world, id_file=id_file)   # device = rank
mine = np.arange(1000, dtype=np.float64) + 1000.0 * rank
parts = comm.all_gather(mine)
assert all(np.array_equal(parts[r], np.arange(1000) + 1000.0 * r) for r in range(world))
send, recv = DeviceBuffer.from_array(mine, rank), DeviceBuffer(8 * 1000 * world, rank)
comm.all_gather_dev(send, recv, 1000); comm.synchronize()
got = recv.to_array((world, 1000))
assert all(np.array_equal(got[r], np.arange(1000) + 1000.0 * r) for r in range(world))
root_recv = DeviceBuffer(8 * 1000 * world, rank) if rank == 1 else None
comm.gather_dev(send, root_recv, 1000, root=1); comm.synchronize()
if rank == 1:
    got = root_recv.to_array((world, 1000))
    assert all(np.array_equal(got[r], np.arange(1000) + 1000.0 * r) for r in range(world))
assert comm.max(float(rank)) == world - 1.0
comm.barrier(); comm.close()
print("rank", rank, "ok")
"""


def test_two_ranks_over_rccl(lib):
    if device_count(lib) < 2:
        pytest.skip("needs two GPUs (one process per GPU)")
    with tempfile.TemporaryDirectory() as tmp:
        script = os.path.join(tmp, "worker.py")
        with open(script, "w") as f:
            f.write(WORKER.format(root=ROOT))
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs = [subprocess.Popen([sys.executable, script, str(r), "2", os.path.join(tmp, "rccl.id")], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
                 for r in range(2)]
        outs = [p.communicate(timeout=300)[0].decode() for p in procs]
        assert all(p.returncode == 0 for p in procs), outs
