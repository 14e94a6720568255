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
        outs = []
        try:
            for p in procs:
                outs.append(p.communicate(timeout=300)[0].decode())
        finally:
            for p in procs:  # a rank that is still alive holds the GPU: end exactly the processes started here
                if p.poll() is None:
                    p.kill()
                    p.wait()
        assert all(p.returncode == 0 for p in procs), outs


ONE_RUNTIME = r"""
import json, os, sys
sys.path.insert(0, {root!r})
import numpy as np
from gpras_amd.comm import Communicator
comm = Communicator.bootstrap(0, rank=0, world=1)
(got,) = comm.all_gather(np.arange(5.0))
comm.barrier()
libs = sorted({{ln.split()[-1] for ln in open("/proc/self/maps") if "libamdhip64" in ln or "librccl" in ln}})
comm.close()
print(json.dumps({{"libs": libs, "torch_imported": "torch" in sys.modules, "ok": bool(np.array_equal(got, np.arange(5.0)))}}))
"""


def test_one_hip_runtime_and_one_rccl_per_rank():
    """a rank of the torch-free launch maps exactly ONE libamdhip64 and ONE librccl (round 2 mixed torch's
    own ROCm tree with /opt/rocm's and failed in ncclCommInitRank)."""
    import json

    res = subprocess.run([sys.executable, "-c", ONE_RUNTIME.format(root=ROOT)], capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", GPRX_COMM_DEBUG="1"))
    assert res.returncode == 0, res.stderr[-3000:]
    info = json.loads(res.stdout.strip().splitlines()[-1])
    assert info["ok"] and not info["torch_imported"]
    hip = [p for p in info["libs"] if "libamdhip64" in p]
    rccl = [p for p in info["libs"] if "librccl" in p]
    assert len(hip) == 1 and len(rccl) == 1, info["libs"]
    assert os.path.dirname(hip[0]) == os.path.dirname(rccl[0])  # one ROCm tree


def test_bench_line_of_a_launched_world_of_one():
    """bench.py as a launched rank (RANK / WORLD_SIZE in the environment, as torch.distributed.run sets them): torch-free, the
    collective is gprx_comm_all_gather, one JSON line on stdout."""
    import json

    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1", "--cells", "8", "--no-extras",
                          "--batched-only"], capture_output=True, text=True, timeout=900, env=env)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["value"] > 0 and "roofline" in out
    assert "gprx_comm_all_gather" in out["config"]["collective"] and out["config"]["launcher"].startswith("torch-free")
    assert len([p for p in out["config"]["hip_and_rccl_libraries_mapped"] if "libamdhip64" in p]) == 1
    assert out["config"]["rccl_ranks_seen"] == {"ncclCommCount": 1, "distinct_ranks_gathered": 1}


def test_bench_c4_leg_of_a_launched_world_of_one():
    """the C4 leg of bench.py -- every rank predicts its share into ONE device block through gprx_predict_batch_dev in
    chunks, then ONE gprx_comm_gather of the blocks to rank 0, timed by itself -- as a launched world of one at a reduced size (the same
    code path the driver's N > 1 launches take; the gather of a world of one is the root's own send / receive pair)."""
    import json

    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29535")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1", "--cells", "4", "--only-c4",
                          "--c4-cells", "10", "--c4-points", "3000"], capture_output=True, text=True, timeout=900, env=env)
    assert res.returncode == 0, res.stderr[-3000:]
    out = json.loads([ln for ln in res.stdout.splitlines() if ln.strip()][-1])
    c4 = out["C4"]
    assert "error" not in c4, c4
    assert c4["cells_per_gpu"] == 10 and c4["n_test"] == 3000 and c4["cells_per_chunk"] == 4  # three chunks, the last one ragged
    assert c4["seconds_this_gpu"] > 0 and c4["C4_gather_GBps"] > 0 and c4["gathered_blocks_checked"] is True
    assert "SCALED" in c4["seconds_for_10k_cells_on_8_gpus_is"]  # a reduced run never claims to be the measurement


def test_bench_line_when_the_communicator_cannot_be_created():
    """bench.py as a launched rank without an RCCL communicator (forced here).  Default: NO measurement through files --
    exit code 3 and the reason on stderr, so that a scaling record can never show shard scaling where the RCCL gather was asked for.
    With the explicit opt-in GPRX_BENCH_FILE_EXCHANGE=1: barriers, maximum and the final gather through files, and the line says so."""
    import json

    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29534", GPRX_BENCH_FORCE_COMM_FAILURE="1")
    env.pop("GPRX_BENCH_FILE_EXCHANGE", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1", "--cells", "8", "--no-extras", "--batched-only"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert res.returncode == 3, (res.returncode, res.stderr[-3000:])
    assert "could not be created" in res.stderr and "forced by GPRX_BENCH_FORCE_COMM_FAILURE" in res.stderr
    assert not [ln for ln in res.stdout.splitlines() if ln.strip().startswith("{")]  # no JSON line
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=dict(env, GPRX_BENCH_FILE_EXCHANGE="1"))
    assert res.returncode == 0, res.stderr[-3000:]
    out = json.loads([ln for ln in res.stdout.splitlines() if ln.strip()][-1])
    assert out["n_gpus"] == 1 and out["value"] > 0 and out["config"]["collective"].startswith("FILE EXCHANGE")
    assert out["config"]["rccl_ranks_seen"] is None
    assert "could not be created" in res.stderr
