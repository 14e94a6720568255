"""World-size-2 gloo test (CPU) of the sharded path: units are split round-robin over ranks, each rank
runs its units with no communication, one all_gather at the end; results equal the single-process run."""

import os
import subprocess
import sys

import numpy as np

from gpras_amd import gpr, parallel
from gpras_amd.synth import make_regression

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_units_round_robin():
    assert parallel.shard_units(10, 0, 4) == [0, 4, 8]
    assert parallel.shard_units(10, 3, 4) == [3, 7]
    assert sorted(sum((parallel.shard_units(7, r, 3) for r in range(3)), [])) == list(range(7))


def test_two_rank_gloo_matches_single_process(tmp_path, monkeypatch):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29577", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29577", os.path.join(ROOT, "tests", "_parallel_worker.py"), str(tmp_path)]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    r0 = np.load(tmp_path / "rank0.npz")
    r1 = np.load(tmp_path / "rank1.npz")
    assert list(r0["owned"]) == [0, 2] and list(r1["owned"]) == [1]
    # every rank ends with all parameters and the full prediction
    for key in ("mean", "var", "params", "z"):
        assert np.array_equal(r0[key], r1[key])
    # and they equal the serial run of the same drivers
    from test_host_logic import OracleBackend

    monkeypatch.setattr(gpr, "Engine", OracleBackend)
    x, y, xs = make_regression(80, 3, n_outputs=3, n_test=17, config=8, unit=0)
    g = gpr.GPRAS("Matern32")
    g.fit(x, y, 8, "kmeans", "adam", max_iter=4)
    mean, var = g.predict(xs)
    assert np.allclose(r0["mean"], mean, rtol=1e-12, atol=1e-14) and np.allclose(r0["var"], var, rtol=1e-12)
    assert np.allclose(r0["z"], np.stack([m.Z for m in g.models]), rtol=1e-12, atol=1e-14)


# ---- torch-free rendezvous of the ranks (gpras_amd.comm.file_rendezvous: bench.py --gpus N and ShardedGPRAS without torch) ----
def _rdv_worker(prefix, rank, world, status, q):
    from gpras_amd.comm import file_rendezvous

    try:
        uid = file_rendezvous(prefix, rank, world, status, lambda: bytes(range(128)) + b"\n tail with a newline", timeout_s=20.0)
        q.put((rank, "id", uid))
    except Exception as exc:  # noqa: BLE001
        q.put((rank, type(exc).__name__, str(exc)))


def _run_rendezvous(prefix, statuses):
    import multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rdv_worker, args=(prefix, r, len(statuses), st, q)) for r, st in enumerate(statuses)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=60) for _ in procs)
    for p in procs:
        p.join(timeout=30)
    return out


def test_file_rendezvous_two_ranks_get_rank0s_id(tmp_path):
    out = _run_rendezvous(str(tmp_path / "rccl"), ["ok", "ok", "ok"])
    want = bytes(range(128)) + b"\n tail with a newline"  # arbitrary bytes, newlines included, arrive unchanged
    assert [(r, kind) for r, kind, _ in out] == [(0, "id"), (1, "id"), (2, "id")]
    assert all(payload == want for _, _, payload in out)


def test_file_rendezvous_one_failing_rank_stops_every_rank(tmp_path):
    """A rank that cannot load RCCL reports it; NO rank goes on to the collective initialisation (they would block there)."""
    out = _run_rendezvous(str(tmp_path / "rccl"), ["ok", "rank 1: cannot load RCCL (librccl.so.1)"])
    assert [(r, kind) for r, kind, _ in out] == [(0, "RuntimeError"), (1, "RuntimeError")]
    assert all("cannot load RCCL" in msg for _, _, msg in out)
    assert not os.path.exists(str(tmp_path / "rccl") + ".id")  # rank 0 never created an id


def test_file_rendezvous_ignores_stale_files_and_times_out_alone(tmp_path):
    from gpras_amd.comm import file_rendezvous

    import pytest

    prefix = str(tmp_path / "rccl")
    with open(prefix + ".ready.1", "wb") as f:  # leftover of a launch 2 hours ago
        f.write(b"%r\nok" % (__import__("time").time() - 7200.0))
    with pytest.raises(TimeoutError):
        file_rendezvous(prefix, 0, 2, "ok", lambda: b"x" * 128, timeout_s=0.5)


def _late_rdv_worker(prefix, rank, world, delay, q):
    import time

    time.sleep(delay)
    _rdv_worker(prefix, rank, world, "ok", q)


def test_file_rendezvous_is_not_fooled_by_fresh_leftovers_of_another_launch(tmp_path):
    """an earlier launch under the SAME explicit prefix died a minute ago and left `ready`, `ack` and `id` files that are
    still fresh by their time stamps.  Rank 1 of the new launch starts late: rank 0 first reads the old `ready.1` ("ok"), but must
    neither create an id on it nor may rank 1 take the old id -- the acks only match once everybody has read the current files."""
    import multiprocessing as mp
    import time

    prefix = str(tmp_path / "rccl")
    now = repr(time.time() - 60.0).encode()
    for name, body in ((".ready.0", b"111.1\nok"), (".ready.1", b"222.2\nok"), (".ack.0", b"0" * 20), (".ack.1", b"0" * 20),
                       (".id", b"0" * 20 + b"\n" + b"OLD" * 40)):
        with open(prefix + name, "wb") as f:
            f.write(now + b"\n" + body)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_late_rdv_worker, args=(prefix, r, 2, 1.5 * r, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=60) for _ in procs)
    for p in procs:
        p.join(timeout=30)
    want = bytes(range(128)) + b"\n tail with a newline"
    assert out == [(0, "id", want), (1, "id", want)]


def test_reports_of_another_launch_do_not_count(tmp_path):
    from gpras_amd.comm import FileExchange, report

    import pytest

    prefix = str(tmp_path / "rep")
    with open(prefix + ".comm.1", "wb") as f:  # rank 1 of an earlier launch said yes
        f.write(b"oldtag:1")
    assert report(prefix, "comm", 0, 2, True, timeout_s=0.3, tag="newtag") == [True, None]
    with open(prefix + ".fx.1.1", "wb") as f:
        f.write(b"oldtag\n")
    with pytest.raises(TimeoutError):
        FileExchange(prefix, 0, 2, timeout_s=0.3, tag="newtag").barrier()


def test_default_id_prefix_is_shared_by_siblings(monkeypatch):
    from gpras_amd.comm import default_id_prefix

    monkeypatch.delenv("GPRX_ID_FILE", raising=False)
    monkeypatch.setenv("MASTER_PORT", "29123")
    a = default_id_prefix()
    assert "29123" in a and str(os.getppid()) in a
    monkeypatch.setenv("GPRX_ID_FILE", "/tmp/explicit")
    assert default_id_prefix() == "/tmp/explicit"


def _agree_worker(prefix, rank, world, ok, q):
    from gpras_amd.comm import agree

    q.put((rank, agree(prefix, "comm", rank, world, ok, timeout_s=20.0)))


def test_agree_is_unanimous(tmp_path):
    import multiprocessing as mp

    ctx = mp.get_context("spawn")
    for votes, want in (([True, True, True], True), ([True, False, True], False)):
        prefix = str(tmp_path / f"agree{int(want)}")
        q = ctx.Queue()
        procs = [ctx.Process(target=_agree_worker, args=(prefix, r, len(votes), v, q)) for r, v in enumerate(votes)]
        for p in procs:
            p.start()
        got = sorted(q.get(timeout=60) for _ in procs)
        for p in procs:
            p.join(timeout=30)
        assert got == [(r, want) for r in range(len(votes))]


def _fx_worker(prefix, rank, world, q):
    from gpras_amd.comm import FileExchange

    fx = FileExchange(prefix, rank, world, timeout_s=60.0)
    fx.barrier()
    parts = fx.all_gather(np.arange(5.0) + 10.0 * rank)
    top = fx.max(1.5 * rank)
    for _ in range(4):  # (old rounds are removed while later ones run)
        fx.barrier()
    fx.close()
    q.put((rank, [p.tolist() for p in parts], top))


def test_file_exchange_barrier_gather_and_max_across_three_processes(tmp_path):
    """bench.py's last resort when the RCCL communicator cannot be created: the ranks' timing barriers, the final gather and the
    maximum over ranks through files of one node."""
    import multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    prefix = str(tmp_path / "fx")
    procs = [ctx.Process(target=_fx_worker, args=(prefix, r, 3, q)) for r in range(3)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(3))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, parts, top in got:
        assert parts == [(np.arange(5.0) + 10.0 * r).tolist() for r in range(3)] and top == 3.0
    assert not [f for f in os.listdir(tmp_path) if f.startswith("fx.fx.")]  # nothing is left behind
