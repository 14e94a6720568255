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
