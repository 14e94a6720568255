"""Device-resident predict -> reverse projection -> metrics (gpras_amd.pipeline) against the host chain of the same steps:
GPRAS.predict -> EOFProjector.reverse_transform -> wse_2_depth -> export_metric_summary
(/root/reference/production/analysis/pipeline.py:256-288)."""

import sqlite3

import numpy as np
import pandas as pd
import pytest

pytestmark = pytest.mark.gpu


def _setup(hp, n_inducing, rng, n=96, d=3, k=5, cells=41, t_star=37):
    from gpras_amd.gpr import GPRAS
    from gpras_amd.preprocess import EOFProjector

    x = rng.normal(size=(n, d))
    y = np.stack([np.sin(x @ rng.normal(size=d)) + 0.05 * rng.normal(size=n) for _ in range(k)], axis=1)
    gpr = GPRAS("Matern32")
    gpr.fit(x, y, n_inducing, "grid", "adam", max_iter=6)
    dry = np.zeros(cells, dtype=bool)
    dry[[3, 17, 40]] = True
    n_wet = cells - 3
    elev = rng.uniform(0.0, 2.0, size=cells)
    proj = EOFProjector(dry, elev, rng.normal(size=n_wet) + (0.0 if hp == "depth" else 1.5),
                        rng.uniform(0.5, 1.5, size=n_wet), rng.normal(size=(k, n_wet)) / np.sqrt(k), rng.normal(size=k), rng.uniform(0.5, 2, size=k),
                        hydraulic_parameter=hp)
    x_test = rng.normal(size=(t_star, d))
    truth = rng.uniform(0.0, 3.0, size=(t_star, cells)) + (0.0 if hp == "velocity" else elev)
    index = pd.MultiIndex.from_tuples([("e1", t) for t in range(20)] + [("e2", t) for t in range(t_star - 20)], names=["event", "timestep"])
    truth_df = pd.DataFrame(truth, index=index, columns=[f"c{j}" for j in range(cells)])
    return gpr, proj, x_test, truth_df, elev


def _host_chain(gpr, proj, x_test, truth, elev, hp):
    """pipeline.py:259-277 with numpy between the steps."""
    mean_pred, var_pred = gpr.predict(x_test)
    y_pred, y_var = proj.reverse_transform(mean_pred, var_pred)
    if hp != "velocity":
        if hp == "depth":
            y_pred += elev
        d = truth - elev
        d[d < 0] = 0
        p = y_pred - elev
        p[p < 0] = 0
        return d, p, np.sqrt(y_var)
    return truth, y_pred, np.sqrt(y_var)


@pytest.mark.parametrize("hp", ["wse", "depth", "velocity"])
@pytest.mark.parametrize("n_inducing", [16, None])
def test_fields_equal_host_chain(lib, hp, n_inducing):
    from gpras_amd.pipeline import DevicePipeline

    rng = np.random.default_rng(7 + len(hp))
    gpr, proj, x_test, truth_df, elev = _setup(hp, n_inducing, rng)
    pipe = DevicePipeline(gpr, proj)
    fields = pipe.predict_fields(x_test)
    pred, conf = fields.to_host()
    fields.close()
    _, want_pred, want_conf = _host_chain(gpr, proj, x_test, truth_df.values.copy(), elev, hp)
    # sparse and exact models alike: the same batched kernels with the same chunking on both sides (GPRAS.predict goes through
    # gprx_predict_batch, the device chain through gprx_predict_batch_dev), and the field kernels restate the numpy expressions
    # operation by operation
    np.testing.assert_array_equal(pred, want_pred)
    np.testing.assert_array_equal(conf, want_conf)
    truth_dev = pipe.truth_depth_dev(truth_df.values)
    got_truth = truth_dev.to_array(truth_df.shape)
    truth_dev.free()
    want_truth, _, _ = _host_chain(gpr, proj, x_test, truth_df.values.copy(), elev, hp)
    np.testing.assert_array_equal(got_truth, want_truth)


def test_fields_equal_host_chain_at_a_production_shape(lib):
    """10 sparse modes, 3 000 test timesteps, 30 011 cells (ragged against every tile size): the (modes, T*) block, its transpose,
    the reconstruction and the depth conversion over 90 M field values -- still equal to the host chain bit for bit."""
    from gpras_amd.pipeline import DevicePipeline

    rng = np.random.default_rng(41)
    gpr, proj, x_test, truth_df, elev = _setup("depth", 24, rng, n=700, d=6, k=10, cells=30011, t_star=3000)
    pipe = DevicePipeline(gpr, proj)
    fields = pipe.predict_fields(x_test)
    pred, conf = fields.to_host()
    fields.close()
    _, want_pred, want_conf = _host_chain(gpr, proj, x_test, truth_df.values.copy(), elev, "depth")
    np.testing.assert_array_equal(pred, want_pred)
    np.testing.assert_array_equal(conf, want_conf)
    assert np.isfinite(pred).all() and (pred >= 0).all() and (conf >= 0).all()


@pytest.mark.parametrize("hp", ["wse", "velocity"])
def test_metric_summary_equals_host_chain(lib, tmp_path, hp):
    from gpras_amd.metrics import export_metric_summary
    from gpras_amd.pipeline import DevicePipeline

    rng = np.random.default_rng(23)
    gpr, proj, x_test, truth_df, elev = _setup(hp, 16, rng)
    pipe = DevicePipeline(gpr, proj)
    dev_db = tmp_path / "dev.db"
    fields = pipe.export_metric_summary(x_test, truth_df, dev_db, depth_threshold=0.5, t_tol=1, v_tol=0.1)
    fields.close()
    d, p, c = _host_chain(gpr, proj, x_test, truth_df.values.copy(), elev, hp)
    host_db = tmp_path / "host.db"
    frame = lambda a: pd.DataFrame(a, index=truth_df.index, columns=truth_df.columns)  # noqa: E731
    export_metric_summary(frame(d), frame(p), frame(c), host_db, depth_threshold=0.5, t_tol=1, v_tol=0.1)
    for table in ("scalar_metrics", "timeseries_metrics", "cell_metrics"):
        with sqlite3.connect(dev_db) as con:
            got = pd.read_sql(f"select * from {table}", con)
        with sqlite3.connect(host_db) as con:
            want = pd.read_sql(f"select * from {table}", con)
        pd.testing.assert_frame_equal(got, want, check_exact=True)
    assert len(got) == truth_df.shape[1] * 2


def test_rejects_mismatched_inputs(lib):
    from gpras_amd.pipeline import DevicePipeline

    rng = np.random.default_rng(3)
    gpr, proj, x_test, truth_df, _ = _setup("wse", 16, rng)
    pipe = DevicePipeline(gpr, proj)
    with pytest.raises(ValueError):
        pipe.predict_fields(x_test[:, :2])
    with pytest.raises(ValueError):
        pipe.export_metric_summary(x_test[:5], truth_df, "/tmp/unused.db")
    shuffled = truth_df.iloc[np.r_[0:10, 25:37, 10:25]]
    with pytest.raises(ValueError, match="contiguous"):
        pipe.export_metric_summary(x_test, shuffled, "/tmp/unused.db")
