"""Field metrics: CPU pins of the oracle restatement by hand loops, GPU parity of the fused
pass through the C ABI.  Index and count outputs (peak timesteps, contingency counts, fidelity matches) must be exact;
sums are compared at 1e-12 relative (summation order)."""

import numpy as np
import pytest

from oracle import metrics as om


def fields(t, cells, seed, ties=True):
    rng = np.random.default_rng(seed)
    base = np.sin(np.linspace(0, 3.0, t))[:, None] * (1.0 + rng.random(cells)) + 0.3 * rng.standard_normal((t, cells))
    x = np.maximum(base, 0.0)
    y = np.maximum(base + 0.1 * rng.standard_normal((t, cells)), 0.0)
    if ties:  # repeated maxima: argmax must pick the first
        x[t // 2, ::7] = x[t // 3, ::7] = x.max() + 1.0
        y[:, ::11] = 0.0  # all-zero columns: argmax 0
    conf = rng.random((t, cells))
    return np.ascontiguousarray(x), np.ascontiguousarray(y), conf


def test_oracle_against_hand_loops():
    x, y, conf = fields(9, 13, 1)
    t, c = x.shape
    se2 = sum((x[i, j] - y[i, j]) ** 2 for i in range(t) for j in range(c))
    assert om.rmse_aoi_toi(x, y) == pytest.approx((se2 / (t * c)) ** 0.5, rel=1e-14)
    for tol_t, tol_v in ((0, 0.05), (2, 0.05), (3, 0.0)):
        count = 0
        for j in range(c):
            for i in range(t):
                m = abs(y[i, j] - x[i, j]) <= tol_v
                for k in range(1, tol_t + 1):
                    if i + k < t:
                        m = m or abs(y[i, j] - x[i + k, j]) <= tol_v or abs(x[i, j] - y[i + k, j]) <= tol_v
                count += bool(m)
        assert om.fi_aoi_toi(x, y, tol_t, tol_v) == pytest.approx(count / (t * c), rel=1e-15)
    xp = np.array([max(x[:, j]) for j in range(c)])
    yp = np.array([max(y[:, j]) for j in range(c)])
    assert np.array_equal(om.err_cell_mts(x, y), xp - yp)
    a = sum(1 for j in range(c) if xp[j] >= 0.5 and yp[j] >= 0.5)
    miss = sum(1 for j in range(c) if xp[j] >= 0.5 and yp[j] < 0.5)
    assert om.pod_mts(x, y, 0.5) == pytest.approx(a / (a + miss))


@pytest.mark.gpu
@pytest.mark.parametrize("t,cells,t_tol,v_tol", [(40, 1000, 0, 0.0), (97, 3001, 2, 0.05), (5, 70, 8, 0.2), (1, 300, 3, 0.1)])
def test_gpu_metrics_match_oracle(t, cells, t_tol, v_tol):
    from gpras_amd import metrics as gm

    x, y, conf = fields(t, cells, 20 + t)
    fm = gm.FieldMetrics(x, y, conf, t_tol=t_tol, v_tol=v_tol)
    assert np.array_equal(fm.x_mts, np.argmax(x, axis=0)) and np.array_equal(fm.y_mts, np.argmax(y, axis=0))
    assert np.array_equal(fm.x_peak, x.max(axis=0)) and np.array_equal(fm.y_peak, y.max(axis=0))
    assert fm.fi_aoi_toi() == om.fi_aoi_toi(x, y, t_tol, v_tol)
    for thr in (0.0, 0.5, 1.5):
        assert tuple(int(v) for v in fm.contingency(thr)) == tuple(int(v) for v in om.contingency(x, y, thr))
    rel = 1e-12
    assert fm.rmse_aoi_toi() == pytest.approx(om.rmse_aoi_toi(x, y), rel=rel)
    assert fm.mae_aoi_toi() == pytest.approx(om.mae_aoi_toi(x, y), rel=rel)
    assert fm.err_aoi_toi() == pytest.approx(om.err_aoi_toi(x, y), rel=1e-10, abs=1e-15)
    assert fm.conf_aoi_toi() == pytest.approx(om.conf_aoi_toi(conf), rel=rel)
    assert fm.rmse_aoi_mts() == pytest.approx(om.rmse_aoi_mts(x, y), rel=rel)
    assert fm.err_aoi_mts() == pytest.approx(om.err_aoi_mts(x, y), rel=1e-10, abs=1e-15)
    if t > 1:
        assert fm.nse_aoi_mts() == pytest.approx(om.nse_aoi_mts(x, y), rel=1e-10)
    np.testing.assert_allclose(fm.rmse_aoi_ts(), om.rmse_aoi_ts(x, y), rtol=rel)
    np.testing.assert_allclose(fm.err_aoi_ts(), om.err_aoi_ts(x, y), rtol=1e-9, atol=1e-14)
    np.testing.assert_allclose(fm.conf_aoi_ts(), om.conf_aoi_ts(conf), rtol=rel)
    np.testing.assert_allclose(fm.rmse_cell_toi(), om.rmse_cell_toi(x, y), rtol=rel)
    np.testing.assert_allclose(fm.err_cell_toi(), om.err_cell_toi(x, y), rtol=1e-9, atol=1e-14)
    np.testing.assert_allclose(fm.conf_cell_toi(), om.conf_cell_toi(conf), rtol=rel)
    assert np.array_equal(fm.err_cell_mts(), om.err_cell_mts(x, y))
    assert fm.pod_mts(0.5) == om.pod_mts(x, y, 0.5) and fm.rfa_mts(0.5) == om.rfa_mts(x, y, 0.5)
    assert fm.f2_mts(0.5) == om.f2_mts(x, y, 0.5) and fm.f3_mts(0.5) == om.f3_mts(x, y, 0.5)
    # the reference's function names
    assert gm.rmse_aoi_toi(x, y) == fm.rmse_aoi_toi() and gm.fi_aoi_toi(x, y, t_tol, v_tol) == fm.fi_aoi_toi()
    assert gm.csi_mts(x, y, 0.5) == pytest.approx(om.csi_mts(x, y, 0.5), rel=1e-14)


@pytest.mark.gpu
def test_gpu_metrics_nan_peaks_and_errors():
    from gpras_amd import metrics as gm

    x, y, _ = fields(12, 130, 3, ties=False)
    x[4, 5] = np.nan
    x[7, 5] = np.nan
    y[0, 9] = np.nan
    fm = gm.FieldMetrics(x, y)
    assert np.array_equal(fm.x_mts, np.argmax(x, axis=0)) and np.array_equal(fm.y_mts, np.argmax(y, axis=0))  # first NaN wins, as numpy
    assert np.isnan(fm.x_peak[5]) and np.isnan(fm.y_peak[9])
    with pytest.raises(ValueError):
        gm.FieldMetrics(x, y[:, :-1])
    with pytest.raises(ValueError):
        gm.FieldMetrics(x, y, t_tol=9)


def _golden():
    import os

    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fields_golden.npz"))


def _scalars(mod_or_obj, x, y):
    f = mod_or_obj
    return np.array([f.rmse_aoi_toi(x, y), f.mae_aoi_toi(x, y), f.err_aoi_toi(x, y), f.rmse_aoi_mts(x, y), f.nse_aoi_mts(x, y), f.err_aoi_mts(x, y),
                     f.fi_aoi_toi(x, y, 2, 0.04), f.pod_mts(x, y, 101.0), f.rfa_mts(x, y, 101.0), f.csi_mts(x, y, 101.0), f.f2_mts(x, y, 101.0),
                     f.f3_mts(x, y, 101.0)])


def test_oracle_reproduces_golden_metrics():
    g = _golden()
    x, y = g["wse_full"], g["met_y"]
    np.testing.assert_allclose(_scalars(om, x, y), g["met_scalars"], rtol=1e-13)
    assert np.array_equal(np.argmax(x, axis=0), g["met_x_mts"]) and np.array_equal(np.argmax(y, axis=0), g["met_y_mts"])


@pytest.mark.gpu
def test_gpu_metrics_reproduce_golden():
    from gpras_amd import metrics as gm

    g = _golden()
    x, y = np.ascontiguousarray(g["wse_full"]), np.ascontiguousarray(g["met_y"])
    np.testing.assert_allclose(_scalars(gm, x, y), g["met_scalars"], rtol=1e-10)
    fm = gm.FieldMetrics(x, y)
    assert np.array_equal(fm.x_mts, g["met_x_mts"]) and np.array_equal(fm.y_mts, g["met_y_mts"])
    np.testing.assert_allclose(fm.rmse_aoi_ts(), g["met_rmse_ts"], rtol=1e-12)
    np.testing.assert_allclose(fm.rmse_cell_toi(), g["met_rmse_cell"], rtol=1e-12)
    assert np.array_equal(fm.err_cell_mts(), g["met_err_cell_mts"])


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(16))
def test_gpu_metrics_random_shapes(seed):
    from gpras_amd import metrics as gm

    rng = np.random.default_rng(500 + seed)
    t = int(rng.choice([1, 2, 3, 9, 10, 40]))
    cells = int(rng.choice([1, 2, 255, 256, 257, rng.integers(300, 4000)]))
    t_tol = int(rng.integers(0, 9))
    v_tol = float(rng.choice([0.0, 0.05, 0.5]))
    x, y, conf = fields(t, cells, 600 + seed, ties=bool(rng.integers(2)) and t >= 3 and cells >= 12)
    fm = gm.FieldMetrics(x, y, conf, t_tol=t_tol, v_tol=v_tol)
    assert np.array_equal(fm.x_mts, np.argmax(x, axis=0)) and np.array_equal(fm.y_mts, np.argmax(y, axis=0)), (t, cells)
    assert fm.fi_aoi_toi() == om.fi_aoi_toi(x, y, t_tol, v_tol), (t, cells, t_tol, v_tol)
    assert tuple(int(v) for v in fm.contingency(0.5)) == tuple(int(v) for v in om.contingency(x, y, 0.5))
    assert fm.rmse_aoi_toi() == pytest.approx(om.rmse_aoi_toi(x, y), rel=1e-12)
    np.testing.assert_allclose(fm.rmse_aoi_ts(), om.rmse_aoi_ts(x, y), rtol=1e-12)
    np.testing.assert_allclose(fm.rmse_cell_toi(), om.rmse_cell_toi(x, y), rtol=1e-12)
    np.testing.assert_allclose(fm.conf_cell_toi(), om.conf_cell_toi(conf), rtol=1e-12)
    np.testing.assert_allclose(fm.conf_aoi_ts(), om.conf_aoi_ts(conf), rtol=1e-12)
