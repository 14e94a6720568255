"""CPU restatement (test infrastructure) of the field metrics -- SURVEY.md section 8(f) row N3,
``/root/reference/gpras/metrics.py:85-318``.  Plain numpy in the reference too; each function follows the reference's
expression (same reductions, same argmax / threshold conventions).  x = truth, y = prediction, both (timesteps, cells).
PINNED by outputs of the reference itself: ``tests/golden/make_golden_metrics_ref.py`` imports the reference module in the
build container (its imports are numpy / pandas / sqlite3 only) and writes ``tests/golden/metrics_ref_golden.npz``;
``tests/test_metrics_ref.py`` checks every function here, and ``export_metric_tables``, against it."""

from __future__ import annotations

import numpy as np


def _peaks(x, y, x_mts=None, y_mts=None):
    if x_mts is None:
        x_mts = np.argmax(x, axis=0)
    if y_mts is None:
        y_mts = np.argmax(y, axis=0)
    cols = np.arange(x.shape[1])
    return x[x_mts, cols], y[y_mts, cols]


def rmse_aoi_toi(x, y):  # metrics.py:85-87
    return float((((x - y) ** 2).mean()) ** 0.5)


def mae_aoi_toi(x, y):  # :90-92
    return float(np.abs(x - y).mean())


def conf_aoi_toi(c):  # :95-97
    return float(c.mean())


def rmse_aoi_ts(x, y):  # :100-102
    return (((x - y) ** 2).mean(axis=1)) ** 0.5


def rmse_cell_toi(x, y):  # :105-107
    return (((x - y) ** 2).mean(axis=0)) ** 0.5


def rmse_aoi_mts(x, y, x_mts=None, y_mts=None):  # :110-121
    xp, yp = _peaks(x, y, x_mts, y_mts)
    return float((((xp - yp) ** 2).mean()) ** 0.5)


def err_cell_mts(x, y, x_mts=None, y_mts=None):  # :124-135
    xp, yp = _peaks(x, y, x_mts, y_mts)
    return xp - yp


def nse_aoi_mts(x, y, x_mts=None, y_mts=None):  # :138-151
    xp, yp = _peaks(x, y, x_mts, y_mts)
    return float(1 - (np.sum((xp - yp) ** 2) / np.sum((xp - xp.mean()) ** 2)))


def err_aoi_toi(x, y):  # :154-156
    return float((x - y).mean())


def err_aoi_mts(x, y, x_mts=None, y_mts=None):  # :159-171
    xp, yp = _peaks(x, y, x_mts, y_mts)
    return float((xp - yp).mean())


def err_aoi_ts(x, y):  # :174-176
    return (x - y).mean(axis=1)


def conf_aoi_ts(c):  # :179-181
    return c.mean(axis=1)


def err_cell_toi(x, y):  # :184-186
    return (x - y).mean(axis=0)


def conf_cell_toi(c):  # :189-191
    return c.mean(axis=0)


def fi_aoi_toi(x, y, t_tol, v_tol):  # :194-204
    matching = np.abs(y - x) <= v_tol
    for i in range(1, t_tol + 1):
        tmp = np.abs(y[:-i, :] - x[i:, :]) <= v_tol
        matching[:-i] = tmp | matching[:-i]
    for i in range(1, t_tol + 1):
        tmp = np.abs(x[:-i, :] - y[i:, :]) <= v_tol
        matching[:-i] = tmp | matching[:-i]
    return float(np.sum(matching) / (matching.shape[0] * matching.shape[1]))


def contingency(x, y, depth_threshold=0, x_mts=None, y_mts=None):
    """a (both wet), b (false alarm), c (missed) at each cell's peak (:207-318)."""
    xp, yp = _peaks(x, y, x_mts, y_mts)
    a = np.sum((xp >= depth_threshold) * (yp >= depth_threshold))
    b = np.sum((xp < depth_threshold) * (yp >= depth_threshold))
    c = np.sum((xp >= depth_threshold) * (yp < depth_threshold))
    return a, b, c


def pod_mts(x, y, depth_threshold=0, x_mts=None, y_mts=None):  # :207-224
    a, _, c = contingency(x, y, depth_threshold, x_mts, y_mts)
    return float(a / (a + c))


def rfa_mts(x, y, depth_threshold=0, x_mts=None, y_mts=None):  # :227-244
    a, b, _ = contingency(x, y, depth_threshold, x_mts, y_mts)
    return float(b / (a + b))


def csi_mts(x, y, depth_threshold=0, x_mts=None, y_mts=None):  # :247-260
    pod = pod_mts(x, y, depth_threshold, x_mts, y_mts)
    rfa = rfa_mts(x, y, depth_threshold, x_mts, y_mts)
    return float(1 / ((1 / pod) + (1 / (1 - rfa)) - 1))


def f2_mts(x, y, depth_threshold=0, x_mts=None, y_mts=None):  # :263-289 (called with x_mts in the threshold slot at :56)
    a, b, c = contingency(x, y, depth_threshold, x_mts, y_mts)
    return 1 if a + b + c == 0 else float((a - c) / (a + b + c))


def f3_mts(x, y, depth_threshold=0, x_mts=None, y_mts=None):  # :292-318
    a, b, c = contingency(x, y, depth_threshold, x_mts, y_mts)
    return 1 if a + b + c == 0 else float((a - b) / (a + b + c))


def export_metric_tables(x_all, y_all, conf_all, depth_threshold=0.5, t_tol=0, v_tol=0, hydraulic_parameter="depth"):
    """The three tables that ``export_metric_summary`` (metrics.py:11-82) writes to sqlite, as DataFrames
    (scalar_metrics, timeseries_metrics, cell_metrics).  The calls are the reference's, positional quirk of :56-57 included."""
    import pandas as pd

    all_scalar, all_timeseries, all_cells = [], [], []
    for event in x_all.index.unique(level=0):  # :26
        x = x_all.loc[event].values
        y = y_all.loc[event].values
        conf = conf_all.loc[event].values
        tsteps = x_all.loc[event].index.values
        x_mts = np.argmax(x, axis=0)  # :35-36
        y_mts = np.argmax(y, axis=0)
        wet = hydraulic_parameter != "velocity"
        all_scalar.append(pd.DataFrame.from_dict({  # :39-58
            "event": event,
            "rmse_aoi_toi": [rmse_aoi_toi(x, y)],
            "mae_aoi_toi": [mae_aoi_toi(x, y)],
            "conf_aoi_toi": [conf_aoi_toi(conf)],
            "rmse_aoi_mts": [rmse_aoi_mts(x, y, x_mts, y_mts)],
            "nse_aoi_mts": [nse_aoi_mts(x, y, x_mts, y_mts)],
            "err_aoi_toi": [err_aoi_toi(x, y)],
            "err_aoi_mts": [err_aoi_mts(x, y, x_mts, y_mts)],
            "fi_aoi_toi": [fi_aoi_toi(x, y, t_tol, v_tol)],
            "pod_mts": [pod_mts(x, y, depth_threshold, x_mts, y_mts)] if wet else [np.nan],
            "rfa_mts": [rfa_mts(x, y, depth_threshold, x_mts, y_mts)] if wet else [np.nan],
            "csi_mts": [csi_mts(x, y, depth_threshold, x_mts, y_mts)] if wet else [np.nan],
            "f2_mts": [f2_mts(x, y, x_mts, y_mts)],  # :56: x_mts in the depth_threshold slot, y_mts in the x_mts slot
            "f3_mts": [f3_mts(x, y, x_mts, y_mts)],  # :57
        }))
        all_timeseries.append(pd.DataFrame.from_dict({  # :61-68
            "event": np.repeat(event, x.shape[0]),
            "timestep": tsteps,
            "rmse_aoi_ts": rmse_aoi_ts(x, y),
            "err_aoi_ts": err_aoi_ts(x, y),
            "conf_aoi_ts": conf_aoi_ts(conf),
        }))
        all_cells.append(pd.DataFrame.from_dict({  # :71-79
            "event": np.repeat(event, x.shape[1]),
            "cell_id": x_all.columns,
            "rmse_cell_toi": rmse_cell_toi(x, y),
            "err_cell_mts": err_cell_mts(x, y, x_mts, y_mts),
            "err_cell_toi": err_cell_toi(x, y),
            "conf_cell_toi": conf_cell_toi(conf),
        }))
    return pd.concat(all_scalar), pd.concat(all_timeseries), pd.concat(all_cells)
