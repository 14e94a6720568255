"""CPU restatement (test infrastructure) of the field metrics -- SURVEY.md section 8(f) row N3,
``/root/reference/gpras/metrics.py:85-318``.  Plain numpy in the reference too; each function follows the reference's
expression (same reductions, same argmax / threshold conventions).  x = truth, y = prediction, both (timesteps, cells).
PARITY UNPINNED against the real reference (it has no tests or fixtures for these functions)."""

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
