"""Field metrics on the GPU -- SURVEY.md section 8(f) row N3, the functions of ``/root/reference/gpras/metrics.py:85-318``
under their own names.  ``FieldMetrics(x, y, conf)`` makes one fused evaluation of the fields in ``libgprx.so``
(``gprx_metrics``: per-timestep sums, per-cell sums, per-cell peaks with their timesteps, fidelity matches); every metric
is then a few scalar operations on those reductions.  The module-level functions keep the reference's signatures.

x = truth, y = prediction, both (timesteps, cells).  The reference's ``x_mts`` / ``y_mts`` arguments (cached argmax) are
accepted and ignored: the peaks come out of the same pass.
"""

from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import as_f64, check, ptr


class FieldMetrics:
    def __init__(self, x, y, conf=None, t_tol: int = 0, v_tol: float = 0.0, device: int = 0):
        x, y = as_f64(x), as_f64(y)
        if x.ndim != 2 or x.shape != y.shape:
            raise ValueError("x and y must be 2-D arrays of the same shape (timesteps, cells)")
        conf = None if conf is None else as_f64(conf)
        if conf is not None and conf.shape != x.shape:
            raise ValueError("conf must have the shape of x")
        self.rows, self.cells = x.shape
        self.t_tol, self.v_tol = int(t_tol), float(v_tol)
        self.has_conf = conf is not None
        row = np.empty((self.rows, 4))
        cell = np.empty((5, self.cells))
        arg = np.empty((2, self.cells), dtype=np.int32)
        matches = C.c_uint64()
        lib = _lib.load()
        check(lib.gprx_metrics(device, ptr(x), ptr(y), None if conf is None else ptr(conf), self.rows, self.cells, self.t_tol, self.v_tol,
                               ptr(row), ptr(cell), ptr(arg), C.byref(matches)))
        self.row_sum_e, self.row_sum_e2, self.row_sum_conf, self.row_sum_abs = row[:, 0], row[:, 1], row[:, 2], row[:, 3]
        self.cell_sum_e, self.cell_sum_e2, self.cell_sum_conf, self.x_peak, self.y_peak = cell
        self.x_mts, self.y_mts = arg[0].astype(np.int64), arg[1].astype(np.int64)
        self.matches = int(matches.value)

    # ---- scalars ----
    def rmse_aoi_toi(self) -> float:
        return float((self.row_sum_e2.sum() / (self.rows * self.cells)) ** 0.5)

    def mae_aoi_toi(self) -> float:
        return float(self.row_sum_abs.sum() / (self.rows * self.cells))

    def err_aoi_toi(self) -> float:
        return float(self.row_sum_e.sum() / (self.rows * self.cells))

    def conf_aoi_toi(self) -> float:
        return float(self.row_sum_conf.sum() / (self.rows * self.cells))

    def fi_aoi_toi(self) -> float:
        return float(self.matches / (self.rows * self.cells))

    def rmse_aoi_mts(self) -> float:
        return float((((self.x_peak - self.y_peak) ** 2).mean()) ** 0.5)

    def err_aoi_mts(self) -> float:
        return float((self.x_peak - self.y_peak).mean())

    def nse_aoi_mts(self) -> float:
        return float(1 - (np.sum((self.x_peak - self.y_peak) ** 2) / np.sum((self.x_peak - self.x_peak.mean()) ** 2)))

    def contingency(self, depth_threshold=0):
        xw, yw = self.x_peak >= depth_threshold, self.y_peak >= depth_threshold
        return np.sum(xw & yw), np.sum(~xw & yw), np.sum(xw & ~yw)

    def pod_mts(self, depth_threshold=0) -> float:
        a, _, c = self.contingency(depth_threshold)
        return float(a / (a + c))

    def rfa_mts(self, depth_threshold=0) -> float:
        a, b, _ = self.contingency(depth_threshold)
        return float(b / (a + b))

    def csi_mts(self, depth_threshold=0) -> float:
        pod, rfa = self.pod_mts(depth_threshold), self.rfa_mts(depth_threshold)
        return float(1 / ((1 / pod) + (1 / (1 - rfa)) - 1))

    def f2_mts(self, depth_threshold=0):
        a, b, c = self.contingency(depth_threshold)
        return 1 if a + b + c == 0 else float((a - c) / (a + b + c))

    def f3_mts(self, depth_threshold=0):
        a, b, c = self.contingency(depth_threshold)
        return 1 if a + b + c == 0 else float((a - b) / (a + b + c))

    # ---- per timestep / per cell ----
    def rmse_aoi_ts(self):
        return (self.row_sum_e2 / self.cells) ** 0.5

    def err_aoi_ts(self):
        return self.row_sum_e / self.cells

    def conf_aoi_ts(self):
        return self.row_sum_conf / self.cells

    def rmse_cell_toi(self):
        return (self.cell_sum_e2 / self.rows) ** 0.5

    def err_cell_toi(self):
        return self.cell_sum_e / self.rows

    def conf_cell_toi(self):
        return self.cell_sum_conf / self.rows

    def err_cell_mts(self):
        return self.x_peak - self.y_peak


# ---- the reference's function names (gpras/metrics.py:85-318) -----------------------------------------------------
def rmse_aoi_toi(x, y): return FieldMetrics(x, y).rmse_aoi_toi()  # noqa: E704
def mae_aoi_toi(x, y): return FieldMetrics(x, y).mae_aoi_toi()  # noqa: E704
def conf_aoi_toi(x): return FieldMetrics(x, x, x).conf_aoi_toi()  # noqa: E704
def rmse_aoi_ts(x, y): return FieldMetrics(x, y).rmse_aoi_ts()  # noqa: E704
def rmse_cell_toi(x, y): return FieldMetrics(x, y).rmse_cell_toi()  # noqa: E704
def rmse_aoi_mts(x, y, x_mts=None, y_mts=None): return FieldMetrics(x, y).rmse_aoi_mts()  # noqa: E704
def err_cell_mts(x, y, x_mts=None, y_mts=None): return FieldMetrics(x, y).err_cell_mts()  # noqa: E704
def nse_aoi_mts(x, y, x_mts=None, y_mts=None): return FieldMetrics(x, y).nse_aoi_mts()  # noqa: E704
def err_aoi_toi(x, y): return FieldMetrics(x, y).err_aoi_toi()  # noqa: E704
def err_aoi_mts(x, y, x_mts=None, y_mts=None): return FieldMetrics(x, y).err_aoi_mts()  # noqa: E704
def err_aoi_ts(x, y): return FieldMetrics(x, y).err_aoi_ts()  # noqa: E704
def conf_aoi_ts(x): return FieldMetrics(x, x, x).conf_aoi_ts()  # noqa: E704
def err_cell_toi(x, y): return FieldMetrics(x, y).err_cell_toi()  # noqa: E704
def conf_cell_toi(x): return FieldMetrics(x, x, x).conf_cell_toi()  # noqa: E704
def fi_aoi_toi(x, y, t_tol, v_tol): return FieldMetrics(x, y, t_tol=t_tol, v_tol=v_tol).fi_aoi_toi()  # noqa: E704
def pod_mts(x, y, depth_threshold=0, x_mts=None, y_mts=None): return FieldMetrics(x, y).pod_mts(depth_threshold)  # noqa: E704
def rfa_mts(x, y, depth_threshold=0, x_mts=None, y_mts=None): return FieldMetrics(x, y).rfa_mts(depth_threshold)  # noqa: E704
def csi_mts(x, y, depth_threshold=0, x_mts=None, y_mts=None): return FieldMetrics(x, y).csi_mts(depth_threshold)  # noqa: E704
def f2_mts(x, y, depth_threshold=0, x_mts=None, y_mts=None): return FieldMetrics(x, y).f2_mts(depth_threshold)  # noqa: E704
def f3_mts(x, y, depth_threshold=0, x_mts=None, y_mts=None): return FieldMetrics(x, y).f3_mts(depth_threshold)  # noqa: E704
