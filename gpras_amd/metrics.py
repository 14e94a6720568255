"""Field metrics on the GPU -- SURVEY.md section 8(f) row N3: ``/root/reference/gpras/metrics.py`` under its own names.

``FieldMetrics(x, y, conf)`` uploads the fields ONCE, keeps them resident in HBM and makes one fused evaluation in
``libgprx.so`` (``gprx_metrics_dev``: per-timestep sums, per-cell sums, per-cell peaks with their timesteps, fidelity
matches); every metric is then a few scalar operations on those reductions.  The ``*_mts`` metrics honour caller-supplied
``x_mts`` / ``y_mts`` exactly as the reference does (``x[x_mts, np.arange(cells)]``, metrics.py:119-121): the values at those
timesteps are gathered on the device from the resident fields (``gprx_gather_rows``); without them the peaks of the fused
pass are used (numpy ``argmax`` semantics).  ``depth_threshold`` may be an array (one value per cell), which is what the
reference's own call ``f2_mts(x, y, x_mts, y_mts)`` (metrics.py:56-57) passes.

``export_metric_summary`` is the reference's only caller of these functions (``production/analysis/pipeline.py:15``); it is
mirrored here with ONE ``FieldMetrics`` per event serving all of the event's metrics.  The module-level functions keep the
reference's signatures; each builds its own ``FieldMetrics`` (no caching across calls: the arrays may change in between).

x = truth, y = prediction, both (timesteps, cells).
"""

from __future__ import annotations

import ctypes as C
import sqlite3
from pathlib import Path

import numpy as np

from . import _lib
from ._lib import DeviceBuffer, as_f64, check, ptr


class _View:
    """A device pointer that is not owned (``free`` does nothing)."""

    def __init__(self, p):
        self.ptr = p.ptr if hasattr(p, "ptr") else p

    def free(self):
        pass


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
        self.device = device
        self._lib = _lib.load()
        # the fields stay on the device for the lifetime of this object (the gathers at caller-supplied timesteps read them)
        self._dx = DeviceBuffer.from_array(x, device)
        self._dy = self._dx if y is x else DeviceBuffer.from_array(y, device)
        self._dc = None if conf is None else (self._dx if conf is x else DeviceBuffer.from_array(conf, device))
        self._evaluate()

    @classmethod
    def from_device(cls, x_dev, y_dev, conf_dev, rows: int, cells: int, t_tol: int = 0, v_tol: float = 0.0, device: int = 0) -> "FieldMetrics":
        """The same over fields that are ALREADY in device memory ((rows, cells) row-major float64; pointers or anything with a
        ``.ptr``): nothing is uploaded and nothing is owned -- the caller keeps the buffers alive while the ``*_mts`` metrics
        with caller-supplied timesteps may still gather from them (``gpras_amd.pipeline``)."""
        self = cls.__new__(cls)
        self.rows, self.cells = int(rows), int(cells)
        if self.rows <= 0 or self.cells <= 0:
            raise ValueError("x and y must be 2-D arrays of the same shape (timesteps, cells)")
        self.t_tol, self.v_tol = int(t_tol), float(v_tol)
        self.has_conf = conf_dev is not None
        self.device = device
        self._lib = _lib.load()
        self._dx, self._dy = _View(x_dev), _View(y_dev)
        self._dc = None if conf_dev is None else _View(conf_dev)
        self._evaluate()
        return self

    def _evaluate(self):
        device = self.device
        drow = DeviceBuffer(8 * self.rows * 4, device)
        dcell = DeviceBuffer(8 * self.cells * 5, device)
        darg = DeviceBuffer(4 * self.cells * 2, device)
        matches = C.c_uint64()
        check(self._lib.gprx_metrics_dev(device, self._dx.ptr, self._dy.ptr, None if self._dc is None else self._dc.ptr, self.rows, self.cells,
                                         self.t_tol, self.v_tol, drow.ptr, dcell.ptr, darg.ptr, C.byref(matches)))
        row = drow.to_array((self.rows, 4))
        cell = dcell.to_array((5, self.cells))
        arg = np.empty((2, self.cells), dtype=np.int32)
        check(self._lib.gprx_memcpy_d2h(device, ptr(arg), darg.ptr, arg.nbytes))
        for b in (drow, dcell, darg):
            b.free()
        self.row_sum_e, self.row_sum_e2, self.row_sum_conf, self.row_sum_abs = row[:, 0], row[:, 1], row[:, 2], row[:, 3]
        self.cell_sum_e, self.cell_sum_e2, self.cell_sum_conf, self.x_peak, self.y_peak = cell
        self.x_mts, self.y_mts = arg[0].astype(np.int64), arg[1].astype(np.int64)
        self.matches = int(matches.value)

    def close(self):
        for name in ("_dx", "_dy", "_dc"):
            buf = getattr(self, name, None)
            if buf is not None:
                buf.free()
            setattr(self, name, None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- values at given timesteps -------------------------------------------------------------------------------
    def _gather(self, buf: DeviceBuffer, idx) -> np.ndarray:
        idx = np.asarray(idx)
        if idx.shape != (self.cells,):
            raise IndexError(f"shape mismatch: indexing arrays could not be broadcast together with shapes {idx.shape} ({self.cells},)")
        if idx.dtype.kind not in "iu" and idx.dtype.kind != "b":
            raise IndexError("arrays used as indices must be of integer (or boolean) type")
        idx = np.ascontiguousarray(idx, dtype=np.int64)
        out = np.empty(self.cells)
        rc = self._lib.gprx_gather_rows(self.device, buf.ptr, self.rows, self.cells, ptr(idx), ptr(out))
        if rc == _lib.GPRX_EINVAL:
            raise IndexError(_lib.last_error())  # numpy: "index ... is out of bounds for axis 0 with size ..."
        check(rc)
        return out

    def peaks(self, x_mts=None, y_mts=None):
        """``x[x_mts, cols], y[y_mts, cols]``; ``None`` = the argmax over time (metrics.py:116-119 and every *_mts function)."""
        xp = self.x_peak if x_mts is None else self._gather(self._dx, x_mts)
        yp = self.y_peak if y_mts is None else self._gather(self._dy, y_mts)
        return xp, yp

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

    def rmse_aoi_mts(self, x_mts=None, y_mts=None) -> float:
        xp, yp = self.peaks(x_mts, y_mts)
        return float((((xp - yp) ** 2).mean()) ** 0.5)

    def err_aoi_mts(self, x_mts=None, y_mts=None) -> float:
        xp, yp = self.peaks(x_mts, y_mts)
        return float((xp - yp).mean())

    def nse_aoi_mts(self, x_mts=None, y_mts=None) -> float:
        xp, yp = self.peaks(x_mts, y_mts)
        return float(1 - (np.sum((xp - yp) ** 2) / np.sum((xp - xp.mean()) ** 2)))

    def contingency(self, depth_threshold=0, x_mts=None, y_mts=None):
        """a (both wet), b (false alarm), c (missed) at each cell's peak (metrics.py:215-224, 235-244, 274-283)."""
        xp, yp = self.peaks(x_mts, y_mts)
        xw, yw = xp >= depth_threshold, yp >= depth_threshold
        xd, yd = xp < depth_threshold, yp < depth_threshold  # (not ~xw: a NaN is neither, as in the reference)
        return np.sum(xw * yw), np.sum(xd * yw), np.sum(xw * yd)

    def pod_mts(self, depth_threshold=0, x_mts=None, y_mts=None) -> float:
        a, _, c = self.contingency(depth_threshold, x_mts, y_mts)
        return float(a / (a + c))

    def rfa_mts(self, depth_threshold=0, x_mts=None, y_mts=None) -> float:
        a, b, _ = self.contingency(depth_threshold, x_mts, y_mts)
        return float(b / (a + b))

    def csi_mts(self, depth_threshold=0, x_mts=None, y_mts=None) -> float:
        pod, rfa = self.pod_mts(depth_threshold, x_mts, y_mts), self.rfa_mts(depth_threshold, x_mts, y_mts)
        return float(1 / ((1 / pod) + (1 / (1 - rfa)) - 1))  # ZeroDivisionError for pod == 0, as metrics.py:262

    def f2_mts(self, depth_threshold=0, x_mts=None, y_mts=None):
        a, b, c = self.contingency(depth_threshold, x_mts, y_mts)
        return 1 if a + b + c == 0 else float((a - c) / (a + b + c))

    def f3_mts(self, depth_threshold=0, x_mts=None, y_mts=None):
        a, b, c = self.contingency(depth_threshold, x_mts, y_mts)
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

    def err_cell_mts(self, x_mts=None, y_mts=None):
        xp, yp = self.peaks(x_mts, y_mts)
        return np.asarray(xp - yp, dtype=np.float64)


def export_metric_summary(x_all, y_all, conf_all, out_path: str | Path, depth_threshold: float = 0.5, t_tol: int = 0, v_tol: float = 0,
                          hydraulic_parameter: str = "depth", device: int = 0) -> None:
    """Export all metrics to a sqlite database (metrics.py:11-82): same tables, columns and call pattern -- including the
    positional ``f2_mts(x, y, x_mts, y_mts)`` / ``f3_mts`` calls of metrics.py:56-57, where the cached argmax of x lands in
    the ``depth_threshold`` slot and that of y in the ``x_mts`` slot.  One fused device evaluation per event."""
    all_scalar, all_timeseries, all_cells = [], [], []
    for event in x_all.index.unique(level=0):
        x = x_all.loc[event].values
        y = y_all.loc[event].values
        conf = conf_all.loc[event].values
        tsteps = x_all.loc[event].index.values
        fm = FieldMetrics(x, y, conf, t_tol=t_tol, v_tol=v_tol, device=device)
        scalar, series, cells = event_tables(fm, event, tsteps, x_all.columns, depth_threshold, hydraulic_parameter)
        all_scalar.append(scalar)
        all_timeseries.append(series)
        all_cells.append(cells)
        fm.close()
    write_metric_db(all_scalar, all_timeseries, all_cells, out_path)


def event_tables(fm: FieldMetrics, event, tsteps, columns, depth_threshold: float = 0.5, hydraulic_parameter: str = "depth"):
    """The three data frames of one event (metrics.py:31-76) from its fused evaluation ``fm``."""
    import pandas as pd

    x_mts, y_mts = fm.x_mts, fm.y_mts  # np.argmax(x, axis=0), np.argmax(y, axis=0) (metrics.py:35-36), from the fused pass
    wet = hydraulic_parameter != "velocity"
    scalar_dict = {
        "event": event,
        "rmse_aoi_toi": [fm.rmse_aoi_toi()],
        "mae_aoi_toi": [fm.mae_aoi_toi()],
        "conf_aoi_toi": [fm.conf_aoi_toi()],
        "rmse_aoi_mts": [fm.rmse_aoi_mts(x_mts, y_mts)],
        "nse_aoi_mts": [fm.nse_aoi_mts(x_mts, y_mts)],
        "err_aoi_toi": [fm.err_aoi_toi()],
        "err_aoi_mts": [fm.err_aoi_mts(x_mts, y_mts)],
        "fi_aoi_toi": [fm.fi_aoi_toi()],
        "pod_mts": [fm.pod_mts(depth_threshold, x_mts, y_mts)] if wet else [np.nan],
        "rfa_mts": [fm.rfa_mts(depth_threshold, x_mts, y_mts)] if wet else [np.nan],
        "csi_mts": [fm.csi_mts(depth_threshold, x_mts, y_mts)] if wet else [np.nan],
        "f2_mts": [fm.f2_mts(x_mts, y_mts)],  # positional, as the reference: threshold = x_mts, x_mts = y_mts
        "f3_mts": [fm.f3_mts(x_mts, y_mts)],
    }
    scalar = pd.DataFrame.from_dict(scalar_dict)
    series = pd.DataFrame.from_dict({
        "event": np.repeat(event, fm.rows),
        "timestep": tsteps,
        "rmse_aoi_ts": fm.rmse_aoi_ts(),
        "err_aoi_ts": fm.err_aoi_ts(),
        "conf_aoi_ts": fm.conf_aoi_ts(),
    })
    cells = pd.DataFrame.from_dict({
        "event": np.repeat(event, fm.cells),
        "cell_id": columns,
        "rmse_cell_toi": fm.rmse_cell_toi(),
        "err_cell_mts": fm.err_cell_mts(x_mts, y_mts),
        "err_cell_toi": fm.err_cell_toi(),
        "conf_cell_toi": fm.conf_cell_toi(),
    })
    return scalar, series, cells


def write_metric_db(all_scalar, all_timeseries, all_cells, out_path) -> None:
    import pandas as pd

    with sqlite3.connect(out_path) as con:
        pd.concat(all_scalar).to_sql("scalar_metrics", con, index=False, if_exists="replace")
        pd.concat(all_timeseries).to_sql("timeseries_metrics", con, index=False, if_exists="replace")
        pd.concat(all_cells).to_sql("cell_metrics", con, index=False, if_exists="replace")


# ---- the reference's function names and signatures (gpras/metrics.py:85-318) -----------------------------------------
def rmse_aoi_toi(x, y): return FieldMetrics(x, y).rmse_aoi_toi()  # noqa: E704
def mae_aoi_toi(x, y): return FieldMetrics(x, y).mae_aoi_toi()  # noqa: E704
def conf_aoi_toi(x): return FieldMetrics(x, x, x).conf_aoi_toi()  # noqa: E704
def rmse_aoi_ts(x, y): return FieldMetrics(x, y).rmse_aoi_ts()  # noqa: E704
def rmse_cell_toi(x, y): return FieldMetrics(x, y).rmse_cell_toi()  # noqa: E704
def rmse_aoi_mts(x, y, x_mts=None, y_mts=None): return FieldMetrics(x, y).rmse_aoi_mts(x_mts, y_mts)  # noqa: E704
def err_cell_mts(x, y, x_mts=None, y_mts=None): return FieldMetrics(x, y).err_cell_mts(x_mts, y_mts)  # noqa: E704
def nse_aoi_mts(x, y, x_mts=None, y_mts=None): return FieldMetrics(x, y).nse_aoi_mts(x_mts, y_mts)  # noqa: E704
def err_aoi_toi(x, y): return FieldMetrics(x, y).err_aoi_toi()  # noqa: E704
def err_aoi_mts(x, y, x_mts=None, y_mts=None): return FieldMetrics(x, y).err_aoi_mts(x_mts, y_mts)  # noqa: E704
def err_aoi_ts(x, y): return FieldMetrics(x, y).err_aoi_ts()  # noqa: E704
def conf_aoi_ts(x): return FieldMetrics(x, x, x).conf_aoi_ts()  # noqa: E704
def err_cell_toi(x, y): return FieldMetrics(x, y).err_cell_toi()  # noqa: E704
def conf_cell_toi(x): return FieldMetrics(x, x, x).conf_cell_toi()  # noqa: E704
def fi_aoi_toi(x, y, t_tol, v_tol): return FieldMetrics(x, y, t_tol=t_tol, v_tol=v_tol).fi_aoi_toi()  # noqa: E704
def pod_mts(x, y, depth_threshold=0, x_mts=None, y_mts=None): return FieldMetrics(x, y).pod_mts(depth_threshold, x_mts, y_mts)  # noqa: E704
def rfa_mts(x, y, depth_threshold=0, x_mts=None, y_mts=None): return FieldMetrics(x, y).rfa_mts(depth_threshold, x_mts, y_mts)  # noqa: E704
def csi_mts(x, y, depth_threshold=0, x_mts=None, y_mts=None): return FieldMetrics(x, y).csi_mts(depth_threshold, x_mts, y_mts)  # noqa: E704
def f2_mts(x, y, depth_threshold=0, x_mts=None, y_mts=None): return FieldMetrics(x, y).f2_mts(depth_threshold, x_mts, y_mts)  # noqa: E704
def f3_mts(x, y, depth_threshold=0, x_mts=None, y_mts=None): return FieldMetrics(x, y).f3_mts(depth_threshold, x_mts, y_mts)  # noqa: E704
