"""Predict -> reverse projection -> metrics without leaving the GPU -- the tail of the reference's analysis pipeline
(``/root/reference/production/analysis/pipeline.py:256-288``) as one object.

The reference's chain is ``gpr.predict(x_test)`` (gpr.py:322-342) -> ``hf_reducer.reverse_transform(mean, var)``
(preprocess.py:1052-1085) -> ``wse_2_depth`` of truth and prediction (pipeline.py:262-277, preprocess.py:1040-1044) ->
``export_metric_summary(truth, prediction, sqrt(var))`` (metrics.py:11-82), each step handing numpy arrays to the next.
``DevicePipeline`` runs the same steps with every intermediate resident in HBM:

* all modes are factorised and predicted by batched launches into one ``(modes, T*)`` device block
  (``gprx_predict_batch_dev``), transposed on the device to the ``(T*, modes)`` that ``reverse_transform`` consumes;
* ``gprx_pca_reverse_dev`` reconstructs the ``(T*, cells)`` field and its propagated variance;
* ``gprx_pca_to_depth_dev`` / ``gprx_pca_sqrt_dev`` apply pipeline.py:262-277 and the ``np.sqrt(y_test_var)`` of :286 in place;
* the truth field goes up once; one fused ``gprx_metrics_dev`` evaluation per event reads row ranges of the resident fields.

Only the reductions the metric tables are made of (per-timestep and per-cell sums, peaks) and, on request, the fields
themselves come back to the host.  The host chain (``GPRAS.predict`` -> ``EOFProjector.reverse_transform`` -> ``metrics``) stays
available; tests/test_gpu_pipeline.py holds the two against each other.
"""

from __future__ import annotations

from pathlib import Path
from typing import Any

import numpy as np

from . import _lib
from ._lib import DeviceBuffer, as_f64, check
from .metrics import FieldMetrics, event_tables, write_metric_db


class DeviceFields:
    """The reconstructed prediction on the device: ``pred`` and ``conf`` are ``(rows, cells)`` row-major float64 buffers."""

    def __init__(self, pred: DeviceBuffer, conf: DeviceBuffer, rows: int, cells: int):
        self.pred, self.conf, self.rows, self.cells = pred, conf, int(rows), int(cells)

    def to_host(self):
        return self.pred.to_array((self.rows, self.cells)), self.conf.to_array((self.rows, self.cells))

    def close(self):
        for buf in (self.pred, self.conf):
            if buf is not None:
                buf.free()
        self.pred = self.conf = None


class DevicePipeline:
    def __init__(self, gpr: Any, projector: Any):
        """``gpr``: a fitted (or loaded) ``gpras_amd.GPRAS``; ``projector``: the ``EOFProjector`` of the high-fidelity data
        (``hf_reducer`` in pipeline.py:218-230).  Both must live on the same device."""
        if not getattr(gpr, "models", None):
            raise ValueError("the GPRAS model has not been fitted or loaded")
        if len(gpr.models) != projector.spatial_mode_count:
            raise ValueError(f"the model has {len(gpr.models)} outputs, the projector {projector.spatial_mode_count} spatial modes")
        self.gpr, self.projector = gpr, projector
        self.device = gpr.device
        self._lib = _lib.load()

    # ---- pipeline.py:256-277 --------------------------------------------------------------------------------------------
    def predict_modes_dev(self, x_test) -> tuple[DeviceBuffer, DeviceBuffer, int]:
        """``gpr.predict(x_test)`` with the result left on the device: ``(mean, var)`` as ``(T*, modes)`` buffers, and ``T*``."""
        x = as_f64(np.asarray(x_test).astype(np.float64))
        models = self.gpr.models
        if x.ndim != 2 or x.shape[1] != self.gpr.x.shape[1]:
            raise ValueError(f"x must be (N*, {self.gpr.x.shape[1]})")
        if x.shape[1] > 64:
            raise ValueError("the batched kernels carry at most 64 input dimensions per cell")
        ns, k = x.shape[0], len(models)
        if ns == 0:
            raise ValueError("x_test has no rows")
        dev = self.device
        xs_dev = DeviceBuffer.from_array(x, dev)
        by_mode_mean = DeviceBuffer(8 * k * ns, dev)  # (modes, T*): the layout gprx_predict_batch_dev writes
        by_mode_var = DeviceBuffer(8 * k * ns, dev)
        try:
            by_engine: dict[int, list[int]] = {}
            for i, mod in enumerate(models):
                by_engine.setdefault(id(mod.backend), []).append(i)
            for idx in by_engine.values():
                eng = models[idx[0]].backend
                sparse = models[idx[0]].Z is not None
                chunk = max(1, eng.max_cells(want_grad=False))
                if sparse:
                    chunk = min(64, chunk)  # the chunking of GPRAS._predict_batched_sparse: the same numbers bit for bit
                # runs of consecutive modes land directly in their rows of the (modes, T*) block
                runs: list[list[int]] = []
                for i in idx:
                    if runs and runs[-1][-1] + 1 == i and len(runs[-1]) < chunk:
                        runs[-1].append(i)
                    else:
                        runs.append([i])
                for part in runs:
                    units = [models[i].unit for i in part]
                    thetas = np.stack([models[i].theta() for i in part])
                    zs = np.stack([models[i].Z for i in part]) if sparse else None
                    eng.predict_batch_dev(units, thetas, xs_dev, ns, by_mode_mean.at(part[0] * ns), by_mode_var.at(part[0] * ns), zs=zs,
                                          include_noise=True, wait=False)
                eng.synchronize()
            mean_t = DeviceBuffer(8 * k * ns, dev)
            var_t = DeviceBuffer(8 * k * ns, dev)
            ph = self.projector.handle
            check(self._lib.gprx_pca_transpose_dev(ph, by_mode_mean.ptr, k, ns, mean_t.ptr))
            check(self._lib.gprx_pca_transpose_dev(ph, by_mode_var.ptr, k, ns, var_t.ptr))
            check(self._lib.gprx_pca_synchronize(ph))
            return mean_t, var_t, ns
        finally:
            for buf in (xs_dev, by_mode_mean, by_mode_var):
                buf.free()

    def predict_fields(self, x_test) -> DeviceFields:
        """pipeline.py:259-277 and the ``np.sqrt`` of :286: the predicted field as the metrics see it (depths unless the
        hydraulic parameter is velocity) and its one-sigma confidence, both ``(T*, cells)`` on the device."""
        mean_t, var_t, ns = self.predict_modes_dev(x_test)
        pr = self.projector
        cells = pr.n_cells
        full = DeviceBuffer(8 * ns * cells, self.device)
        conf = DeviceBuffer(8 * ns * cells, self.device)
        try:
            ph = pr.handle
            check(self._lib.gprx_pca_reverse_dev(ph, mean_t.ptr, var_t.ptr, ns, full.ptr, conf.ptr))
            if pr.hydraulic_parameter != "velocity":
                self._require_elevations()
                check(self._lib.gprx_pca_to_depth_dev(ph, full.ptr, ns, int(pr.hydraulic_parameter == "depth")))
            check(self._lib.gprx_pca_sqrt_dev(ph, conf.ptr, ns * cells))
            check(self._lib.gprx_pca_synchronize(ph))
        except Exception:
            full.free()
            conf.free()
            raise
        finally:
            mean_t.free()
            var_t.free()
        return DeviceFields(full, conf, ns, cells)

    def _require_elevations(self):
        if self.projector.elevations is None:
            raise ValueError("wse_2_depth needs the cell elevations (the projector was built without them)")

    def truth_depth_dev(self, hf_test_data) -> DeviceBuffer:
        """``hf_reducer.wse_2_depth(hf_test_data)`` (pipeline.py:271; the truth is left as it is for velocity), on the device."""
        truth = as_f64(hf_test_data)
        if truth.ndim != 2 or truth.shape[1] != self.projector.n_cells:
            raise ValueError(f"hf_test_data must be (timesteps, {self.projector.n_cells})")
        buf = DeviceBuffer.from_array(truth, self.device)
        if self.projector.hydraulic_parameter != "velocity":
            self._require_elevations()
            check(self._lib.gprx_pca_to_depth_dev(self.projector.handle, buf.ptr, truth.shape[0], 0))
            check(self._lib.gprx_pca_synchronize(self.projector.handle))
        return buf

    # ---- pipeline.py:279-288 --------------------------------------------------------------------------------------------
    def export_metric_summary(self, x_test, hf_test_data_df, out_path: str | Path, depth_threshold: float = 0.5, t_tol: int = 0, v_tol: float = 0,
                              hydraulic_parameter: str = "depth"):
        """Predict at ``x_test`` and write the reference's sqlite tables for the truth ``hf_test_data_df`` (a data frame indexed
        by (event, timestep) with one column per cell, pipeline.py:279-288).  ``hydraulic_parameter`` is export_metric_summary's own argument (metrics.py:19), which the reference's
        pipeline leaves at its default whatever the configuration says.  Returns the ``DeviceFields`` (the caller closes them)."""
        if len(hf_test_data_df) != np.shape(x_test)[0]:
            raise ValueError("x_test and hf_test_data_df must have one row per test timestep")
        fields = self.predict_fields(x_test)
        truth = self.truth_depth_dev(hf_test_data_df.values)
        hp = hydraulic_parameter
        cells = fields.cells
        try:
            events = hf_test_data_df.index.get_level_values(0)
            all_scalar, all_series, all_cells = [], [], []
            for event in hf_test_data_df.index.unique(level=0):
                pos = np.flatnonzero(events == event)
                tsteps = hf_test_data_df.loc[event].index.values
                lo, n = int(pos[0]), int(pos.size)
                if not np.array_equal(pos, np.arange(lo, lo + n)):
                    raise ValueError(f"the rows of event {event!r} are not contiguous in hf_test_data_df (sort the index by event first)")
                fm = FieldMetrics.from_device(truth.at(lo * cells), fields.pred.at(lo * cells), fields.conf.at(lo * cells), n, cells, t_tol=t_tol,
                                              v_tol=v_tol, device=self.device)
                scalar, series, cell_tab = event_tables(fm, event, tsteps, hf_test_data_df.columns, depth_threshold, hp)
                all_scalar.append(scalar)
                all_series.append(series)
                all_cells.append(cell_tab)
            write_metric_db(all_scalar, all_series, all_cells, out_path)
        except Exception:
            fields.close()
            raise
        finally:
            truth.free()
        return fields
