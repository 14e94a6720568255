"""EOF (PCA) projection either side of the GP path on the GPU -- the next row after the GP path itself
(SURVEY.md section 8(f) N1): ``transform`` / ``reverse_transform`` of the reference's ``PreProcessor``
(``/root/reference/gpras/preprocess.py:1009-1038, 1052-1094``) with the same names, arguments and return shapes.

``EOFProjector`` holds the fitted state (it does not fit the PCA: the reference does that once with scikit-learn's
``IncrementalPCA`` on the host, preprocess.py:947-1007) and runs the two projections through ``libgprx.so``.
Difference from the reference: a ``PreProcessor`` fitted WITHOUT weights keeps ``weights = np.empty(0)``
(preprocess.py:917), so its ``transform`` fails on the broadcast at :1031; here missing or empty weights mean "unweighted".
"""

from __future__ import annotations

import ctypes as C
from typing import Any

import numpy as np

from . import _lib
from ._lib import as_f64, check, ptr


class EOFProjector:
    def __init__(self, dry_indices, elevations, input_mean, weights, eofs, x_mean, x_std, hydraulic_parameter: str = "wse", device: int = 0):
        if hydraulic_parameter not in ("wse", "depth", "velocity"):
            raise ValueError(f"unknown hydraulic_parameter {hydraulic_parameter!r}")
        self._lib = _lib.load()
        self.hydraulic_parameter = hydraulic_parameter
        self.dry_indices = np.ascontiguousarray(dry_indices, dtype=bool)
        self.elevations = None if elevations is None or np.size(elevations) == 0 else as_f64(elevations)
        self.input_mean = as_f64(input_mean)
        self.weights = None if weights is None or np.size(weights) == 0 else as_f64(weights)
        self.eofs = as_f64(np.atleast_2d(eofs))
        self.x_mean = as_f64(x_mean)
        self.x_std = as_f64(x_std)
        self.n_cells = int(self.dry_indices.size)
        self.spatial_mode_count = int(self.eofs.shape[0])
        n_wet = int(self.n_cells - self.dry_indices.sum())
        if self.eofs.shape[1] != n_wet or self.input_mean.shape != (n_wet,):
            raise ValueError("eofs must be (k, n_wet) and input_mean (n_wet,) over the cells that are not always dry")
        if self.weights is not None and self.weights.shape != (n_wet,):
            raise ValueError("weights must be (n_wet,)")
        if self.x_mean.shape != (self.spatial_mode_count,) or self.x_std.shape != (self.spatial_mode_count,):
            raise ValueError("x_mean and x_std must be (k,)")
        if self.elevations is not None and self.elevations.shape != (self.n_cells,):
            raise ValueError("elevations must be (n_cells,)")
        dry_u8 = np.ascontiguousarray(self.dry_indices, dtype=np.uint8)
        self._h = C.c_void_p()
        check(
            self._lib.gprx_pca_create(
                device, self.n_cells, self.spatial_mode_count, ptr(dry_u8), None if self.elevations is None else ptr(self.elevations),
                ptr(self.input_mean), None if self.weights is None else ptr(self.weights), ptr(self.eofs), ptr(self.x_mean), ptr(self.x_std),
                int(hydraulic_parameter == "depth"), C.byref(self._h),
            )
        )

    @classmethod
    def from_preprocessor(cls, pre: Any, device: int = 0) -> "EOFProjector":
        """Take the fitted state of a reference ``PreProcessor`` (attribute names of preprocess.py:868-927)."""
        return cls(pre.dry_indices, pre.elevations, pre.input_mean, pre.weights, pre.eofs, pre.x_mean, pre.x_std, pre.hydraulic_parameter, device=device)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.gprx_pca_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def transform(self, x):
        """(samples, cells) -> EOF space (samples, spatial_mode_count)  (preprocess.py:1009-1038)."""
        x = as_f64(x)
        if x.ndim != 2 or x.shape[1] != self.n_cells:
            raise ValueError(f"x must be (samples, {self.n_cells})")
        z = np.empty((x.shape[0], self.spatial_mode_count))
        check(self._lib.gprx_pca_transform(self._h, ptr(x), x.shape[0], ptr(z)))
        return z

    def reverse_transform(self, mean, var=None):
        """EOF space -> (samples, cells); with ``var`` also the propagated variance  (preprocess.py:1052-1085)."""
        mean = as_f64(mean)
        if mean.ndim != 2 or mean.shape[1] != self.spatial_mode_count:
            raise ValueError(f"mean must be (samples, {self.spatial_mode_count})")
        full = np.empty((mean.shape[0], self.n_cells))
        if var is None:
            check(self._lib.gprx_pca_reverse(self._h, ptr(mean), None, mean.shape[0], ptr(full), None))
            return full
        var = as_f64(var)
        if var.shape != mean.shape:
            raise ValueError("var must have the shape of mean")
        var_full = np.empty_like(full)
        check(self._lib.gprx_pca_reverse(self._h, ptr(mean), ptr(var), mean.shape[0], ptr(full), ptr(var_full)))
        return full, var_full

    @property
    def handle(self):
        return self._h
