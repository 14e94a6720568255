"""Handle-level wrapper of the C ABI: one `Engine` = one gprx handle = one training set on one GPU.

The engine is the only backend the package ships.  It raises (never falls back) when
libgprx.so is missing or a device call fails.
"""

from __future__ import annotations

import ctypes as C
import functools
import threading

import numpy as np

from . import _lib
from ._lib import KERNEL_IDS, as_f64, check, ptr


def _locked(method):
    """A gprx handle is not thread-safe (one stream, shared staging and arenas): every call that touches it holds the
    engine's lock, so optimiser threads that share an engine (``gpras_amd.lockstep``) serialise on it."""

    @functools.wraps(method)
    def wrapper(self, *args, **kwargs):
        with self._lock:
            return method(self, *args, **kwargs)

    return wrapper


class Engine:
    """Owns the device copy of ``x (N, d)``, ``y (N, K)`` and all factorisation workspaces.

    ``n_inducing = 0`` selects the exact GP; otherwise the sparse model with that many inducing points
    (``SGPR`` at ``/root/reference/gpras/gpr.py:299``).
    """

    def __init__(self, kernel: str, x, y, n_inducing: int = 0, ard: bool = False, device: int = 0, distance_form: str = "difference"):
        self._lib = _lib.load()
        self.kernel = kernel
        kernel_id = KERNEL_IDS[kernel]  # KeyError for unknown names, as gpr.py:230
        x = as_f64(x)
        y = as_f64(y)
        if x.ndim != 2 or y.ndim != 2 or x.shape[0] != y.shape[0]:
            raise ValueError("x must be (N, d) and y (N, K) with matching N")
        self.x = x  # host copy (the multi-start driver samples Z inside its bounding box)
        self.n, self.d = x.shape
        self.n_units = y.shape[1]
        self.m = int(n_inducing or 0)
        self.ard = bool(ard)
        self.n_len = self.d if self.ard else 1
        self.n_theta = 2 + self.n_len
        self.device = device
        self._lock = threading.RLock()
        self._h = C.c_void_p()
        check(self._lib.gprx_create(device, self.n, self.d, self.m, kernel_id, int(self.ard), C.byref(self._h)))
        check(self._lib.gprx_set_data(self._h, ptr(x), ptr(y), self.n_units), self._h)
        self.distance_form = "difference"
        if distance_form != "difference":
            self.set_distance_form(distance_form)

    @_locked
    def set_distance_form(self, form: str) -> None:
        """``"difference"`` (default: r2 = sum ((a - b) / l)^2) or ``"expanded"`` (gpflow's literal
        ``|a|^2 + |b|^2 - 2 a.b``) for every kernel evaluation of this engine (``gprx_set_distance_form``)."""
        check(self._lib.gprx_set_distance_form(self._h, _lib.DISTANCE_FORMS[form]), self._h)
        self.distance_form = form

    # -- lifetime -------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.gprx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- evaluations -----------------------------------------------------------------------------
    def _z_ptr(self, z):
        if self.m == 0:
            return None, None
        z = as_f64(z)
        if z.shape != (self.m, self.d):
            raise ValueError(f"Z must have shape {(self.m, self.d)}, got {z.shape}")
        return z, ptr(z)

    @_locked
    def objective(self, unit: int, theta, z, mask: int, want_grad: bool = True):
        """``training_loss`` and (optionally) its gradient ``[d theta | d Z]`` for one output unit."""
        theta = as_f64(theta)
        if theta.shape != (self.n_theta,):
            raise ValueError(f"theta must have {self.n_theta} entries")
        zk, zp = self._z_ptr(z)
        loss = C.c_double()
        if want_grad and self.m > 64 and self.d <= 64:
            # a sparse model with M > 64 inducing points: its evaluation with the gradient goes through the batched launches with ONE cell
            # (round 4): the launch sequence replayed from a graph instead of ~40 launches and a dozen copies; the values are
            # gprx_objective's bit for bit (tests/test_gpu_sgpr.py).  NOTE: this route leaves NO resident factorisation -- ``predict``
            # after it raises (GPRX_ESTATE: "gprx_predict before a successful gprx_factorize / gprx_objective"); call
            # ``objective(..., want_grad=False)`` (= gprx_factorize) first, as GPModel.predict does.  M <= 64 (round 5) takes
            # gprx_objective below: the five-launch evaluation of sgpr_fused.h, which keeps the factorisation in cell block 0 for predict.
            losses = np.full(1, np.nan)
            grad = np.full((1, self.n_theta + self.m * self.d), np.nan)
            units = np.array([unit], dtype=np.int32)
            rc = self._lib.gprx_objective_batch(self._h, 1, ptr(units), ptr(theta), zp, mask, ptr(losses), ptr(grad))
            check(rc, self._h)
            return float(losses[0]), grad[0]
        if want_grad:
            grad = np.zeros(self.n_theta + self.m * self.d)
            check(self._lib.gprx_objective(self._h, unit, ptr(theta), zp, mask, C.byref(loss), ptr(grad)), self._h)
            return loss.value, grad
        check(self._lib.gprx_factorize(self._h, unit, ptr(theta), zp, mask, C.byref(loss)), self._h)
        return loss.value, None

    @_locked
    def factorize_batch(self, units, thetas, mask: int):
        """Exact models only: factorise ``len(units)`` cells -- cell ``i`` = output unit ``units[i]`` with
        hyperparameters ``thetas[i]`` -- by one batched launch sequence (``gprx_factorize_batch``).  Returns
        ``(losses, ok)``: the training losses (NaN where the kernel matrix was not positive definite) and a boolean
        array; the factorisations stay resident in slots ``0..len(units)-1`` (see ``select_slot``)."""
        units = np.ascontiguousarray(units, dtype=np.int32)
        thetas = as_f64(thetas)
        if thetas.shape != (units.size, self.n_theta):
            raise ValueError(f"thetas must be ({units.size}, {self.n_theta})")
        losses = np.full(units.size, np.nan)
        status = np.zeros(units.size, dtype=np.int32)
        rc = self._lib.gprx_factorize_batch(self._h, units.size, ptr(units), ptr(thetas), mask, ptr(losses), ptr(status))
        if rc not in (_lib.GPRX_OK, _lib.GPRX_ENOTPD):
            check(rc, self._h)
        return losses, status == 0

    @_locked
    def objective_batch(self, units, thetas, mask: int, want_grad: bool = True, zs=None):
        """``training_loss`` (and gradient ``[d theta | d Z]``) of ``len(units)`` cells by batched launches
        (``gprx_objective_batch``): exact models, or sparse models with one ``Z`` per cell in ``zs (cells, M, d)``.
        Returns ``(losses, grads or None, ok)``; failed cells hold NaN."""
        units = np.ascontiguousarray(units, dtype=np.int32)
        thetas = as_f64(thetas)
        if thetas.shape != (units.size, self.n_theta):
            raise ValueError(f"thetas must be ({units.size}, {self.n_theta})")
        zp = None
        if self.m != 0:
            zs = as_f64(zs)
            if zs.shape != (units.size, self.m, self.d):
                raise ValueError(f"zs must be ({units.size}, {self.m}, {self.d})")
            zp = ptr(zs)
        # (NaN, not uninitialised memory: a cell the library did not reach must never look like a valid evaluation -- the losses are
        # pre-filled; the gradient block, 64 KB at 16 cells of M = 50, is filled below for the cells without a finite loss only)
        losses = np.full(units.size, np.nan)
        grads = np.empty((units.size, self.n_theta + self.m * self.d)) if want_grad else None
        rc = self._lib.gprx_objective_batch(self._h, units.size, ptr(units), ptr(thetas), zp, mask, ptr(losses), ptr(grads) if want_grad else None)
        if rc == _lib.GPRX_ENOMEM and units.size > 1:
            # the batch does not fit in device memory: two half batches (values do not depend on the batch composition)
            half = units.size // 2
            lo = self.objective_batch(units[:half], thetas[:half], mask, want_grad, None if zs is None else zs[:half])
            hi = self.objective_batch(units[half:], thetas[half:], mask, want_grad, None if zs is None else zs[half:])
            return (np.concatenate([lo[0], hi[0]]), None if not want_grad else np.concatenate([lo[1], hi[1]]), np.concatenate([lo[2], hi[2]]))
        if rc not in (_lib.GPRX_OK, _lib.GPRX_ENOTPD):
            check(rc, self._h)
        ok = np.isfinite(losses)
        if want_grad and rc != _lib.GPRX_OK:
            grads[~ok] = np.nan
        return losses, grads, ok

    @_locked
    def adam_batch(self, units, thetas, mask: int, max_iter: int, zs=None):
        """``_optimize_adam`` (gpr.py:147-173) for ``len(units)`` cells in lock step inside the library (``gprx_adam_batch``): one
        batched evaluation per step, the update on the C side.  Returns ``(thetas, zs, n_evals, batches)`` -- the optimiser's
        variables after the run (copies), evaluations per cell, batched evaluations made.  Raises as ``objective_batch`` does when
        a cell stops being positive definite (the arrays of that step are attached to the exception as ``.state``)."""
        units = np.ascontiguousarray(units, dtype=np.int32)
        thetas = np.array(thetas, dtype=np.float64, order="C")
        if thetas.shape != (units.size, self.n_theta):
            raise ValueError(f"thetas must be ({units.size}, {self.n_theta})")
        zp = None
        if self.m != 0:
            zs = np.array(zs, dtype=np.float64, order="C")
            if zs.shape != (units.size, self.m, self.d):
                raise ValueError(f"zs must be ({units.size}, {self.m}, {self.d})")
            zp = ptr(zs)
        n_evals = np.zeros(units.size, dtype=np.int32)
        batches = C.c_int()
        rc = self._lib.gprx_adam_batch(self._h, units.size, ptr(units), ptr(thetas), zp, int(mask), int(max_iter), ptr(n_evals), C.byref(batches))
        if rc != _lib.GPRX_OK:
            try:
                check(rc, self._h)
            except Exception as exc:  # noqa: BLE001
                exc.state = (thetas, zs, n_evals, batches.value)
                raise
        return thetas, (zs if self.m != 0 else None), n_evals, batches.value

    def max_cells(self, want_grad: bool = False, reserve: float = 0.15) -> int:
        """How many cells of a batched call fit into the free device memory (``gprx_cell_bytes`` against ``gprx_mem_info``,
        keeping ``reserve`` of the total free for predict tiles and other handles); at least 1."""
        nbytes, free, total = C.c_int64(), C.c_int64(), C.c_int64()
        check(self._lib.gprx_cell_bytes(self._h, int(want_grad), C.byref(nbytes)), self._h)
        check(self._lib.gprx_mem_info(self.device, C.byref(free), C.byref(total)))
        return max(1, int((free.value - reserve * total.value) // max(nbytes.value, 1)))

    @_locked
    def select_slot(self, slot: int):
        """Make slot ``slot`` of the last ``factorize_batch`` the current factorisation (for ``predict``)."""
        check(self._lib.gprx_select_slot(self._h, int(slot)), self._h)

    def last_batch_ms(self) -> float:
        ms = C.c_double()
        check(self._lib.gprx_last_batch_ms(self._h, C.byref(ms)), self._h)
        return ms.value

    @_locked
    def predict(self, xs, include_noise: bool = True):
        """Mean and variance at ``xs`` for the unit factorised by the last ``objective`` call -- any call for exact models and for
        sparse models with M <= 64; for sparse models with M > 64 only a call with ``want_grad=False`` (the gradient route of those goes
        through a one-cell batch that keeps no resident factorisation: a ``RuntimeError`` naming gprx_factorize says so)."""
        xs = as_f64(xs)
        if xs.ndim != 2 or xs.shape[1] != self.d:
            raise ValueError(f"x must be (N*, {self.d})")
        mean = np.empty(xs.shape[0])
        var = np.empty(xs.shape[0])
        check(self._lib.gprx_predict(self._h, ptr(xs), xs.shape[0], ptr(mean), ptr(var), int(include_noise)), self._h)
        return mean, var

    @_locked
    def predict_batch(self, units, thetas, xs, zs=None, include_noise: bool = True):
        """The whole predict loop of gpr.py:336-339 in one call (``gprx_predict_batch``): cell ``i`` = (``units[i]``,
        ``thetas[i]``, ``zs[i]`` for sparse models) is factorised -- all cells by one batched launch sequence -- and predicts at
        the shared ``xs``.  Returns ``(means, variances)``, each ``(cells, N*)``."""
        units = np.ascontiguousarray(units, dtype=np.int32)
        thetas = as_f64(thetas)
        xs = as_f64(xs)
        if thetas.shape != (units.size, self.n_theta):
            raise ValueError(f"thetas must be ({units.size}, {self.n_theta})")
        if xs.ndim != 2 or xs.shape[1] != self.d:
            raise ValueError(f"x must be (N*, {self.d})")
        zp = None
        if self.m != 0:
            zs = as_f64(zs)
            if zs.shape != (units.size, self.m, self.d):
                raise ValueError(f"zs must be ({units.size}, {self.m}, {self.d})")
            zp = ptr(zs)
        means = np.empty((units.size, xs.shape[0]))
        variances = np.empty((units.size, xs.shape[0]))
        check(self._lib.gprx_predict_batch(self._h, units.size, ptr(units), ptr(thetas), zp, ptr(xs), xs.shape[0], ptr(means), ptr(variances),
                                           int(include_noise)), self._h)
        return means, variances

    @_locked
    def predict_batch_t(self, units, thetas, xs, zs=None, include_noise: bool = True):
        """``predict_batch`` with the results in ``GPRAS.predict``'s layout, ``(N*, cells)`` (``gprx_predict_batch_t``: transposed on the
        device before they leave -- the host-side transposition of 2 x 40 MB was half of a 50-mode predict at 100 000 points)."""
        units = np.ascontiguousarray(units, dtype=np.int32)
        thetas = as_f64(thetas)
        xs = as_f64(xs)
        if thetas.shape != (units.size, self.n_theta):
            raise ValueError(f"thetas must be ({units.size}, {self.n_theta})")
        if xs.ndim != 2 or xs.shape[1] != self.d:
            raise ValueError(f"x must be (N*, {self.d})")
        zp = None
        if self.m != 0:
            zs = as_f64(zs)
            if zs.shape != (units.size, self.m, self.d):
                raise ValueError(f"zs must be ({units.size}, {self.m}, {self.d})")
            zp = ptr(zs)
        means = np.empty((xs.shape[0], units.size))
        variances = np.empty((xs.shape[0], units.size))
        check(self._lib.gprx_predict_batch_t(self._h, units.size, ptr(units), ptr(thetas), zp, ptr(xs), xs.shape[0], ptr(means), ptr(variances),
                                             int(include_noise)), self._h)
        return means, variances

    @_locked
    def predict_batch_dev(self, units, thetas, xs_dev, ns: int, means_dev, vars_dev, zs=None, include_noise: bool = True, wait: bool = True):
        """``predict_batch`` with the test points and the ``(cells, N*)`` results in device memory (``gprx_predict_batch_dev``)."""
        units = np.ascontiguousarray(units, dtype=np.int32)
        thetas = as_f64(thetas)
        if thetas.shape != (units.size, self.n_theta):
            raise ValueError(f"thetas must be ({units.size}, {self.n_theta})")
        zp = None
        if self.m != 0:
            zs = as_f64(zs)
            if zs.shape != (units.size, self.m, self.d):
                raise ValueError(f"zs must be ({units.size}, {self.m}, {self.d})")
            zp = ptr(zs)
        p = lambda b: b.ptr if hasattr(b, "ptr") else b  # noqa: E731
        check(self._lib.gprx_predict_batch_dev(self._h, units.size, ptr(units), ptr(thetas), zp, p(xs_dev), int(ns), p(means_dev), p(vars_dev),
                                               int(include_noise)), self._h)
        if wait:
            check(self._lib.gprx_synchronize(self._h), self._h)

    @_locked
    def predict_dev(self, xs_dev, ns: int, mean_dev, var_dev, include_noise: bool = True, wait: bool = True):
        """The same with device pointers (``gprx_predict_dev``): inputs and outputs stay resident in HBM."""
        p = lambda b: b.ptr if hasattr(b, "ptr") else b  # noqa: E731
        check(self._lib.gprx_predict_dev(self._h, p(xs_dev), int(ns), p(mean_dev), p(var_dev), int(include_noise)), self._h)
        if wait:
            check(self._lib.gprx_synchronize(self._h), self._h)

    @_locked
    def synchronize(self):
        check(self._lib.gprx_synchronize(self._h), self._h)

    def timings(self):
        ms = (C.c_double * 4)()
        self._lib.gprx_last_timings(self._h, ms)
        return {"kernel_build_ms": ms[0], "cholesky_ms": ms[1], "solves_ms": ms[2], "gradient_ms": ms[3]}

    @property
    def handle(self):
        return self._h
