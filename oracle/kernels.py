"""Stationary kernels of the gpras hot path (oracle, numpy).

Restates what ``KERNEL_FACTORY`` selects in ``/root/reference/gpras/gpr.py:21-37``
for the five kernels the reference can actually construct with
``kernel(variance=1, lengthscales=ini_length)`` (``gpr.py:298``; SURVEY.md
section 8a row a1): gpflow ``Matern12``, ``Matern32``, ``Matern52``,
``SquaredExponential`` ("RBF") and ``Exponential``.

Third-party algorithm restated (gpflow 2.x ``kernels/stationaries.py``; the
package is unpinned and absent from this image -- SURVEY.md section 8-A):

* scaled squared distance ``r2(a, b) = sum_k ((a_k - b_k) / l_k)^2``.  gpflow
  evaluates it in the expanded form ``|a|^2 + |b|^2 - 2 a.b`` (``form="expanded"``
  here); the HIP path and the default of this oracle use the algebraically
  identical difference form (``form="direct"``), which has no cancellation and
  gives ``r2(a, a) == 0`` exactly.  The two differ by O(eps * |a|^2 / l^2); tests
  measure the effect on every output.
* ``r = sqrt(max(r2, 1e-36))`` for the kernels written in ``r``;
  the squared exponential uses ``r2`` directly.
* ``k = variance * g(r)``;  ``k(x, x) = variance``.
"""

from __future__ import annotations

import numpy as np

KERNEL_IDS = {"RBF": 0, "Matern12": 1, "Matern32": 2, "Matern52": 3, "Exponential": 4}
KERNEL_NAMES = tuple(KERNEL_IDS)
R2_FLOOR = 1e-36
SQRT3 = np.sqrt(3.0)
SQRT5 = np.sqrt(5.0)


def scaled_sqdist(a: np.ndarray, b: np.ndarray, lengthscales, form: str = "direct") -> np.ndarray:
    """(n1, n2) matrix of ``sum_k ((a_ik - b_jk) / l_k)^2``; ``lengthscales`` scalar or (d,)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    ls = np.broadcast_to(np.asarray(lengthscales, dtype=np.float64), (a.shape[1],))
    sa = a / ls
    sb = b / ls
    if form == "direct":
        out = np.zeros((a.shape[0], b.shape[0]))
        for k in range(a.shape[1]):
            diff = sa[:, k][:, None] - sb[:, k][None, :]
            out += diff * diff
        return out
    if form == "expanded":
        # gpflow.utilities.ops.square_distance
        return (sa * sa).sum(1)[:, None] + (sb * sb).sum(1)[None, :] - 2.0 * (sa @ sb.T)
    raise ValueError(form)


def g_of_r2(kernel: str, r2: np.ndarray) -> np.ndarray:
    """Correlation ``g`` (k = variance * g) as a function of the scaled squared distance."""
    if kernel == "RBF":
        return np.exp(-0.5 * r2)
    r = np.sqrt(np.maximum(r2, R2_FLOOR))
    if kernel == "Matern12":
        return np.exp(-r)
    if kernel == "Matern32":
        return (1.0 + SQRT3 * r) * np.exp(-SQRT3 * r)
    if kernel == "Matern52":
        return (1.0 + SQRT5 * r + (5.0 / 3.0) * r * r) * np.exp(-SQRT5 * r)
    if kernel == "Exponential":
        return np.exp(-0.5 * r)
    raise KeyError(kernel)


def h_of_r2(kernel: str, r2: np.ndarray) -> np.ndarray:
    """``h = 2 dg/d(r2) = g'(r)/r``, so that ``dk/dx_k = variance * h * (a_k - b_k) / l_k^2``.

    Where ``r2 < 1e-36`` gpflow's ``maximum`` stops the gradient, so ``h = 0`` there
    for the kernels written in ``r``.
    """
    if kernel == "RBF":
        return -np.exp(-0.5 * r2)
    live = r2 >= R2_FLOOR
    r = np.sqrt(np.maximum(r2, R2_FLOOR))
    if kernel == "Matern12":
        h = -np.exp(-r) / r
    elif kernel == "Matern32":
        h = -3.0 * np.exp(-SQRT3 * r)
    elif kernel == "Matern52":
        h = -(5.0 / 3.0) * (1.0 + SQRT5 * r) * np.exp(-SQRT5 * r)
    elif kernel == "Exponential":
        h = -0.5 * np.exp(-0.5 * r) / r
    else:
        raise KeyError(kernel)
    return np.where(live, h, 0.0)


def kmat(kernel: str, a, b, variance: float, lengthscales, form: str = "direct") -> np.ndarray:
    """``k(a, b)``, shape (n1, n2)."""
    return variance * g_of_r2(kernel, scaled_sqdist(a, b, lengthscales, form))
