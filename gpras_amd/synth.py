"""Synthetic inputs for benchmarks and parity tests.

The reference standardises GP features to zero mean / unit standard deviation
(``/root/reference/gpras/preprocess.py:1037``), so ``X ~ N(0, 1)`` is the realistic
distribution.  Seeds: ``1000 * config + unit``.
"""

from __future__ import annotations

import numpy as np


def make_regression(n, d, n_outputs=1, n_test=0, config=0, unit=0):
    """``X (n, d)``, ``y (n, n_outputs)`` standardised, ``X* (n_test, d)``; all float64, C order."""
    rng = np.random.default_rng(1000 * config + unit)
    x = rng.standard_normal((n, d))
    w = rng.standard_normal((d, n_outputs)) / np.sqrt(d)
    y = np.sin(x @ w) + 0.1 * rng.standard_normal((n, n_outputs))
    y = (y - y.mean(axis=0)) / y.std(axis=0)
    xs = rng.standard_normal((n_test, d))
    return np.ascontiguousarray(x), np.ascontiguousarray(y), np.ascontiguousarray(xs)


def make_hydrograph_features(n, d, n_outputs=1, config=0, unit=0):
    """Config 1 flavour: feature j is a gamma-shaped pulse ``t^a exp(-t/b)`` sampled at n times."""
    rng = np.random.default_rng(1000 * config + unit)
    t = np.linspace(0.05, 10.0, n)
    a = 1.5 + rng.uniform(-0.5, 0.5, size=d)
    b = 1.0 + rng.uniform(-0.3, 0.3, size=d)
    x = t[:, None] ** a[None, :] * np.exp(-t[:, None] / b[None, :])
    x = x + 0.02 * rng.standard_normal(x.shape)
    x = (x - x.mean(axis=0)) / x.std(axis=0)
    w = rng.standard_normal((d, n_outputs)) / np.sqrt(d)
    y = np.tanh(x @ w) + 0.05 * rng.standard_normal((n, n_outputs))
    y = (y - y.mean(axis=0)) / y.std(axis=0)
    return np.ascontiguousarray(x), np.ascontiguousarray(y)
