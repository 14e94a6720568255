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


def make_eof_state(n_cells: int, k: int, n_samples: int, seed: int, dry_fraction: float = 0.15, weighted: bool = True):
    """Synthetic fitted state of an EOF projector and matching fields: orthonormal EOF rows
    over the wet cells, cell-area-like weights, terrain elevations, water-surface samples of shape (n_samples, n_cells)."""
    rng = np.random.default_rng(seed)
    dry = rng.random(n_cells) < dry_fraction
    n_wet = int(n_cells - dry.sum())
    elevations = 100.0 + 5.0 * rng.standard_normal(n_cells)
    q, _ = np.linalg.qr(rng.standard_normal((n_wet, k)))
    eofs = np.ascontiguousarray(q.T)
    weights = 0.5 + rng.random(n_wet) if weighted else None
    modes = rng.standard_normal((n_samples, k)) * np.linspace(3.0, 0.3, k)
    wse = np.tile(elevations, (n_samples, 1))
    depth_wet = 1.5 + (modes @ eofs) / (weights if weighted else 1.0) + 0.05 * rng.standard_normal((n_samples, n_wet))
    wse[:, ~dry] += depth_wet
    wse[:, dry] -= 0.2  # below the terrain: always dry
    input_mean = wse[:, ~dry].mean(axis=0)
    proj = ((wse[:, ~dry] - input_mean) * (weights if weighted else 1.0)) @ eofs.T
    return {
        "dry": dry, "elevations": elevations, "input_mean": input_mean, "weights": weights, "eofs": eofs,
        "x_mean": proj.mean(axis=0), "x_std": proj.std(axis=0), "x": wse,
    }
