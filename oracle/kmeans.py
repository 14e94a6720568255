"""CPU restatement (test infrastructure) of the k-means inducing-point initialisation -- SURVEY.md section 8(f) row N4,
``/root/reference/gpras/gpr.py:312-315``: ``KMeans(n_clusters=M, random_state=0, n_init="auto").fit(x).cluster_centers_``.

The algorithm lives in scikit-learn (installed here, 1.7.2; unpinned in the reference): ``KMeans.fit`` centres the data,
seeds ONE run (``n_init="auto"`` with k-means++ is 1) with ``_kmeans_plusplus`` on ``RandomState(0)``, and runs
``_kmeans_single_lloyd``: E-step (nearest centre, first index on ties), M-step (cluster means), stop when the labels repeat
("strict convergence") or the summed squared centre shift is <= tol = mean(var(x)) * 1e-4, at most 300 iterations; the data
mean is added back.  This restatement follows that loop with plain numpy and is pinned against ``KMeans`` itself
(tests/test_kmeans.py): centres agree to 1e-12 (summation order of the cluster means differs)."""

from __future__ import annotations

import numpy as np


def plusplus_init(xc: np.ndarray, m: int) -> np.ndarray:
    """k-means++ seeding exactly as ``KMeans(random_state=0)`` draws it (public ``sklearn.cluster.kmeans_plusplus``)."""
    from sklearn.cluster import kmeans_plusplus
    from sklearn.utils.extmath import row_norms

    centers, _ = kmeans_plusplus(xc, m, x_squared_norms=row_norms(xc, squared=True), random_state=0)
    return np.ascontiguousarray(centers, dtype=np.float64)


def assign(xc, centers):
    """Nearest centre of every point, first index on ties (difference form of the distance)."""
    d2 = np.zeros((xc.shape[0], centers.shape[0]))
    for k in range(xc.shape[1]):
        diff = xc[:, k][:, None] - centers[:, k][None, :]
        d2 += diff * diff
    return np.argmin(d2, axis=1).astype(np.int32)


def lloyd(xc, centers_init, tol, max_iter=300):
    """``_kmeans_single_lloyd`` (sklearn/cluster/_kmeans.py).  Returns (centers, labels, n_iter, had_empty_cluster)."""
    centers = np.array(centers_init, dtype=np.float64)
    m = centers.shape[0]
    labels_old = np.full(xc.shape[0], -1, dtype=np.int32)
    strict = False
    n_iter = 0
    empty = False
    for i in range(max_iter):
        n_iter = i + 1
        labels = assign(xc, centers)
        new = np.zeros_like(centers)
        counts = np.bincount(labels, minlength=m)
        np.add.at(new, labels, xc)
        if np.any(counts == 0):
            empty = True  # sklearn relocates empty clusters (_relocate_empty_clusters_dense); not restated: callers fall back
            break
        new /= counts[:, None]
        shift_tot = float(((new - centers) ** 2).sum())
        centers = new
        if np.array_equal(labels, labels_old):
            strict = True
            break
        if shift_tot <= tol:
            break
        labels_old = labels
    if not strict and not empty:
        labels = assign(xc, centers)
    return centers, labels, n_iter, empty


def kmeans_centers(x, m):
    """The inducing points of gpr.py:312-315."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    mean = x.mean(axis=0)
    xc = x - mean
    tol = float(np.mean(np.var(x, axis=0)) * 1e-4)
    centers, labels, n_iter, empty = lloyd(xc, plusplus_init(xc, m), tol)
    if empty:
        raise RuntimeError("empty cluster: not restated (sklearn relocates it)")
    return centers + mean, labels, n_iter
