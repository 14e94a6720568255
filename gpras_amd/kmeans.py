"""K-means inducing-point initialisation -- SURVEY.md section 8(f) row N4, ``/root/reference/gpras/gpr.py:312-315``:
``KMeans(n_clusters=M, random_state=0, n_init="auto").fit(x).cluster_centers_``.

What stays on the host is what scikit-learn does once: centring the data, the tolerance, and the k-means++ seeding on
``RandomState(0)`` (``sklearn.cluster.kmeans_plusplus``, the public form of what ``KMeans`` calls).  The Lloyd iterations --
where the time goes: 50-400 ms for N = 4096-16384, a third of a default 16-mode sparse fit -- run in ``libgprx.so``
(``gprx_kmeans_lloyd``) with scikit-learn's stopping rules, and end at its centres (<= 1e-12: the cluster means are summed
in another order).  If a cluster runs empty scikit-learn relocates it to far points; that rare case is handed back to
scikit-learn itself (the reference's own call), never approximated.
"""

from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import as_f64, check, ptr


def kmeans_centers(x, n_clusters: int, device: int = 0, return_info: bool = False):
    from sklearn.cluster import kmeans_plusplus
    from sklearn.utils.extmath import row_norms

    x = as_f64(x)
    n, d = x.shape
    mean = x.mean(axis=0)
    xc = np.ascontiguousarray(x - mean)
    tol = float(np.mean(np.var(x, axis=0)) * 1e-4)  # sklearn.cluster._kmeans._tolerance
    info = {"device": False, "n_iter": None, "labels": None}
    if d <= 64 and 0 < n_clusters <= n:
        init, _ = kmeans_plusplus(xc, n_clusters, x_squared_norms=row_norms(xc, squared=True), random_state=0)
        centers = np.ascontiguousarray(init, dtype=np.float64)
        labels = np.empty(n, dtype=np.int32)
        n_iter, empty = C.c_int(), C.c_int()
        check(_lib.load().gprx_kmeans_lloyd(device, ptr(xc), n, d, ptr(centers), int(n_clusters), tol, 300, ptr(labels), C.byref(n_iter), C.byref(empty)))
        if not empty.value:
            info = {"device": True, "n_iter": n_iter.value, "labels": labels}
            out = np.ascontiguousarray(centers + mean)
            return (out, info) if return_info else out
    # an emptied cluster (or d > 64): scikit-learn's own routine, exactly the reference's call
    from sklearn.cluster import KMeans

    km = KMeans(n_clusters=n_clusters, random_state=0, n_init="auto").fit(x)
    info = {"device": False, "n_iter": km.n_iter_, "labels": km.labels_}
    out = np.ascontiguousarray(km.cluster_centers_.astype(np.float64))
    return (out, info) if return_info else out
