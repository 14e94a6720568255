"""K-means inducing-point initialisation -- SURVEY.md section 8(f) row N4, ``/root/reference/gpras/gpr.py:312-315``:
``KMeans(n_clusters=M, random_state=0, n_init="auto").fit(x).cluster_centers_``.

What stays on the host is what scikit-learn does once: centring the data, the tolerance, the squared row norms and the
``RandomState(0)`` draws of the k-means++ seeding (the first index and ``2 + int(log(M))`` uniforms per centre: they do not
depend on the data).  The seeding itself (``gprx_kmeans_pp``: candidate distances, running minimum, potentials, the
cumulative-sum search; round 2 profiled it at 120 of the 125 ms of an N = 16384 initialisation when it ran on the host through
``sklearn.cluster.kmeans_plusplus``) and the Lloyd iterations (``gprx_kmeans_lloyd``, scikit-learn's stopping rules) run in
``libgprx.so`` and end at scikit-learn's centres (<= 1e-12: the cluster means are summed in another order).  If a cluster runs empty scikit-learn relocates it to far points; that rare case is handed back to
scikit-learn itself (the reference's own call), never approximated.
"""

from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import as_f64, check, ptr


def _draws(n: int, n_clusters: int):
    """The ``RandomState(0)`` draws of ``sklearn.cluster._kmeans._kmeans_plusplus`` as scikit-learn >= 1.3 makes them (they do not
    depend on the data): the first index by ``choice(n, p=uniform weights)``, then ``uniform(size=n_local_trials)`` per centre."""
    rs = np.random.RandomState(0)
    trials = 2 + int(np.log(n_clusters))
    weight = np.ones(n, dtype=np.float64)
    first = int(rs.choice(n, p=weight / weight.sum()))
    uniforms = np.ascontiguousarray(np.stack([rs.uniform(size=trials) for _ in range(n_clusters - 1)]) if n_clusters > 1 else np.zeros((1, trials)))
    return trials, first, uniforms


_DRAWS_MATCH_SKLEARN = None


def draws_match_installed_sklearn() -> bool:
    """Once per process: does the INSTALLED scikit-learn consume its random stream the way ``_draws`` assumes?  A tiny
    problem (48 well-separated points, 6 centres) is seeded by ``sklearn.cluster.kmeans_plusplus(random_state=0)`` and by a plain
    numpy replay of its algorithm on ``_draws``; if the picked rows differ -- an older release drew with ``randint`` /
    ``random_sample``, a future one may change again -- the device seeding is not used and ``kmeans_centers`` asks scikit-learn's own
    ``kmeans_plusplus`` for the seeds, as it did in round 2 (same result as the reference's call either way, only slower)."""
    global _DRAWS_MATCH_SKLEARN
    if _DRAWS_MATCH_SKLEARN is None:
        try:
            from sklearn.cluster import kmeans_plusplus

            g = np.random.default_rng(12345)
            xc = np.ascontiguousarray(g.normal(size=(48, 3)) + 5.0 * g.integers(0, 4, size=(48, 1)))
            xc -= xc.mean(axis=0)
            m = 6
            _, want = kmeans_plusplus(xc, m, random_state=0)
            trials, first, uniforms = _draws(len(xc), m)
            d2 = lambda c: ((xc - c) ** 2).sum(axis=1)  # noqa: E731
            picked, closest = [first], d2(xc[first])
            for c in range(1, m):
                cand = np.searchsorted(np.cumsum(closest), uniforms[c - 1] * closest.sum())
                np.clip(cand, None, len(xc) - 1, out=cand)
                pots = [np.minimum(closest, d2(xc[i])).sum() for i in cand]
                best = int(cand[int(np.argmin(pots))])
                picked.append(best)
                closest = np.minimum(closest, d2(xc[best]))
            _DRAWS_MATCH_SKLEARN = bool(np.array_equal(np.asarray(picked), np.asarray(want)))
        except Exception:  # noqa: BLE001  (no scikit-learn, or an API this code does not know: do not trust the replay)
            _DRAWS_MATCH_SKLEARN = False
    return _DRAWS_MATCH_SKLEARN


def kmeans_pp_indices(xc, n_clusters: int, device: int = 0):
    """Rows of the (centred) data that ``sklearn.cluster.kmeans_plusplus(xc, n_clusters, random_state=0)`` picks, computed by
    ``gprx_kmeans_pp``.  The draws are made here exactly as ``_kmeans_plusplus`` makes them: ``choice(n, p=uniform)`` for the first
    centre, then ``uniform(size=n_local_trials)`` once per further centre."""
    from sklearn.utils.extmath import row_norms

    n, _ = xc.shape
    trials, first, uniforms = _draws(n, n_clusters)
    xsq = np.ascontiguousarray(row_norms(xc, squared=True), dtype=np.float64)
    indices = np.empty(n_clusters, dtype=np.int64)
    check(_lib.load().gprx_kmeans_pp(device, ptr(xc), n, xc.shape[1], ptr(xsq), int(n_clusters), trials, first, ptr(uniforms), ptr(indices)))
    return indices


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
        trials = 2 + int(np.log(n_clusters))  # _kmeans_plusplus: n_local_trials
        if trials <= 16 and draws_match_installed_sklearn():
            indices = kmeans_pp_indices(xc, n_clusters, device)
            init = xc[indices]
        else:  # (more than e^14 clusters, or a scikit-learn whose random draws this module does not know)
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
