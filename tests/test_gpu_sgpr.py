"""GPU parity of the sparse (SGPR) path -- the model the reference actually runs
(gpflow SGPR behind /root/reference/gpras/gpr.py:299) -- against the CPU oracle, through the C ABI.
Also checks the committed golden vectors."""

import ctypes as C
import os

import numpy as np
import pytest

from gpras_amd import _lib
from gpras_amd._lib import check, ptr
from gpras_amd.synth import make_regression
from oracle import gpras_oracle
from oracle import kernels as okn
from oracle import sgpr as osg
from oracle import transforms as otr

pytestmark = pytest.mark.gpu

HYPER = _lib.TRAIN_VARIANCE | _lib.TRAIN_LENGTHSCALE | _lib.TRAIN_NOISE
ALL = HYPER | _lib.TRAIN_Z
GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "gp_golden_n256_d4.npz")


def make_handle(lib, n, d, m, kernel, ard, x, y):
    h = C.c_void_p()
    check(lib.gprx_create(0, n, d, m, okn.KERNEL_IDS[kernel], int(ard), C.byref(h)))
    check(lib.gprx_set_data(h, ptr(x), ptr(y), y.shape[1]), h)
    return h


def eval_gpu(lib, h, unit, theta, z, mask, m, d):
    loss = C.c_double()
    grad = np.zeros(theta.size + m * d)
    check(lib.gprx_objective(h, unit, ptr(theta), ptr(z), mask, C.byref(loss), ptr(grad)), h)
    return loss.value, grad


def ref_eval(kernel, x, y, z, theta, ard, mask_tuple):
    wl = theta[1:-1] if ard else float(theta[1])
    loss, g = osg.loss_and_grad(kernel, x, y, z, float(theta[0]), wl, float(theta[-1]), mask_tuple)
    vec = np.concatenate([[g["variance"]], np.atleast_1d(g["lengthscales"]), [g["noise"]], np.asarray(g["Z"]).ravel()])
    return loss, vec


@pytest.mark.parametrize("kernel", okn.KERNEL_NAMES)
@pytest.mark.parametrize("n,d,m,ard", [(256, 4, 32, False), (700, 5, 100, True), (1000, 8, 200, False)])
def test_sgpr_loss_grad_predict(lib, kernel, n, d, m, ard):
    x, y, xs = make_regression(n, d, n_outputs=2, n_test=300, config=3, unit=n + m)
    z = gpras_oracle.create_inducing(x, m, "kmeans")
    # k-means with many centres leaves single-point clusters, i.e. z_i == x_n up to rounding.  For the
    # kernels that are not differentiable at r = 0 the Z-gradient is discontinuous there, so no two
    # implementations agree on it; keep the centres a hair away from the data for the parity check.
    z = np.ascontiguousarray(z + 1e-3 * np.random.default_rng(m).standard_normal(z.shape))
    h = make_handle(lib, n, d, m, kernel, ard, x, y)
    try:
        ls = np.linspace(0.7, 1.3, d) if ard else 0.85
        variance, noise = 1.2, 0.08
        wv, wl, wn = otr.unconstrain(variance, ls, noise)
        theta = np.ascontiguousarray(np.concatenate([[wv], np.atleast_1d(wl), [wn]]))
        for unit in range(2):
            loss, grad = eval_gpu(lib, h, unit, theta, z, ALL, m, d)
            ref_loss, ref_grad = ref_eval(kernel, x, y[:, unit], z, theta, ard, (True, True, True, True))
            assert abs(loss - ref_loss) <= 1e-9 * abs(ref_loss)
            nt = theta.size
            assert np.max(np.abs(grad[:nt] - ref_grad[:nt])) <= 1e-7 * np.max(np.abs(ref_grad[:nt]))
            assert np.max(np.abs(grad[nt:] - ref_grad[nt:])) <= 1e-7 * np.max(np.abs(ref_grad[nt:]))
            mean = np.zeros(xs.shape[0])
            var = np.zeros(xs.shape[0])
            check(lib.gprx_predict(h, ptr(xs), xs.shape[0], ptr(mean), ptr(var), 1), h)
            ref_mean, ref_var = osg.predict(kernel, x, y[:, unit], z, variance, ls, noise, xs, True)
            assert np.max(np.abs(mean - ref_mean)) <= 1e-8 * np.max(np.abs(ref_mean))
            assert np.max(np.abs(var - ref_var) / ref_var) <= 1e-8
    finally:
        lib.gprx_destroy(h)


def test_sgpr_trainable_masks(lib):
    """Stage 1 of the staged optimisers trains Z only: no priors in the loss, zero hyperparameter gradients."""
    n, d, m = 300, 3, 20
    x, y, _ = make_regression(n, d, config=3, unit=9)
    z = np.ascontiguousarray(gpras_oracle.create_inducing(x, m, "grid"))
    h = make_handle(lib, n, d, m, "Matern32", False, x, y)
    try:
        theta = np.array([0.2, -0.3, 0.1])
        loss_z, grad_z = eval_gpu(lib, h, 0, theta, z, _lib.TRAIN_Z, m, d)
        ref_loss, ref_grad = ref_eval("Matern32", x, y[:, 0], z, theta, False, (False, False, False, True))
        assert abs(loss_z - ref_loss) <= 1e-10 * abs(ref_loss)
        assert np.all(grad_z[:3] == 0.0)
        assert np.max(np.abs(grad_z[3:] - ref_grad[3:])) <= 1e-7 * np.max(np.abs(ref_grad[3:]))
        loss_h, grad_h = eval_gpu(lib, h, 0, theta, z, HYPER, m, d)
        ref_loss, ref_grad = ref_eval("Matern32", x, y[:, 0], z, theta, False, (True, True, True, False))
        assert abs(loss_h - ref_loss) <= 1e-10 * abs(ref_loss)
        assert np.all(grad_h[3:] == 0.0)
        assert np.max(np.abs(grad_h[:3] - ref_grad[:3])) <= 1e-7 * np.max(np.abs(ref_grad[:3]))
    finally:
        lib.gprx_destroy(h)


@pytest.mark.parametrize("kernel", okn.KERNEL_NAMES)
@pytest.mark.parametrize("tag", ["iso", "ard"])
def test_golden_vectors_on_gpu(lib, kernel, tag):
    gold = np.load(GOLDEN)
    n, d = int(gold["n"]), int(gold["d"])
    x, y, xs = make_regression(n, d, 1, int(gold["n_test"]), int(gold["config"]), int(gold["unit"]))
    ard = tag == "ard"
    wl = gold["w_len_ard"] if ard else np.atleast_1d(float(gold["w_len"]))
    theta = np.ascontiguousarray(np.concatenate([[float(gold["w_var"])], wl, [float(gold["w_noise"])]]))
    for mtag, z in (("m32", gold["z_kmeans32"]), ("m256", x)):
        z = np.ascontiguousarray(z)
        m = z.shape[0]
        h = make_handle(lib, n, d, m, kernel, ard, x, y)
        try:
            key = f"sgpr_{kernel}_{tag}_{mtag}"
            loss, grad = eval_gpu(lib, h, 0, theta, z, ALL, m, d)
            assert abs(loss - float(gold[key + "_loss"])) <= 1e-9 * abs(float(gold[key + "_loss"]))
            assert abs(grad[0] - float(gold[key + "_g_var"])) <= 1e-7 * max(1.0, abs(float(gold[key + "_g_var"])))
            mean = np.zeros(xs.shape[0])
            var = np.zeros(xs.shape[0])
            check(lib.gprx_predict(h, ptr(xs), xs.shape[0], ptr(mean), ptr(var), 1), h)
            assert np.max(np.abs(mean - gold[key + "_mean"])) <= 1e-8 * np.max(np.abs(gold[key + "_mean"]))
            assert np.max(np.abs(var - gold[key + "_var"]) / gold[key + "_var"]) <= 1e-8
        finally:
            lib.gprx_destroy(h)
    # exact path against the exact golden vectors
    h = make_handle(lib, n, d, 0, kernel, ard, x, y)
    try:
        key = f"exact_{kernel}_{tag}"
        loss = C.c_double()
        grad = np.zeros(theta.size)
        check(lib.gprx_objective(h, 0, ptr(theta), None, HYPER, C.byref(loss), ptr(grad)), h)
        assert abs(loss.value - float(gold[key + "_loss"])) <= 1e-9 * abs(float(gold[key + "_loss"]))
        assert np.max(np.abs(grad[1:-1] - np.atleast_1d(gold[key + "_g_len"]))) <= 1e-7 * max(1.0, np.max(np.abs(gold[key + "_g_len"])))
        mean = np.zeros(xs.shape[0])
        var = np.zeros(xs.shape[0])
        check(lib.gprx_predict(h, ptr(xs), xs.shape[0], ptr(mean), ptr(var), 1), h)
        assert np.max(np.abs(mean - gold[key + "_mean"])) <= 1e-8 * np.max(np.abs(gold[key + "_mean"]))
        assert np.max(np.abs(var - gold[key + "_var"]) / gold[key + "_var"]) <= 1e-8
    finally:
        lib.gprx_destroy(h)


@pytest.mark.parametrize("kernel,n,d,m,ard,cells", [("RBF", 500, 4, 30, False, 5), ("Matern32", 700, 6, 70, True, 3), ("Exponential", 300, 3, 64, False, 26), ("Matern52", 4200, 5, 50, False, 4)])
def test_sparse_objective_batch_equals_single_calls(lib, kernel, n, d, m, ard, cells):
    """Batched SGPR loss + gradient (theta and Z): every kernel once for all cells -- bit-identical to gprx_objective per
    cell, 1e-9 / 1e-7 against the oracle; also loss only and a partial mask."""
    import ctypes as C

    from gpras_amd import _lib
    from gpras_amd._lib import check, ptr
    from oracle import kernels as okn
    from oracle import sgpr as osg

    x, y, _ = make_regression(n, d, n_outputs=3, n_test=0, config=14, unit=n + m)
    h = C.c_void_p()
    check(lib.gprx_create(0, n, d, m, okn.KERNEL_IDS[kernel], int(ard), C.byref(h)))
    check(lib.gprx_set_data(h, ptr(x), ptr(y), 3), h)
    nt = 2 + (d if ard else 1)
    try:
        rng = np.random.default_rng(9)
        units = np.ascontiguousarray(rng.integers(0, 3, size=cells), dtype=np.int32)
        thetas = np.ascontiguousarray(rng.normal(0.2, 0.3, size=(cells, nt)))
        zs = np.ascontiguousarray(np.stack([x[rng.choice(n, size=m, replace=False)] + 1e-3 * rng.standard_normal((m, d)) for _ in range(cells)]))
        losses, grads = np.zeros(cells), np.zeros((cells, nt + m * d))
        check(lib.gprx_objective_batch(h, cells, ptr(units), ptr(thetas), ptr(zs), 15, ptr(losses), ptr(grads)), h)
        for c in range(cells):
            th, zc = np.ascontiguousarray(thetas[c]), np.ascontiguousarray(zs[c])
            single, g1 = C.c_double(), np.zeros(nt + m * d)
            check(lib.gprx_objective(h, int(units[c]), ptr(th), ptr(zc), 15, C.byref(single), ptr(g1)), h)
            assert single.value == losses[c]
            assert np.array_equal(g1, grads[c])
            if c < 3:
                wl = th[1:-1] if ard else float(th[1])
                ref_loss, g = osg.loss_and_grad(kernel, x, y[:, units[c]], zc, float(th[0]), wl, float(th[-1]))
                ref = np.concatenate([[g["variance"]], np.atleast_1d(g["lengthscales"]), [g["noise"]], np.asarray(g["Z"]).ravel()])
                assert abs(losses[c] - ref_loss) <= 1e-9 * abs(ref_loss)
                assert np.max(np.abs(grads[c] - ref)) <= 1e-7 * max(1.0, np.max(np.abs(ref)))
        l2 = np.zeros(cells)
        check(lib.gprx_objective_batch(h, cells, ptr(units), ptr(thetas), ptr(zs), _lib.TRAIN_Z, ptr(l2), None), h)
        for c in range(cells):
            single = C.c_double()
            check(lib.gprx_factorize(h, int(units[c]), ptr(np.ascontiguousarray(thetas[c])), ptr(np.ascontiguousarray(zs[c])), _lib.TRAIN_Z, C.byref(single)), h)
            assert single.value == l2[c]
    finally:
        lib.gprx_destroy(h)


@pytest.mark.parametrize("kernel,n,d,m,ns", [("RBF", 900, 6, 50, 5000), ("Matern32", 500, 3, 130, 700)])
def test_batched_sparse_predict_equals_single_calls_and_oracle(lib, kernel, n, d, m, ns):
    """``gprx_predict_batch`` on sparse models (what gpras runs, gpr.py:336-339): all cells factorised and predicted by batched
    launches -- bit-identical to factorise + predict per cell, and within 1e-8 of the oracle.  ns > 4096 crosses a predict tile."""
    cells = 5
    x, y, xs = make_regression(n, d, n_outputs=cells, n_test=ns, config=13, unit=m)
    rng = np.random.default_rng(m)
    h = make_handle(lib, n, d, m, kernel, False, x, y)
    try:
        zs = np.ascontiguousarray(np.stack([gpras_oracle.create_inducing(x, m, "kmeans") + 0.01 * rng.standard_normal((m, d)) for _ in range(cells)]))
        par = [(1.0 + 0.1 * c, 0.8 + 0.05 * c, 0.05 + 0.02 * c) for c in range(cells)]
        thetas = np.ascontiguousarray([np.concatenate([np.atleast_1d(w) for w in otr.unconstrain(*p)]) for p in par])
        units = np.arange(cells, dtype=np.int32)[::-1].copy()  # cell i is unit cells - 1 - i: the unit table is honoured
        means, variances = np.zeros((cells, ns)), np.zeros((cells, ns))
        check(lib.gprx_predict_batch(h, cells, ptr(units), ptr(thetas), ptr(zs), ptr(xs), ns, ptr(means), ptr(variances), 1), h)
        for c in range(cells):
            loss = C.c_double()
            check(lib.gprx_factorize(h, int(units[c]), ptr(thetas[c]), ptr(zs[c]), 0, C.byref(loss)), h)
            mean, var = np.zeros(ns), np.zeros(ns)
            check(lib.gprx_predict(h, ptr(xs), ns, ptr(mean), ptr(var), 1), h)
            assert np.array_equal(mean, means[c]) and np.array_equal(var, variances[c]), c
            v, l, s = par[c]
            rm, rv = osg.predict(kernel, x, y[:, units[c]], zs[c], v, l, s, xs)
            assert np.max(np.abs(means[c] - rm)) <= 1e-8 * np.max(np.abs(rm)) and np.max(np.abs(variances[c] - rv) / rv) <= 1e-8
        # latent variance (include_noise = 0) differs from the observation variance by the cell's noise
        lat_m, lat_v = np.zeros((cells, ns)), np.zeros((cells, ns))
        check(lib.gprx_predict_batch(h, cells, ptr(units), ptr(thetas), ptr(zs), ptr(xs), ns, ptr(lat_m), ptr(lat_v), 0), h)
        assert np.array_equal(lat_m, means)
        for c in range(cells):
            np.testing.assert_allclose(variances[c] - lat_v[c], par[c][2], rtol=1e-9)
    finally:
        lib.gprx_destroy(h)


def test_sparse_batch_failed_cell_is_isolated_and_leaves_no_trace(lib):
    """One cell of a batched sparse evaluation whose Kuu is numerically singular (variance 1e12, lengthscale 1e6: every entry of
    Kuu rounds to v, the 1e-6 jitter drowns): that cell reports NaN / GPRX_ENOTPD, the others equal their single calls bit for
    bit, and the NEXT evaluation on the same handle (captured graph, reused cell blocks) is clean -- NaN left in a cell block
    must not leak into the following call."""
    import ctypes as C

    from gpras_amd import _lib
    from gpras_amd._lib import check, ptr
    from oracle import kernels as okn

    n, d, m, cells = 600, 4, 40, 3
    x, y, _ = make_regression(n, d, n_outputs=3, n_test=0, config=14, unit=77)
    h = C.c_void_p()
    check(lib.gprx_create(0, n, d, m, okn.KERNEL_IDS["RBF"], 0, C.byref(h)))
    check(lib.gprx_set_data(h, ptr(x), ptr(y), 3), h)
    try:
        rng = np.random.default_rng(4)
        units = np.arange(cells, dtype=np.int32)
        good = np.ascontiguousarray(rng.normal(0.2, 0.3, size=(cells, 3)))
        zs = np.ascontiguousarray(np.stack([x[rng.choice(n, size=m, replace=False)] for _ in range(cells)]))
        singles = []
        for c in range(cells):
            s, g = C.c_double(), np.zeros(3 + m * d)
            check(lib.gprx_objective(h, c, ptr(np.ascontiguousarray(good[c])), ptr(np.ascontiguousarray(zs[c])), 15, C.byref(s), ptr(g)), h)
            singles.append((s.value, g))
        bad = good.copy()
        bad[1] = [1e12, 1e6, 0.0]
        for rep in range(2):  # twice: the second failing call replays the captured graph
            losses, grads = np.zeros(cells), np.zeros((cells, 3 + m * d))
            rc = lib.gprx_objective_batch(h, cells, ptr(units), ptr(bad), ptr(zs), 15, ptr(losses), ptr(grads))
            assert rc == _lib.GPRX_ENOTPD
            assert np.isnan(losses[1])
            for c in (0, 2):
                assert losses[c] == singles[c][0] and np.array_equal(grads[c], singles[c][1])
            losses, grads = np.zeros(cells), np.zeros((cells, 3 + m * d))
            check(lib.gprx_objective_batch(h, cells, ptr(units), ptr(good), ptr(zs), 15, ptr(losses), ptr(grads)), h)
            for c in range(cells):
                assert losses[c] == singles[c][0] and np.array_equal(grads[c], singles[c][1])
    finally:
        lib.gprx_destroy(h)


def test_resident_adam_reports_a_cell_that_stops_being_positive_definite(lib):
    """gprx_adam_batch on sparse models with M <= 64 runs resident on the device (gprx.hip sgpr_adam_resident); a cell whose Kuu is
    numerically singular from the first step on ends the call with GPRX_ENOTPD at the first read of the stop flags, names the cell,
    leaves the other cells' variables finite, and the handle serves a clean run afterwards (same result as a run that never failed)."""
    import ctypes as C

    from gpras_amd import _lib
    from gpras_amd._lib import check, ptr
    from oracle import kernels as okn

    n, d, m, cells = 500, 3, 24, 3
    x, y, _ = make_regression(n, d, n_outputs=3, n_test=0, config=14, unit=5)
    h = C.c_void_p()
    check(lib.gprx_create(0, n, d, m, okn.KERNEL_IDS["RBF"], 0, C.byref(h)))
    check(lib.gprx_set_data(h, ptr(x), ptr(y), 3), h)
    try:
        rng = np.random.default_rng(9)
        units = np.arange(cells, dtype=np.int32)
        good = np.ascontiguousarray(rng.normal(0.2, 0.3, size=(cells, 3)))
        zs0 = np.ascontiguousarray(np.stack([x[rng.choice(n, size=m, replace=False)] for _ in range(cells)]))

        def run(thetas):
            th, zs = thetas.copy(), zs0.copy()
            n_evals, batches = np.zeros(cells, dtype=np.int32), C.c_int()
            rc = lib.gprx_adam_batch(h, cells, ptr(units), ptr(th), ptr(zs), 15, 30, ptr(n_evals), C.byref(batches))
            return rc, th, zs, n_evals

        rc, th_ref, zs_ref, ev_ref = run(good)
        assert rc == _lib.GPRX_OK and (ev_ref == 30).all()
        bad = good.copy()
        bad[1] = [1e12, 1e6, 0.0]
        rc, th, zs, ev = run(bad)
        assert rc == _lib.GPRX_ENOTPD
        assert b"cell 1" in lib.gprx_last_error(h)
        assert np.isfinite(th[[0, 2]]).all() and np.isfinite(zs[[0, 2]]).all()
        rc, th2, zs2, ev2 = run(good)
        assert rc == _lib.GPRX_OK and np.array_equal(th2, th_ref) and np.array_equal(zs2, zs_ref) and np.array_equal(ev2, ev_ref)
    finally:
        lib.gprx_destroy(h)


def test_resident_adam_in_two_groups_of_cells_equals_one_group(lib):
    """When a pass over the batch needs more than one round of the chip's CUs (17 cells at N = 4096) the resident Adam loop runs the batch
    as two groups of cells on two streams, one launch apart (gprx.hip sf_group_count: one group's one-workgroup-per-cell launches beside
    the other's streamed passes).  Forced here on a small problem ("sgpr_groups_from" = 1: two groups whatever the size): same variables
    and evaluation counts bit for bit as with the grouping switched off (0); a cell of the SECOND group that is not positive definite is
    named by its index in the batch; a run of zero steps leaves nothing running."""
    import ctypes as C

    from gpras_amd import _lib
    from gpras_amd._lib import check, ptr
    from oracle import kernels as okn

    n, d, m, cells, outs = 400, 3, 20, 19, 4
    x, y, _ = make_regression(n, d, n_outputs=outs, n_test=0, config=14, unit=19)
    h = C.c_void_p()
    check(lib.gprx_create(0, n, d, m, okn.KERNEL_IDS["Matern32"], 0, C.byref(h)))
    check(lib.gprx_set_data(h, ptr(x), ptr(y), outs), h)
    try:
        rng = np.random.default_rng(19)
        units = np.ascontiguousarray(rng.integers(0, outs, size=cells), dtype=np.int32)
        th0 = np.ascontiguousarray(rng.normal(0.2, 0.3, size=(cells, 3)))
        zs0 = np.ascontiguousarray(np.stack([x[rng.choice(n, size=m, replace=False)] for _ in range(cells)]))

        def run(thetas, groups_from, steps=40):
            check(lib.gprx_set_tuning(b"sgpr_groups_from", groups_from))
            th, zs = thetas.copy(), zs0.copy()
            n_evals, batches = np.zeros(cells, dtype=np.int32), C.c_int()
            rc = lib.gprx_adam_batch(h, cells, ptr(units), ptr(th), ptr(zs), 15, steps, ptr(n_evals), C.byref(batches))
            return rc, th, zs, n_evals

        rc0, th_0, zs_0, ev0 = run(th0, 1, steps=0)  # (no step: the variables come back as they went in, nothing is left running)
        assert rc0 == _lib.GPRX_OK and np.array_equal(th_0, th0) and np.array_equal(zs_0, zs0) and (ev0 == 0).all()
        rc1, th1, zs1, ev1 = run(th0, 0)
        rc2, th2, zs2, ev2 = run(th0, 1)
        assert rc1 == rc2 == _lib.GPRX_OK
        assert np.array_equal(th1, th2) and np.array_equal(zs1, zs2) and np.array_equal(ev1, ev2) and (ev1 == 40).all()
        assert not np.array_equal(th1, th0)
        bad = th0.copy()
        bad[15] = [1e12, 1e6, 0.0]  # (second group: cells 10 .. 18)
        rc, th, zs, ev = run(bad, 1)
        assert rc == _lib.GPRX_ENOTPD and b"cell 15" in lib.gprx_last_error(h)
        rc3, th3, zs3, ev3 = run(th0, 1)
        assert rc3 == _lib.GPRX_OK and np.array_equal(th3, th1) and np.array_equal(zs3, zs1)
    finally:
        lib.gprx_set_tuning(b"sgpr_groups_from", 17)
        lib.gprx_destroy(h)


def test_batched_sparse_evaluations_from_two_threads_on_two_handles(lib):
    """Two host threads, each with its own handle (own stream, own captured graph), evaluate batches at the same time: the
    captures must not disturb each other (they did -- "operation failed due to a previous error during capture" -- until the
    library serialised them) and every result equals the single-thread one bit for bit."""
    import threading

    from gpras_amd.engine import Engine

    n, d, m, cells = 700, 5, 40, 6
    x, y, _ = make_regression(n, d, n_outputs=cells, n_test=0, config=14, unit=31)
    rng = np.random.default_rng(12)
    thetas = np.ascontiguousarray(rng.normal(0.2, 0.3, size=(cells, 3)))
    zs = np.ascontiguousarray(np.stack([x[rng.choice(n, size=m, replace=False)] for _ in range(cells)]))
    units = np.arange(cells, dtype=np.int32)
    ref_eng = Engine("Matern32", x, y, m)
    want = ref_eng.objective_batch(units, thetas, 15, zs=zs)
    ref_eng.close()
    engines = [Engine("Matern32", x, y, m) for _ in range(2)]
    results, errors = [[], []], []

    def run(k):
        try:
            for _ in range(6):  # first call eager, second captures, the rest replay
                results[k].append(engines[k].objective_batch(units, thetas, 15, zs=zs))
        except Exception as exc:  # noqa: BLE001
            errors.append(exc)

    threads = [threading.Thread(target=run, args=(k,)) for k in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for e in engines:
        e.close()
    assert not errors, errors
    for k in range(2):
        assert len(results[k]) == 6
        for losses, grads, ok in results[k]:
            assert ok.all() and np.array_equal(losses, want[0]) and np.array_equal(grads, want[1])


@pytest.mark.parametrize("kernel,n,d,m,ard,form", [("RBF", 1100, 10, 50, False, 0), ("Matern52", 700, 20, 64, True, 0), ("Matern12", 500, 3, 33, False, 1),
                                                  ("Exponential", 300, 40, 17, True, 1), ("Matern32", 257, 1, 1, False, 0)])
def test_fused_evaluation_agrees_with_the_launch_sequence_and_does_not_depend_on_the_batch(lib, kernel, n, d, m, ard, form):
    """M <= 64 takes the five-launch evaluation of sgpr_fused.h (tuning key "sgpr_fused", default 1).  Its values agree with the
    round-4 launch sequence ("sgpr_fused" = 0) to rounding -- and differ from it in the last bits, which shows that the fused
    kernels ran --, with the oracle to 1e-9 / 1e-7, and a cell evaluated alone equals the same cell inside a batch bit for bit
    (fixed chunks of 256 columns, chunk order: sgpr_fused.h "Determinism").  d > 16 takes the restaging variants, d = 40 with ARD the
    four-chunk accumulators; form 1 is gpflow's expanded distance."""
    cells = 7
    x, y, _ = make_regression(n, d, n_outputs=3, n_test=0, config=14, unit=n + m + d)
    h = make_handle(lib, n, d, m, kernel, ard, x, y)
    nt = 2 + (d if ard else 1)
    try:
        check(lib.gprx_set_distance_form(h, form), h)
        rng = np.random.default_rng(n)
        units = np.ascontiguousarray(rng.integers(0, 3, size=cells), dtype=np.int32)
        thetas = np.ascontiguousarray(rng.normal(0.2, 0.3, size=(cells, nt)))
        zs = np.ascontiguousarray(np.stack([x[rng.choice(n, size=m, replace=False)] + 1e-3 * rng.standard_normal((m, d)) for _ in range(cells)]))
        out = {}
        for fused in (1, 0):
            check(lib.gprx_set_handle_tuning(h, b"sgpr_fused", fused), h)
            losses, grads = np.zeros(cells), np.zeros((cells, nt + m * d))
            check(lib.gprx_objective_batch(h, cells, ptr(units), ptr(thetas), ptr(zs), 15, ptr(losses), ptr(grads)), h)
            out[fused] = (losses, grads)
        assert np.max(np.abs(out[1][0] - out[0][0]) / np.abs(out[0][0])) <= 1e-11
        scale = np.max(np.abs(out[0][1]), axis=1, keepdims=True)
        assert np.max(np.abs(out[1][1] - out[0][1]) / scale) <= 1e-8
        assert not np.array_equal(out[1][1], out[0][1])  # (another summation order: the last bits differ somewhere)
        check(lib.gprx_set_handle_tuning(h, b"sgpr_fused", 1), h)
        for c in (0, cells - 1):
            th, zc = np.ascontiguousarray(thetas[c]), np.ascontiguousarray(zs[c])
            l1, g1 = np.zeros(1), np.zeros((1, nt + m * d))
            u1 = np.ascontiguousarray(units[c:c + 1])
            check(lib.gprx_objective_batch(h, 1, ptr(u1), ptr(th), ptr(zc), 15, ptr(l1), ptr(g1)), h)
            assert l1[0] == out[1][0][c] and np.array_equal(g1[0], out[1][1][c])
            wl = th[1:-1] if ard else float(th[1])
            ref_loss, g = osg.loss_and_grad(kernel, x, y[:, units[c]], zc, float(th[0]), wl, float(th[-1]), form="expanded" if form else "direct")
            ref = np.concatenate([[g["variance"]], np.atleast_1d(g["lengthscales"]), [g["noise"]], np.asarray(g["Z"]).ravel()])
            assert abs(l1[0] - ref_loss) <= 1e-9 * abs(ref_loss)
            assert np.max(np.abs(g1[0] - ref)) <= 1e-7 * max(1.0, np.max(np.abs(ref)))
    finally:
        lib.gprx_destroy(h)


@pytest.mark.parametrize("kernel,ard", [("RBF", False), ("Matern52", True)])
def test_sgpr_at_the_top_of_the_references_sweep_m300(lib, kernel, ard):
    """M = 300 inducing points -- the upper end of the reference's cross-validation sweep (production/analysis/cross_validation.py:108,
    data_models.py:114-119) -- takes the general launch sequence (M > 64): loss 1e-9, gradient 1e-7 against the oracle, batch = single."""
    n, d, m, cells = 1500, 10, 300, 3
    x, y, _ = make_regression(n, d, n_outputs=cells, n_test=0, config=3, unit=300)
    rng = np.random.default_rng(300)
    h = make_handle(lib, n, d, m, kernel, ard, x, y)
    nt = 2 + (d if ard else 1)
    try:
        zs = np.ascontiguousarray(np.stack([x[rng.choice(n, size=m, replace=False)] + 1e-2 * rng.standard_normal((m, d)) for _ in range(cells)]))
        thetas = np.ascontiguousarray(rng.normal(0.1, 0.2, size=(cells, nt)))
        units = np.arange(cells, dtype=np.int32)
        losses, grads = np.zeros(cells), np.zeros((cells, nt + m * d))
        check(lib.gprx_objective_batch(h, cells, ptr(units), ptr(thetas), ptr(zs), 15, ptr(losses), ptr(grads)), h)
        for c in range(cells):
            th, zc = np.ascontiguousarray(thetas[c]), np.ascontiguousarray(zs[c])
            wl = th[1:-1] if ard else float(th[1])
            ref_loss, g = osg.loss_and_grad(kernel, x, y[:, c], zc, float(th[0]), wl, float(th[-1]))
            ref = np.concatenate([[g["variance"]], np.atleast_1d(g["lengthscales"]), [g["noise"]], np.asarray(g["Z"]).ravel()])
            assert abs(losses[c] - ref_loss) <= 1e-9 * abs(ref_loss)
            assert np.max(np.abs(grads[c][:nt] - ref[:nt])) <= 1e-7 * max(1.0, np.max(np.abs(ref[:nt])))
            assert np.max(np.abs(grads[c][nt:] - ref[nt:])) <= 1e-7 * max(1.0, np.max(np.abs(ref[nt:])))
        single, g1 = C.c_double(), np.zeros(nt + m * d)
        check(lib.gprx_objective(h, 1, ptr(np.ascontiguousarray(thetas[1])), ptr(np.ascontiguousarray(zs[1])), 15, C.byref(single), ptr(g1)), h)
        assert single.value == losses[1] and np.array_equal(g1, grads[1])
    finally:
        lib.gprx_destroy(h)
