"""GPU parity of the exact-GP path (kernel build + blocked MFMA Cholesky + solves + gradient +
predict) against the CPU oracle, through the C ABI.  Tolerance: 1e-8 relative as BASELINE.json's
north_star states for predictions; losses and gradients are held to 1e-9 / 1e-7."""

import ctypes as C

import numpy as np
import pytest

from gpras_amd import _lib
from gpras_amd._lib import check, ptr
from gpras_amd.synth import make_regression
from oracle import exact as oex
from oracle import kernels as okn
from oracle import transforms as otr

pytestmark = pytest.mark.gpu

ALL = _lib.TRAIN_VARIANCE | _lib.TRAIN_LENGTHSCALE | _lib.TRAIN_NOISE


def make_handle(lib, n, d, kernel, ard, x, y):
    h = C.c_void_p()
    check(lib.gprx_create(0, n, d, 0, okn.KERNEL_IDS[kernel], int(ard), C.byref(h)))
    check(lib.gprx_set_data(h, ptr(x), ptr(y), y.shape[1]), h)
    return h


def pack_theta(variance, ls, noise):
    wv, wl, wn = otr.unconstrain(variance, ls, noise)
    return np.ascontiguousarray(np.concatenate([[wv], np.atleast_1d(wl), [wn]]))


@pytest.mark.parametrize("kernel", okn.KERNEL_NAMES)
@pytest.mark.parametrize("n,d,ard", [(256, 4, False), (300, 5, True), (1000, 8, False)])
def test_exact_loss_grad_predict(lib, kernel, n, d, ard):
    x, y, xs = make_regression(n, d, n_outputs=2, n_test=333, config=1, unit=n)
    h = make_handle(lib, n, d, kernel, ard, x, y)
    try:
        ls = np.linspace(0.7, 1.4, d) if ard else 0.9
        variance, noise = 1.3, 0.07
        theta = pack_theta(variance, ls, noise)
        for unit in range(2):
            loss = C.c_double()
            grad = np.zeros(theta.size)
            check(lib.gprx_objective(h, unit, ptr(theta), None, ALL, C.byref(loss), ptr(grad)), h)
            wl = theta[1:-1] if ard else float(theta[1])
            ref_loss, ref_g = oex.loss_and_grad(kernel, x, y[:, unit], float(theta[0]), wl, float(theta[-1]))
            ref_grad = np.concatenate([[ref_g["variance"]], np.atleast_1d(ref_g["lengthscales"]), [ref_g["noise"]]])
            assert abs(loss.value - ref_loss) <= 1e-9 * abs(ref_loss)
            assert np.max(np.abs(grad - ref_grad)) <= 1e-7 * np.max(np.abs(ref_grad))
            mean = np.zeros(xs.shape[0])
            var = np.zeros(xs.shape[0])
            check(lib.gprx_predict(h, ptr(xs), xs.shape[0], ptr(mean), ptr(var), 1), h)
            ref_mean, ref_var = oex.predict(kernel, x, y[:, unit], variance, ls, noise, xs, True)
            assert np.max(np.abs(mean - ref_mean)) <= 1e-8 * np.max(np.abs(ref_mean))
            assert np.max(np.abs(var - ref_var) / ref_var) <= 1e-8
            assert np.all(var >= noise * (1 - 1e-12))  # predictive variance never below the noise floor
    finally:
        lib.gprx_destroy(h)


def test_exact_mask_and_errors(lib):
    n, d = 128, 3
    x, y, _ = make_regression(n, d, config=1, unit=7)
    h = make_handle(lib, n, d, "RBF", False, x, y)
    try:
        theta = pack_theta(1.0, 0.8, 0.5)
        loss = C.c_double()
        grad = np.ones(3)
        mask = _lib.TRAIN_NOISE
        check(lib.gprx_objective(h, 0, ptr(theta), None, mask, C.byref(loss), ptr(grad)), h)
        ref_loss, ref_g = oex.loss_and_grad("RBF", x, y[:, 0], theta[0], float(theta[1]), theta[2], mask=(False, False, True))
        assert abs(loss.value - ref_loss) <= 1e-10 * abs(ref_loss)
        assert grad[0] == 0.0 and grad[1] == 0.0 and abs(grad[2] - ref_g["noise"]) <= 1e-8 * abs(ref_g["noise"])
        # errors: unit out of range, non-finite theta, predict before factorize on a fresh handle
        assert lib.gprx_objective(h, 5, ptr(theta), None, mask, C.byref(loss), None) == _lib.GPRX_EINVAL
        bad = theta.copy()
        bad[1] = np.nan
        assert lib.gprx_objective(h, 0, ptr(bad), None, mask, C.byref(loss), None) == _lib.GPRX_EINVAL
    finally:
        lib.gprx_destroy(h)
    h2 = C.c_void_p()
    check(lib.gprx_create(0, n, d, 0, 0, 0, C.byref(h2)))
    out = np.zeros(4)
    assert lib.gprx_predict(h2, ptr(x), 4, ptr(out), ptr(out), 1) == _lib.GPRX_ESTATE
    lib.gprx_destroy(h2)
    assert lib.gprx_create(0, n, d, 0, 9, 0, C.byref(h2)) == _lib.GPRX_EINVAL


def test_factorize_many_matches_single_calls_and_replays(lib):
    """Many cells per call: the first call runs eagerly, later calls replay a captured hipGraph whose only
    inputs are the pinned parameter block -- results must track changing hyperparameters exactly and equal
    gprx_factorize on the same handle; predict afterwards uses the replayed factorisation."""
    n, d, cells = 700, 5, 3
    handles = (C.c_void_p * cells)()
    data = []
    for c in range(cells):
        x, y, xs = make_regression(n, d, n_outputs=2, n_test=40, config=7, unit=c)
        handles[c] = make_handle(lib, n, d, "Matern32", True, x, y)
        data.append((x, y, xs))
    try:
        units = np.array([1, 0, 1], dtype=np.int32)
        rng = np.random.default_rng(0)
        for rep in range(4):  # rep 0 eager, rep 1 captures, rep 2-3 replay with new parameters
            thetas = np.ascontiguousarray(rng.normal(0.2, 0.3, size=(cells, d + 2)))
            losses = np.zeros(cells)
            check(lib.gprx_factorize_many(cells, handles, ptr(units), ptr(thetas), ALL, ptr(losses)))
            for c in range(cells):
                x, y, xs = data[c]
                th = thetas[c]
                ref = oex.loss("Matern32", x, y[:, units[c]], float(th[0]), th[1:-1], float(th[-1]))
                assert abs(losses[c] - ref) <= 1e-9 * abs(ref)
                single = C.c_double()
                mean, var = np.zeros(40), np.zeros(40)
                check(lib.gprx_predict(C.c_void_p(handles[c]), ptr(xs), 40, ptr(mean), ptr(var), 1))
                v, l, s = otr.constrain(th[0], th[1:-1], th[-1])
                rm, rv = oex.predict("Matern32", x, y[:, units[c]], float(v), l, float(s), xs)
                assert np.max(np.abs(mean - rm)) <= 1e-8 * np.max(np.abs(rm)) and np.max(np.abs(var - rv) / rv) <= 1e-8
                check(lib.gprx_factorize(C.c_void_p(handles[c]), int(units[c]), ptr(np.ascontiguousarray(th)), None, ALL, C.byref(single)))
                assert single.value == losses[c]
    finally:
        for c in range(cells):
            lib.gprx_destroy(C.c_void_p(handles[c]))


def test_factorize_many_reports_a_non_positive_definite_cell(lib):
    """one of three cells IS not positive definite in fp64 (rows duplicated, v = 2^40 beside s = 1e-6: the second pivot is
    exactly zero -- scipy's Cholesky of the oracle's K raises, checked first); gprx_factorize_many must say GPRX_ENOTPD on the eager
    pass and on the graph replay, give that cell a NaN loss and finish the others."""
    n, d, cells = 300, 3, 3
    x, y, _ = make_regression(n, d, n_outputs=1, n_test=0, config=7, unit=9)
    xdup = x.copy()
    xdup[1::2] = xdup[0::2]
    good = np.array([0.3, 0.1, -1.0])
    bad = np.array([2.0**40, 0.1, -800.0])
    with pytest.raises(np.linalg.LinAlgError, match="2-th leading minor"):
        oex.loss("Matern32", xdup, y[:, 0], bad[0], bad[1], bad[2])
    handles = (C.c_void_p * cells)()
    for c in range(cells):
        handles[c] = make_handle(lib, n, d, "Matern32", False, xdup if c == 1 else x, y)
    try:
        units = np.zeros(cells, dtype=np.int32)
        ref = oex.loss("Matern32", x, y[:, 0], good[0], good[1], good[2])
        for rep in range(3):  # eager, capture, replay
            thetas = np.ascontiguousarray(np.stack([good, bad, good]))
            losses = np.zeros(cells)
            assert lib.gprx_factorize_many(cells, handles, ptr(units), ptr(thetas), ALL, ptr(losses)) == _lib.GPRX_ENOTPD
            assert np.isnan(losses[1]) and abs(losses[0] - ref) <= 1e-9 * abs(ref) and losses[2] == losses[0]
        # the same handle factors a proper matrix afterwards
        thetas = np.ascontiguousarray(np.stack([good, good, good]))
        check(lib.gprx_factorize_many(cells, handles, ptr(units), ptr(thetas), ALL, ptr(losses)))
        assert np.all(np.isfinite(losses))
    finally:
        for c in range(cells):
            lib.gprx_destroy(C.c_void_p(handles[c]))


@pytest.mark.parametrize("kernel,n,d,ard,cells", [("RBF", 700, 5, False, 5), ("Matern52", 1100, 8, True, 3), ("Matern12", 130, 2, False, 17), ("Matern32", 330, 3, False, 26), ("RBF", 1000, 4, True, 25)])
def test_factorize_batch_bit_identical_and_selectable(lib, kernel, n, d, ard, cells):
    """gprx_factorize_batch: every kernel of the schedule launched once for all cells.  Losses must equal the
    single-cell call bit for bit (same kernels, same operation order per element), match the oracle to 1e-9, and
    each slot must be usable for predict after gprx_select_slot -- also after the arena has grown."""
    x, y, xs = make_regression(n, d, n_outputs=3, n_test=50, config=11, unit=n)
    h = make_handle(lib, n, d, kernel, ard, x, y)
    nt = 2 + (d if ard else 1)
    try:
        rng = np.random.default_rng(3)
        for count in (2, cells):  # second round grows the arena
            units = np.ascontiguousarray(rng.integers(0, 3, size=count), dtype=np.int32)
            thetas = np.ascontiguousarray(rng.normal(0.3, 0.3, size=(count, nt)))
            losses = np.zeros(count)
            status = np.zeros(count, dtype=np.int32)
            check(lib.gprx_factorize_batch(h, count, ptr(units), ptr(thetas), ALL, ptr(losses), ptr(status)), h)
            assert not status.any()
            # slots first (a single-cell call below would not disturb them, but keep the order strict)
            preds = []
            for c in range(count):
                check(lib.gprx_select_slot(h, c), h)
                mean, var = np.zeros(50), np.zeros(50)
                check(lib.gprx_predict(h, ptr(xs), 50, ptr(mean), ptr(var), 1), h)
                preds.append((mean, var))
            for c in range(count):
                th = thetas[c]
                wl = th[1:-1] if ard else float(th[1])
                ref = oex.loss(kernel, x, y[:, units[c]], float(th[0]), wl, float(th[-1]))
                assert abs(losses[c] - ref) <= 1e-9 * abs(ref)
                v, l, s = otr.constrain(th[0], wl, th[-1])
                rm, rv = oex.predict(kernel, x, y[:, units[c]], float(v), l, float(s), xs)
                mean, var = preds[c]
                assert np.max(np.abs(mean - rm)) <= 1e-8 * np.max(np.abs(rm)) and np.max(np.abs(var - rv) / rv) <= 1e-8
            for c in range(count):
                single = C.c_double()
                check(lib.gprx_factorize(h, int(units[c]), ptr(np.ascontiguousarray(thetas[c])), None, ALL, C.byref(single)), h)
                # (from 24 cells per launch on -- the split panel -- the right-hand side travels as a vector, round 4: the factor and log det
                # are the single call's bits, y^T K^-1 y is summed in another order)
                assert single.value == losses[c] or (count >= 24 and abs(single.value - losses[c]) <= 1e-14 * abs(single.value))
        assert lib.gprx_select_slot(h, 99) == _lib.GPRX_EINVAL
    finally:
        lib.gprx_destroy(h)


def test_factorize_batch_reports_failed_cells(lib):
    """A cell whose kernel matrix is numerically singular (duplicated rows, noise at its floor) fails alone."""
    n, d = 256, 3
    x, y, _ = make_regression(n, d, config=1, unit=1)
    x[1::2] = x[0::2]  # duplicated inputs
    h = make_handle(lib, n, d, "RBF", False, x, y)
    try:
        good = pack_theta(1.0, 0.9, 0.1)
        bad = good.copy()
        bad[0] = 40.0   # variance ~ 40
        bad[-1] = -800.0  # noise -> 1e-6: K has exactly repeated rows up to 1e-6 on the diagonal, variance 40 -> pivot <= 0 in fp64
        bad[1] = 50.0
        thetas = np.ascontiguousarray(np.stack([good, bad, good]))
        units = np.zeros(3, dtype=np.int32)
        losses = np.zeros(3)
        status = np.zeros(3, dtype=np.int32)
        rc = lib.gprx_factorize_batch(h, 3, ptr(units), ptr(thetas), ALL, ptr(losses), ptr(status))
        ref = oex.loss("RBF", x, y[:, 0], float(good[0]), float(good[1]), float(good[2]))
        assert abs(losses[0] - ref) <= 1e-9 * abs(ref) and losses[2] == losses[0]
        if rc == _lib.GPRX_ENOTPD:
            assert status[1] == _lib.GPRX_ENOTPD and status[0] == 0 and status[2] == 0 and np.isnan(losses[1])
            assert lib.gprx_select_slot(h, 1) == _lib.GPRX_ESTATE
        else:
            assert rc == _lib.GPRX_OK
    finally:
        lib.gprx_destroy(h)


@pytest.mark.parametrize("kernel,n,d,ard,cells", [("RBF", 520, 4, False, 6), ("Matern52", 700, 6, True, 4), ("Exponential", 200, 3, True, 9)])
def test_objective_batch_equals_single_calls(lib, kernel, n, d, ard, cells):
    """Batched loss + gradient (every stage once for all cells) against gprx_objective per cell (bit-identical) and
    against the oracle (1e-9 loss, 1e-7 gradient)."""
    x, y, _ = make_regression(n, d, n_outputs=3, n_test=0, config=12, unit=n)
    h = make_handle(lib, n, d, kernel, ard, x, y)
    nt = 2 + (d if ard else 1)
    try:
        rng = np.random.default_rng(5)
        units = np.ascontiguousarray(rng.integers(0, 3, size=cells), dtype=np.int32)
        thetas = np.ascontiguousarray(rng.normal(0.2, 0.3, size=(cells, nt)))
        losses = np.zeros(cells)
        grads = np.zeros((cells, nt))
        check(lib.gprx_objective_batch(h, cells, ptr(units), ptr(thetas), None, ALL, ptr(losses), ptr(grads)), h)
        for c in range(cells):
            th = np.ascontiguousarray(thetas[c])
            single = C.c_double()
            g1 = np.zeros(nt)
            check(lib.gprx_objective(h, int(units[c]), ptr(th), None, ALL, C.byref(single), ptr(g1)), h)
            assert single.value == losses[c]
            assert np.array_equal(g1, grads[c])
            wl = th[1:-1] if ard else float(th[1])
            ref_loss, ref_g = oex.loss_and_grad(kernel, x, y[:, units[c]], float(th[0]), wl, float(th[-1]))
            ref_grad = np.concatenate([[ref_g["variance"]], np.atleast_1d(ref_g["lengthscales"]), [ref_g["noise"]]])
            assert abs(losses[c] - ref_loss) <= 1e-9 * abs(ref_loss)
            assert np.max(np.abs(grads[c] - ref_grad)) <= 1e-7 * np.max(np.abs(ref_grad))
        # loss only, partial mask
        l2 = np.zeros(cells)
        check(lib.gprx_objective_batch(h, cells, ptr(units), ptr(thetas), None, _lib.TRAIN_NOISE, ptr(l2), None), h)
        for c in range(cells):
            single = C.c_double()
            check(lib.gprx_factorize(h, int(units[c]), ptr(np.ascontiguousarray(thetas[c])), None, _lib.TRAIN_NOISE, C.byref(single)), h)
            assert single.value == l2[c]
    finally:
        lib.gprx_destroy(h)


def test_predict_batch_matches_oracle(lib):
    """gprx_predict_batch: the per-mode predict loop of gpr.py:336-339 in one call."""
    n, d, cells, ns = 400, 4, 3, 60
    x, y, xs = make_regression(n, d, n_outputs=cells, n_test=ns, config=13, unit=2)
    h = make_handle(lib, n, d, "Matern52", False, x, y)
    try:
        thetas = np.ascontiguousarray([pack_theta(1.0 + 0.3 * c, 0.8 + 0.1 * c, 0.05 * (c + 1)) for c in range(cells)])
        units = np.arange(cells, dtype=np.int32)
        means, vars_ = np.zeros((cells, ns)), np.zeros((cells, ns))
        check(lib.gprx_predict_batch(h, cells, ptr(units), ptr(thetas), None, ptr(xs), ns, ptr(means), ptr(vars_), 1), h)
        for c in range(cells):
            rm, rv = oex.predict("Matern52", x, y[:, c], 1.0 + 0.3 * c, 0.8 + 0.1 * c, 0.05 * (c + 1), xs)
            assert np.max(np.abs(means[c] - rm)) <= 1e-8 * np.max(np.abs(rm)) and np.max(np.abs(vars_[c] - rv) / rv) <= 1e-8
    finally:
        lib.gprx_destroy(h)


@pytest.mark.parametrize("n", [320, 1000])
def test_gradient_does_not_depend_on_old_workspace_contents(lib, n):
    """The L^-1 workspace of the gradient is no longer zeroed (every tile that is read was written first, the triangular K ranges
    never reach a tile above the diagonal): with the workspace poisoned by NaN patterns ("poison_workspace") the single and the
    batched gradient are the same numbers, bit for bit."""
    d = 4
    x, y, _ = make_regression(n, d, n_outputs=3, n_test=8, config=1, unit=n)
    theta = pack_theta(1.1, 0.8, 0.05)
    out = {}
    for poison in (0, 1):
        h = make_handle(lib, n, d, "RBF", False, x, y)
        try:
            check(lib.gprx_set_handle_tuning(h, b"poison_workspace", poison), h)
            loss, grad = C.c_double(), np.zeros(theta.size)
            check(lib.gprx_objective(h, 1, ptr(theta), None, ALL, C.byref(loss), ptr(grad)), h)
            units = np.array([0, 1, 2, 1], dtype=np.int32)
            thetas = np.ascontiguousarray(np.tile(theta, (4, 1)) + 0.01 * np.arange(4)[:, None])
            losses, grads = np.zeros(4), np.zeros((4, theta.size))
            check(lib.gprx_objective_batch(h, 4, ptr(units), ptr(thetas), None, ALL, ptr(losses), ptr(grads)), h)
            out[poison] = (loss.value, grad, losses, grads)
        finally:
            lib.gprx_destroy(h)
    assert np.all(np.isfinite(out[1][1])) and np.all(np.isfinite(out[1][3]))
    assert out[0][0] == out[1][0] and np.array_equal(out[0][1], out[1][1])
    assert np.array_equal(out[0][2], out[1][2]) and np.array_equal(out[0][3], out[1][3])


@pytest.mark.parametrize("kernel,noise", [("RBF", 0.05), ("RBF", 2e-6), ("Matern32", 1e-4)])
def test_alpha_from_the_gradients_inverse_agrees_with_backward_substitution(lib, kernel, noise):
    """An evaluation WITH a gradient forms alpha = X^T beta from the inverse it builds (solve.h alpha_from_inverse), a plain
    factorisation by backward substitution: the predictive means that follow agree far inside the 1e-8 of the boundary, also
    near the noise floor where K is worst conditioned."""
    n, d, ns = 700, 5, 257
    x, y, xs = make_regression(n, d, n_outputs=1, n_test=ns, config=1, unit=n)
    h = make_handle(lib, n, d, kernel, False, x, y)
    try:
        theta = pack_theta(1.2, 0.9, noise)
        loss, grad = C.c_double(), np.zeros(theta.size)
        check(lib.gprx_objective(h, 0, ptr(theta), None, ALL, C.byref(loss), ptr(grad)), h)
        m1, v1 = np.zeros(ns), np.zeros(ns)
        check(lib.gprx_predict(h, ptr(xs), ns, ptr(m1), ptr(v1), 1), h)
        loss2 = C.c_double()
        check(lib.gprx_factorize(h, 0, ptr(theta), None, ALL, C.byref(loss2)), h)
        m2, v2 = np.zeros(ns), np.zeros(ns)
        check(lib.gprx_predict(h, ptr(xs), ns, ptr(m2), ptr(v2), 1), h)
        assert loss.value == loss2.value
        assert np.max(np.abs(m1 - m2)) <= 1e-9 * np.max(np.abs(m2))
        assert np.array_equal(v1, v2)  # (the variance does not involve alpha)
        ref_mean, _ = oex.predict(kernel, x, y[:, 0], 1.2, 0.9, noise, xs, True)
        assert np.max(np.abs(m1 - ref_mean)) <= 1e-8 * np.max(np.abs(ref_mean))
    finally:
        lib.gprx_destroy(h)


PAIR_SOLVE = r"""
import ctypes as C, json, sys
import numpy as np
sys.path.insert(0, {root!r})
from gpras_amd import _lib
from gpras_amd._lib import check, ptr
from gpras_amd.synth import make_regression
from oracle import transforms as otr
lib = _lib.load()
out = []
for n, d in ((50, 2), (128, 3), (190, 2), (448, 4), (1000, 5), (2100, 8)):  # 1, 2, 3, 7, 16 and 33 block steps
    x, y, xs = make_regression(n, d, n_outputs=1, n_test=40, config=3, unit=n)
    theta = np.array(otr.unconstrain(1.3, float(np.mean(np.abs(x))), 0.2), dtype=np.float64)
    h = C.c_void_p()
    check(lib.gprx_create(0, n, d, 0, 0, 0, C.byref(h)))
    check(lib.gprx_set_data(h, ptr(x), ptr(y), 1), h)
    loss = C.c_double()
    check(lib.gprx_factorize(h, 0, ptr(theta), None, 7, C.byref(loss)), h)
    mean, var = np.zeros(40), np.zeros(40)
    check(lib.gprx_predict(h, ptr(xs), 40, ptr(mean), ptr(var), 1), h)
    lib.gprx_destroy(h)
    out += [float.hex(v) for v in mean] + [float.hex(loss.value)]
print(json.dumps(out))
"""


def test_backward_solve_two_steps_per_launch_equals_one_step_per_launch_bit_for_bit():
    """A lone fit's alpha = L^-T beta runs two block steps per launch (trsv_bwd_pair_step: every workgroup forms x_{i-1} itself instead
    of waiting for a launch boundary).  Every sum is the one-step kernel's, in its order: the predictive means (K*^T alpha) of six sizes
    -- odd and even block counts, one block, two blocks -- are the same bits with GPRX_TRSV_PAIR=0, and they match the oracle."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for env in ({"GPRX_TRSV_PAIR": "0"}, {}):
        base = {k: v for k, v in os.environ.items() if k != "GPRX_TRSV_PAIR"}
        res = subprocess.run([sys.executable, "-c", PAIR_SOLVE.format(root=root)], capture_output=True, text=True, timeout=600, env=dict(base, **env))
        assert res.returncode == 0, res.stderr[-2000:]
        outs.append(json.loads(res.stdout.strip().splitlines()[-1]))
    assert outs[0] == outs[1]
