"""Edge cases of the path on the GPU: ragged and tiny sizes, wide feature vectors, many units, empty
predict batches, prediction tiles larger than one pass, duplicated inputs."""

import ctypes as C

import numpy as np
import pytest

from gpras_amd import _lib
from gpras_amd._lib import check, ptr
from gpras_amd.gpr import GPRAS
from gpras_amd.synth import make_regression
from oracle import exact as oex
from oracle import kernels as okn
from oracle import sgpr as osg
from oracle import transforms as otr

pytestmark = pytest.mark.gpu


def theta_of(variance, ls, noise):
    wv, wl, wn = otr.unconstrain(variance, ls, noise)
    return np.ascontiguousarray(np.concatenate([[wv], np.atleast_1d(wl), [wn]]))


@pytest.mark.parametrize("n,d,m", [(3, 1, 0), (63, 2, 0), (65, 3, 0), (129, 12, 0), (200, 50, 0), (70, 2, 1), (130, 7, 65), (100, 3, 100), (50, 4, 64)])
def test_ragged_sizes_exact_and_sparse(lib, n, d, m):
    """Sizes around the 64-padding granule, d up to 50 (the reference sweeps 1..50 modes), M = 1, M = N, M > N."""
    x, y, xs = make_regression(n, d, n_outputs=1, n_test=7, config=5, unit=n + d + m)
    h = C.c_void_p()
    check(lib.gprx_create(0, n, d, m, okn.KERNEL_IDS["Matern52"], 0, C.byref(h)))
    check(lib.gprx_set_data(h, ptr(x), ptr(y), 1), h)
    try:
        variance, ls, noise = 0.9, 1.1, 0.2
        theta = theta_of(variance, ls, noise)
        z = None
        if m:
            rng = np.random.default_rng(m)
            z = np.ascontiguousarray(x[rng.integers(0, n, size=m)] + 0.05 * rng.standard_normal((m, d)))
        loss = C.c_double()
        grad = np.zeros(3 + m * d)
        check(lib.gprx_objective(h, 0, ptr(theta), ptr(z) if m else None, 15 if m else 7, C.byref(loss), ptr(grad)), h)
        if m:
            ref_loss, g = osg.loss_and_grad("Matern52", x, y[:, 0], z, theta[0], float(theta[1]), theta[2])
            ref = np.concatenate([[g["variance"], g["lengthscales"], g["noise"]], g["Z"].ravel()])
            ref_mean, ref_var = osg.predict("Matern52", x, y[:, 0], z, variance, ls, noise, xs)
        else:
            ref_loss, g = oex.loss_and_grad("Matern52", x, y[:, 0], theta[0], float(theta[1]), theta[2])
            ref = np.array([g["variance"], g["lengthscales"], g["noise"]])
            ref_mean, ref_var = oex.predict("Matern52", x, y[:, 0], variance, ls, noise, xs)
        assert abs(loss.value - ref_loss) <= 1e-9 * abs(ref_loss)
        assert np.max(np.abs(grad - ref)) <= 1e-7 * max(1.0, np.max(np.abs(ref)))
        mean, var = np.zeros(7), np.zeros(7)
        check(lib.gprx_predict(h, ptr(xs), 7, ptr(mean), ptr(var), 1), h)
        assert np.max(np.abs(mean - ref_mean)) <= 1e-8 * max(np.max(np.abs(ref_mean)), 1e-3)
        assert np.max(np.abs(var - ref_var) / ref_var) <= 1e-8
        # empty batch is a no-op, not an error
        assert lib.gprx_predict(h, None, 0, None, None, 1) == _lib.GPRX_OK
    finally:
        lib.gprx_destroy(h)


def test_predict_tiling_and_latent_variance(lib):
    """More test points than one 8192-column pass; include_noise = 0 gives predict_f."""
    n, d, ns = 300, 3, 8192 + 8192 + 77
    x, y, xs = make_regression(n, d, n_outputs=1, n_test=ns, config=5, unit=1)
    h = C.c_void_p()
    check(lib.gprx_create(0, n, d, 0, 0, 0, C.byref(h)))
    check(lib.gprx_set_data(h, ptr(x), ptr(y), 1), h)
    try:
        theta = theta_of(1.0, 0.8, 0.1)
        loss = C.c_double()
        check(lib.gprx_factorize(h, 0, ptr(theta), None, 7, C.byref(loss)), h)
        mean, var, varf = np.zeros(ns), np.zeros(ns), np.zeros(ns)
        check(lib.gprx_predict(h, ptr(xs), ns, ptr(mean), ptr(var), 1), h)
        check(lib.gprx_predict(h, ptr(xs), ns, ptr(mean), ptr(varf), 0), h)
        ref_mean, ref_var = oex.predict("RBF", x, y[:, 0], 1.0, 0.8, 0.1, xs, include_noise=True)
        assert np.max(np.abs(mean - ref_mean)) <= 1e-8 * np.max(np.abs(ref_mean))
        assert np.max(np.abs(var - ref_var) / ref_var) <= 1e-8
        assert np.allclose(var - varf, 0.1, rtol=0, atol=1e-12)
    finally:
        lib.gprx_destroy(h)


def test_predict_inverse_path_matches_substitution(lib):
    """N* >= 2 N switches predict to one triangular GEMM against the explicit inverse of L; N* < 2 N uses blocked
    substitution.  Both must agree with the oracle (and hence with each other) on an ill-conditioned-ish K."""
    n, d = 640, 4
    x, y, xs = make_regression(n, d, n_outputs=1, n_test=2 * n + 50, config=5, unit=4)
    h = C.c_void_p()
    check(lib.gprx_create(0, n, d, 0, okn.KERNEL_IDS["RBF"], 0, C.byref(h)))
    check(lib.gprx_set_data(h, ptr(x), ptr(y), 1), h)
    try:
        variance, ls, noise = 2.0, 1.5, 1e-4  # cond(K) ~ 1e7
        theta = theta_of(variance, ls, noise)
        loss = C.c_double()
        check(lib.gprx_factorize(h, 0, ptr(theta), None, 7, C.byref(loss)), h)
        ns = xs.shape[0]
        m_big, v_big = np.zeros(ns), np.zeros(ns)
        check(lib.gprx_predict(h, ptr(xs), ns, ptr(m_big), ptr(v_big), 1), h)  # inverse path
        m_small, v_small = np.zeros(100), np.zeros(100)
        check(lib.gprx_predict(h, ptr(xs), 100, ptr(m_small), ptr(v_small), 1), h)  # substitution path
        ref_m, ref_v = oex.predict("RBF", x, y[:, 0], variance, ls, noise, xs, True)
        assert np.max(np.abs(m_big - ref_m)) <= 1e-8 * np.max(np.abs(ref_m))
        assert np.max(np.abs(v_big - ref_v) / ref_v) <= 1e-8
        assert np.max(np.abs(v_small - ref_v[:100]) / ref_v[:100]) <= 1e-8
        assert np.max(np.abs(v_big[:100] - v_small) / v_small) <= 1e-9
    finally:
        lib.gprx_destroy(h)


def test_many_units_share_one_handle():
    """K = 12 output columns (spatial modes) through the class: one handle, unit-major y on the device."""
    x, y, xs = make_regression(150, 4, n_outputs=12, n_test=9, config=5, unit=2)
    g = GPRAS("RBF")
    g.fit(x, y, 10, "grid", "adam", max_iter=2)
    mean, var = g.predict(xs)
    assert mean.shape == (9, 12) and var.shape == (9, 12)
    for u in (0, 5, 11):
        m = g.models[u]
        rm, rv = osg.predict("RBF", x, y[:, u], m.Z, m.variance, m.lengthscales, m.noise, xs)
        assert np.allclose(mean[:, u], rm, rtol=1e-8, atol=1e-10) and np.allclose(var[:, u], rv, rtol=1e-8)


def test_duplicate_training_points_need_the_noise_term():
    """Exact GP on duplicated inputs: K is singular without s I; tiny noise must still factor or raise LinAlgError."""
    x, y, _ = make_regression(64, 2, config=5, unit=3)
    x2 = np.vstack([x, x])
    y2 = np.vstack([y, y])
    g = GPRAS("RBF")
    g._init_models(x2.astype(np.float64), y2.astype(np.float64), None)
    g.models[0].assign(noise=1e-3)
    assert np.isfinite(g.models[0].training_loss())
    g.models[0].w_noise = -1e3  # softplus underflows: noise = 1e-6 + 0 -> Cholesky may fail; must be an exception, not garbage
    try:
        val = g.models[0].training_loss()
        assert np.isfinite(val)
    except np.linalg.LinAlgError:
        pass


def test_full_size_4096_batched_cells_against_oracle(lib):
    """BASELINE configs[1] size: four cells of N = 4096 in one batched launch sequence against the CPU oracle
    (loss 1e-9; the oracle needs about a second per cell)."""
    import ctypes as C

    from gpras_amd._lib import check, ptr
    from oracle import exact as oex

    n, d, cells = 4096, 8, 4
    x, y, _ = make_regression(n, d, n_outputs=cells, n_test=0, config=2, unit=77)
    h = C.c_void_p()
    check(lib.gprx_create(0, n, d, 0, 0, 0, C.byref(h)))
    check(lib.gprx_set_data(h, ptr(x), ptr(y), cells), h)
    try:
        rng = np.random.default_rng(4)
        thetas = np.ascontiguousarray(np.array([0.5413, 0.3, 0.5413]) + rng.uniform(-0.2, 0.2, size=(cells, 3)))
        units = np.arange(cells, dtype=np.int32)
        losses = np.zeros(cells)
        status = np.zeros(cells, dtype=np.int32)
        check(lib.gprx_factorize_batch(h, cells, ptr(units), ptr(thetas), 7, ptr(losses), ptr(status)), h)
        for c in range(cells):
            ref = oex.loss("RBF", x, y[:, c], float(thetas[c, 0]), float(thetas[c, 1]), float(thetas[c, 2]))
            assert abs(losses[c] - ref) <= 1e-9 * abs(ref)
    finally:
        lib.gprx_destroy(h)


def test_full_size_16384_gradient_matches_central_differences(lib):
    """BASELINE configs[4] size (N = 16384, d = 12), too large for the oracle: the analytic gradient (L^-1, K^-1, trace
    pass) must match central differences of the loss (three more factorisations per parameter pair), losses must be
    reproducible bit for bit, and the predictive variance at training points must lie between the noise and noise + variance."""
    import ctypes as C

    from gpras_amd._lib import check, ptr

    n, d = 16384, 12
    x, y, _ = make_regression(n, d, n_outputs=1, n_test=0, config=5, unit=0)
    h = C.c_void_p()
    check(lib.gprx_create(0, n, d, 0, 2, 0, C.byref(h)))  # Matern-3/2
    check(lib.gprx_set_data(h, ptr(x), ptr(y), 1), h)
    try:
        theta = np.array([0.4, 0.9, -1.2])
        loss = C.c_double()
        grad = np.zeros(3)
        check(lib.gprx_objective(h, 0, ptr(theta), None, 7, C.byref(loss), ptr(grad)), h)
        again = C.c_double()
        check(lib.gprx_factorize(h, 0, ptr(theta), None, 7, C.byref(again)), h)
        assert again.value == loss.value
        eps = 1e-4
        for k in range(3):
            lp, lm = C.c_double(), C.c_double()
            tp, tm = theta.copy(), theta.copy()
            tp[k] += eps
            tm[k] -= eps
            check(lib.gprx_factorize(h, 0, ptr(tp), None, 7, C.byref(lp)), h)
            check(lib.gprx_factorize(h, 0, ptr(tm), None, 7, C.byref(lm)), h)
            fd = (lp.value - lm.value) / (2 * eps)
            assert abs(fd - grad[k]) <= 1e-5 * max(1.0, abs(grad[k])), (k, fd, grad[k])
        check(lib.gprx_factorize(h, 0, ptr(theta), None, 7, C.byref(again)), h)
        xs = np.ascontiguousarray(x[:500])
        mean, var = np.zeros(500), np.zeros(500)
        check(lib.gprx_predict(h, ptr(xs), 500, ptr(mean), ptr(var), 1), h)
        noise = 1e-6 + np.log1p(np.exp(theta[2]))
        variance = np.log1p(np.exp(theta[0]))
        assert np.all(var >= noise * (1 - 1e-9)) and np.all(var <= (noise + variance) * (1 + 1e-9))
        assert np.all(var <= 2.0 * noise + 1e-9)  # at a training point the latent variance is below the noise
    finally:
        lib.gprx_destroy(h)


@pytest.mark.parametrize("path", [1, 2])
def test_both_predict_paths_agree_with_the_oracle(lib, path):
    """The exact predict has two routes to V = L^-1 Ks (triangular GEMM against the explicit inverse, blocked forward
    substitution); the default choice depends on N and the batch, so each is forced here."""
    n, d, ns = 700, 5, 333
    x, y, xs = make_regression(n, d, n_outputs=1, n_test=ns, config=15, unit=path)
    h = C.c_void_p()
    check(lib.gprx_create(0, n, d, 0, okn.KERNEL_IDS["Matern32"], 0, C.byref(h)))
    check(lib.gprx_set_data(h, ptr(x), ptr(y), 1), h)
    try:
        check(lib.gprx_set_handle_tuning(h, b"predict_path", path), h)  # per-handle: process defaults are copied at creation
        theta = theta_of(1.2, 0.8, 0.03)
        loss = C.c_double()
        check(lib.gprx_factorize(h, 0, ptr(theta), None, 7, C.byref(loss)), h)
        mean, var = np.zeros(ns), np.zeros(ns)
        check(lib.gprx_predict(h, ptr(xs), ns, ptr(mean), ptr(var), 1), h)
        rm, rv = oex.predict("Matern32", x, y[:, 0], 1.2, 0.8, 0.03, xs)
        assert np.max(np.abs(mean - rm)) <= 1e-8 * np.max(np.abs(rm)) and np.max(np.abs(var - rv) / rv) <= 1e-8
    finally:
        lib.gprx_destroy(h)


def test_wait_hands_over_to_stream_synchronize_and_the_next_call_is_clean(lib):
    """after the event poll of wait_stream has handed over to hipStreamSynchronize, the hipErrorNotReady its polls
    left behind must not fail the next launch helper (they end in hipGetLastError()).  The hand-over threshold is lowered to 0."""
    n, d = 2048, 6
    x, y, xs = make_regression(n, d, n_outputs=1, n_test=5, config=5, unit=77)
    h = C.c_void_p()
    check(lib.gprx_create(0, n, d, 0, okn.KERNEL_IDS["RBF"], 0, C.byref(h)))
    check(lib.gprx_set_data(h, ptr(x), ptr(y), 1), h)
    theta = theta_of(1.0, 0.9, 0.3)
    try:
        check(lib.gprx_set_tuning(b"wait_handover_us", 0))
        loss1, loss2 = C.c_double(), C.c_double()
        grad = np.zeros(3)
        for _ in range(3):  # every wait of these calls goes through the hand-over; each following launch must come back GPRX_OK
            check(lib.gprx_factorize(h, 0, ptr(theta), None, 7, C.byref(loss1)), h)
            check(lib.gprx_objective(h, 0, ptr(theta), None, 7, C.byref(loss2), ptr(grad)), h)
            mean, var = np.zeros(5), np.zeros(5)
            check(lib.gprx_predict(h, ptr(xs), 5, ptr(mean), ptr(var), 1), h)
        assert loss1.value == loss2.value
    finally:
        lib.gprx_set_tuning(b"wait_handover_us", 200000)
        lib.gprx_destroy(h)


@pytest.mark.parametrize("m", [40, 100])
def test_engine_objective_with_gradient_then_predict(lib, m):
    """Engine.objective(want_grad=True) followed by Engine.predict either works (M <= 64: the fused evaluation keeps the
    factorisation in cell block 0) or raises a clear error (M > 64: the one-cell batch keeps none), never a stale prediction."""
    from gpras_amd.engine import Engine

    n, d = 600, 4
    x, y, xs = make_regression(n, d, n_outputs=2, n_test=9, config=5, unit=m)
    rng = np.random.default_rng(m)
    z = np.ascontiguousarray(x[rng.choice(n, size=m, replace=False)] + 0.01 * rng.standard_normal((m, d)))
    theta = theta_of(1.1, 0.9, 0.2)
    eng = Engine("Matern32", x, y, m)
    try:
        eng.objective(1, theta, z, 15, want_grad=True)
        if m <= 64:
            mean, var = eng.predict(xs)
            ref_mean, ref_var = osg.predict("Matern32", x, y[:, 1], z, 1.1, 0.9, 0.2, xs)
            assert np.max(np.abs(mean - ref_mean)) <= 1e-8 * np.max(np.abs(ref_mean)) and np.max(np.abs(var - ref_var) / ref_var) <= 1e-8
        else:
            with pytest.raises(Exception, match="gprx_factorize"):
                eng.predict(xs)
            eng.objective(1, theta, z, 15, want_grad=False)
            mean, var = eng.predict(xs)
            ref_mean, ref_var = osg.predict("Matern32", x, y[:, 1], z, 1.1, 0.9, 0.2, xs)
            assert np.max(np.abs(mean - ref_mean)) <= 1e-8 * np.max(np.abs(ref_mean)) and np.max(np.abs(var - ref_var) / ref_var) <= 1e-8
    finally:
        eng.close()
