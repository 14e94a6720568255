"""End-to-end GPU tests of the drop-in class: GPRAS.fit / predict / to_file / from_file on the HIP
engine against the oracle's CPU restatement of the same driver (same seeds, same inputs)."""

import numpy as np
import pytest

from gpras_amd.gpr import GPRAS
from gpras_amd.synth import make_hydrograph_features, make_regression
from oracle import gpras_oracle

pytestmark = pytest.mark.gpu


def test_config1_fit_predict_roundtrip(tmp_path):
    """BASELINE config 1: single cell, N=256 d=4 RBF, fit + predict on synthetic hydrograph features."""
    x, y = make_hydrograph_features(256, 4, n_outputs=2, config=1, unit=0)
    xs = x[::5] + 0.01
    g = GPRAS("RBF")
    g.fit(x, y, n_inducing=24, inducing_initializer="kmeans", optimization_method="two-stage", max_iter=15)
    ref = gpras_oracle.GPRASOracle("RBF")
    ref.fit(x, y, n_inducing=24, inducing_initializer="kmeans", optimization_method="two-stage", max_iter=15)
    for a, b in zip(g.models, ref.models):
        assert a.variance == pytest.approx(b.variance, rel=1e-8)
        assert a.lengthscales == pytest.approx(b.lengthscales, rel=1e-8)
        assert a.noise == pytest.approx(b.noise, rel=1e-8)
        assert np.allclose(a.inducing_variable.Z, b.Z, rtol=1e-8, atol=1e-10)
    mean, var = g.predict(xs)
    rmean, rvar = ref.predict(xs)
    assert mean.shape == (xs.shape[0], 2) and var.shape == mean.shape
    assert np.max(np.abs(mean - rmean)) <= 1e-8 * np.max(np.abs(rmean))
    assert np.max(np.abs(var - rvar) / rvar) <= 1e-8
    path = tmp_path / "model.json"
    g.to_file(path)
    g2 = GPRAS.from_file(path)
    mean2, var2 = g2.predict(xs)
    assert np.array_equal(mean, mean2) and np.array_equal(var, var2)


@pytest.mark.parametrize("kernel", ["Matern52", "Exponential"])
def test_lbfgs_reaches_the_same_optimum(kernel):
    x, y, xs = make_regression(400, 4, n_outputs=1, n_test=50, config=4, unit=1)
    g = GPRAS(kernel)
    g.fit(x, y, n_inducing=30, optimization_method="L-BFGS-B", max_iter=25)
    ref = gpras_oracle.GPRASOracle(kernel)
    ref.fit(x, y, n_inducing=30, optimization_method="L-BFGS-B", max_iter=25)
    # trajectories of a quasi-Newton method are chaotic in the last digits: compare the objective reached
    lg, lr = g.models[0].training_loss(), ref.models[0].training_loss()
    assert lg == pytest.approx(lr, rel=1e-6)
    mean, _ = g.predict(xs)
    rmean, _ = ref.predict(xs)
    assert np.max(np.abs(mean - rmean)) <= 1e-4 * np.max(np.abs(rmean))


def test_exact_gp_ard_matern52_lbfgs():
    """BASELINE config 3 shape at test size: Matern-5/2 ARD, L-BFGS-B on the exact log marginal likelihood."""
    x, y, xs = make_regression(512, 8, n_outputs=1, n_test=64, config=3, unit=0)
    g = GPRAS("Matern52")
    g.fit(x, y, None, optimization_method="L-BFGS-B", ard=True, max_iter=10)
    ref = gpras_oracle.GPRASOracle("Matern52")
    ref.fit(x, y, None, optimization_method="L-BFGS-B", ard=True, max_iter=10)
    assert g.models[0].lengthscales.shape == (8,)
    assert g.models[0].training_loss() == pytest.approx(ref.models[0].training_loss(), rel=1e-7)
    mean, var = g.predict(xs)
    rmean, rvar = ref.predict(xs)
    assert np.max(np.abs(mean - rmean)) <= 1e-5 * np.max(np.abs(rmean))
    assert np.all(var > 0)


def test_errors_surface_as_python_exceptions():
    x, y, _ = make_regression(100, 2, config=4, unit=2)
    g = GPRAS("RBF")
    with pytest.raises(KeyError):
        g.fit(x, y, 8, optimization_method="no-such")
    g.fit(x, y, 8, "grid", "adam", max_iter=1)
    with pytest.raises(ValueError):
        g.predict(np.zeros((3, 5)))
    # duplicated inducing points with a vanishing jitter-free kernel matrix are still PD thanks to the jitter
    g.models[0].Z = np.repeat(x[:1], 8, axis=0)
    mean, var = g.predict(x[:4])
    assert np.all(np.isfinite(mean)) and np.all(np.isfinite(var))


def test_concurrent_workers_give_identical_models():
    """workers > 1: modes fitted from several host threads / handles concurrently; per-mode results unchanged."""
    import time

    x, y, xs = make_regression(1500, 6, n_outputs=8, n_test=20, config=4, unit=5)
    g1 = GPRAS("Matern32")
    t0 = time.perf_counter()
    g1.fit(x, y, 40, "grid", "adam", max_iter=30)
    t1 = time.perf_counter() - t0
    g4 = GPRAS("Matern32")
    t0 = time.perf_counter()
    g4.fit(x, y, 40, "grid", "adam", max_iter=30, workers=4)
    t4 = time.perf_counter() - t0
    for a, b in zip(g1.models, g4.models):
        assert a.variance == b.variance and a.noise == b.noise and np.array_equal(a.Z, b.Z)
    m1, v1 = g1.predict(xs)
    m4, v4 = g4.predict(xs)
    assert np.array_equal(m1, m4) and np.array_equal(v1, v4)
    print(f"serial {t1:.3f} s, 4 workers {t4:.3f} s")


def test_exact_predict_is_batched_and_equals_the_per_mode_loop():
    """GPRAS.predict on exact models factorises all modes by one batched launch sequence; the numbers must equal
    the reference-shaped loop (model.predict_y per mode) exactly, and the oracle to 1e-8."""
    x, y, xs = make_regression(600, 5, n_outputs=4, n_test=70, config=9, unit=2)
    g = GPRAS("Matern32")
    g.fit(x, y, None, optimization_method="adam", max_iter=3)
    for i, m in enumerate(g.models):  # distinct hyperparameters per mode
        m.assign(variance=0.8 + 0.2 * i, lengthscales=0.7 + 0.1 * i, noise=0.05 * (i + 1))
    mean, var = g.predict(xs)
    loop = [m.predict_y(xs) for m in g.models]
    assert np.array_equal(mean, np.concatenate([p[0] for p in loop], axis=1))
    assert np.array_equal(var, np.concatenate([p[1] for p in loop], axis=1))
    from oracle import exact as oex

    for i, m in enumerate(g.models):
        rm, rv = oex.predict("Matern32", x, y[:, i], m.variance, m.lengthscales, m.noise, xs)
        assert np.max(np.abs(mean[:, i] - rm)) <= 1e-8 * np.max(np.abs(rm))
        assert np.max(np.abs(var[:, i] - rv) / rv) <= 1e-8


def test_batched_differential_evolution_on_exact_model():
    """``batched=True``: scipy hands a whole generation to the objective, which the engine evaluates in one batched
    launch sequence.  The objective values must equal single evaluations, and the search must not end above its start."""
    x, y, _ = make_regression(300, 3, n_outputs=1, n_test=0, config=5, unit=4)
    g = GPRAS("RBF")
    g._init_models(x.astype(np.float64), y.astype(np.float64), None)
    model = g.models[0]
    model.set_all_trainable(False)
    start = model.training_loss()
    rng = np.random.default_rng(0)
    cand = np.stack([rng.uniform(-1, 1, 7), rng.uniform(-1, 1, 7), rng.uniform(-3, 0, 7)])
    many = model.training_loss_many(10.0 ** cand[0], 10.0 ** cand[1], 10.0 ** cand[2])
    for k in range(7):
        model.assign(variance=10.0 ** cand[0, k], lengthscales=10.0 ** cand[1, k], noise=10.0 ** cand[2, k])
        assert model.training_loss() == many[k]
    g2 = GPRAS("RBF")
    g2.fit(x, y, None, optimization_method="diffential_evolution", popsize=5, max_iter=4, seed=1, adam_iter=2, verbose=False, batched=True)
    m2 = g2.models[0]
    m2.set_all_trainable(False)
    assert m2.training_loss() <= start
    assert 0.1 <= m2.variance <= 10.0 and 0.1 <= m2.lengthscales <= 10.0 and 1e-3 <= m2.noise <= 1.0


def test_lockstep_fit_on_the_engine_equals_the_serial_loop():
    """Exact models, 5 modes, L-BFGS-B: the lock-step fit (batched evaluations) and the reference-shaped serial loop
    must end at bit-identical parameters and predictions; the oracle's serial restatement agrees to optimiser accuracy."""
    x, y, xs = make_regression(320, 4, n_outputs=5, n_test=40, config=10, unit=3)
    a = GPRAS("Matern52")
    a.fit(x, y, None, optimization_method="L-BFGS-B", ard=True, max_iter=8)
    assert a.lockstep_stats["batches"] < a.lockstep_stats["evaluations"]
    b = GPRAS("Matern52")
    b.fit(x, y, None, optimization_method="L-BFGS-B", ard=True, max_iter=8, lockstep=False)
    for ma, mb in zip(a.models, b.models):
        assert ma.variance == mb.variance and ma.noise == mb.noise and np.array_equal(ma.lengthscales, mb.lengthscales)
    pa, pb = a.predict(xs), b.predict(xs)
    assert np.array_equal(pa[0], pb[0]) and np.array_equal(pa[1], pb[1])
    ref = gpras_oracle.GPRASOracle("Matern52")
    ref.fit(x, y, None, optimization_method="L-BFGS-B", ard=True, max_iter=8)
    for ma, mr in zip(a.models, ref.models):
        assert ma.training_loss() == pytest.approx(mr.training_loss(), rel=1e-6)


def test_lockstep_fit_of_sparse_models_on_the_engine():
    """The reference's default configuration (SGPR, kmeans, two-stage Adam) over 6 modes: lock step with batched
    evaluations against the serial loop (bit-identical) and the oracle's serial restatement."""
    x, y = make_hydrograph_features(300, 4, n_outputs=6, config=1, unit=3)
    xs = x[::7] + 0.01
    a = GPRAS("Matern32")
    a.fit(x, y, n_inducing=20, inducing_initializer="kmeans", optimization_method="two-stage", max_iter=12)
    assert a.lockstep_stats["batches"] < a.lockstep_stats["evaluations"]
    b = GPRAS("Matern32")
    b.fit(x, y, n_inducing=20, inducing_initializer="kmeans", optimization_method="two-stage", max_iter=12, lockstep=False)
    for ma, mb in zip(a.models, b.models):
        assert ma.variance == mb.variance and ma.noise == mb.noise and ma.lengthscales == mb.lengthscales
        assert np.array_equal(ma.Z, mb.Z)
    ref = gpras_oracle.GPRASOracle("Matern32")
    ref.fit(x, y, n_inducing=20, inducing_initializer="kmeans", optimization_method="two-stage", max_iter=12)
    mean, var = a.predict(xs)
    rmean, rvar = ref.predict(xs)
    assert np.max(np.abs(mean - rmean)) <= 1e-8 * np.max(np.abs(rmean))
    assert np.max(np.abs(var - rvar) / rvar) <= 1e-8


def test_batched_differential_evolution_of_several_modes_in_lock_step():
    """with several modes the drivers run in lock-step threads; batched DE calls Engine.factorize_batch
    directly from every thread on the SHARED handle.  The engine serialises those calls on its lock, so the multi-mode fit
    must equal the serial (lockstep=False) one bit for bit -- seeded DE, deterministic batched evaluations."""
    x, y, _ = make_regression(192, 3, n_outputs=4, n_test=0, config=12, unit=9)
    kw = dict(popsize=4, max_iter=3, seed=3, adam_iter=2, verbose=False, batched=True)
    a = GPRAS("RBF")
    a.fit(x, y, None, optimization_method="diffential_evolution", **kw)  # lockstep on by default for > 1 mode
    b = GPRAS("RBF")
    b.fit(x, y, None, optimization_method="diffential_evolution", lockstep=False, **kw)
    for ma, mb in zip(a.models, b.models):
        assert ma.variance == mb.variance and ma.lengthscales == mb.lengthscales and ma.noise == mb.noise


def test_batch_sizes_follow_the_free_device_memory():
    """the lock-step batch and the batched predict are sized from hipMemGetInfo and the per-cell
    footprint; an over-large batch is split instead of raising MemoryError."""
    x, y, xs = make_regression(256, 3, n_outputs=6, n_test=20, config=12, unit=10)
    g = GPRAS("RBF")
    g.fit(x, y, None, optimization_method="adam", max_iter=2)
    eng = g.engine
    cells = eng.max_cells(want_grad=True)
    assert cells >= 6 and eng.max_cells(want_grad=False) >= cells
    full = g.predict(xs)
    eng.max_cells = lambda want_grad=False, reserve=0.15: 4  # pretend only four cells fit: predict must chunk, same numbers
    chunked = g.predict(xs)
    assert np.array_equal(full[0], chunked[0]) and np.array_equal(full[1], chunked[1])


def test_sparse_predict_is_batched_and_equals_the_per_mode_loop():
    """GPRAS.predict on sparse models goes through gprx_predict_batch; the reference-shaped loop over model.predict_y gives the
    same numbers bit for bit."""
    x, y, xs = make_regression(400, 4, n_outputs=6, n_test=300, config=14, unit=2)
    g = GPRAS("Matern52")
    g.fit(x, y, 24, "kmeans", "adam", max_iter=4)
    mean, var = g.predict(xs)
    loop = [m.predict_y(xs) for m in g.models]
    assert np.array_equal(mean, np.concatenate([p[0] for p in loop], axis=1))
    assert np.array_equal(var, np.concatenate([p[1] for p in loop], axis=1))
    ref = gpras_oracle.GPRASOracle("Matern52")
    ref.fit(x, y, 24, "kmeans", "adam", max_iter=4)
    rmean, rvar = ref.predict(xs)
    assert np.max(np.abs(mean - rmean)) <= 1e-8 * np.max(np.abs(rmean)) and np.max(np.abs(var - rvar) / rvar) <= 1e-8


@pytest.mark.parametrize("frozen", [dict(Z=False), dict(variance=False, noise=False), dict(lengthscales=False, Z=False)])
def test_resident_adam_with_frozen_parameters_equals_the_python_loop(frozen):
    """The device-resident Adam loop of the sparse model (M <= 64: update, stop rule and the next step's Kuu in one launch) with parts of
    the model frozen, as the reference's two-stage fit freezes them (gpr.py:111-145): same variables bit for bit as the packed Python
    loop over host evaluations, frozen parameters untouched, 70 steps (the stop flags are read every 25)."""
    from gpras_amd import optimizers

    x, y = make_hydrograph_features(300, 4, n_outputs=4, config=1, unit=23)

    def prepared():
        g = GPRAS("RBF")
        g._init_models(x, y, 20, "grid")
        for m in g.models:
            m.set_all_trainable(True)
            m.set_trainable(**frozen)
            m.n_evals = 0
        return g

    a, b = prepared(), prepared()
    before = [(m.variance, np.array(m.lengthscales, copy=True), m.noise, np.array(m.Z, copy=True)) for m in a.models]
    optimizers._optimize_adam_many(a.models, 70)  # library loop (resident on the device)
    batch = optimizers._PackedBatch(b.models)
    optimizers._adam_packed(batch, np.stack([m.get_vector() for m in b.models]), 70, None)  # Python loop, batched host evaluations
    for ma, mb, (v0, l0, s0, z0) in zip(a.models, b.models, before):
        assert ma.n_evals == mb.n_evals and ma.n_evals > 50
        assert np.array_equal(ma.get_vector(), mb.get_vector())
        assert np.array_equal(ma.Z, mb.Z) and ma.variance == mb.variance and ma.noise == mb.noise
        if frozen.get("Z") is False:
            assert np.array_equal(ma.Z, z0)
        if frozen.get("variance") is False:
            assert ma.variance == v0 and ma.noise == s0
        if frozen.get("lengthscales") is False:
            assert np.array_equal(np.asarray(ma.lengthscales), l0)


@pytest.mark.parametrize("n_inducing", [None, 16])
def test_adam_inside_the_library_equals_the_python_loops(n_inducing):
    """gprx_adam_batch (the lock-step Adam driver inside libgprx.so) against the packed Python loop and against the serial
    per-model driver: same variables bit for bit and the same number of evaluations per model -- including models that stop
    early (they start at an L-BFGS optimum, so the loss cannot improve by 1e-5 for 50 steps) beside models that keep going."""
    from gpras_amd import optimizers

    x, y = make_hydrograph_features(260, 3, n_outputs=5, config=1, unit=11)

    def prepared():
        g = GPRAS("Matern52")
        g.fit(x, y, n_inducing, "grid", "L-BFGS-B", max_iter=40, lockstep=False)
        for m in g.models[3:]:  # two models are pushed away from the optimum: they keep improving
            m.set_vector(m.get_vector() + 0.3)
        for m in g.models:
            m.n_evals = 0
            m.set_all_trainable(True)
        return g

    a, b, c = prepared(), prepared(), prepared()
    assert hasattr(a.engine, "adam_batch")
    optimizers._optimize_adam_many(a.models, 90)  # library loop
    batch = optimizers._PackedBatch(b.models)
    optimizers._adam_packed(batch, np.stack([m.get_vector() for m in b.models]), 90, None)  # Python loop, batched evaluations
    for m in c.models:
        optimizers._optimize_adam(m, 90)  # serial driver
    evals = [m.n_evals for m in a.models]
    assert evals == [m.n_evals for m in b.models] == [m.n_evals for m in c.models]
    assert max(evals) == 90
    if n_inducing is None:
        assert min(evals) < 90  # both kinds of model are present (the sparse models' 40 L-BFGS iterations do not end at an optimum)
    for ma, mb, mc in zip(a.models, b.models, c.models):
        va, vb, vc = ma.get_vector(), mb.get_vector(), mc.get_vector()
        assert np.array_equal(va, vb) and np.array_equal(va, vc)


@pytest.mark.parametrize("n_inducing", [None, 12])
def test_predict_in_the_references_layout_equals_the_transposed_batched_predict(n_inducing, monkeypatch):
    """gprx_predict_batch_t hands the predictions over as (N*, modes) -- the layout GPRAS.predict returns (gpr.py:340-342) --, transposed on
    the device: the same bits as gprx_predict_batch's (modes, N*) block, whole and slab by slab (GPRX_PREDICT_SLAB forces the 2-D copies)."""
    x, y = make_hydrograph_features(200, 3, n_outputs=5, config=1, unit=31)
    xs = make_hydrograph_features(333, 3, n_outputs=1, config=1, unit=32)[0]
    g = GPRAS("Matern32")
    g.fit(x, y, n_inducing, "grid", "adam", max_iter=2)
    eng = g.engine
    units = np.arange(5, dtype=np.int32)
    thetas = np.stack([m.theta() for m in g.models])
    zs = None if n_inducing is None else np.stack([m.Z for m in g.models])
    mean, var = eng.predict_batch(units, thetas, xs, zs=zs)
    for slab in (None, "2"):
        if slab:
            monkeypatch.setenv("GPRX_PREDICT_SLAB", slab)
        mean_t, var_t = eng.predict_batch_t(units, thetas, xs, zs=zs)
        assert mean_t.shape == (333, 5) and np.array_equal(mean_t, mean.T) and np.array_equal(var_t, var.T)
    monkeypatch.delenv("GPRX_PREDICT_SLAB")
    pm, pv = g.predict(xs)
    assert np.array_equal(pm, mean.T) and np.array_equal(pv, var.T)
