"""CPU tests of the host side: the C-ABI library loads and exports every declared symbol, and the
product's optimiser drivers / GPRAS class reproduce the oracle's independent restatement when driven
by an oracle-backed stand-in for the HIP engine (test infrastructure; the package itself never
imports the oracle)."""

import ctypes as C
import os
import re

import numpy as np
import pytest

from gpras_amd import _lib, gpr, model, optimizers
from gpras_amd.synth import make_hydrograph_features, make_regression
from oracle import exact as oex
from oracle import gpras_oracle
from oracle import sgpr as osg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleBackend:
    """Implements the Engine interface (objective / predict / attributes) on the numpy oracle."""

    def __init__(self, kernel, x, y, n_inducing=0, ard=False, device=0, distance_form="difference"):
        _lib.KERNEL_IDS[kernel]
        self.distance_form = distance_form
        self.kernel, self.x, self.y = kernel, np.asarray(x, float), np.asarray(y, float)
        self.n, self.d = self.x.shape
        self.n_units = self.y.shape[1]
        self.m = int(n_inducing or 0)
        self.ard = bool(ard)
        self.n_len = self.d if ard else 1
        self.n_theta = 2 + self.n_len
        self._last = None

    def close(self):
        pass

    def objective(self, unit, theta, z, mask, want_grad=True):
        wl = theta[1:-1] if self.ard else float(theta[1])
        flags = tuple(bool(mask & b) for b in (1, 2, 4, 8))
        self._last = (unit, np.array(theta), None if z is None else np.array(z))
        if self.m == 0:
            loss, g = oex.loss_and_grad(self.kernel, self.x, self.y[:, unit], float(theta[0]), wl, float(theta[-1]), flags[:3])
            gz = np.zeros(0)
        else:
            loss, g = osg.loss_and_grad(self.kernel, self.x, self.y[:, unit], z, float(theta[0]), wl, float(theta[-1]), flags)
            gz = np.asarray(g["Z"]).ravel()
        if not want_grad:
            return loss, None
        return loss, np.concatenate([[g["variance"]], np.atleast_1d(g["lengthscales"]), [g["noise"]], gz])

    def predict(self, xs, include_noise=True):
        unit, theta, z = self._last
        v = float(model.softplus(theta[0]))
        ls = model.softplus(theta[1:-1])
        ls = ls if self.ard else float(ls[0])
        s = float(model.NOISE_LOWER + model.softplus(theta[-1]))
        if self.m == 0:
            return oex.predict(self.kernel, self.x, self.y[:, unit], v, ls, s, xs, include_noise)
        return osg.predict(self.kernel, self.x, self.y[:, unit], z, v, ls, s, xs, include_noise)


@pytest.fixture()
def oracle_engine(monkeypatch):
    monkeypatch.setattr(gpr, "Engine", OracleBackend)


# ---------------------------------------------------------------------------------------------------
def test_library_loads_and_exports_every_declared_symbol():
    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "gprx.h")).read()
    declared = set(re.findall(r"\b(gprx_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.PROTOTYPES), "include/gprx.h and gpras_amd/_lib.py disagree"
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.gprx_version() == 100
    # argument validation works without a GPU; device calls fail loudly instead of falling back
    h = C.c_void_p()
    assert lib.gprx_create(0, 0, 4, 0, 0, 0, C.byref(h)) == _lib.GPRX_EINVAL
    assert lib.gprx_create(0, 16, 4, 0, 7, 0, C.byref(h)) == _lib.GPRX_EINVAL
    assert "kernel" in _lib.last_error()
    if not os.path.exists("/dev/kfd"):
        with pytest.raises(RuntimeError):
            _lib.check(lib.gprx_create(0, 16, 4, 0, 0, 0, C.byref(h)))


def test_missing_library_is_an_error_not_a_fallback(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", tmp_path / "libgprx.so")
    with pytest.raises(_lib.GprxLibraryError):
        _lib.load()


def test_registries_match_the_reference():
    assert list(optimizers.OPTIMIZERS) == ["two-stage", "three-stage", "adam", "adadelta", "L-BFGS-B", "stochastic", "diffential_evolution"]
    assert list(gpr.KERNEL_FACTORY) == ["Matern12", "Matern32", "Matern52", "RBF", "Linear", "Polynomial", "Periodic", "Exponential"]  # gpr.py:21-29
    with pytest.raises(KeyError):
        gpr.GPRAS("Cosine")
    g = gpr.GPRAS("Periodic")  # constructs, as in the reference; fails where the reference's kernel constructor call does (gpr.py:298)
    with pytest.raises(NotImplementedError):
        g.fit(np.zeros((4, 2)), np.zeros((4, 1)), 2, "grid", "adam", max_iter=1)
    with pytest.raises(TypeError):  # adam has no default max_iter (gpr.py:147)
        optimizers.OPTIMIZERS["adam"](object())


@pytest.mark.parametrize("method,kwargs", [
    ("two-stage", {"max_iter": 12}),
    ("three-stage", {"max_iter": 6}),
    ("adam", {"max_iter": 15}),
    ("adadelta", {"max_iter": 10}),
    ("L-BFGS-B", {"max_iter": 8}),
])
def test_drivers_reproduce_the_oracle_restatement(oracle_engine, method, kwargs):
    x, y, xs = make_regression(90, 3, n_outputs=2, n_test=25, config=7, unit=1)
    ours = gpr.GPRAS("Matern52")
    ours.fit(x, y, n_inducing=10, inducing_initializer="kmeans", optimization_method=method, **kwargs)
    ref = gpras_oracle.GPRASOracle("Matern52")
    ref.fit(x, y, n_inducing=10, inducing_initializer="kmeans", optimization_method=method, **kwargs)
    for a, b in zip(ours.models, ref.models):
        assert a.variance == pytest.approx(b.variance, rel=1e-9)
        assert a.lengthscales == pytest.approx(b.lengthscales, rel=1e-9)
        assert a.noise == pytest.approx(b.noise, rel=1e-9)
        assert np.allclose(a.Z, b.Z, rtol=1e-9, atol=1e-12)
    m1, v1 = ours.predict(xs)
    m2, v2 = ref.predict(xs)
    assert m1.shape == (25, 2) and np.allclose(m1, m2, rtol=1e-8, atol=1e-10) and np.allclose(v1, v2, rtol=1e-8)


def test_multistart_and_de_follow_the_reference_quirks(oracle_engine):
    x, y, _ = make_regression(60, 2, n_outputs=1, n_test=0, config=7, unit=2)
    ours = gpr.GPRAS("RBF")
    ours.fit(x, y, 6, "grid", "stochastic", n_starts=3, iter_initial=2, iter_final=4, rng=np.random.default_rng(5))
    ref = gpras_oracle.GPRASOracle("RBF")
    ref.fit(x, y, 6, "grid", "stochastic", n_starts=3, iter_initial=2, iter_final=4, rng=np.random.default_rng(5))
    a, b = ours.models[0], ref.models[0]
    assert not (a.mask & model.TRAIN_Z)  # Z frozen after the Parameter was replaced by an array
    assert a.variance == pytest.approx(b.variance, rel=1e-8) and np.allclose(a.Z, b.Z)
    ours = gpr.GPRAS("RBF")
    ours.fit(x, y, 6, "grid", "diffential_evolution", popsize=3, max_iter=2, seed=3, adam_iter=5, verbose=False)
    ref = gpras_oracle.GPRASOracle("RBF")
    ref.fit(x, y, 6, "grid", "diffential_evolution", popsize=3, max_iter=2, seed=3, adam_iter=5)
    a, b = ours.models[0], ref.models[0]
    assert a.mask == model.TRAIN_Z  # hyperparameters stay frozen, as in gpr.py:48-49
    assert a.noise == pytest.approx(b.noise, rel=1e-6) and -3 <= np.log10(a.noise) <= 0.001


def test_exact_gp_extension_and_ard(oracle_engine):
    x, y, xs = make_regression(70, 3, n_outputs=1, n_test=10, config=7, unit=3)
    ours = gpr.GPRAS("RBF")
    ours.fit(x, y, None, optimization_method="L-BFGS-B", ard=True, max_iter=5)
    ref = gpras_oracle.GPRASOracle("RBF")
    ref.fit(x, y, None, optimization_method="L-BFGS-B", ard=True, max_iter=5)
    assert ours.models[0].Z is None and ours.models[0].lengthscales.shape == (3,)
    assert np.allclose(ours.models[0].lengthscales, ref.models[0].lengthscales, rtol=1e-9)
    assert np.allclose(ours.predict(xs)[0], ref.predict(xs)[0], rtol=1e-8, atol=1e-10)


def test_file_round_trip_and_plumbing_config1(oracle_engine, tmp_path):
    """BASELINE config 1 shape: N=256, d=4, RBF; fit -> to_file -> from_file -> predict, as pipeline.py:245-260."""
    x, y = make_hydrograph_features(256, 4, n_outputs=2, config=1, unit=0)
    g = gpr.GPRAS("RBF")
    g.fit(x, y, 12, "kmeans", "two-stage", max_iter=4)
    path = tmp_path / "model.json"
    g.to_file(path)
    g2 = gpr.GPRAS.from_file(path)
    assert g2.kernel_str == "RBF" and len(g2.models) == 2
    assert np.array_equal(g2.models[1].inducing_variable.Z, g.models[1].inducing_variable.Z)
    assert g2.models[0].inducing_variable.Z.shape == (12, 4)  # what pipeline.py:115 reads
    m1, v1 = g.predict(x[:30])
    m2, v2 = g2.predict(x[:30])
    assert m1.shape == (30, 2) and np.allclose(m1, m2, rtol=1e-12, atol=1e-14) and np.allclose(v1, v2, rtol=1e-12)
    assert x.dtype == np.float64 and g.x is not x  # inputs are copied by the float64 cast, never retained
    with open(tmp_path / "foreign.bin", "wb") as f:
        import pickle

        pickle.dump({"kernel": "RBF"}, f)
    with pytest.raises(ValueError):
        gpr.GPRAS.from_file(tmp_path / "foreign.bin")


class BatchedOracleBackend(OracleBackend):
    """The stand-in with the batched entry point of the HIP engine (a loop here): exercises the lock-step driver."""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.batch_sizes = []

    def objective_batch(self, units, thetas, mask, want_grad=True, zs=None):
        self.batch_sizes.append(len(units))
        zs = [None] * len(units) if zs is None else zs
        out = [self.objective(u, t, z, mask, want_grad) for u, t, z in zip(units, thetas, zs)]
        losses = np.array([o[0] for o in out])
        grads = np.stack([o[1] for o in out]) if want_grad else None
        return losses, grads, np.isfinite(losses)


@pytest.mark.parametrize("method,kwargs", [("L-BFGS-B", {"max_iter": 6}), ("adam", {"max_iter": 4}), ("two-stage", {"max_iter": 3})])
def test_lockstep_fit_equals_serial_loop(monkeypatch, method, kwargs):
    """K optimiser drivers in K threads, every round of evaluations as one batch: parameters bit-identical to the
    serial per-mode loop, although the modes need different numbers of evaluations."""
    x, y, _ = make_regression(60, 3, n_outputs=4, n_test=0, config=8, unit=0)
    monkeypatch.setattr(gpr, "Engine", BatchedOracleBackend)
    a = gpr.GPRAS("Matern32")
    a.fit(x, y, None, optimization_method=method, **kwargs)
    assert a.lockstep_stats["batches"] > 0 and max(a.engine.batch_sizes) == 4
    assert a.lockstep_stats["evaluations"] == sum(m.n_evals for m in a.models)
    b = gpr.GPRAS("Matern32")
    b.fit(x, y, None, optimization_method=method, lockstep=False, **kwargs)
    for ma, mb in zip(a.models, b.models):
        assert ma.variance == mb.variance and ma.noise == mb.noise and np.array_equal(ma.lengthscales, mb.lengthscales)
        assert ma.n_evals == mb.n_evals and ma.backend is a.engine


def test_lockstep_fit_of_sparse_models_equals_serial_loop(monkeypatch):
    """The reference's default path (SGPR, two-stage Adam): lock step over the modes, Z and hyperparameters per mode."""
    x, y, _ = make_regression(70, 3, n_outputs=3, n_test=0, config=8, unit=5)
    monkeypatch.setattr(gpr, "Engine", BatchedOracleBackend)
    a = gpr.GPRAS("RBF")
    a.fit(x, y, 7, "grid", "two-stage", max_iter=4)
    assert max(a.engine.batch_sizes) == 3 and a.lockstep_stats["evaluations"] == sum(m.n_evals for m in a.models)
    b = gpr.GPRAS("RBF")
    b.fit(x, y, 7, "grid", "two-stage", max_iter=4, lockstep=False)
    for ma, mb in zip(a.models, b.models):
        assert ma.variance == mb.variance and ma.noise == mb.noise and ma.lengthscales == mb.lengthscales
        assert np.array_equal(ma.Z, mb.Z)


def test_lockstep_propagates_failures_without_deadlock(monkeypatch):
    class Failing(BatchedOracleBackend):
        def objective(self, unit, theta, z, mask, want_grad=True):
            if unit == 2:
                raise np.linalg.LinAlgError("unit 2 fails")
            return super().objective(unit, theta, z, mask, want_grad)

        def objective_batch(self, units, thetas, mask, want_grad=True, zs=None):
            losses, grads = [], []
            for u, t in zip(units, thetas):
                if u == 2:
                    losses.append(np.nan)
                    grads.append(np.full(self.n_theta, np.nan))
                else:
                    lo, g = super().objective(u, t, None, mask, True)
                    losses.append(lo)
                    grads.append(g)
            losses = np.array(losses)
            return losses, (np.stack(grads) if want_grad else None), np.isfinite(losses)

    x, y, _ = make_regression(40, 2, n_outputs=3, n_test=0, config=8, unit=1)
    monkeypatch.setattr(gpr, "Engine", Failing)
    g = gpr.GPRAS("RBF")
    with pytest.raises(np.linalg.LinAlgError):
        g.fit(x, y, None, optimization_method="adam", max_iter=3)


def test_bench_launcher_counts_a_signalled_rank_as_a_failure():
    """Popen.wait() is negative for a rank killed by a signal; the launcher's exit code must not be max() of the raw codes."""
    import importlib.util
    import os

    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert bench.worst_code([0, 0]) == 0
    assert bench.worst_code([0, -6]) == 134 and bench.worst_code([0, -11, 3]) == 139
    assert bench.worst_code([1, 124, 0]) == 124
