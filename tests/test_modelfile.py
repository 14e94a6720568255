"""Row N2: portable model files and the best-effort reader of files written by the reference
(``/root/reference/gpras/gpr.py:344-384``).  CPU only: the GP engine is the oracle-backed stand-in of test_host_logic."""

import functools
import json
import pickle
import sys
import types

import numpy as np
import pytest

from gpras_amd import gpr, modelfile
from gpras_amd.synth import make_hydrograph_features
from test_host_logic import OracleBackend


@pytest.fixture
def fitted(monkeypatch):
    monkeypatch.setattr(gpr, "Engine", OracleBackend)
    x, y = make_hydrograph_features(120, 3, n_outputs=2, config=1, unit=4)
    g = gpr.GPRAS("Matern32")
    g.fit(x, y, 9, "kmeans", "adam", max_iter=3)
    return g, x


@pytest.mark.parametrize("name", ["gpr.pkl", "gpr.npz", "gpr.json", "gpr_model"])
def test_every_container_round_trips_bit_for_bit(fitted, tmp_path, name):
    g, x = fitted
    path = tmp_path / name
    g.to_file(path)
    head = open(path, "rb").read(2)
    assert (head == b"PK") == name.endswith(".npz") and (head[:1] == b"{") == name.endswith(".json")
    g2 = gpr.GPRAS.from_file(path)
    assert g2.kernel_str == g.kernel_str and len(g2.models) == len(g.models) and g2.ard == g.ard
    assert np.array_equal(g2.x, g.x) and np.array_equal(g2.y, g.y)
    for a, b in zip(g.models, g2.models):
        assert a.variance == b.variance and a.lengthscales == b.lengthscales and a.noise == b.noise and np.array_equal(a.Z, b.Z)
        assert np.array_equal(a.theta(), b.theta())
    m1, v1 = g.predict(x[:20])
    m2, v2 = g2.predict(x[:20])
    assert np.array_equal(m1, m2) and np.array_equal(v1, v2)


def test_portable_containers_hold_no_pickle(fitted, tmp_path):
    g, _ = fitted
    g.to_file(tmp_path / "m.npz")
    with np.load(tmp_path / "m.npz", allow_pickle=False) as z:  # would raise on object arrays
        meta = json.loads(str(z["meta"]))
        assert meta["kernel"] == "Matern32" and meta["n_inducing"] == 9 and meta["n_models"] == 2
        assert z["models.inducing_variable.Z"].shape == (2, 9, 3) and z["models.kernel.variance"].shape == (2,)
    g.to_file(tmp_path / "m.json")
    d = json.load(open(tmp_path / "m.json"))
    assert set(d["models"][0]) == set(modelfile.PARAM_KEYS) and d["format"] == modelfile.FILE_FORMAT


def test_exact_model_files_have_no_inducing_points(monkeypatch, tmp_path):
    monkeypatch.setattr(gpr, "Engine", OracleBackend)
    x, y = make_hydrograph_features(60, 2, n_outputs=1, config=1, unit=5)
    g = gpr.GPRAS("RBF")
    g.fit(x, y, None, optimization_method="adam", ard=True, max_iter=2)
    for name in ("e.pkl", "e.npz", "e.json"):
        g.to_file(tmp_path / name)
        g2 = gpr.GPRAS.from_file(tmp_path / name)
        assert g2.models[0].Z is None and g2.ard and np.array_equal(g2.models[0].lengthscales, g.models[0].lengthscales)


def _reference_shaped_pickle(path, x, y, n_inducing, params, kernel="RBF"):
    """A file laid out like the reference's (gpr.py:359-366): the parameter dicts hold objects of classes that live in
    ``gpflow`` / ``tensorflow`` modules.  SYNTHETIC: gpflow is not installed here, so the objects are built from stand-in
    modules registered only while pickling; their shape follows TensorFlow's ``ResourceVariable.__reduce__`` (a
    ``functools.partial(ResourceVariable, initial_value=<unconstrained array>, ...)``) inside a gpflow ``Parameter``'s state."""
    tf_mod = types.ModuleType("tensorflow.python.ops.resource_variable_ops")
    gp_mod = types.ModuleType("gpflow.base")

    class ResourceVariable:
        def __init__(self, initial_value=None, trainable=True, name=None, dtype=None):
            self.initial_value = initial_value

        def __reduce__(self):
            return functools.partial(ResourceVariable, initial_value=self.initial_value, trainable=True, name="v", dtype="float64"), ()

    class Parameter:
        def __init__(self, unconstrained):
            self._pretransformed_input = ResourceVariable(unconstrained)
            self.prior_on = "constrained"

    ResourceVariable.__module__, ResourceVariable.__qualname__ = tf_mod.__name__, "ResourceVariable"
    Parameter.__module__, Parameter.__qualname__ = gp_mod.__name__, "Parameter"
    tf_mod.ResourceVariable, gp_mod.Parameter = ResourceVariable, Parameter
    added = {}
    for mod in (tf_mod, gp_mod):
        parts = mod.__name__.split(".")
        for i in range(1, len(parts) + 1):
            name = ".".join(parts[:i])
            if name not in sys.modules:
                added[name] = sys.modules[name] = mod if i == len(parts) else types.ModuleType(name)
    try:
        d = {"kernel": kernel, "data": {"x": x, "y": y}, "n_inducing": n_inducing,
             "models": [{k: Parameter(v) for k, v in p.items()} for p in params]}
        with open(path, "wb") as f:
            pickle.dump(d, f)
    finally:
        for name in added:
            del sys.modules[name]


def test_reference_written_file_is_read_without_gpflow(monkeypatch, tmp_path):
    monkeypatch.setattr(gpr, "Engine", OracleBackend)
    rng = np.random.default_rng(3)
    x, y = rng.standard_normal((40, 2)), rng.standard_normal((40, 2))
    unconstrained = [{".kernel.variance": np.array(0.3 + i), ".kernel.lengthscales": np.array(-0.2), ".likelihood.variance": np.array(-1.0),
                      ".inducing_variable.Z": rng.standard_normal((5, 2))} for i in range(2)]
    path = tmp_path / "reference_gpr.pkl"
    _reference_shaped_pickle(path, x, y, 5, unconstrained)
    assert "gpflow" not in sys.modules
    with pytest.raises(ModuleNotFoundError):  # plain pickle cannot read it here: that is the problem N2 names
        pickle.load(open(path, "rb"))
    g = gpr.GPRAS.from_file(path)
    assert g.kernel_str == "RBF" and len(g.models) == 2 and np.array_equal(g.x, x)
    for m, u in zip(g.models, unconstrained):
        assert m.variance == pytest.approx(np.log1p(np.exp(u[".kernel.variance"])), rel=1e-15)
        assert m.lengthscales == pytest.approx(np.log1p(np.exp(-0.2)), rel=1e-15)
        assert m.noise == pytest.approx(1e-6 + np.log1p(np.exp(-1.0)), rel=1e-15)
        assert np.allclose(m.Z, u[".inducing_variable.Z"], rtol=0, atol=0)


@pytest.mark.parametrize("kernel, want", [("Matern12", "expanded"), ("Exponential", "expanded"), ("Matern52", "difference")])
def test_reference_written_file_loads_with_gpflows_distance_form(monkeypatch, tmp_path, kernel, want):
    """a model trained by gpflow and loaded through from_file must predict with gpflow's arithmetic where that matters
    (Matern12 / Exponential: the difference form is 1.2e-8 / 2.5e-8 away) -- a file that carries no ``distance_form`` gets the
    per-kernel default of ``GPRAS(kernel)``, not a blanket "difference"."""
    monkeypatch.setattr(gpr, "Engine", OracleBackend)
    rng = np.random.default_rng(4)
    x, y = rng.standard_normal((30, 2)), rng.standard_normal((30, 1))
    params = [{".kernel.variance": np.array(0.1), ".kernel.lengthscales": np.array(0.2), ".likelihood.variance": np.array(-2.0),
               ".inducing_variable.Z": rng.standard_normal((4, 2))}]
    path = tmp_path / "reference_gpr.pkl"
    _reference_shaped_pickle(path, x, y, 4, params, kernel=kernel)
    assert modelfile.load(path)["distance_form"] is None
    g = gpr.GPRAS.from_file(path)
    assert g.distance_form == want == gpr.GPRAS(kernel).distance_form
    # a file of this package written before the key existed loads the same way
    d = modelfile.model_dict(g)
    del d["distance_form"]
    with open(tmp_path / "old.pkl", "wb") as f:
        pickle.dump(d, f)
    assert gpr.GPRAS.from_file(tmp_path / "old.pkl").distance_form == want
    # and an explicit choice survives a round trip
    g2 = gpr.GPRAS(kernel, distance_form="difference")
    g2.fit(x, y, 4, "grid", "adam", max_iter=1)
    g2.to_file(tmp_path / "explicit.npz")
    assert gpr.GPRAS.from_file(tmp_path / "explicit.npz").distance_form == "difference"


def test_unrecoverable_parameters_fail_with_a_clear_message(tmp_path):
    x = np.zeros((4, 2))
    with open(tmp_path / "odd.pkl", "wb") as f:
        pickle.dump({"kernel": "RBF", "data": {"x": x, "y": x}, "n_inducing": 2, "models": [{".kernel.variance": "not a number"}]}, f)
    with pytest.raises(ValueError, match=r"written by the reference.*Recovered without gpflow.*\.kernel\.variance"):
        modelfile.load(tmp_path / "odd.pkl")
    with open(tmp_path / "junk.bin", "wb") as f:
        f.write(b"\x00\x01junk")
    with pytest.raises(ValueError):
        modelfile.load(tmp_path / "junk.bin")


def test_pickle_reader_refuses_globals_outside_the_allow_list(tmp_path):
    """the reader of reference pickles resolves numpy reconstruction, plain containers and functools.partial only --
    a file that names os.system (or anything else) is refused instead of executed."""
    import pickle

    import pytest

    class Evil:
        def __reduce__(self):
            import os

            return (os.system, ("echo pwned > /dev/null",))

    path = tmp_path / "gpr.pkl"
    with open(path, "wb") as f:
        pickle.dump({"kernel": "RBF", "models": [Evil()]}, f)
    with pytest.raises(ValueError, match="not allowed"):
        modelfile.load(str(path))


def test_multiple_assign_prefers_edited_constrained_values():
    """parameter_dict -> edit -> multiple_assign (the reference's workflow, gpr.py:363, :383): a stale ``.unconstrained`` entry
    must not override the edited values; an untouched dictionary still reloads bit-identical variables."""
    import numpy as np

    from gpras_amd.model import GPModel

    class FakeEngine:
        n_len, d, ard = 1, 3, False

    m = GPModel(FakeEngine(), 0, None, 1.3, 0.7, 0.2)
    params = m.parameter_dict()
    w_before = m.theta().copy()
    m2 = GPModel(FakeEngine(), 0, None, 1.0, 1.0, 1.0)
    m2.multiple_assign(params)
    assert np.array_equal(m2.theta(), w_before)  # untouched: the unconstrained variables come back bit for bit
    params[".kernel.variance"] = np.array(2.5)  # edited, ".unconstrained" left stale
    m3 = GPModel(FakeEngine(), 0, None, 1.0, 1.0, 1.0)
    m3.multiple_assign(params)
    assert abs(m3.variance - 2.5) < 1e-12 and abs(m3.noise - 0.2) < 1e-12
