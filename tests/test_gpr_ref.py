"""The parts of ``/root/reference/gpras/gpr.py`` that are NOT gpflow, pinned by outputs of the reference itself
(``tests/golden/gpr_ref_golden.npz``, written by ``tests/golden/make_golden_gpr_ref.py`` in the build container): registries and public
literals (gpr.py:21-41, 206-214), ``GPRAS.__init__`` (:220-235), ``_create_inducing`` (:310-320) and the call signatures of ``fit`` and the
optimiser drivers.  CPU: the host mirror and the oracle against the fixture (registries exact, "grid" bit for bit, the oracle's k-means --
the same scikit-learn call -- to rounding).  GPU: the device k-means initialisation against the reference's centres (1e-12)."""

import inspect
import json
import os
import sys
import typing

import numpy as np
import pytest

from gpras_amd import gpr, optimizers
from oracle import gpras_oracle

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from make_golden_gpr_ref import gpr_ref_cases  # noqa: E402  (pure numpy: the seeded inputs; the reference is not imported)


@pytest.fixture(scope="module")
def golden():
    z = np.load(os.path.join(HERE, "golden", "gpr_ref_golden.npz"))
    return z, json.loads(str(z["meta_json"]))


def test_fixture_was_recorded_without_touching_a_stand_in(golden):
    _, meta = golden
    assert meta["reference_file"] == "gpras/gpr.py" and {"gpflow", "tensorflow", "tensorflow_probability"} <= {m.split(".")[0] for m in meta["inert_modules"]}


def test_registries_and_literals_equal_the_references(golden):
    _, meta = golden
    assert sorted(gpr.KERNEL_FACTORY) == meta["kernel_factory_keys"] and list(gpr.KERNEL_FACTORY) == meta["kernel_factory_key_order"]
    assert [k for k in gpr.KERNEL_FACTORY if gpr.GPRAS(k).kernel_str == k] == meta["constructible_names"]  # every listed name constructs
    assert sorted(optimizers.OPTIMIZERS) == meta["optimizer_keys"] and list(optimizers.OPTIMIZERS) == meta["optimizer_key_order"]
    assert list(typing.get_args(gpr.KernelType)) == meta["KernelType"]
    assert list(typing.get_args(gpr.OptimizerType)) == meta["OptimizerType"]  # incl. the misspelt "diffential_evolution", without "three-stage" / "adadelta"
    assert list(typing.get_args(gpr.InductionInitializerType)) == meta["InductionInitializerType"]
    assert {k: v.__name__ for k, v in optimizers.OPTIMIZERS.items()} == meta["optimizer_function_names"]
    assert sorted(gpras_oracle.OPTIMIZERS) == meta["optimizer_keys"]


def _params(fn):
    return [[n, p.kind.name, None if p.default is inspect.Parameter.empty else p.default, p.default is not inspect.Parameter.empty]
            for n, p in inspect.signature(fn).parameters.items()]


def test_driver_signatures_keep_the_references_names_and_defaults(golden):
    """Every parameter of the reference's drivers exists here under the same name, in the same position, with the same default (quirk 4:
    ``_optimize_adam`` / ``_optimize_bfgs`` have NO default for max_iter); extensions only come after them."""
    _, meta = golden
    for name, ref_sig in meta["signatures"].items():
        if name.startswith("GPRAS."):
            continue
        ours = _params(getattr(optimizers, name))
        assert ours[: len(ref_sig)] == ref_sig, (name, ours, ref_sig)
        assert all(p[3] for p in ours[len(ref_sig):]), name  # (anything added has a default)


def test_gpras_method_signatures_keep_the_references_surface(golden):
    _, meta = golden
    for name in ("__init__", "fit", "_create_inducing", "predict", "to_file", "from_file"):
        ref_sig = meta["signatures"][f"GPRAS.{name}"]
        fn = getattr(gpr.GPRAS, name)
        ours = _params(fn.__func__ if inspect.ismethod(fn) else fn)
        ours_named = {p[0]: p for p in ours}
        positional = [p for p in ref_sig if p[1] == "POSITIONAL_OR_KEYWORD"]
        assert [p[0] for p in ours[: len(positional)]] == [p[0] for p in positional], name  # same order for positional use
        for p in ref_sig:  # same kind and default for every reference parameter (**optimizer_kwargs is called **opt_kwargs here: VAR_KEYWORD both)
            if p[1] == "VAR_KEYWORD":
                assert any(q[1] == "VAR_KEYWORD" for q in ours), name
            else:
                assert ours_named[p[0]][1:] == p[1:], (name, p, ours_named[p[0]])


def test_construction_as_the_reference(golden, monkeypatch):
    _, meta = golden
    g = gpr.GPRAS("RBF")
    init = meta["init_attributes"]
    assert g.kernel_str == init["kernel_str"] and g.models == init["models"] and (g.x is None) == init["x_is_none"] and (g.y is None) == init["y_is_none"]
    assert set(init["attribute_names"]) <= set(vars(g))
    assert meta["unknown_kernel_raises"] == "KeyError"
    with pytest.raises(KeyError):
        gpr.GPRAS("NoSuchKernel")
    with pytest.raises(KeyError):
        gpras_oracle.GPRASOracle("NoSuchKernel")


def test_grid_inducing_points_bit_for_bit(golden):
    z, meta = golden
    g = gpr.GPRAS.__new__(gpr.GPRAS)
    for name, (x, m) in gpr_ref_cases().items():
        want = z[f"{name}/grid"]
        got = g._create_inducing(x, m, "grid")
        assert got.dtype == np.float64 == np.dtype(meta["inducing_dtypes"][f"{name}/grid"]) and got.shape == want.shape == (m, x.shape[1])
        assert np.array_equal(got, want), name
        assert np.array_equal(gpras_oracle.create_inducing(x, m, "grid"), want), name
    assert meta["unknown_initializer_returns"] == "None"  # the reference falls through; this package raises instead (documented deviation)
    with pytest.raises(ValueError):
        g._create_inducing(np.zeros((4, 2)), 2, "nonsense")


def test_oracle_kmeans_is_the_references_call(golden):
    z, _ = golden
    for name, (x, m) in gpr_ref_cases().items():
        got = gpras_oracle.create_inducing(x, m, "kmeans")
        assert np.allclose(got, z[f"{name}/kmeans"], rtol=1e-13, atol=1e-13), name


@pytest.mark.gpu
def test_device_kmeans_initialisation_equals_the_references_centres(golden):
    z, _ = golden
    g = gpr.GPRAS("RBF")
    for name, (x, m) in gpr_ref_cases().items():
        want = z[f"{name}/kmeans"]
        got = g._create_inducing(x, m, "kmeans")
        assert got.shape == want.shape and got.dtype == np.float64
        assert np.max(np.abs(got - want)) <= 1e-12 * max(1.0, np.max(np.abs(want))), name
