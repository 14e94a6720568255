"""Generate tests/golden/pca_ref_golden.npz FROM THE REFERENCE ITSELF.

The three functions pinned here -- ``PreProcessor.transform`` (/root/reference/gpras/preprocess.py:1009-1038, with
``wse_2_depth`` :1040-1044), ``reverse_transform`` (:1052-1085) and ``_linear_transform_for_var`` (:1087-1094) -- are plain
numpy, but ``import gpras.preprocess`` fails in the build container on ordinary absent imports (``typing.Self`` needs
Python 3.11; geopandas / pyproj / rasterio / shapely / hecdss are not installed; ``gpras.gpr`` pulls gpflow / tensorflow /
tensorflow_probability; ``gpras.ras`` / ``gpras.utils`` pull h5py, rashdf, ...).  This script therefore

  (i)   appends a LAST-RESORT meta-path finder that hands out inert recording modules for those top-level packages which the
        reference's own source files import (found by parsing /root/reference/gpras/**/*.py) and which are not installed
        (their names are written into the fixture), and gives ``typing`` a ``Self`` attribute;
  (ii)  imports ``gpras.preprocess`` from /root/reference and builds ``PreProcessor`` objects through the reference's own
        constructor (:869-924) with seeded fitted attributes -- wse / depth / velocity x weighted / unweighted, with dry
        cells;
  (iii) records ``transform(x)``, ``reverse_transform(mean)``, ``reverse_transform(mean, var)`` and
        ``_linear_transform_for_var``;
  (iv)  asserts that NO attribute of any inert module was touched while those calls ran -- only the reference's numpy
        statements executed.

Runs in the build container only: the reference never travels to the GPU box, the fixture (data: expected outputs; the
inputs are re-seeded by ``pca_ref_cases()`` below, which the tests import) does.

    python tests/golden/make_golden_pca_ref.py

"unweighted": ``PreProcessor(weights=None)`` stores ``np.empty(0)`` (:916), and ``x *= self.weights`` (:1030) then raises
a broadcasting ValueError -- recorded as such (``<case>/transform_raises``).  The unweighted arithmetic the reference's
``if self.weights is not None`` guards describe is reached by setting the attribute to None on the constructed object;
both states are recorded.
"""
import importlib.abc
import importlib.machinery
import json
import os
import sys
import types
import typing

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE = "/root/reference"
MODES = ("wse", "depth", "velocity")


def pca_ref_cases():
    """Seeded fitted states and inputs: name -> dict(mode, weighted, ctor kwargs, x, mean, var).  Pure numpy."""
    cases = {}
    rng = np.random.default_rng(20261004)
    for mode in MODES:
        for weighted in (True, False):
            for tag, (cells, k, t) in {"a": (53, 4, 7), "b": (130, 11, 3)}.items():
                dry = rng.random(cells) < 0.2
                dry[0], dry[-1] = True, False
                classes = np.where(dry, "AD", np.where(rng.random(cells) < 0.5, "AF", "TF"))
                n_wet = int((~dry).sum())
                elevations = 100.0 + 5.0 * rng.random(cells)
                q, _ = np.linalg.qr(rng.standard_normal((n_wet, k)))
                eofs = np.ascontiguousarray(q.T)
                weights = 0.5 + 2.0 * rng.random(n_wet) if weighted else None
                x = elevations + 1.5 * rng.standard_normal((t, cells))  # water surface on both sides of the ground
                field = np.maximum(x - elevations, 0.0) if mode == "depth" else x
                kwargs = dict(
                    spatial_mode_count=k,
                    input_mean=field[:, ~dry].mean(axis=0),
                    elevations=elevations,
                    hydraulic_parameter=mode,
                    wetness_classes=classes,
                    weights=weights,
                    eofs=eofs,
                    eigenvalues=np.sort(rng.random(n_wet))[::-1].copy(),
                    n_samples_fit=t,
                    x_mean=0.3 * rng.standard_normal(k),
                    x_std=0.5 + rng.random(k),
                )
                cases[f"{mode}_{'w' if weighted else 'u'}_{tag}"] = dict(
                    mode=mode, weighted=weighted, dry=dry, kwargs=kwargs, x=np.ascontiguousarray(x),
                    mean=rng.standard_normal((t, k)), var=0.01 + rng.random((t, k)))
    return cases


# ---- (i) inert modules for imports nothing else can satisfy -------------------------------------------------------------
TOUCHED: list[str] = []
STUBBED: list[str] = []


class _Inert:
    """Whatever an inert module hands out: callable, subscriptable, attribute-bearing -- and every use is logged."""

    def __init__(self, name):
        object.__setattr__(self, "_name", name)

    def __getattr__(self, item):
        if item.startswith("__") and item.endswith("__"):
            raise AttributeError(item)
        TOUCHED.append(f"{self._name}.{item}")
        return _Inert(f"{self._name}.{item}")

    def __call__(self, *a, **k):
        TOUCHED.append(f"{self._name}()")
        return _Inert(f"{self._name}()")

    def __getitem__(self, item):
        TOUCHED.append(f"{self._name}[]")
        return _Inert(f"{self._name}[]")

    def __mro_entries__(self, bases):  # "class X(stub.Base)" in the reference's other modules
        return (object,)

    def __or__(self, other):  # "A | None" in annotations evaluated at import time
        return typing.Any

    __ror__ = __or__


class _InertModule(types.ModuleType):
    def __getattr__(self, item):
        if item.startswith("__") and item.endswith("__"):
            raise AttributeError(item)
        TOUCHED.append(f"{self.__name__}.{item}")
        return _Inert(f"{self.__name__}.{item}")


def absent_reference_imports():
    """Top-level packages imported anywhere under /root/reference/gpras that this container cannot import."""
    import ast
    import importlib.util

    tops = set()
    for root, _, files in os.walk(os.path.join(REFERENCE, "gpras")):
        for f in files:
            if not f.endswith(".py"):
                continue
            with open(os.path.join(root, f)) as fh:
                tree = ast.parse(fh.read())
            for node in ast.walk(tree):
                if isinstance(node, ast.Import):
                    tops.update(a.name.split(".")[0] for a in node.names)
                elif isinstance(node, ast.ImportFrom) and node.level == 0 and node.module:
                    tops.add(node.module.split(".")[0])
    tops.discard("gpras")  # the reference's own modules must be the real files
    return {t for t in tops if importlib.util.find_spec(t) is None}


class _LastResortFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def __init__(self, tops):
        self.tops = tops

    def find_spec(self, fullname, path=None, target=None):
        if fullname.split(".")[0] not in self.tops:
            return None
        return importlib.machinery.ModuleSpec(fullname, self, is_package=True)

    def create_module(self, spec):
        STUBBED.append(spec.name)
        m = _InertModule(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


def import_reference_preprocess():
    if not hasattr(typing, "Self"):
        typing.Self = typing.TypeVar("Self")  # typing.Self exists from Python 3.11; only used in annotations
    absent = absent_reference_imports()
    sys.meta_path.append(_LastResortFinder(absent))  # LAST: everything that is installed resolves normally
    sys.path.insert(0, REFERENCE)
    import gpras.preprocess as ref_pre

    assert os.path.abspath(ref_pre.__file__).startswith(REFERENCE), ref_pre.__file__
    return ref_pre


def main():
    ref_pre = import_reference_preprocess()
    out = {}
    raised = {}
    cases = pca_ref_cases()
    TOUCHED.clear()  # import-time uses (HecDss.set_global_debug_level, the kernel registry of gpras.gpr, ...) are expected
    for name, c in cases.items():
        pp = ref_pre.PreProcessor(**{k: (None if v is None else (v.copy() if isinstance(v, np.ndarray) else v)) for k, v in c["kwargs"].items()})
        if not c["weighted"]:
            # state straight out of the constructor: weights == np.empty(0), the in-place multiply cannot broadcast
            try:
                pp.transform(c["x"].copy())
                raised[name] = ""
            except ValueError as e:
                raised[name] = type(e).__name__
            pp = ref_pre.PreProcessor(**{k: (None if v is None else (v.copy() if isinstance(v, np.ndarray) else v)) for k, v in c["kwargs"].items()})
            pp.weights = None  # the branch the reference's "is not None" guards describe
        assert np.array_equal(pp.dry_indices, c["dry"])
        out[f"{name}/transform"] = pp.transform(c["x"].copy())
        out[f"{name}/reverse_mean_only"] = pp.reverse_transform(c["mean"].copy())
        full, vfull = pp.reverse_transform(c["mean"].copy(), c["var"].copy())
        out[f"{name}/reverse_full"], out[f"{name}/reverse_var"] = full, vfull
        out[f"{name}/linear_transform_for_var"] = np.asarray(pp._linear_transform_for_var)
        if c["mode"] == "depth":
            out[f"{name}/wse_2_depth"] = pp.wse_2_depth(c["x"].copy())
    # (iv) nothing but the reference's own numpy ran in the recorded calls
    assert not TOUCHED, f"inert modules were used during the recorded calls: {TOUCHED[:10]}"
    meta = {
        "reference_file": "gpras/preprocess.py",
        "functions": ["PreProcessor.__init__ :869-924", "transform :1009-1038", "wse_2_depth :1040-1044", "reverse_transform :1052-1085",
                      "_linear_transform_for_var :1087-1094"],
        "inert_modules": sorted(set(STUBBED)),
        "unweighted_constructor_state_raises": raised,
        "python": sys.version.split()[0],
        "numpy": np.__version__,
    }
    out["meta_json"] = np.array(json.dumps(meta, sort_keys=True))
    path = os.path.join(HERE, "pca_ref_golden.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {len(out)} arrays from {len(cases)} cases; inert modules: {len(set(STUBBED))}; "
          f"constructor-state unweighted transform raised: {sorted(set(raised.values()))}")


if __name__ == "__main__":
    main()
