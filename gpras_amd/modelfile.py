"""Model files -- SURVEY.md section 8(f) row N2 (``/root/reference/gpras/gpr.py:344-384``).

The reference pickles ``{"kernel", "data": {"x", "y"}, "n_inducing", "models": [gpflow parameter_dict, ...]}``; the values of
the parameter dicts are gpflow ``Parameter`` objects, so such a file cannot be read without gpflow.  Here the same
fields travel in three containers, chosen by the file suffix (``GPRAS.to_file``) and recognised by content (``from_file``):

* ``*.npz``  -- portable: numpy arrays + one JSON string ``meta`` (no pickle: ``allow_pickle=False`` on load);
* ``*.json`` -- portable text: every array as nested lists (``repr`` round-trips float64 exactly);
* anything else -- a pickle of the same dict with plain arrays under gpflow's parameter-dict keys (the reference's layout,
  loadable without gpflow).

``load`` also reads files WRITTEN BY THE REFERENCE as far as that is possible without gpflow: ``kernel``, ``data`` and
``n_inducing`` are plain Python / numpy and always come back; the per-mode parameters are gpflow ``Parameter`` objects,
which are unpickled into inert stand-in records (no gpflow / tensorflow code runs), and their numbers are recovered from
the ``initial_value`` that TensorFlow's variables record when pickled (the UNCONSTRAINED value; gpflow's default transforms of
gpr.py:298-305 are then applied: softplus for kernel variance / lengthscales, 1e-6 + softplus for the likelihood variance,
identity for Z).  If a parameter cannot be recovered the error names it and says what was recovered.
"""

from __future__ import annotations

import io
import json
import pickle
from pathlib import Path
from typing import Any

import numpy as np

FILE_FORMAT = "gpras_amd-1"
PARAM_KEYS = (".kernel.variance", ".kernel.lengthscales", ".likelihood.variance", ".inducing_variable.Z", ".unconstrained")
_FOREIGN = ("gpflow", "tensorflow", "tensorflow_probability", "tf_keras", "keras", "check_shapes")


def model_dict(gpras) -> dict[str, Any]:
    """The reference's dictionary (gpr.py:359-364) with plain arrays, plus the extension fields."""
    z0 = gpras.models[0].Z
    return {
        "format": FILE_FORMAT,
        "kernel": gpras.kernel_str,
        "data": {"x": gpras.x, "y": gpras.y},
        "n_inducing": None if z0 is None else int(z0.shape[0]),
        "ard": bool(gpras.ard),
        "distance_form": getattr(gpras, "distance_form", "difference"),
        "models": [m.parameter_dict() for m in gpras.models],
    }


# ---- writers -------------------------------------------------------------------------------------------------------------
def save(d: dict[str, Any], path: str | Path) -> None:
    path = Path(path)
    suffix = path.suffix.lower()
    if suffix == ".npz":
        _save_npz(d, path)
    elif suffix == ".json":
        _save_json(d, path)
    else:
        with open(path, mode="wb") as f:
            pickle.dump(d, f)


def _meta(d):
    return {k: d[k] for k in ("format", "kernel", "n_inducing", "ard", "distance_form")} | {"n_models": len(d["models"])}


def _save_npz(d, path):
    arrays = {"x": np.asarray(d["data"]["x"], dtype=np.float64), "y": np.asarray(d["data"]["y"], dtype=np.float64)}
    for key in PARAM_KEYS:
        if all(key in m for m in d["models"]):
            arrays["models" + key] = np.stack([np.asarray(m[key], dtype=np.float64) for m in d["models"]])
    arrays["meta"] = np.array(json.dumps(_meta(d)))
    with open(path, "wb") as f:  # (np.savez would append ".npz" to other names; the caller's path is kept as given)
        np.savez(f, **arrays)


def _save_json(d, path):
    out = _meta(d)
    out["data"] = {"x": np.asarray(d["data"]["x"], dtype=np.float64).tolist(), "y": np.asarray(d["data"]["y"], dtype=np.float64).tolist()}
    out["models"] = [{k: np.asarray(v, dtype=np.float64).tolist() for k, v in m.items()} for m in d["models"]]
    with open(path, "w") as f:
        json.dump(out, f)


# ---- readers -------------------------------------------------------------------------------------------------------------
def load(path: str | Path) -> dict[str, Any]:
    """Any of the three containers, or a file written by the reference (see the module docstring)."""
    with open(path, "rb") as f:
        head = f.read(4)
    if head[:2] == b"PK":
        return _load_npz(path)
    if head.lstrip()[:1] == b"{":
        return _load_json(path)
    return _load_pickle(path)


def _load_npz(path):
    with np.load(path, allow_pickle=False) as z:
        meta = json.loads(str(z["meta"]))
        if meta.get("format") != FILE_FORMAT:
            raise ValueError(f"{path}: not a gpras_amd model file (format {meta.get('format')!r})")
        models = []
        for i in range(meta["n_models"]):
            models.append({key: np.array(z["models" + key][i]) for key in PARAM_KEYS if "models" + key in z.files})
        return {**{k: meta.get(k) for k in ("format", "kernel", "n_inducing", "ard", "distance_form")}, "data": {"x": z["x"], "y": z["y"]},
                "models": models}


def _load_json(path):
    with open(path) as f:
        d = json.load(f)
    if d.get("format") != FILE_FORMAT:
        raise ValueError(f"{path}: not a gpras_amd model file (format {d.get('format')!r})")
    d["data"] = {"x": np.asarray(d["data"]["x"], dtype=np.float64), "y": np.asarray(d["data"]["y"], dtype=np.float64)}
    d["models"] = [{k: np.asarray(v, dtype=np.float64) for k, v in m.items()} for m in d["models"]]
    return d


class _Record:
    """Inert stand-in for an object of a package that is not installed: remembers how it was built, runs nothing."""

    def __new__(cls, *args, **kwargs):  # (pickle builds most objects with cls.__new__(cls) and never calls __init__)
        obj = object.__new__(cls)
        obj.args, obj.kwargs, obj.state, obj.items = args, kwargs, None, []
        return obj

    def __init__(self, *args, **kwargs):
        self.args, self.kwargs = args, kwargs

    def __setstate__(self, state):
        self.state = state

    def __call__(self, *args, **kwargs):  # a stand-in used as a factory (functools.partial(Class, ...)())
        return _Record(*args, **kwargs)

    def append(self, item):
        self.items.append(item)

    def extend(self, items):
        self.items.extend(items)

    def __setitem__(self, key, value):
        self.items.append((key, value))


# the only real classes / functions a model pickle may name: numpy's array reconstruction, plain containers, functools.partial
# (TensorFlow pickles a variable as partial(ResourceVariable, ...)).  Everything else raises -- os.system, builtins.eval and the
# like never resolve, so loading a file cannot run code of the file's choosing.
_ALLOWED_GLOBALS = {
    ("numpy", "ndarray"), ("numpy", "dtype"), ("numpy", "float64"), ("numpy", "int64"), ("numpy", "bool_"),
    ("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
    ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"),
    ("numpy.core.numeric", "_frombuffer"), ("numpy._core.numeric", "_frombuffer"),
    ("collections", "OrderedDict"), ("functools", "partial"),
    ("builtins", "dict"), ("builtins", "list"), ("builtins", "tuple"), ("builtins", "set"), ("builtins", "frozenset"),
    ("builtins", "int"), ("builtins", "float"), ("builtins", "complex"), ("builtins", "str"), ("builtins", "bytes"), ("builtins", "bool"),
    ("builtins", "slice"), ("builtins", "range"),
}


class _ReferenceUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module.split(".")[0] in _FOREIGN:
            return type(name, (_Record,), {"__module__": module})
        if (module, name) in _ALLOWED_GLOBALS:
            return super().find_class(module, name)
        raise pickle.UnpicklingError(f"global {module}.{name} is not allowed in a model file")


def _find_value(obj, depth=0):
    """The numeric payload of a stand-in: TensorFlow pickles a variable as partial(ResourceVariable, initial_value=<array>, ...)."""
    if depth > 12:
        return None
    if isinstance(obj, np.ndarray) and obj.dtype.kind == "f":
        return obj
    if isinstance(obj, (float, np.floating)):
        return np.asarray(float(obj))
    if isinstance(obj, _Record):
        if "initial_value" in obj.kwargs:
            found = _find_value(obj.kwargs["initial_value"], depth + 1)
            if found is not None:
                return found
        for part in (obj.kwargs, obj.state, obj.args, obj.items):
            found = _find_value(part, depth + 1)
            if found is not None:
                return found
        return None
    if isinstance(obj, dict):
        for key in ("initial_value", "_pretransformed_input", "_unconstrained", "value"):
            if key in obj:
                found = _find_value(obj[key], depth + 1)
                if found is not None:
                    return found
        for v in obj.values():
            found = _find_value(v, depth + 1)
            if found is not None:
                return found
        return None
    if isinstance(obj, (list, tuple)):
        for v in obj:
            found = _find_value(v, depth + 1)
            if found is not None:
                return found
    return None


def _softplus(w):
    return np.logaddexp(0.0, np.asarray(w, dtype=np.float64))


def _load_pickle(path):
    with open(path, "rb") as f:
        raw = f.read()
    try:
        d = _ReferenceUnpickler(io.BytesIO(raw)).load()
    except Exception as exc:  # noqa: BLE001
        raise ValueError(f"{path}: neither a gpras_amd model file nor a readable reference pickle ({type(exc).__name__}: {exc})") from exc
    if not isinstance(d, dict) or "kernel" not in d or "models" not in d:
        raise ValueError(f"{path}: not a gpras model file (expected the dictionary of gpr.py:359-364)")
    if d.get("format") == FILE_FORMAT:
        return d
    # written by the reference: kernel / data / n_inducing are plain; the parameter dicts hold gpflow Parameter objects
    recovered = f"kernel={d['kernel']!r}, data x{np.shape(d['data']['x'])} y{np.shape(d['data']['y'])}, n_inducing={d.get('n_inducing')}"
    transforms = {".kernel.variance": _softplus, ".kernel.lengthscales": _softplus, ".likelihood.variance": lambda w: 1e-6 + _softplus(w),
                  ".inducing_variable.Z": lambda w: np.asarray(w, dtype=np.float64)}
    models = []
    for i, params in enumerate(d["models"]):
        out = {}
        for key, to_constrained in transforms.items():
            value = params.get(key) if isinstance(params, dict) else None
            if isinstance(value, np.ndarray) or isinstance(value, (float, np.floating)):
                out[key] = np.asarray(value, dtype=np.float64)  # already a plain (constrained) value
                continue
            unconstrained = _find_value(value)
            if unconstrained is None:
                raise ValueError(
                    f"{path}: a model file written by the reference -- its parameters are pickled gpflow Parameter objects "
                    f"(gpr.py:363), which need gpflow to load.  Recovered without gpflow: {recovered}; could not recover "
                    f"{key!r} of model {i}.  Re-save it where gpflow is installed, e.g. as plain arrays under the same keys."
                )
            out[key] = to_constrained(unconstrained)
        models.append(out)
    return {"format": "reference", "kernel": d["kernel"], "data": d["data"], "n_inducing": d.get("n_inducing"), "ard": False,
            "distance_form": None, "models": models}  # (None: GPRAS picks gpflow's form where it matters -- the file came from gpflow)
