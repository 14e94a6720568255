"""Methods for GPR training, tuning, and prediction -- MI355X-native drop-in for ``gpras/gpr.py``.

Keeps the public surface of the reference class (``/root/reference/gpras/gpr.py:217-384``):
``GPRAS(kernel)``, ``fit``, ``predict``, ``to_file``, ``from_file``, attributes ``kernel_str``, ``x``, ``y``,
``models`` (with ``models[i].inducing_variable.Z``), and the registries ``KERNEL_FACTORY`` /
``OPTIMIZERS`` plus the three ``Literal`` types that ``production/analysis/data_models.py:12`` imports.

All arithmetic runs in ``libgprx.so`` (hand-written HIP for gfx950) through ``gpras_amd.engine.Engine``;
there is no CPU path.  Extensions beyond the reference: ``n_inducing=None`` fits the exact GP
(``Z = X``) and ``ard=True`` gives one lengthscale per feature.
"""

from __future__ import annotations

from pathlib import Path
from typing import Any, Literal

import numpy as np
from numpy.typing import NDArray

from . import modelfile
from ._lib import KERNEL_IDS
from .engine import Engine
from .kmeans import kmeans_centers
from .model import GPModel
from .optimizers import BATCHED_OPTIMIZERS, OPTIMIZERS

# the names that the reference maps to gpflow kernel classes, in its order (gpr.py:21-37; pinned by tests/golden/gpr_ref_golden.npz).
# Linear / Polynomial / Periodic are listed there but cannot be constructed with (variance=, lengthscales=) at gpr.py:298 ("currently
# not working" in the reference's own comments): GPRAS(name) succeeds as in the reference, and the failure comes where the reference's
# does -- when fit() builds the models.
_REFERENCE_ONLY = ("Linear", "Polynomial", "Periodic")
KERNEL_FACTORY = {name: name for name in ("Matern12", "Matern32", "Matern52", "RBF", "Linear", "Polynomial", "Periodic", "Exponential")}
assert all(name in KERNEL_IDS for name in KERNEL_FACTORY if name not in _REFERENCE_ONLY)

KernelType = Literal["Matern12", "Matern32", "Matern52", "RBF", "Linear", "Polynomial", "Periodic", "Exponential"]
OptimizerType = Literal["two-stage", "adam", "L-BFGS-B", "stochastic", "diffential_evolution"]
InductionInitializerType = Literal["kmeans", "grid"]

FILE_FORMAT = modelfile.FILE_FORMAT

# gpflow evaluates r^2 in the expanded form (utilities/ops.py square_distance, behind gpr.py:22,29,298); the kernels whose
# derivative is singular at r = 0 amplify the difference between the two forms beyond 1e-8, so they default to gpflow's form
DEFAULT_DISTANCE_FORM = {"Matern12": "expanded", "Exponential": "expanded"}


class GPRAS:
    """Gaussian Process Regression for HEC-RAS model upskilling and emulation."""

    def __init__(self, kernel: KernelType, device: int = 0, distance_form: str | None = None) -> None:
        """``distance_form`` (extension): ``"difference"`` or ``"expanded"`` -- the arithmetic form of the scaled squared
        distance inside the kernels; ``"expanded"`` is gpflow's literal ``|a|^2 + |b|^2 - 2 a.b``.
        Default (``None``): ``"expanded"`` for the two kernels that are not differentiable at r = 0 (Matern12, Exponential:
        there the difference form sits 1.2e-8 / 2.5e-8 from gpflow's arithmetic, outside the 1e-8 this drop-in promises),
        ``"difference"`` for the smooth kernels (both forms agree to 8e-15 there)."""
        self.kernel_str = kernel
        if distance_form is None:
            distance_form = DEFAULT_DISTANCE_FORM.get(kernel, "difference")
        self.distance_form = distance_form
        self.kernel = KERNEL_FACTORY[kernel]  # KeyError for unknown names, as the reference
        self.device = device
        self.models: list[GPModel] = []
        self.x: NDArray[Any] | None = None
        self.y: NDArray[Any] | None = None
        self.engine: Engine | None = None
        self.engines: list[Engine] = []
        self.ard = False

    def fit(
        self,
        x: NDArray[Any],
        y: NDArray[Any],
        n_inducing: int | None,
        inducing_initializer: InductionInitializerType = "kmeans",
        optimization_method: OptimizerType = "two-stage",
        ard: bool = False,
        workers: int = 1,
        lockstep: bool | None = None,
        **opt_kwargs: Any,
    ) -> None:
        """Fit one GP per column of ``y`` (gpr.py:237-275).

        ``lockstep`` (extension): the per-mode optimisers run in lock step, every round of evaluations as one
        batched launch sequence on the GPU (``gpras_amd.lockstep``); the fitted parameters equal the serial loop's
        bit for bit.  Default: on whenever there is more than one mode and ``workers == 1``.

        ``workers > 1`` (extension): the per-mode loop, serial in the reference (gpr.py:272-274), runs from
        that many host threads, each with its own engine (handle + HIP stream) on the same GPU, so the many
        small launches of different modes overlap on the device.  Results per mode are unchanged.
        """
        self.x = x.astype(np.float64)
        self.y = y.astype(np.float64)
        opt = OPTIMIZERS[optimization_method]  # KeyError before any device work, as the reference (gpr.py:272)
        self._init_models(self.x, self.y, n_inducing, inducing_initializer, ard, workers=workers)
        self._run_optimizers(self.models, optimization_method, lockstep, opt_kwargs)

    def _run_optimizers(self, models: list[GPModel], optimization_method: str, lockstep: bool | None, opt_kwargs: dict[str, Any]) -> None:
        """The per-mode loop of gpr.py:272-274 over ``models`` (all of them, or one rank's share): in lock step on
        batched evaluations where possible, else serially, else from ``workers`` host threads."""
        opt = OPTIMIZERS[optimization_method]
        can_lockstep = len(models) > 1 and len(self.engines) == 1 and hasattr(self.engine, "objective_batch")
        if lockstep and not can_lockstep:
            raise ValueError("lockstep fitting needs several modes on one engine (workers=1)")
        if len(models) == 1 and len(self.engines) == 1 and lockstep is None and optimization_method in BATCHED_OPTIMIZERS and hasattr(self.engine, "adam_batch"):
            # one mode with an Adam-based driver: the loop inside the library as well (sparse models with M <= 64: resident on the
            # device, no host round trip per step); same variables as the serial driver bit for bit (tests/test_gpu_gpras.py)
            stats1: dict[str, int] = {"batches": 0}
            BATCHED_OPTIMIZERS[optimization_method](models, stats=stats1, **opt_kwargs)
            return
        if can_lockstep and (lockstep is None or lockstep):
            # modes per batched launch sequence: 32, or fewer when the per-cell workspaces (kernel matrix, L^-1 and K^-1 of
            # an exact model: 5 N^2 doubles) would not fit into the free device memory
            per_batch = 32
            if hasattr(self.engine, "max_cells"):
                per_batch = max(1, min(per_batch, self.engine.max_cells(want_grad=True)))
            if optimization_method in BATCHED_OPTIMIZERS:
                # Adam-based drivers: one loop over all modes (rows of 2-D state arrays; no thread per mode), so up to 128 cells
                # per batched evaluation -- the fixed latency of an evaluation is shared by more modes (50 modes: one loop of
                # 50 instead of 32 + 18)
                if hasattr(self.engine, "max_cells"):
                    per_batch = max(1, min(128, self.engine.max_cells(want_grad=True)))
                before = sum(m.n_evals for m in models)
                stats: dict[str, int] = {"batches": 0}
                for lo in range(0, len(models), per_batch):
                    BATCHED_OPTIMIZERS[optimization_method](models[lo : lo + per_batch], stats=stats, **opt_kwargs)
                self.lockstep_stats = {"batches": stats["batches"], "evaluations": sum(m.n_evals for m in models) - before}
                return
            from .lockstep import fit_lockstep

            self.lockstep_stats = fit_lockstep(models, opt, opt_kwargs, max_batch=per_batch)
            return
        if len(self.engines) == 1:
            for model in models:
                opt(model, **opt_kwargs)
            return
        from concurrent.futures import ThreadPoolExecutor

        def run(worker: int) -> None:
            for model in models[worker :: len(self.engines)]:
                opt(model, **opt_kwargs)

        with ThreadPoolExecutor(max_workers=len(self.engines)) as pool:
            for fut in [pool.submit(run, w) for w in range(len(self.engines))]:
                fut.result()

    def _init_models(
        self,
        x: NDArray[Any],
        y: NDArray[Any],
        n_inducing: int | None,
        inducing_initializer: InductionInitializerType = "kmeans",
        ard: bool = False,
        workers: int = 1,
    ) -> None:
        """Create one model per spatial mode using base model settings (gpr.py:277-308)."""
        if self.kernel_str in _REFERENCE_ONLY:
            # the reference fails at this point too: its kernel constructor call (gpr.py:298) passes variance= and lengthscales=, which
            # gpflow's Linear / Polynomial / Periodic do not accept (TypeError there)
            raise NotImplementedError(
                f"kernel {self.kernel_str!r} is listed by the reference but cannot be built by its own fit() "
                "(gpr.py:26-27, :298); only the stationary kernels are implemented"
            )
        inducing = None if n_inducing is None else self._create_inducing(x, n_inducing, inducing_initializer)
        ini_length = np.mean(abs(x))
        self.ard = bool(ard)
        for eng in getattr(self, "engines", []):
            eng.close()
        n_eng = max(1, min(int(workers), y.shape[1]))
        m = 0 if inducing is None else inducing.shape[0]
        self.engines = [Engine(self.kernel_str, x, y, m, ard=self.ard, device=self.device, distance_form=self.distance_form) for _ in range(n_eng)]
        self.engine = self.engines[0]
        # variance 1, lengthscale mean|x| (gpr.py:289, :298); Gaussian likelihood variance 1.0 (gpflow default)
        self.models = [GPModel(self.engines[i % n_eng], i, inducing, 1.0, ini_length, 1.0) for i in range(y.shape[1])]

    def _create_inducing(self, x: NDArray[Any], n_inducing: int, method: InductionInitializerType) -> NDArray[Any]:
        """Create an array representing locations in dataspace (gpr.py:310-320)."""
        if method == "kmeans":
            # KMeans(n_clusters=n_inducing, random_state=0, n_init="auto").fit(x).cluster_centers_ (gpr.py:313-315): seeding on
            # the host as scikit-learn draws it, Lloyd iterations on the device (gpras_amd.kmeans)
            return kmeans_centers(x, n_inducing, device=self.device)
        elif method == "grid":
            inducing_variable = np.c_[np.linspace(x[:, 0].min(), x[:, 0].max(), n_inducing)]
            for j in range(1, x.shape[1]):
                inducing_variable = np.c_[inducing_variable, np.linspace(x[:, j].min(), x[:, j].max(), n_inducing)]
            return np.ascontiguousarray(inducing_variable)
        raise ValueError(f"unknown inducing initializer {method!r}")  # the reference silently returns None here

    def predict(self, x: NDArray[Any]) -> tuple[NDArray[Any], NDArray[Any]]:
        """Predictive mean and observation variance, each (n_samples, n_outputs) (gpr.py:322-342)."""
        x = np.ascontiguousarray(x, dtype=np.float64)  # (the reference's astype copy is not needed: x is neither kept nor written)
        batched = self._predict_batched(x)
        if batched is not None:
            return batched
        means = []
        variances = []
        for model in self.models:
            pred = model.predict_y(x)
            means.append(pred[0])
            variances.append(pred[1])
        return np.concatenate(means, axis=1), np.concatenate(variances, axis=1)

    @staticmethod
    def _scatter_rows(dst_mean, dst_var, part, mean, var):
        """``(cells, N*)`` results of the modes ``part`` into the columns of the ``(N*, n_outputs)`` arrays: one transposed block copy when
        the modes are consecutive (the usual case: 11 ms instead of 90 for 50 modes x 100 000 points), column by column otherwise."""
        if part == list(range(part[0], part[0] + len(part))):
            dst_mean[:, part[0] : part[0] + len(part)] = mean.T
            dst_var[:, part[0] : part[0] + len(part)] = var.T
        else:
            for row, i in enumerate(part):
                dst_mean[:, i], dst_var[:, i] = mean[row], var[row]

    def _predict_batched(self, x: NDArray[Any], indices: list[int] | None = None):
        """Exact models: the factorisations that ``predict_y`` recomputes per mode (gpr.py:337) are independent,
        so all modes of an engine are factorised by one batched launch sequence (``Engine.factorize_batch``) and
        each mode then predicts from its slot.  Same numbers as the per-mode loop (bit-identical factorisations)."""
        todo = list(range(len(self.models))) if indices is None else list(indices)
        models = [self.models[i] for i in todo]
        if not models:
            return None
        if all(m.Z is not None for m in models) and all(hasattr(m.backend, "predict_batch") for m in models) and x.shape[1] <= 64:
            return self._predict_batched_sparse(x, todo)
        if any(m.Z is not None for m in models):
            return None
        if not all(hasattr(m.backend, "factorize_batch") for m in models):
            return None
        if x.shape[1] > 64:  # the batched kernels carry at most 64 lengthscales per cell (gprx_factorize_batch)
            return None
        means = np.full((x.shape[0], len(self.models)), np.nan)  # columns outside `indices` stay NaN
        variances = np.full((x.shape[0], len(self.models)), np.nan)
        by_engine: dict[int, list[int]] = {}
        for i in todo:
            by_engine.setdefault(id(self.models[i].backend), []).append(i)
        for idx in by_engine.values():
            eng = self.models[idx[0]].backend
            chunk = eng.max_cells(want_grad=False) if hasattr(eng, "max_cells") else len(idx)  # cells that fit in HBM
            for lo in range(0, len(idx), chunk):
                part = idx[lo : lo + chunk]
                units = [self.models[i].unit for i in part]
                thetas = np.stack([self.models[i].theta() for i in part])
                if hasattr(eng, "predict_batch"):
                    # one call: batched factorisations, L^-1 of every cell by batched launches, x uploaded once (gprx_predict_batch)
                    consecutive = part == list(range(part[0], part[0] + len(part)))
                    try:
                        if consecutive and hasattr(eng, "predict_batch_t"):
                            mean_t, var_t = eng.predict_batch_t(units, thetas, x)
                            means[:, part[0] : part[0] + len(part)], variances[:, part[0] : part[0] + len(part)] = mean_t, var_t
                            continue
                        mean, var = eng.predict_batch(units, thetas, x)
                    except np.linalg.LinAlgError as exc:
                        raise RuntimeError(f"kernel matrix not positive definite for one of the modes {part}: {exc}") from exc
                    self._scatter_rows(means, variances, part, mean, var)
                    continue
                _, ok = eng.factorize_batch(units, thetas, 0)
                if not ok.all():
                    bad = [part[k] for k in np.flatnonzero(~ok)]
                    raise RuntimeError(f"kernel matrix not positive definite for mode(s) {bad}")
                for slot, i in enumerate(part):
                    eng.select_slot(slot)
                    means[:, i], variances[:, i] = eng.predict(x, include_noise=True)
        return means, variances

    def _predict_batched_sparse(self, x: NDArray[Any], todo: list[int]):
        """Sparse models (what the reference runs): all modes of an engine are factorised by one batched launch sequence and
        predicted by one batched predict (``gprx_predict_batch``: the ~30 small launches of a mode's factorise + predict_y serve
        every mode).  Same kernels and operation order per mode: the numbers equal the per-mode loop's bit for bit."""
        full = sorted(todo) == list(range(len(self.models)))  # (columns outside `todo` stay NaN)
        means = np.empty((x.shape[0], len(self.models))) if full else np.full((x.shape[0], len(self.models)), np.nan)
        variances = np.empty((x.shape[0], len(self.models))) if full else np.full((x.shape[0], len(self.models)), np.nan)
        by_engine: dict[int, list[int]] = {}
        for i in todo:
            by_engine.setdefault(id(self.models[i].backend), []).append(i)
        for idx in by_engine.values():
            eng = self.models[idx[0]].backend
            chunk = max(1, min(64, eng.max_cells(want_grad=False))) if hasattr(eng, "max_cells") else len(idx)
            for lo in range(0, len(idx), chunk):
                part = idx[lo : lo + chunk]
                units = [self.models[i].unit for i in part]
                thetas = np.stack([self.models[i].theta() for i in part])
                zs = np.stack([self.models[i].Z for i in part])
                consecutive = part == list(range(part[0], part[0] + len(part)))
                if consecutive and hasattr(eng, "predict_batch_t"):
                    mean_t, var_t = eng.predict_batch_t(units, thetas, x, zs=zs)  # (N*, modes of this part): already the layout returned
                    if len(part) == len(self.models):
                        return mean_t, var_t
                    means[:, part[0] : part[0] + len(part)], variances[:, part[0] : part[0] + len(part)] = mean_t, var_t
                    continue
                mean, var = eng.predict_batch(units, thetas, x, zs=zs)
                self._scatter_rows(means, variances, part, mean, var)
        return means, variances

    def to_file(self, json_path: str | Path, model_dir: str | Path | None = None) -> None:
        """Serialize the trained model (gpr.py:344-366).

        Same dictionary as the reference -- ``kernel``, ``data{x, y}``, ``n_inducing``, ``models`` with gpflow's
        parameter-dict keys -- with plain numpy arrays as values, so the file loads without gpflow.  The container follows
        the suffix of ``json_path`` (``gpras_amd.modelfile``): ``.npz`` (arrays + JSON metadata, no pickle), ``.json`` (text),
        anything else a pickle of the dictionary as in the reference.  ``model_dir`` is accepted and ignored, as in the reference.
        """
        modelfile.save(modelfile.model_dict(self), json_path)

    @classmethod
    def from_file(cls, json_path: str | Path, device: int = 0) -> "GPRAS":
        """Load a model written by ``to_file`` (gpr.py:368-384): re-initialise with the cheap ``"grid"`` inducing points,
        then assign every saved parameter.  The container is recognised by content; files written by the reference are read
        as far as that is possible without gpflow (``gpras_amd.modelfile.load``)."""
        d = modelfile.load(json_path)
        inst = cls(d["kernel"], device=device, distance_form=d.get("distance_form"))  # (absent / None: the per-kernel default)
        inst.x = np.asarray(d["data"]["x"], dtype=np.float64)
        inst.y = np.asarray(d["data"]["y"], dtype=np.float64)
        inst._init_models(inst.x, inst.y, d["n_inducing"], "grid", d.get("ard", False))
        for ind, params in enumerate(d["models"]):
            inst.models[ind].multiple_assign(params)
        return inst
