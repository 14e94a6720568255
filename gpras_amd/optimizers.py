"""Optimiser drivers with the names, signatures and quirks of ``/root/reference/gpras/gpr.py:44-214``.

Each driver works on a ``gpras_amd.model.GPModel``; one loss/gradient evaluation is one call into the
HIP engine.  Behaviour kept on purpose (SURVEY.md section 8a, quirk list):

1. ``stochastic`` never records ``best_loss``: the last start wins; its generator is unseeded; it replaces
   Z by a plain array, which freezes Z (gpr.py:86-109).
2. Stage 1 of ``two-stage`` / ``three-stage`` trains Z only, so its loss carries no prior term.
3. ``tol = 10e-6`` (that is 1e-5) and patience 50 in Adam's early stop.
4. ``adam`` / ``L-BFGS-B`` / ``adadelta`` have no default ``max_iter``.
5. ``three-stage`` and ``adadelta`` are registered although absent from ``OptimizerType``; the key
   ``"diffential_evolution"`` is misspelled in the reference and that spelling is the public one.
6. The differential-evolution objective evaluates the loss twice per call and prints it.
"""

from __future__ import annotations

from typing import Any

import numpy as np
from scipy.optimize import differential_evolution, minimize


def _optimize_adam(model, max_iter: int) -> None:
    """Keras ``Adam()`` defaults on the trainable variables, fresh state per call (gpr.py:147-173)."""
    lr, beta1, beta2, eps = 1e-3, 0.9, 0.999, 1e-7
    x = model.get_vector()
    if x.size == 0:  # nothing trainable (e.g. the Z-only stage of an exact model): no step can change anything
        return
    m = np.zeros_like(x)
    v = np.zeros_like(x)
    best = np.inf
    count = 0
    tol = 10e-6
    patience = 50
    for t in range(1, int(max_iter) + 1):
        loss, grad = model.loss_and_grad()
        m = beta1 * m + (1.0 - beta1) * grad
        v = beta2 * v + (1.0 - beta2) * grad * grad
        alpha = lr * np.sqrt(1.0 - beta2**t) / (1.0 - beta1**t)
        x = x - alpha * m / (np.sqrt(v) + eps)
        model.set_vector(x)
        if ((best - loss) / abs(loss)) > tol:
            best = loss
            count = 0
        else:
            count += 1
            if count > patience:
                break


def _optimize_adadelta(model, max_iter: int) -> Any:
    """Keras ``Adadelta()`` defaults, exactly ``max_iter`` steps (gpr.py:176-192).  Returns the last loss."""
    lr, rho, eps = 1e-3, 0.95, 1e-7
    x = model.get_vector()
    acc_grad = np.zeros_like(x)
    acc_delta = np.zeros_like(x)
    loss = None
    for _ in range(int(max_iter)):
        loss, grad = model.loss_and_grad()
        acc_grad = rho * acc_grad + (1.0 - rho) * grad * grad
        delta = -np.sqrt(acc_delta + eps) * grad / np.sqrt(acc_grad + eps)
        acc_delta = rho * acc_delta + (1.0 - rho) * delta * delta
        x = x + lr * delta
        model.set_vector(x)
    return loss


def _optimize_bfgs(model, max_iter: int) -> Any:
    """``gpflow.optimizers.Scipy().minimize(..., method="L-BFGS-B", options={"maxiter": max_iter})`` (gpr.py:195-203)."""

    def fun(vec):
        model.set_vector(vec)
        return model.loss_and_grad()

    res = minimize(fun, model.get_vector(), jac=True, method="L-BFGS-B", options={"maxiter": int(max_iter)})
    model.set_vector(res.x)
    return res


def _optimize_two_stage(model, max_iter: int = 100) -> Any:
    """Z first, then the hyperparameters, both with Adam (gpr.py:112-127)."""
    model.set_all_trainable(False)
    model.set_trainable(Z=True)
    _optimize_adam(model, max_iter)
    model.set_all_trainable(True)
    model.set_trainable(Z=False)
    _optimize_adam(model, max_iter)
    model.set_trainable(Z=True)
    return model.training_loss()


def _optimize_three_stage(model, max_iter: int = 100) -> None:
    """Adam on Z, L-BFGS-B on the hyperparameters, L-BFGS-B on everything (gpr.py:130-144)."""
    model.set_all_trainable(False)
    model.set_trainable(Z=True)
    _optimize_adam(model, max_iter)
    model.set_all_trainable(True)
    model.set_trainable(Z=False)
    _optimize_bfgs(model, max_iter)
    model.set_trainable(Z=True)
    _optimize_bfgs(model, max_iter)


def _optimize_multi_start(model, n_starts: int = 40, iter_initial: int = 20, iter_final: int = 1000, rng=None) -> None:
    """Random restarts + L-BFGS-B polish (gpr.py:73-109).  ``rng`` is a test hook; the reference's is unseeded."""
    np.random.seed(1)  # gpr.py:76 -- has no effect on default_rng(), kept for fidelity
    rng = np.random.default_rng() if rng is None else rng
    x = model.backend_x()
    mins, maxs = x.min(axis=0), x.max(axis=0)
    z_shape = (model.Z.shape[0], x.shape[1])
    best_params = None
    for _ in range(int(n_starts)):
        model.assign(variance=10 ** rng.uniform(-1, 1))
        model.assign(lengthscales=10 ** rng.uniform(-1, 1))
        model.assign(noise=10 ** rng.uniform(-3, 0))
        model.Z = rng.uniform(mins, maxs, size=z_shape)
        model.set_trainable(Z=False)  # gpr.py:91 swaps the Parameter for an ndarray: Z is no longer trainable
        _optimize_adam(model, iter_initial)
        model.training_loss()
        # gpr.py:96 never assigns best_loss, so every start overwrites best_params
        best_params = [model.variance, model.lengthscales, model.noise, model.Z.copy()]
    model.assign(variance=best_params[0], lengthscales=best_params[1], noise=best_params[2])
    model.Z = best_params[3]
    _optimize_bfgs(model, iter_final)


def _optimize_differential_evolutions(
    model, popsize: int = 15, max_iter: int = 500, seed=None, adam_iter: int = 3000, verbose=True, batched: bool = False
) -> Any:
    """Adam on Z, then scipy differential evolution over log10 of the three hyperparameters (gpr.py:44-70).

    Every hyperparameter is frozen while DE runs, so the objective is ``-ELBO`` without priors.
    ``seed``, ``adam_iter``, ``verbose`` and ``batched`` are additions (the reference prints every objective value).
    ``batched=True`` (exact models): scipy evaluates a whole generation at once (``vectorized=True``, which implies
    ``updating="deferred"`` -- a different, equally valid DE variant than the reference's immediate updating) and the
    ``popsize * 3`` candidates are factorised by one batched launch sequence on the GPU.
    """
    model.set_all_trainable(False)
    model.set_trainable(Z=True)
    _optimize_adam(model, adam_iter)
    param_bounds = [(-1, 1), (-1, 1), (-3, 0)]

    def objective(params):
        model.assign(variance=10 ** params[0], lengthscales=10 ** params[1], noise=10 ** params[2])
        value = model.training_loss()
        if verbose:
            print(value)
        return model.training_loss()

    if batched:
        if model.Z is not None or not hasattr(model.backend, "factorize_batch"):
            raise ValueError("batched differential evolution needs an exact model on the HIP engine")

        def objective_many(params):  # (3, S) -> (S,)
            values = model.training_loss_many(10.0 ** params[0], 10.0 ** params[1], 10.0 ** params[2])
            if verbose:
                print(values)
            return values

        result = differential_evolution(
            objective_many, param_bounds, popsize=popsize, maxiter=max_iter, seed=seed, vectorized=True, updating="deferred", polish=False
        )
    else:
        result = differential_evolution(objective, param_bounds, popsize=popsize, maxiter=max_iter, seed=seed)
    model.assign(variance=10 ** result.x[0], lengthscales=10 ** result.x[1], noise=10 ** result.x[2])
    return result


# ---- the same drivers over many modes at once ------------------------------------------------------------------------
# The Adam-based drivers are simple enough to run for all modes of one engine in ONE host loop: the per-mode state lives
# in the rows of 2-D arrays (numpy applies the same IEEE operations per element as the per-mode code above, so every
# mode's trajectory -- including its own early stop -- is bit-identical), and every step is one batched evaluation
# (Engine.objective_batch).  GPRAS.fit uses them in lock-step mode; the other drivers run under gpras_amd.lockstep.
def _evaluate_many(models, want_grad: bool = True, stats: dict | None = None):
    if stats is not None:
        stats["batches"] = stats.get("batches", 0) + 1
    eng = models[0].backend
    mask = models[0].mask
    if any(m.mask != mask for m in models):
        raise ValueError("all models of a batched step must share the trainable mask")
    if len(models) == 1:
        loss, grad = eng.objective(models[0].unit, models[0].theta(), models[0].Z, mask, want_grad=want_grad)
        losses, grads = np.array([loss]), (None if grad is None else grad[None, :])
    else:
        units = [m.unit for m in models]
        thetas = np.stack([m.theta() for m in models])
        if models[0].Z is None:
            losses, grads, ok = eng.objective_batch(units, thetas, mask, want_grad=want_grad)
        else:
            losses, grads, ok = eng.objective_batch(units, thetas, mask, want_grad=want_grad, zs=np.stack([m.Z for m in models]))
        if not ok.all():
            raise np.linalg.LinAlgError(f"kernel matrix not positive definite for unit(s) {[units[i] for i in np.flatnonzero(~ok)]}")
    for m in models:
        m.n_evals += 1
    packed = None if grads is None else [m._pack_grad(g) for m, g in zip(models, grads)]
    return losses, packed


class _PackedBatch:
    """The hyperparameters of many models as rows of 2-D arrays for the duration of one batched driver: the per-model
    ``theta()`` / ``set_vector`` / ``_pack_grad`` calls of a step (three small numpy concatenations per model: 0.5 ms per step
    at 50 modes, as much as the device's share) become three column-indexed array operations.  Same numbers: the columns are
    only moved.  ``write_back`` leaves the models as ``set_vector`` would have."""

    def __init__(self, models):
        from .model import TRAIN_LENGTHSCALE, TRAIN_NOISE, TRAIN_VARIANCE, TRAIN_Z

        self.models = models
        self.eng = models[0].backend
        self.mask = models[0].mask
        nt = self.eng.n_theta
        self.units = np.array([m.unit for m in models], dtype=np.int32)
        self.thetas = np.stack([m.theta() for m in models])
        self.zs = None if models[0].Z is None else np.stack([m.Z for m in models]).astype(np.float64)
        zsize = 0 if self.zs is None else self.zs[0].size
        theta_cols, x_cols, grad_cols, pos = [], [], [], 0
        self.z_cols = None
        if self.mask & TRAIN_Z:
            self.z_cols = slice(0, zsize)
            grad_cols += list(range(nt, nt + zsize))
            pos = zsize
        if self.mask & TRAIN_LENGTHSCALE:
            for k in range(1, nt - 1):
                theta_cols.append(k); x_cols.append(pos); grad_cols.append(k); pos += 1
        if self.mask & TRAIN_VARIANCE:
            theta_cols.append(0); x_cols.append(pos); grad_cols.append(0); pos += 1
        if self.mask & TRAIN_NOISE:
            theta_cols.append(nt - 1); x_cols.append(pos); grad_cols.append(nt - 1); pos += 1
        self.theta_cols, self.x_cols, self.grad_cols = np.array(theta_cols, dtype=int), np.array(x_cols, dtype=int), np.array(grad_cols, dtype=int)

    @staticmethod
    def usable(models) -> bool:
        m0 = models[0]
        return (len(models) >= 1 and hasattr(m0.backend, "objective_batch") and all(m.backend is m0.backend and m.mask == m0.mask for m in models)
                and all((m.Z is None) == (m0.Z is None) for m in models))

    def assign(self, rows, x_rows):
        """``set_vector(x)`` for the given models, in the arrays."""
        if self.z_cols is not None:
            self.zs[rows] = x_rows[:, self.z_cols].reshape((len(rows),) + self.zs.shape[1:])
        if self.theta_cols.size:
            self.thetas[np.ix_(rows, self.theta_cols)] = x_rows[:, self.x_cols]

    def evaluate(self, rows, stats=None):
        if stats is not None:
            stats["batches"] = stats.get("batches", 0) + 1
        units = self.units[rows]
        losses, grads, ok = self.eng.objective_batch(units, self.thetas[rows], self.mask, want_grad=True, zs=None if self.zs is None else self.zs[rows])
        if not ok.all():
            raise np.linalg.LinAlgError(f"kernel matrix not positive definite for unit(s) {[int(units[i]) for i in np.flatnonzero(~ok)]}")
        for i in rows:
            self.models[i].n_evals += 1
        return losses, grads[:, self.grad_cols]

    def write_back(self, x):
        for m, row in zip(self.models, x):
            m.set_vector(row)


def _optimize_adam_many(models, max_iter: int, stats: dict | None = None) -> None:
    """``_optimize_adam`` for every model, one batched evaluation per step (same constants, same early stop per model)."""
    lr, beta1, beta2, eps = 1e-3, 0.9, 0.999, 1e-7
    x = np.stack([m.get_vector() for m in models])
    if x.shape[1] == 0:
        return
    if _PackedBatch.usable(models):
        batch = _PackedBatch(models)
        if hasattr(batch.eng, "adam_batch"):
            return _adam_in_library(batch, x, int(max_iter), stats)
        return _adam_packed(batch, x, int(max_iter), stats)
    mom = np.zeros_like(x)
    v = np.zeros_like(x)
    best = np.full(len(models), np.inf)
    count = np.zeros(len(models), dtype=int)
    active = np.ones(len(models), dtype=bool)
    tol = 10e-6
    patience = 50
    for t in range(1, int(max_iter) + 1):
        idx = np.flatnonzero(active)
        if idx.size == 0:
            break
        losses, grads = _evaluate_many([models[i] for i in idx], stats=stats)
        g = np.stack(grads)
        mom[idx] = beta1 * mom[idx] + (1.0 - beta1) * g
        v[idx] = beta2 * v[idx] + (1.0 - beta2) * g * g
        alpha = lr * np.sqrt(1.0 - beta2**t) / (1.0 - beta1**t)
        x[idx] = x[idx] - alpha * mom[idx] / (np.sqrt(v[idx]) + eps)
        for j, i in enumerate(idx):
            models[i].set_vector(x[i])
            loss = losses[j]
            if ((best[i] - loss) / abs(loss)) > tol:
                best[i] = loss
                count[i] = 0
            else:
                count[i] += 1
                if count[i] > patience:
                    active[i] = False


def _adam_in_library(batch: "_PackedBatch", x, max_iter: int, stats) -> None:
    """The same loop inside ``libgprx.so`` (``gprx_adam_batch``): nothing of a step runs in Python.  Same numbers as
    ``_adam_packed`` (the C loop restates the NumPy expressions operation by operation; tests hold both against the serial driver)."""
    from .model import TRAIN_Z

    def store(thetas, zs, n_evals, batches):
        if stats is not None:
            stats["batches"] = stats.get("batches", 0) + int(batches)
        for m, k in zip(batch.models, n_evals):
            m.n_evals += int(k)
        batch.thetas[:] = thetas
        if zs is not None and batch.zs is not None:
            batch.zs[:] = zs
        # back into the packed vector, then into the models exactly as set_vector would
        if batch.z_cols is not None and (batch.mask & TRAIN_Z):
            x[:, batch.z_cols] = batch.zs.reshape(len(batch.models), -1)
        if batch.theta_cols.size:
            x[:, batch.x_cols] = batch.thetas[:, batch.theta_cols]
        batch.write_back(x)

    try:
        out = batch.eng.adam_batch(batch.units, batch.thetas, batch.mask, max_iter, zs=batch.zs)
    except MemoryError:
        # the batch does not fit in device memory: the Python loop splits its evaluations (Engine.objective_batch); the library
        # ran out on the first evaluation, before any update
        return _adam_packed(batch, x, max_iter, stats)
    except Exception as exc:  # noqa: BLE001
        state = getattr(exc, "state", None)
        if state is not None:
            store(*state)  # the models keep the variables of the step that failed, as in the Python loop
        raise
    store(*out)


def _adam_packed(batch: "_PackedBatch", x, max_iter: int, stats) -> None:
    """The loop of ``_optimize_adam_many`` on packed arrays: per element the same operations in the same order."""
    lr, beta1, beta2, eps = 1e-3, 0.9, 0.999, 1e-7
    n = x.shape[0]
    mom = np.zeros_like(x)
    v = np.zeros_like(x)
    best = np.full(n, np.inf)
    count = np.zeros(n, dtype=int)
    active = np.ones(n, dtype=bool)
    tol = 10e-6
    patience = 50
    try:
        for t in range(1, max_iter + 1):
            idx = np.flatnonzero(active)
            if idx.size == 0:
                break
            losses, g = batch.evaluate(idx, stats)
            mom[idx] = beta1 * mom[idx] + (1.0 - beta1) * g
            v[idx] = beta2 * v[idx] + (1.0 - beta2) * g * g
            alpha = lr * np.sqrt(1.0 - beta2**t) / (1.0 - beta1**t)
            x[idx] = x[idx] - alpha * mom[idx] / (np.sqrt(v[idx]) + eps)
            batch.assign(idx, x[idx])
            with np.errstate(invalid="ignore", divide="ignore"):
                improved = ((best[idx] - losses) / np.abs(losses)) > tol
            best[idx[improved]] = losses[improved]
            count[idx[improved]] = 0
            stale = idx[~improved]
            count[stale] += 1
            active[stale[count[stale] > patience]] = False
    finally:
        batch.write_back(x)


def _optimize_two_stage_many(models, max_iter: int = 100, stats: dict | None = None) -> None:
    """``_optimize_two_stage`` for every model (gpr.py:112-127)."""
    for m in models:
        m.set_all_trainable(False)
        m.set_trainable(Z=True)
    _optimize_adam_many(models, max_iter, stats)
    for m in models:
        m.set_all_trainable(True)
        m.set_trainable(Z=False)
    _optimize_adam_many(models, max_iter, stats)
    for m in models:
        m.set_trainable(Z=True)
    _evaluate_many(models, want_grad=False, stats=stats)  # the final training_loss() of the reference's driver


BATCHED_OPTIMIZERS: dict[str, Any] = {"adam": _optimize_adam_many, "two-stage": _optimize_two_stage_many}

OPTIMIZERS: dict[str, Any] = {
    "two-stage": _optimize_two_stage,
    "three-stage": _optimize_three_stage,
    "adam": _optimize_adam,
    "adadelta": _optimize_adadelta,
    "L-BFGS-B": _optimize_bfgs,
    "stochastic": _optimize_multi_start,
    "diffential_evolution": _optimize_differential_evolutions,
}
