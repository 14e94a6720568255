"""Lock-step fitting of the independent per-mode models of one ``GPRAS`` (exact and sparse models).

The reference fits its modes one after the other (``/root/reference/gpras/gpr.py:272-274``); every optimiser
step of a mode is one ``training_loss`` (+ gradient) evaluation, and the modes share nothing but ``x``.  Here
each mode's optimiser driver -- unchanged, with all its quirks -- runs in its own host thread against a proxy
backend; an evaluation request blocks until every still-running mode has asked for one, then ALL pending
requests are evaluated by one batched launch sequence (``Engine.objective_batch``: every kernel once for all
cells) and the threads continue.  The batched evaluations are bit-identical to single ones, and each driver
sees exactly the values it would have seen alone, so the fitted parameters equal the serial loop's bit for bit.
"""

from __future__ import annotations

import threading
from typing import Any, Callable

import numpy as np


class LockstepEvaluator:
    """Collects one evaluation request per running worker and evaluates them as one batch."""

    def __init__(self, engine, n_workers: int) -> None:
        self.engine = engine
        self.cv = threading.Condition()
        self.active = n_workers
        self.pending: dict[int, tuple[int, np.ndarray, int, bool, Any]] = {}
        self.results: dict[int, Any] = {}
        self.batches = 0  # launch sequences issued (for tests / reporting)
        self.cells = 0    # evaluations served

    # -- worker side ---------------------------------------------------------------------------------
    def evaluate(self, wid: int, unit: int, theta, z, mask: int, want_grad: bool):
        with self.cv:
            zc = None if z is None else np.array(z, dtype=np.float64)
            self.pending[wid] = (int(unit), np.array(theta, dtype=np.float64), int(mask), bool(want_grad), zc)
            if len(self.pending) >= self.active:
                self._flush()
            else:
                while wid not in self.results:
                    self.cv.wait()
            out = self.results.pop(wid)
        if isinstance(out, BaseException):
            raise out
        return out

    def finish(self, wid: int) -> None:
        """The worker's optimiser has returned (or raised): the others no longer wait for it."""
        with self.cv:
            self.active -= 1
            if self.pending and len(self.pending) >= self.active:
                self._flush()

    # -- batch (called with the lock held; the waiting workers hold nothing) -----------------------------
    def _flush(self) -> None:
        groups: dict[tuple[int, bool], list[int]] = {}
        for wid, (_, _, mask, want_grad, _) in self.pending.items():
            groups.setdefault((mask, want_grad), []).append(wid)
        for (mask, want_grad), wids in groups.items():
            try:
                if len(wids) == 1:
                    unit, theta, _, _, zc = self.pending[wids[0]]
                    self.results[wids[0]] = self.engine.objective(unit, theta, zc, mask, want_grad=want_grad)
                else:
                    units = [self.pending[w][0] for w in wids]
                    thetas = np.stack([self.pending[w][1] for w in wids])
                    zs = None if self.pending[wids[0]][4] is None else np.stack([self.pending[w][4] for w in wids])
                    if zs is None:
                        losses, grads, ok = self.engine.objective_batch(units, thetas, mask, want_grad=want_grad)
                    else:
                        losses, grads, ok = self.engine.objective_batch(units, thetas, mask, want_grad=want_grad, zs=zs)
                    for k, w in enumerate(wids):
                        if ok[k]:
                            self.results[w] = (float(losses[k]), grads[k].copy() if want_grad else None)
                        else:
                            self.results[w] = np.linalg.LinAlgError(f"kernel matrix of unit {units[k]} is not positive definite")
            except BaseException as exc:  # a device failure reaches every worker of the group
                for w in wids:
                    self.results[w] = exc
            self.batches += 1
            self.cells += len(wids)
        self.pending.clear()
        self.cv.notify_all()


class LockstepBackend:
    """What a ``GPModel`` sees instead of the engine while a lock-step fit runs."""

    def __init__(self, engine, evaluator: LockstepEvaluator, wid: int) -> None:
        self._engine = engine
        self._evaluator = evaluator
        self._wid = wid

    def __getattr__(self, name):
        # attributes (n_theta, n_len, m, d, ard, x, ...) and the engine's other calls -- factorize_batch (batched
        # differential evolution), predict, max_cells: those run outside the lock-step rounds, directly on the shared
        # engine, which serialises concurrent callers on its own lock (Engine._lock: a gprx handle is not thread-safe)
        return getattr(self._engine, name)

    def objective(self, unit, theta, z, mask, want_grad=True):
        return self._evaluator.evaluate(self._wid, unit, theta, z, mask, want_grad)


def fit_lockstep(models, optimizer: Callable[..., Any], opt_kwargs: dict[str, Any], max_batch: int = 32) -> dict[str, int]:
    """Run ``optimizer(model, **opt_kwargs)`` for every model, ``max_batch`` models at a time in lock step.
    All models must share one engine.  Returns counters (batches issued, evaluations served)."""
    stats = {"batches": 0, "evaluations": 0}
    for lo in range(0, len(models), max_batch):
        chunk = models[lo : lo + max_batch]
        engine = chunk[0].backend
        evaluator = LockstepEvaluator(engine, len(chunk))
        errors: list[BaseException | None] = [None] * len(chunk)

        def run(wid: int, model) -> None:
            model.backend = LockstepBackend(engine, evaluator, wid)
            try:
                optimizer(model, **opt_kwargs)
            except BaseException as exc:
                errors[wid] = exc
            finally:
                model.backend = engine
                evaluator.finish(wid)

        threads = [threading.Thread(target=run, args=(wid, m), name=f"gpras-lockstep-{wid}") for wid, m in enumerate(chunk)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        stats["batches"] += evaluator.batches
        stats["evaluations"] += evaluator.cells
        for exc in errors:
            if exc is not None:
                raise exc
    return stats
