"""CPU restatement of ``GPRAS`` (``/root/reference/gpras/gpr.py:217-384``) and its optimiser drivers.

Test infrastructure: runs the whole fit / predict path on the numpy oracle so that the
HIP-backed ``gpras_amd.GPRAS`` can be compared end to end.  Every driver cites the lines it
follows; the behavioural quirks listed in SURVEY.md section 8a are kept on purpose:

* Adam = ``tf.keras.optimizers.Adam()`` defaults (lr 1e-3, beta 0.9 / 0.999, eps 1e-7), a fresh
  optimiser state per call, early stop with ``tol = 10e-6`` and patience 50 (``gpr.py:147-173``);
* stage 1 of the staged drivers trains Z only, so its loss carries no prior term
  (``gpr.py:115-116``; gpflow sums priors over trainable parameters);
* the multi-start driver never records ``best_loss`` so the last start wins, and it replaces
  the Z parameter by a plain array, which freezes Z (``gpr.py:86-109``);
* differential evolution runs with every hyperparameter frozen, i.e. on ``-ELBO`` without
  priors (``gpr.py:48-62``).
"""

from __future__ import annotations

import numpy as np
from scipy.optimize import differential_evolution, minimize

from . import exact, sgpr
from . import transforms as tr

V, L, S, ZZ = 0, 1, 2, 3  # mask slots: variance, lengthscales, noise, Z


class OracleModel:
    """One gpflow ``SGPR`` model (one output column), or the exact GP when ``Z is None``."""

    def __init__(self, kernel, x, y_col, Z, variance=1.0, lengthscales=1.0, noise=1.0, form="direct"):
        self.kernel = kernel
        self.x = np.asarray(x, dtype=np.float64)
        self.y = np.asarray(y_col, dtype=np.float64).reshape(-1)
        self.Z = None if Z is None else np.array(Z, dtype=np.float64)
        self.form = form
        self.w_var, self.w_len, self.w_noise = (np.array(t, dtype=np.float64) for t in tr.unconstrain(variance, lengthscales, noise))
        self.mask = [True, True, True, self.Z is not None]

    # -- gpflow.set_trainable -------------------------------------------------------------
    def set_trainable(self, variance=None, lengthscales=None, noise=None, Z=None):
        for slot, flag in ((V, variance), (L, lengthscales), (S, noise), (ZZ, Z)):
            if flag is not None:
                self.mask[slot] = bool(flag) and not (slot == ZZ and self.Z is None)

    def set_all_trainable(self, flag):
        self.set_trainable(flag, flag, flag, flag)

    # -- constrained views / Parameter.assign ---------------------------------------------
    @property
    def variance(self):
        return float(tr.softplus(self.w_var))

    @property
    def lengthscales(self):
        ls = tr.softplus(self.w_len)
        return ls if ls.ndim else float(ls)

    @property
    def noise(self):
        return float(tr.NOISE_LOWER + tr.softplus(self.w_noise))

    def assign(self, variance=None, lengthscales=None, noise=None):
        if variance is not None:
            self.w_var = np.array(tr.softplus_inv(variance))
        if lengthscales is not None:
            self.w_len = np.array(tr.softplus_inv(lengthscales))
        if noise is not None:
            self.w_noise = np.array(tr.softplus_inv(np.asarray(noise) - tr.NOISE_LOWER))

    # -- loss ----------------------------------------------------------------------------
    def loss_and_grad(self):
        wl = self.w_len if self.w_len.ndim else float(self.w_len)
        if self.Z is None:
            return exact.loss_and_grad(self.kernel, self.x, self.y, float(self.w_var), wl, float(self.w_noise), tuple(self.mask[:3]), self.form)
        return sgpr.loss_and_grad(self.kernel, self.x, self.y, self.Z, float(self.w_var), wl, float(self.w_noise), tuple(self.mask), form=self.form)

    def training_loss(self):
        return self.loss_and_grad()[0]

    # -- flat vector of trainable variables (order: Z, lengthscales, variance, noise) ------
    def get_vector(self):
        parts = []
        if self.mask[ZZ]:
            parts.append(self.Z.ravel())
        if self.mask[L]:
            parts.append(np.atleast_1d(self.w_len).ravel())
        if self.mask[V]:
            parts.append(np.atleast_1d(self.w_var))
        if self.mask[S]:
            parts.append(np.atleast_1d(self.w_noise))
        return np.concatenate(parts) if parts else np.zeros(0)

    def set_vector(self, vec):
        pos = 0
        if self.mask[ZZ]:
            self.Z = vec[pos : pos + self.Z.size].reshape(self.Z.shape).copy()
            pos += self.Z.size
        if self.mask[L]:
            k = self.w_len.size
            self.w_len = vec[pos : pos + k].reshape(self.w_len.shape).copy()
            pos += k
        if self.mask[V]:
            self.w_var = np.array(vec[pos])
            pos += 1
        if self.mask[S]:
            self.w_noise = np.array(vec[pos])
            pos += 1

    def grad_vector(self, g):
        parts = []
        if self.mask[ZZ]:
            parts.append(np.asarray(g["Z"]).ravel())
        if self.mask[L]:
            parts.append(np.atleast_1d(g["lengthscales"]).ravel())
        if self.mask[V]:
            parts.append(np.atleast_1d(g["variance"]))
        if self.mask[S]:
            parts.append(np.atleast_1d(g["noise"]))
        return np.concatenate(parts) if parts else np.zeros(0)

    def predict_y(self, xs):
        if self.Z is None:
            return exact.predict(self.kernel, self.x, self.y, self.variance, self.lengthscales, self.noise, xs, True, self.form)
        return sgpr.predict(self.kernel, self.x, self.y, self.Z, self.variance, self.lengthscales, self.noise, xs, True, form=self.form)


# ---------------------------------------------------------------------------------------
# optimiser drivers
# ---------------------------------------------------------------------------------------
def optimize_adam(model, max_iter):
    """``_optimize_adam`` (gpr.py:147-173).  Returns the list of per-step losses."""
    lr, b1, b2, eps = 1e-3, 0.9, 0.999, 1e-7
    x = model.get_vector()
    m = np.zeros_like(x)
    v = np.zeros_like(x)
    losses = []
    best, count, tol, patience = np.inf, 0, 10e-6, 50
    for t in range(1, int(max_iter) + 1):
        loss, g = model.loss_and_grad()
        gv = model.grad_vector(g)
        m = b1 * m + (1.0 - b1) * gv
        v = b2 * v + (1.0 - b2) * gv * gv
        alpha = lr * np.sqrt(1.0 - b2**t) / (1.0 - b1**t)
        x = x - alpha * m / (np.sqrt(v) + eps)
        model.set_vector(x)
        losses.append(loss)
        if ((best - loss) / abs(loss)) > tol:
            best, count = loss, 0
        else:
            count += 1
            if count > patience:
                break
    return losses


def optimize_adadelta(model, max_iter):
    """``_optimize_adadelta`` / ``_optimize_tf`` (gpr.py:176-192): Keras Adadelta defaults, no early stop."""
    lr, rho, eps = 1e-3, 0.95, 1e-7
    x = model.get_vector()
    acc_g = np.zeros_like(x)
    acc_d = np.zeros_like(x)
    loss = None
    for _ in range(int(max_iter)):
        loss, g = model.loss_and_grad()
        gv = model.grad_vector(g)
        acc_g = rho * acc_g + (1.0 - rho) * gv * gv
        delta = -np.sqrt(acc_d + eps) * gv / np.sqrt(acc_g + eps)
        acc_d = rho * acc_d + (1.0 - rho) * delta * delta
        x = x + lr * delta
        model.set_vector(x)
    return loss


def optimize_bfgs(model, max_iter):
    """``_optimize_bfgs`` (gpr.py:195-203): scipy L-BFGS-B on the packed unconstrained vector, jac=True."""

    def fun(vec):
        model.set_vector(vec)
        loss, g = model.loss_and_grad()
        return loss, model.grad_vector(g)

    res = minimize(fun, model.get_vector(), jac=True, method="L-BFGS-B", options={"maxiter": int(max_iter)})
    model.set_vector(res.x)
    return res


def optimize_two_stage(model, max_iter=100):
    """``_optimize_two_stage`` (gpr.py:112-127)."""
    model.set_all_trainable(False)
    model.set_trainable(Z=True)
    optimize_adam(model, max_iter)
    model.set_all_trainable(True)
    model.set_trainable(Z=False)
    optimize_adam(model, max_iter)
    model.set_trainable(Z=True)
    return model.training_loss()


def optimize_three_stage(model, max_iter=100):
    """``_optimize_three_stage`` (gpr.py:130-144)."""
    model.set_all_trainable(False)
    model.set_trainable(Z=True)
    optimize_adam(model, max_iter)
    model.set_all_trainable(True)
    model.set_trainable(Z=False)
    optimize_bfgs(model, max_iter)
    model.set_trainable(Z=True)
    optimize_bfgs(model, max_iter)


def optimize_multi_start(model, n_starts=40, iter_initial=20, iter_final=1000, rng=None):
    """``_optimize_multi_start`` (gpr.py:73-109).  ``rng`` is injectable for tests; the reference's is unseeded."""
    rng = np.random.default_rng() if rng is None else rng
    mins, maxs = model.x.min(axis=0), model.x.max(axis=0)
    z_shape = (model.Z.shape[0], model.x.shape[1])
    best_params = None
    for _ in range(int(n_starts)):
        model.assign(variance=10 ** rng.uniform(-1, 1))
        model.assign(lengthscales=np.full_like(model.w_len, 10 ** rng.uniform(-1, 1)))
        model.assign(noise=10 ** rng.uniform(-3, 0))
        model.Z = rng.uniform(mins, maxs, size=z_shape)
        model.set_trainable(Z=False)  # gpr.py:91 replaces the Parameter by an ndarray
        optimize_adam(model, iter_initial)
        model.training_loss()
        # gpr.py:96: best_loss stays None, so every start overwrites best_params
        best_params = [model.variance, model.lengthscales, model.noise, model.Z.copy()]
    model.assign(variance=best_params[0], lengthscales=best_params[1], noise=best_params[2])
    model.Z = best_params[3]
    optimize_bfgs(model, iter_final)


def optimize_differential_evolution(model, popsize=15, max_iter=500, seed=None, adam_iter=3000):
    """``_optimize_differential_evolutions`` (gpr.py:44-70); ``seed`` / ``adam_iter`` are test hooks."""
    model.set_all_trainable(False)
    model.set_trainable(Z=True)
    optimize_adam(model, adam_iter)
    bounds = [(-1, 1), (-1, 1), (-3, 0)]

    def objective(p):
        model.assign(variance=10 ** p[0], lengthscales=np.full_like(model.w_len, 10 ** p[1]), noise=10 ** p[2])
        return model.training_loss()

    res = differential_evolution(objective, bounds, popsize=popsize, maxiter=max_iter, seed=seed)
    model.assign(variance=10 ** res.x[0], lengthscales=np.full_like(model.w_len, 10 ** res.x[1]), noise=10 ** res.x[2])
    return res


OPTIMIZERS = {
    "two-stage": optimize_two_stage,
    "three-stage": optimize_three_stage,
    "adam": optimize_adam,
    "adadelta": optimize_adadelta,
    "L-BFGS-B": optimize_bfgs,
    "stochastic": optimize_multi_start,
    "diffential_evolution": optimize_differential_evolution,
}


def create_inducing(x, n_inducing, method):
    """``GPRAS._create_inducing`` (gpr.py:310-320)."""
    if method == "kmeans":
        from sklearn.cluster import KMeans

        km = KMeans(n_clusters=n_inducing, random_state=0, n_init="auto")
        km.fit(x)
        return km.cluster_centers_.astype(np.float64)
    if method == "grid":
        return np.column_stack([np.linspace(x[:, j].min(), x[:, j].max(), n_inducing) for j in range(x.shape[1])])
    return None


class GPRASOracle:
    """``GPRAS`` on the CPU oracle.  ``n_inducing=None`` selects the exact GP (extension)."""

    def __init__(self, kernel, form="direct"):
        from .kernels import KERNEL_IDS

        self.kernel_str = kernel
        KERNEL_IDS[kernel]  # KeyError for unknown names, as gpr.py:230
        self.form = form
        self.models = []
        self.x = None
        self.y = None

    def _init_models(self, x, y, n_inducing, inducing_initializer="kmeans", ard=False):
        Z = None if n_inducing is None else create_inducing(x, n_inducing, inducing_initializer)
        ini_length = np.mean(abs(x))
        ls0 = np.full(x.shape[1], ini_length) if ard else ini_length
        self.models = [OracleModel(self.kernel_str, x, y[:, i], Z, 1.0, ls0, 1.0, self.form) for i in range(y.shape[1])]

    def fit(self, x, y, n_inducing, inducing_initializer="kmeans", optimization_method="two-stage", ard=False, **opt_kwargs):
        self.x = x.astype(np.float64)
        self.y = y.astype(np.float64)
        self._init_models(self.x, self.y, n_inducing, inducing_initializer, ard)
        opt = OPTIMIZERS[optimization_method]
        for model in self.models:
            opt(model, **opt_kwargs)

    def predict(self, x):
        x = x.astype(np.float64)
        preds = [m.predict_y(x) for m in self.models]
        return np.column_stack([p[0] for p in preds]), np.column_stack([p[1] for p in preds])
