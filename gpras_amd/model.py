"""One GP model per output column -- the host-side mirror of a gpflow ``SGPR`` object as the reference
uses it (``/root/reference/gpras/gpr.py:293-308``): parameters, trainable flags, loss and predict.

All arithmetic of loss / gradient / predict is delegated to a backend with three methods
(``objective``, ``predict`` and the attributes ``n_theta, m, d, n_len``); the package ships exactly one
backend, ``gpras_amd.engine.Engine`` (HIP).  Scalar transforms here are bookkeeping only.
"""

from __future__ import annotations

import numpy as np

TRAIN_VARIANCE, TRAIN_LENGTHSCALE, TRAIN_NOISE, TRAIN_Z = 1, 2, 4, 8
NOISE_LOWER = 1e-6  # gpflow Gaussian likelihood: variance = 1e-6 + softplus(w)


def softplus(w):
    return np.logaddexp(0.0, np.asarray(w, dtype=np.float64))


def softplus_inv(u):
    u = np.asarray(u, dtype=np.float64)
    return u + np.log(-np.expm1(-u))


class InducingVariable:
    """Carries ``Z`` so that ``gpr.models[0].inducing_variable.Z`` keeps working
    (``/root/reference/production/analysis/pipeline.py:115``)."""

    def __init__(self, Z):
        self.Z = Z


class GPModel:
    def __init__(self, backend, unit: int, Z=None, variance=1.0, lengthscales=1.0, noise=1.0):
        self.backend = backend
        self.unit = int(unit)
        self.inducing_variable = InducingVariable(None if Z is None else np.array(Z, dtype=np.float64))
        n_len = backend.n_len
        ls = np.broadcast_to(np.asarray(lengthscales, dtype=np.float64), (n_len,)).copy()
        self.w_var = float(softplus_inv(variance))
        self.w_len = softplus_inv(ls)
        self.w_noise = float(softplus_inv(noise - NOISE_LOWER))
        self.mask = TRAIN_VARIANCE | TRAIN_LENGTHSCALE | TRAIN_NOISE | (TRAIN_Z if Z is not None else 0)
        self.n_evals = 0

    def backend_x(self):
        """Training inputs (``model.data[0]`` in the reference, gpr.py:80)."""
        return self.backend.x

    # -- parameters --------------------------------------------------------------------------------
    @property
    def Z(self):
        return self.inducing_variable.Z

    @Z.setter
    def Z(self, value):
        self.inducing_variable.Z = np.array(value, dtype=np.float64)

    @property
    def variance(self) -> float:
        return float(softplus(self.w_var))

    @property
    def lengthscales(self):
        ls = softplus(self.w_len)
        return ls if self.backend.ard else float(ls[0])

    @property
    def noise(self) -> float:
        return float(NOISE_LOWER + softplus(self.w_noise))

    def assign(self, variance=None, lengthscales=None, noise=None):
        """``Parameter.assign`` on constrained values (gpr.py:57-59, 88-90, 105-107)."""
        if variance is not None:
            self.w_var = float(softplus_inv(variance))
        if lengthscales is not None:
            self.w_len = softplus_inv(np.broadcast_to(np.asarray(lengthscales, dtype=np.float64), self.w_len.shape)).copy()
        if noise is not None:
            self.w_noise = float(softplus_inv(np.asarray(noise, dtype=np.float64) - NOISE_LOWER))

    def parameter_dict(self):
        """Same keys as ``gpflow.utilities.parameter_dict`` on an SGPR (gpr.py:363), values as plain arrays."""
        out = {
            ".kernel.variance": np.array(self.variance),
            ".kernel.lengthscales": np.array(self.lengthscales),
            ".likelihood.variance": np.array(self.noise),
        }
        if self.Z is not None:
            out[".inducing_variable.Z"] = self.Z.copy()
        # extension: the optimiser's own variables, so that a reload continues from bit-identical values (the constrained
        # values above go through softplus and back, which costs an ulp)
        out[".unconstrained"] = self.theta()
        return out

    def multiple_assign(self, params):
        """``gpflow.utilities.multiple_assign`` (gpr.py:383)."""
        self.assign(params[".kernel.variance"], params[".kernel.lengthscales"], params[".likelihood.variance"])
        # ".unconstrained" (extension) only short-cuts the softplus round trip of the values just assigned: it is taken when its
        # forward transform reproduces them (to a few ulp), so a dictionary whose constrained entries were edited -- the
        # reference's parameter_dict -> modify -> multiple_assign workflow -- loads the edited values, as gpflow would
        w = params.get(".unconstrained")
        if w is not None and np.size(w) == 2 + self.w_len.size:
            w = np.asarray(w, dtype=np.float64)
            keep = (self.w_var, self.w_len.copy(), self.w_noise)
            want = (self.variance, np.array(self.lengthscales, dtype=np.float64), self.noise)
            self.w_var, self.w_len, self.w_noise = float(w[0]), w[1:-1].copy(), float(w[-1])
            got = (self.variance, np.array(self.lengthscales, dtype=np.float64), self.noise)
            same = all(np.allclose(g, t, rtol=1e-13, atol=0.0) for g, t in zip(got, want))
            if not same:
                self.w_var, self.w_len, self.w_noise = keep
        if ".inducing_variable.Z" in params and self.Z is not None:
            self.Z = params[".inducing_variable.Z"]

    # -- gpflow.set_trainable ------------------------------------------------------------------------
    def set_trainable(self, variance=None, lengthscales=None, noise=None, Z=None):
        for bit, flag in ((TRAIN_VARIANCE, variance), (TRAIN_LENGTHSCALE, lengthscales), (TRAIN_NOISE, noise), (TRAIN_Z, Z)):
            if flag is None:
                continue
            if flag and not (bit == TRAIN_Z and self.Z is None):
                self.mask |= bit
            else:
                self.mask &= ~bit

    def set_all_trainable(self, flag: bool):
        self.set_trainable(flag, flag, flag, flag)

    # -- packed trainable vector (order: Z, lengthscales, variance, noise) ---------------------------
    def theta(self):
        return np.concatenate([[self.w_var], self.w_len, [self.w_noise]])

    def get_vector(self):
        parts = []
        if self.mask & TRAIN_Z:
            parts.append(self.Z.ravel())
        if self.mask & TRAIN_LENGTHSCALE:
            parts.append(self.w_len)
        if self.mask & TRAIN_VARIANCE:
            parts.append([self.w_var])
        if self.mask & TRAIN_NOISE:
            parts.append([self.w_noise])
        return np.concatenate(parts).astype(np.float64) if parts else np.zeros(0)

    def set_vector(self, vec):
        vec = np.asarray(vec, dtype=np.float64)
        pos = 0
        if self.mask & TRAIN_Z:
            k = self.Z.size
            self.Z = vec[pos : pos + k].reshape(self.Z.shape)
            pos += k
        if self.mask & TRAIN_LENGTHSCALE:
            k = self.w_len.size
            self.w_len = vec[pos : pos + k].copy()
            pos += k
        if self.mask & TRAIN_VARIANCE:
            self.w_var = float(vec[pos])
            pos += 1
        if self.mask & TRAIN_NOISE:
            self.w_noise = float(vec[pos])
            pos += 1

    def _pack_grad(self, grad):
        nt = self.backend.n_theta
        parts = []
        if self.mask & TRAIN_Z:
            parts.append(grad[nt:])
        if self.mask & TRAIN_LENGTHSCALE:
            parts.append(grad[1 : nt - 1])
        if self.mask & TRAIN_VARIANCE:
            parts.append(grad[0:1])
        if self.mask & TRAIN_NOISE:
            parts.append(grad[nt - 1 : nt])
        return np.concatenate(parts) if parts else np.zeros(0)

    # -- loss ------------------------------------------------------------------------------------------
    def loss_and_grad(self):
        """``training_loss`` and its gradient w.r.t. ``get_vector()``."""
        self.n_evals += 1
        loss, grad = self.backend.objective(self.unit, self.theta(), self.Z, self.mask, want_grad=True)
        return loss, self._pack_grad(grad)

    def training_loss(self) -> float:
        self.n_evals += 1
        loss, _ = self.backend.objective(self.unit, self.theta(), self.Z, self.mask, want_grad=False)
        return loss

    def training_loss_many(self, variances, lengthscales, noises, chunk: int = 64):
        """``training_loss`` of this unit at many hyperparameter settings (constrained values, isotropic lengthscale
        or one row of lengthscales per candidate), evaluated by batched launch sequences of up to ``chunk`` cells.
        Candidates whose kernel matrix is not positive definite get ``+inf``.  The model's own parameters are untouched."""
        variances = np.atleast_1d(np.asarray(variances, dtype=np.float64))
        count = variances.size
        ls = np.asarray(lengthscales, dtype=np.float64)
        ls = np.broadcast_to(ls.reshape(count, -1), (count, self.backend.n_len))
        noises = np.broadcast_to(np.asarray(noises, dtype=np.float64), (count,))
        thetas = np.empty((count, self.backend.n_theta))
        thetas[:, 0] = softplus_inv(variances)
        thetas[:, 1:-1] = softplus_inv(ls)
        thetas[:, -1] = softplus_inv(noises - NOISE_LOWER)
        out = np.empty(count)
        if hasattr(self.backend, "max_cells"):
            chunk = max(1, min(chunk, self.backend.max_cells(want_grad=False)))
        for lo in range(0, count, chunk):
            hi = min(count, lo + chunk)
            losses, ok = self.backend.factorize_batch(np.full(hi - lo, self.unit, dtype=np.int32), thetas[lo:hi], self.mask)
            out[lo:hi] = np.where(ok, losses, np.inf)
        self.n_evals += count
        return out

    def predict_y(self, xs):
        """Mean and observation variance (``SGPR.predict_y``, gpr.py:337), each of shape (N*, 1)."""
        self.backend.objective(self.unit, self.theta(), self.Z, self.mask, want_grad=False)
        mean, var = self.backend.predict(xs, include_noise=True)
        return mean[:, None], var[:, None]
