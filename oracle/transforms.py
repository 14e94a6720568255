"""Parameter transforms and priors of the gpras hot path (oracle).

Restates, for ``/root/reference/gpras/gpr.py:298-305``:

* gpflow ``positive()``: kernel variance and lengthscales are ``softplus(w)``;
  the Gaussian likelihood variance is ``1e-6 + softplus(w)`` (gpflow's
  ``DEFAULT_VARIANCE_LOWER_BOUND``).  Optimisers act on the unconstrained ``w``.
* ``tfp.distributions.LogNormal(0, 1)`` priors placed on the *constrained* value
  (gpflow ``prior_on = CONSTRAINED``: no Jacobian term), summed over trainable
  parameters only.
"""

from __future__ import annotations

import numpy as np

NOISE_LOWER = 1e-6
LOG_2PI = np.log(2.0 * np.pi)


def softplus(w):
    w = np.asarray(w, dtype=np.float64)
    return np.logaddexp(0.0, w)


def softplus_grad(w):
    """d softplus / dw = sigmoid(w)."""
    w = np.asarray(w, dtype=np.float64)
    return 0.5 * (1.0 + np.tanh(0.5 * w))


def softplus_inv(u):
    """w with softplus(w) = u (u > 0), stable for small and large u."""
    u = np.asarray(u, dtype=np.float64)
    return u + np.log(-np.expm1(-u))


def lognormal01_logpdf(u):
    u = np.asarray(u, dtype=np.float64)
    lu = np.log(u)
    return -lu - 0.5 * LOG_2PI - 0.5 * lu * lu


def lognormal01_dlogpdf(u):
    """d/du of ``lognormal01_logpdf``."""
    u = np.asarray(u, dtype=np.float64)
    return -(1.0 + np.log(u)) / u


def constrain(w_var, w_len, w_noise):
    """Unconstrained -> (variance, lengthscales, noise variance)."""
    return softplus(w_var), softplus(w_len), NOISE_LOWER + softplus(w_noise)


def unconstrain(variance, lengthscales, noise):
    return softplus_inv(variance), softplus_inv(lengthscales), softplus_inv(np.asarray(noise) - NOISE_LOWER)
