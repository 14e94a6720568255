"""Sparse GP regression (Titsias collapsed bound) -- oracle restatement of gpflow ``SGPR``.

This is the arithmetic behind ``SGPR.training_loss()`` and ``SGPR.predict_y()``
as called from ``/root/reference/gpras/gpr.py:61-62, 95, 127, 154, 187, 199``
(loss / gradient) and ``:337`` (prediction).  gpflow (unpinned, not installed
here) is the third-party owner of the algorithm; formulas follow its
``models/sgpr.py`` (``_common_calculation``, ``logdet_term``, ``quad_term``,
``predict_f``) as summarised in SURVEY.md section 8-A.

Notation: X (N, d) inputs, y (N,) one output column (the reference builds one
model per column of ``y``: ``gpr.py:293-299``), Z (M, d) inducing inputs,
v kernel variance, l lengthscale(s), s noise variance, jitter 1e-6.

    Kuf = k(Z, X)            Kuu = k(Z, Z) + jitter I      L  = chol(Kuu)
    A   = L^-1 Kuf / sqrt(s) B   = I + A A^T               LB = chol(B)
    c   = LB^-1 A y / sqrt(s)
    ELBO = -N/2 log 2pi - sum log diag LB - N/2 log s
           - (N v / s - tr(A A^T)) / 2 - (y.y / s - c.c) / 2
    loss = -(ELBO + sum_{p trainable} log LogNormal(0,1)(p))

The gradient is written analytically (reverse mode by hand), not by autodiff:
with P = Kuf, Q = Kuu, Sigma = Q + P P^T / s, m = Sigma^-1 P y / s,

    dELBO/dQ = (Q^-1 - Sigma^-1)/2 - Q^-1 P P^T Q^-1 / (2 s) - m m^T / 2
    dELBO/dP = ((Q^-1 - Sigma^-1 - m m^T) P + m y^T) / s
    dELBO/ds = (tr(Sigma^-1 P P^T) - tr(Q^-1 P P^T) + |y - P^T m|^2 + N v) / (2 s^2) - N / (2 s)

and the chain rule through the stationary kernel uses ``h = g'(r)/r``
(``oracle/kernels.py``).  ``tests/test_oracle.py`` checks it against central
differences.
"""

from __future__ import annotations

import numpy as np
from scipy.linalg import cholesky, solve_triangular

from . import kernels as kn
from . import transforms as tr

JITTER = 1e-6
LOG_2PI = np.log(2.0 * np.pi)


def _ls_vec(lengthscales, d):
    return np.broadcast_to(np.asarray(lengthscales, dtype=np.float64), (d,))


def common_terms(kernel, X, y, Z, variance, lengthscales, noise, jitter=JITTER, form="direct"):
    """The tensors of gpflow ``SGPR._common_calculation`` plus ``c``."""
    X = np.asarray(X, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64).reshape(-1)
    Z = np.asarray(Z, dtype=np.float64)
    sigma = np.sqrt(noise)
    Kuf = kn.kmat(kernel, Z, X, variance, lengthscales, form)
    Kuu = kn.kmat(kernel, Z, Z, variance, lengthscales, form) + jitter * np.eye(Z.shape[0])
    L = cholesky(Kuu, lower=True)
    A = solve_triangular(L, Kuf, lower=True) / sigma
    B = np.eye(Z.shape[0]) + A @ A.T
    LB = cholesky(B, lower=True)
    c = solve_triangular(LB, A @ y, lower=True) / sigma
    return {"Kuf": Kuf, "Kuu": Kuu, "L": L, "A": A, "B": B, "LB": LB, "c": c, "sigma": sigma}


def elbo(kernel, X, y, Z, variance, lengthscales, noise, jitter=JITTER, form="direct"):
    """gpflow ``SGPR.elbo`` for one output column."""
    t = common_terms(kernel, X, y, Z, variance, lengthscales, noise, jitter, form)
    y = np.asarray(y, dtype=np.float64).reshape(-1)
    n = y.shape[0]
    A, LB, c = t["A"], t["LB"], t["c"]
    bound = -0.5 * n * LOG_2PI
    bound -= np.log(np.diag(LB)).sum()
    bound -= 0.5 * n * np.log(noise)
    bound -= 0.5 * (n * variance / noise - np.sum(A * A))
    bound -= 0.5 * (y @ y / noise - c @ c)
    return float(bound)


def elbo_grads(kernel, X, y, Z, variance, lengthscales, noise, jitter=JITTER, form="direct"):
    """ELBO and its derivatives w.r.t. the *constrained* parameters and Z.

    Returns ``(elbo, d_variance, d_lengthscales (same shape as input), d_noise, d_Z (M, d))``.
    """
    X = np.asarray(X, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64).reshape(-1)
    Z = np.asarray(Z, dtype=np.float64)
    n, d = X.shape
    m_ind = Z.shape[0]
    ard = np.ndim(lengthscales) > 0
    ls = _ls_vec(lengthscales, d)
    t = common_terms(kernel, X, y, Z, variance, lengthscales, noise, jitter, form)
    L, A, LB, c, sigma = t["L"], t["A"], t["LB"], t["c"], t["sigma"]
    P, Q = t["Kuf"], t["Kuu"]
    s = noise

    bound = (
        -0.5 * n * LOG_2PI
        - np.log(np.diag(LB)).sum()
        - 0.5 * n * np.log(s)
        - 0.5 * (n * variance / s - np.sum(A * A))
        - 0.5 * (y @ y / s - c @ c)
    )

    eye = np.eye(m_ind)
    Linv = solve_triangular(L, eye, lower=True)
    Qinv = Linv.T @ Linv
    R = solve_triangular(LB, Linv, lower=True)  # LB^-1 L^-1
    Sinv = R.T @ R  # Sigma^-1
    mvec = solve_triangular(L, solve_triangular(LB, c, lower=True, trans="T"), lower=True, trans="T")
    QinvP = Qinv @ P
    W = Qinv - Sinv - np.outer(mvec, mvec)
    G_Q = 0.5 * (Qinv - Sinv) - QinvP @ QinvP.T / (2.0 * s) - 0.5 * np.outer(mvec, mvec)
    G_P = (W @ P + np.outer(mvec, y)) / s

    resid = y - P.T @ mvec
    tr_SinvPP = s * (m_ind - np.trace(solve_triangular(LB, solve_triangular(LB, eye, lower=True), lower=True, trans="T")))
    tr_QinvPP = s * np.sum(A * A)
    d_noise = (tr_SinvPP - tr_QinvPP + resid @ resid + n * variance) / (2.0 * s * s) - n / (2.0 * s)

    # chain rule through the kernel
    r2_P = kn.scaled_sqdist(Z, X, ls, form)
    r2_Q = kn.scaled_sqdist(Z, Z, ls, form)
    h_P = variance * kn.h_of_r2(kernel, r2_P)
    h_Q = variance * kn.h_of_r2(kernel, r2_Q)
    d_variance = -n / (2.0 * s) + np.sum(G_P * P) / variance + np.sum(G_Q * (Q - jitter * eye)) / variance

    GhP = G_P * h_P
    GhQ = G_Q * h_Q
    GhQs = GhQ + GhQ.T
    d_ls = np.zeros(d)
    d_Z = np.zeros_like(Z)
    for k in range(d):
        dP = Z[:, k][:, None] - X[:, k][None, :]
        dQ = Z[:, k][:, None] - Z[:, k][None, :]
        d_ls[k] = -(np.sum(GhP * dP * dP) + np.sum(GhQ * dQ * dQ)) / ls[k] ** 3
        d_Z[:, k] = (np.sum(GhP * dP, axis=1) + np.sum(GhQs * dQ, axis=1)) / ls[k] ** 2
    d_len = d_ls if ard else float(d_ls.sum())
    return float(bound), float(d_variance), d_len, float(d_noise), d_Z


def loss_and_grad(kernel, X, y, Z, w_var, w_len, w_noise, mask=(True, True, True, True), jitter=JITTER, form="direct"):
    """``training_loss`` and gradient in the optimiser's (unconstrained) variables.

    ``mask`` = trainable flags for (variance, lengthscales, noise, Z) -- gpflow
    ``set_trainable`` as used in ``gpr.py:48-49, 115-125, 133-143``.  Priors are summed
    over trainable parameters only.  Gradients of frozen parameters are returned as 0.
    """
    variance, ls, noise = tr.constrain(w_var, w_len, w_noise)
    variance = float(variance)
    noise = float(noise)
    ls_arg = ls if np.ndim(w_len) > 0 else float(ls)
    bound, d_v, d_l, d_s, d_Z = elbo_grads(kernel, X, y, Z, variance, ls_arg, noise, jitter, form)
    logp = 0.0
    if mask[0]:
        logp += float(tr.lognormal01_logpdf(variance))
        d_v += float(tr.lognormal01_dlogpdf(variance))
    if mask[1]:
        logp += float(np.sum(tr.lognormal01_logpdf(ls)))
        d_l = d_l + tr.lognormal01_dlogpdf(ls)
    if mask[2]:
        logp += float(tr.lognormal01_logpdf(noise))
        d_s += float(tr.lognormal01_dlogpdf(noise))
    loss = -(bound + logp)
    g_var = -d_v * float(tr.softplus_grad(w_var)) if mask[0] else 0.0
    g_len = -np.asarray(d_l) * tr.softplus_grad(w_len) if mask[1] else np.zeros_like(np.asarray(w_len, dtype=np.float64))
    g_noise = -d_s * float(tr.softplus_grad(w_noise)) if mask[2] else 0.0
    g_Z = -d_Z if mask[3] else np.zeros_like(d_Z)
    if np.ndim(w_len) == 0:
        g_len = float(g_len)
    return loss, {"variance": g_var, "lengthscales": g_len, "noise": g_noise, "Z": g_Z}


def loss(kernel, X, y, Z, w_var, w_len, w_noise, mask=(True, True, True, True), jitter=JITTER, form="direct"):
    variance, ls, noise = tr.constrain(w_var, w_len, w_noise)
    ls_arg = ls if np.ndim(w_len) > 0 else float(ls)
    bound = elbo(kernel, X, y, Z, float(variance), ls_arg, float(noise), jitter, form)
    logp = 0.0
    if mask[0]:
        logp += float(tr.lognormal01_logpdf(variance))
    if mask[1]:
        logp += float(np.sum(tr.lognormal01_logpdf(ls)))
    if mask[2]:
        logp += float(tr.lognormal01_logpdf(noise))
    return -(bound + logp)


def predict(kernel, X, y, Z, variance, lengthscales, noise, Xs, include_noise=True, jitter=JITTER, form="direct"):
    """gpflow ``SGPR.predict_f`` (+ noise = ``predict_y``, which is what ``gpr.py:337`` returns)."""
    t = common_terms(kernel, X, y, Z, variance, lengthscales, noise, jitter, form)
    Kus = kn.kmat(kernel, Z, Xs, variance, lengthscales, form)
    tmp1 = solve_triangular(t["L"], Kus, lower=True)
    tmp2 = solve_triangular(t["LB"], tmp1, lower=True)
    mean = tmp2.T @ t["c"]
    var = variance + np.sum(tmp2 * tmp2, axis=0) - np.sum(tmp1 * tmp1, axis=0)
    if include_noise:
        var = var + noise
    return mean, var
