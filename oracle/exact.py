"""Exact GP regression -- the ``Z = X`` specialisation BASELINE.json's metric is quoted on.

The reference never builds an exact GP; this is the limit of
``oracle/sgpr.py`` with every training input an inducing input and the jitter dropped
("Exact specialisation"):

    K = k(X, X) + s I = L L^T        alpha = K^-1 y
    LML = -y.alpha / 2 - sum log diag L - N/2 log 2pi
    dLML/dtheta = tr((alpha alpha^T - K^-1) dK/dtheta) / 2
    mean = k(X*, X) alpha            var_y = v - colsum((L^-1 k(X, X*))^2) + s

Loss, priors and parameter transforms are those of the sparse model
(``/root/reference/gpras/gpr.py:298-305``), so the same optimiser drivers apply.
"""

from __future__ import annotations

import numpy as np
from scipy.linalg import cholesky, solve_triangular

from . import kernels as kn
from . import transforms as tr

LOG_2PI = np.log(2.0 * np.pi)


def factorize(kernel, X, y, variance, lengthscales, noise, form="direct"):
    X = np.asarray(X, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64).reshape(-1)
    K = kn.kmat(kernel, X, X, variance, lengthscales, form)
    K[np.diag_indices_from(K)] += noise
    L = cholesky(K, lower=True)
    beta = solve_triangular(L, y, lower=True)
    alpha = solve_triangular(L, beta, lower=True, trans="T")
    return L, alpha


def lml(kernel, X, y, variance, lengthscales, noise, form="direct"):
    y = np.asarray(y, dtype=np.float64).reshape(-1)
    L, alpha = factorize(kernel, X, y, variance, lengthscales, noise, form)
    return float(-0.5 * y @ alpha - np.log(np.diag(L)).sum() - 0.5 * y.shape[0] * LOG_2PI)


def lml_grads(kernel, X, y, variance, lengthscales, noise, form="direct"):
    """LML and derivatives w.r.t. constrained (variance, lengthscales, noise)."""
    X = np.asarray(X, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64).reshape(-1)
    n, d = X.shape
    ard = np.ndim(lengthscales) > 0
    ls = np.broadcast_to(np.asarray(lengthscales, dtype=np.float64), (d,))
    L, alpha = factorize(kernel, X, y, variance, lengthscales, noise, form)
    value = float(-0.5 * y @ alpha - np.log(np.diag(L)).sum() - 0.5 * n * LOG_2PI)
    Linv = solve_triangular(L, np.eye(n), lower=True)
    W = np.outer(alpha, alpha) - Linv.T @ Linv
    r2 = kn.scaled_sqdist(X, X, ls, form)
    g = kn.g_of_r2(kernel, r2)
    hW = W * (variance * kn.h_of_r2(kernel, r2))
    d_variance = 0.5 * float(np.sum(W * g))
    d_ls = np.zeros(d)
    for k in range(d):
        diff = X[:, k][:, None] - X[:, k][None, :]
        d_ls[k] = -0.5 * float(np.sum(hW * diff * diff)) / ls[k] ** 3
    d_noise = 0.5 * float(np.trace(W))
    return value, d_variance, (d_ls if ard else float(d_ls.sum())), d_noise


def loss_and_grad(kernel, X, y, w_var, w_len, w_noise, mask=(True, True, True), form="direct"):
    """``-(LML + priors over trainable parameters)`` and its gradient in unconstrained variables."""
    variance, ls, noise = tr.constrain(w_var, w_len, w_noise)
    variance = float(variance)
    noise = float(noise)
    ls_arg = ls if np.ndim(w_len) > 0 else float(ls)
    value, d_v, d_l, d_s = lml_grads(kernel, X, y, variance, ls_arg, noise, form)
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
    g_var = -d_v * float(tr.softplus_grad(w_var)) if mask[0] else 0.0
    g_len = -np.asarray(d_l) * tr.softplus_grad(w_len) if mask[1] else np.zeros_like(np.asarray(w_len, dtype=np.float64))
    g_noise = -d_s * float(tr.softplus_grad(w_noise)) if mask[2] else 0.0
    if np.ndim(w_len) == 0:
        g_len = float(g_len)
    return -(value + logp), {"variance": g_var, "lengthscales": g_len, "noise": g_noise}


def loss(kernel, X, y, w_var, w_len, w_noise, mask=(True, True, True), form="direct"):
    variance, ls, noise = tr.constrain(w_var, w_len, w_noise)
    ls_arg = ls if np.ndim(w_len) > 0 else float(ls)
    value = lml(kernel, X, y, float(variance), ls_arg, float(noise), form)
    logp = 0.0
    if mask[0]:
        logp += float(tr.lognormal01_logpdf(variance))
    if mask[1]:
        logp += float(np.sum(tr.lognormal01_logpdf(ls)))
    if mask[2]:
        logp += float(tr.lognormal01_logpdf(noise))
    return -(value + logp)


def predict_from_factor(kernel, X, L, alpha, variance, lengthscales, noise, Xs, include_noise=True, form="direct"):
    """The predictive equations on a factorisation that already exists (``factorize``): what ``predict`` does after its own."""
    Ks = kn.kmat(kernel, X, Xs, variance, lengthscales, form)
    mean = Ks.T @ alpha
    V = solve_triangular(L, Ks, lower=True)
    var = variance - np.sum(V * V, axis=0)
    if include_noise:
        var = var + noise
    return mean, var


def predict(kernel, X, y, variance, lengthscales, noise, Xs, include_noise=True, form="direct"):
    L, alpha = factorize(kernel, X, y, variance, lengthscales, noise, form)
    return predict_from_factor(kernel, X, L, alpha, variance, lengthscales, noise, Xs, include_noise, form)
