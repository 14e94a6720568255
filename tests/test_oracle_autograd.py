"""The oracle's hand-derived gradients against reverse-mode automatic differentiation.

gpflow obtains every gradient the reference's optimisers consume from TensorFlow's autodiff of ``training_loss``
(/root/reference/gpras/gpr.py:151-157, :183-188, :195-203).  ``oracle/sgpr.py`` and ``oracle/exact.py`` derive them by hand (and
the HIP kernels follow those formulas), pinned so far by central differences on a handful of coordinates at 2e-6.  Here the
same training loss is written once more in PyTorch float64 -- straight from the published model equations (Titsias' collapsed
bound with gpflow's jitter, softplus transforms, 1e-6 noise floor and LogNormal(0, 1) priors on the trainable parameters;
``r = sqrt(max(r2, 1e-36))`` with the gradient stopped at the floor as ``tf.maximum`` does) -- and differentiated by autograd: the
WHOLE gradient, every coordinate of Z included, must agree to 1e-9.  No formula of the hand derivation appears below.
"""

import numpy as np
import pytest
import torch

from gpras_amd.synth import make_regression
from oracle import exact, sgpr
from oracle import kernels as kn

torch.set_default_dtype(torch.float64)
SQRT3, SQRT5 = 3.0**0.5, 5.0**0.5


def _softplus(w):
    return torch.nn.functional.softplus(w)


def _lognormal01_logpdf(p):
    return -torch.log(p) - 0.5 * np.log(2.0 * np.pi) - 0.5 * torch.log(p) ** 2


def _kmat(kernel, a, b, v, ls):
    sa, sb = a / ls, b / ls
    diff = sa[:, None, :] - sb[None, :, :]
    r2 = (diff * diff).sum(-1)
    if kernel == "RBF":
        return v * torch.exp(-0.5 * r2)
    r = torch.sqrt(torch.clamp(r2, min=1e-36))  # clamp: zero gradient where the floor is active, as tf.maximum
    if kernel == "Matern12":
        return v * torch.exp(-r)
    if kernel == "Matern32":
        return v * (1.0 + SQRT3 * r) * torch.exp(-SQRT3 * r)
    if kernel == "Matern52":
        return v * (1.0 + SQRT5 * r + (5.0 / 3.0) * r * r) * torch.exp(-SQRT5 * r)
    if kernel == "Exponential":
        return v * torch.exp(-0.5 * r)
    raise KeyError(kernel)


def _constrained(w_var, w_len, w_noise):
    return _softplus(w_var), _softplus(w_len), 1e-6 + _softplus(w_noise)


def _sgpr_loss(kernel, x, y, z, w_var, w_len, w_noise, mask):
    v, ls, s = _constrained(w_var, w_len, w_noise)
    n, m = x.shape[0], z.shape[0]
    kuf = _kmat(kernel, z, x, v, ls)
    kuu = _kmat(kernel, z, z, v, ls) + 1e-6 * torch.eye(m)
    # Titsias (2009), eq. 9, written densely: log N(y | 0, Qff + s I) - tr(Kff - Qff) / (2 s)
    qff = kuf.T @ torch.linalg.solve(kuu, kuf)
    cov = qff + s * torch.eye(n)
    elbo = -0.5 * (n * np.log(2.0 * np.pi) + torch.logdet(cov) + y @ torch.linalg.solve(cov, y)) - (n * v - torch.trace(qff)) / (2.0 * s)
    logp = 0.0
    if mask[0]:
        logp = logp + _lognormal01_logpdf(v)
    if mask[1]:
        logp = logp + _lognormal01_logpdf(ls).sum()
    if mask[2]:
        logp = logp + _lognormal01_logpdf(s)
    return -(elbo + logp)


def _exact_loss(kernel, x, y, w_var, w_len, w_noise):
    v, ls, s = _constrained(w_var, w_len, w_noise)
    n = x.shape[0]
    cov = _kmat(kernel, x, x, v, ls) + s * torch.eye(n)
    lml = -0.5 * (n * np.log(2.0 * np.pi) + torch.logdet(cov) + y @ torch.linalg.solve(cov, y))
    return -(lml + _lognormal01_logpdf(v) + _lognormal01_logpdf(ls).sum() + _lognormal01_logpdf(s))


def _data(n, d, m, seed):
    x, y, _ = make_regression(n, d, 1, 0, config=9, unit=seed)
    rng = np.random.default_rng(seed)
    z = x[rng.choice(n, m, replace=False)] + 0.05 * rng.standard_normal((m, d))
    return x, y[:, 0], z


@pytest.mark.parametrize("kernel", kn.KERNEL_NAMES)
@pytest.mark.parametrize("ard", [False, True])
def test_sparse_loss_and_whole_gradient_against_autograd(kernel, ard):
    x, y, z = _data(90, 3, 12, 5)
    wv, wn = 0.3, -1.2
    wl = np.array([0.2, -0.1, 0.4]) if ard else 0.25
    loss, g = sgpr.loss_and_grad(kernel, x, y, z, wv, wl, wn)
    tz = torch.tensor(z, requires_grad=True)
    tv, tn = torch.tensor(wv, requires_grad=True), torch.tensor(wn, requires_grad=True)
    tl = torch.tensor(wl, requires_grad=True)
    tloss = _sgpr_loss(kernel, torch.tensor(x), torch.tensor(y), tz, tv, tl, tn, (True, True, True))
    tloss.backward()
    assert loss == pytest.approx(tloss.item(), rel=1e-10)
    scale = max(1.0, float(tz.grad.abs().max()))
    assert g["variance"] == pytest.approx(tv.grad.item(), rel=1e-8, abs=1e-9)
    assert g["noise"] == pytest.approx(tn.grad.item(), rel=1e-8, abs=1e-9)
    np.testing.assert_allclose(np.atleast_1d(g["lengthscales"]), np.atleast_1d(tl.grad.numpy()), rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(g["Z"], tz.grad.numpy(), rtol=1e-7, atol=1e-9 * scale)


def test_sparse_masked_gradient_against_autograd():
    """Z-only stage of the two-stage driver (gpr.py:115-118): priors of frozen parameters leave the loss, their gradients are 0."""
    x, y, z = _data(70, 2, 9, 8)
    mask = (False, False, False, True)
    loss, g = sgpr.loss_and_grad("Matern32", x, y, z, 0.1, 0.3, -0.5, mask=mask)
    tz = torch.tensor(z, requires_grad=True)
    tloss = _sgpr_loss("Matern32", torch.tensor(x), torch.tensor(y), tz, torch.tensor(0.1), torch.tensor(0.3), torch.tensor(-0.5), mask[:3])
    tloss.backward()
    assert loss == pytest.approx(tloss.item(), rel=1e-10)
    assert g["variance"] == 0.0 and g["noise"] == 0.0 and g["lengthscales"] == 0.0
    np.testing.assert_allclose(g["Z"], tz.grad.numpy(), rtol=1e-7, atol=1e-10)


@pytest.mark.parametrize("kernel", kn.KERNEL_NAMES)
def test_exact_loss_and_gradient_against_autograd(kernel):
    x, y, _ = _data(80, 3, 4, 6)
    wv, wn, wl = 0.4, -0.8, np.array([0.3, 0.0, -0.2])
    loss, g = exact.loss_and_grad(kernel, x, y, wv, wl, wn)
    tv, tn, tl = torch.tensor(wv, requires_grad=True), torch.tensor(wn, requires_grad=True), torch.tensor(wl, requires_grad=True)
    tloss = _exact_loss(kernel, torch.tensor(x), torch.tensor(y), tv, tl, tn)
    tloss.backward()
    assert loss == pytest.approx(tloss.item(), rel=1e-10)
    assert g["variance"] == pytest.approx(tv.grad.item(), rel=1e-8, abs=1e-9)
    assert g["noise"] == pytest.approx(tn.grad.item(), rel=1e-8, abs=1e-9)
    np.testing.assert_allclose(g["lengthscales"], tl.grad.numpy(), rtol=1e-8, atol=1e-9)
