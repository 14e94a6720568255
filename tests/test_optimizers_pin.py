"""Independent pins of the first-order drivers (gpras_amd/optimizers.py): the update rules are checked against PyTorch's
implementations of the same published algorithms, not against this repository's own restatement in oracle/.

The reference calls ``tf.keras.optimizers.Adam()`` / ``Adadelta()`` with default arguments (/root/reference/gpras/gpr.py:149,
:178): learning rate 1e-3, (beta1, beta2) = (0.9, 0.999) or rho = 0.95, epsilon 1e-7.

* Adadelta: Keras and ``torch.optim.Adadelta`` state the same recurrences (epsilon inside both square roots), so the
  trajectories must agree to rounding.
* Adam: Keras folds the bias corrections into the step size and adds epsilon to sqrt(v) (``alpha_t = lr sqrt(1 - b2^t) / (1 - b1^t)``,
  ``x -= alpha_t m / (sqrt(v) + eps)``); PyTorch adds epsilon to sqrt(v / (1 - b2^t)).  The two differ only through epsilon
  (1e-7 against gradients of order one): the trajectories must agree to ~1e-6, which pins learning rate, both betas and
  the bias correction; the placement of epsilon is pinned by a hand-computed first step.
"""

import numpy as np
import pytest
import torch

from gpras_amd import optimizers


class _Quartic:
    """loss(x) = sum_i a_i (x_i - c_i)^2 + 0.1 (x_i - c_i)^4, gradient analytic; the driver-facing surface of GPModel."""

    def __init__(self, x0):
        self.x = np.asarray(x0, dtype=np.float64).copy()
        n = self.x.size
        self.a = np.linspace(0.5, 3.0, n)
        self.c = np.linspace(-1.0, 1.0, n)
        self.n_evals = 0

    def get_vector(self):
        return self.x.copy()

    def set_vector(self, v):
        self.x = np.asarray(v, dtype=np.float64).copy()

    @staticmethod
    def f(x, a, c):
        return (a * (x - c) ** 2 + 0.1 * (x - c) ** 4).sum()

    def loss_and_grad(self):
        self.n_evals += 1
        e = self.x - self.c
        return float(self.f(self.x, self.a, self.c)), 2.0 * self.a * e + 0.4 * e**3


def _torch_run(opt_factory, x0, a, c, steps):
    x = torch.tensor(x0, dtype=torch.float64, requires_grad=True)
    ta, tc = torch.tensor(a), torch.tensor(c)
    opt = opt_factory([x])
    for _ in range(steps):
        opt.zero_grad()
        loss = (ta * (x - tc) ** 2 + 0.1 * (x - tc) ** 4).sum()
        loss.backward()
        opt.step()
    return x.detach().numpy()


X0 = np.array([2.0, -1.5, 0.3, 4.0, -0.2, 1.1])


def test_adadelta_driver_follows_the_published_recurrences():
    model = _Quartic(X0)
    optimizers._optimize_adadelta(model, 300)
    want = _torch_run(lambda p: torch.optim.Adadelta(p, lr=1e-3, rho=0.95, eps=1e-7), X0, model.a, model.c, 300)
    assert model.n_evals == 300  # exactly max_iter steps, no early stop (gpr.py:181-190)
    np.testing.assert_allclose(model.x, want, rtol=1e-11, atol=1e-13)
    assert np.abs(model.x - X0).max() > 1e-4  # (it moved: the comparison is not between two untouched starting points)


def test_adam_driver_follows_the_published_recurrences():
    model = _Quartic(X0)
    optimizers._optimize_adam(model, 200)
    want = _torch_run(lambda p: torch.optim.Adam(p, lr=1e-3, betas=(0.9, 0.999), eps=1e-7), X0, model.a, model.c, 200)
    assert model.n_evals == 200  # the loss improves by more than 1e-5 relative at every step here: no early stop
    np.testing.assert_allclose(model.x, want, rtol=2e-6, atol=2e-7)
    assert np.abs(model.x - X0).max() > 0.1


def test_adam_first_step_has_keras_epsilon_placement():
    """t = 1: m = 0.1 g, v = 0.001 g^2, alpha = lr sqrt(0.001) / 0.1; Keras: x - alpha m / (sqrt(v) + eps)."""
    model = _Quartic(X0)
    _, g = model.loss_and_grad()
    optimizers._optimize_adam(model, 1)
    alpha = 1e-3 * np.sqrt(1.0 - 0.999) / (1.0 - 0.9)
    keras = X0 - alpha * (0.1 * g) / (np.sqrt(0.001 * g * g) + 1e-7)
    np.testing.assert_allclose(model.x, keras, rtol=1e-15, atol=0)
    torch_like = X0 - (1e-3 / 0.1) * (0.1 * g) / (np.sqrt(0.001 * g * g) / np.sqrt(0.001) + 1e-7)
    assert np.abs(keras - torch_like).max() > 0  # the two placements are distinguishable at this precision


def test_adam_early_stop_is_the_reference_rule():
    """gpr.py:160-171: stop once the relative improvement stayed <= 1e-5 for more than 50 consecutive steps."""

    class Flat(_Quartic):
        def loss_and_grad(self):
            self.n_evals += 1
            return 1.0, np.zeros_like(self.x)

    model = Flat(X0)
    optimizers._optimize_adam(model, 500)
    # step 1 improves on best = inf; steps 2..52 do not (count 1..51 > patience at 51): 52 evaluations
    assert model.n_evals == 52


@pytest.mark.parametrize("many", [2, 5])
def test_batched_adam_equals_the_single_driver(many):
    """``_optimize_adam_many`` without a batching backend falls back to per-model evaluations: same trajectories bit for bit."""

    class M(_Quartic):
        backend = object()
        mask = 7
        unit = 0
        Z = None

        def theta(self):
            return self.x

        def _pack_grad(self, g):
            return g

    starts = [X0 + 0.1 * k for k in range(many)]
    singles = [_Quartic(s) for s in starts]
    for s in singles:
        optimizers._optimize_adam(s, 40)

    models = [M(s) for s in starts]

    def fake_eval(ms, want_grad=True, stats=None):
        out = [m.loss_and_grad() for m in ms]
        return np.array([o[0] for o in out]), [o[1] for o in out]

    orig = optimizers._evaluate_many
    optimizers._evaluate_many = fake_eval
    try:
        optimizers._optimize_adam_many(models, 40)
    finally:
        optimizers._evaluate_many = orig
    for a, b in zip(models, singles):
        np.testing.assert_array_equal(a.x, b.x)
