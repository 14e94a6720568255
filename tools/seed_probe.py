import ctypes as C, sys
import numpy as np
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from test_gpu_random_sweep import draw
from gpras_amd import _lib
from gpras_amd._lib import check, ptr
from gpras_amd.synth import make_regression
from oracle import kernels as okn, sgpr as osg, transforms as otr
lib = _lib.load()
seed = int(sys.argv[1])
rng, kernel, n, d, ard, units = draw(2000 + seed)
n = max(n, 8)
m = int(rng.integers(1, min(n, 130) + 1))
print(kernel, n, d, ard, units, m)
x, y, xs = make_regression(n, d, n_outputs=units, n_test=9, config=21, unit=seed)
nl = d if ard else 1
h = C.c_void_p()
check(lib.gprx_create(0, n, d, m, okn.KERNEL_IDS[kernel], int(ard), C.byref(h)))
check(lib.gprx_set_data(h, ptr(x), ptr(y), units), h)
unit = int(rng.integers(units))
variance, ls, noise = float(rng.uniform(0.3, 3.0)), rng.uniform(0.5, 2.0, nl), float(10.0 ** rng.uniform(-2.0, 0.0))
print("var ls noise", variance, ls, noise)
wv, wl, wn = otr.unconstrain(variance, ls, noise)
theta = np.ascontiguousarray(np.concatenate([[wv], np.atleast_1d(wl), [wn]]))
z = np.ascontiguousarray(x[rng.choice(n, size=m, replace=False)] + 1e-3 * rng.standard_normal((m, d)))
loss, grad = C.c_double(), np.zeros(2 + nl + m * d)
check(lib.gprx_objective(h, unit, ptr(theta), ptr(z), 15, C.byref(loss), ptr(grad)), h)
wl_arg = theta[1:-1] if ard else float(theta[1])
ref_loss, g = osg.loss_and_grad(kernel, x, y[:, unit], z, float(theta[0]), wl_arg, float(theta[-1]))
ref = np.concatenate([[g["variance"]], np.atleast_1d(g["lengthscales"]), [g["noise"]], np.asarray(g["Z"]).ravel()])
print("loss", loss.value, ref_loss, abs(loss.value - ref_loss) / max(abs(ref_loss), 1))
err = np.abs(grad - ref)
print("grad max err", err.max(), "at", err.argmax(), "ref scale", np.abs(ref).max(), "theta part err", err[: 2 + nl], "ref theta", ref[: 2 + nl])
mean, var = np.zeros(9), np.zeros(9)
check(lib.gprx_predict(h, ptr(xs), 9, ptr(mean), ptr(var), 1), h)
lsc = ls if ard else float(ls[0])
rm, rv = osg.predict(kernel, x, y[:, unit], z, variance, lsc, noise, xs)
print("pred err", np.max(np.abs(mean - rm)) / max(np.max(np.abs(rm)), 1e-3), np.max(np.abs(var - rv) / rv))
lsc0 = ls if ard else float(ls[0])
Kuu = okn.kmat(kernel, z, z, variance, lsc0) + 1e-6 * np.eye(m)
print("cond Kuu", np.linalg.cond(Kuu))
# which gradient is closer to central differences of the ORACLE loss?
def f(th):
    return osg.loss_and_grad(kernel, x, y[:, unit], z, float(th[0]), (th[1:-1] if ard else float(th[1])), float(th[-1]))[0]
for k in range(2 + nl):
    e = np.zeros_like(theta); e[k] = 1e-5
    fd = (f(theta + e) - f(theta - e)) / 2e-5
    print("theta", k, "fd", fd, "oracle", ref[k], "gpu", grad[k])
