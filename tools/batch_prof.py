"""Batched exact fits at size N with `cells` cells per launch sequence: timing loop for rocprofv3 runs (development aid).
argv: N cells [steps] [key=value tuning ...]"""
import ctypes as C, sys, time
import numpy as np
sys.path.insert(0, ".")
from gpras_amd import _lib
from gpras_amd._lib import check, ptr
from gpras_amd.model import NOISE_LOWER, softplus_inv
from gpras_amd.synth import make_regression
lib = _lib.load()
n, cells = int(sys.argv[1]), int(sys.argv[2])
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
for kv in sys.argv[4:]:  # tuning: key=value
    k, v = kv.split("=")
    check(lib.gprx_set_tuning(k.encode(), int(v)))
x, y, _ = make_regression(n, 8, n_outputs=cells, n_test=0, config=2, unit=500)
h = C.c_void_p()
check(lib.gprx_create(0, n, 8, 0, 0, 0, C.byref(h)))
check(lib.gprx_set_data(h, ptr(x), ptr(y), cells), h)
theta = np.array([softplus_inv(1.0), softplus_inv(np.mean(np.abs(x))), softplus_inv(1.0 - NOISE_LOWER)])
thetas = np.ascontiguousarray(theta[None, :] + np.random.default_rng(1).uniform(-0.15, 0.15, size=(cells, 3)))
units = np.arange(cells, dtype=np.int32)
losses, status = np.zeros(cells), np.zeros(cells, dtype=np.int32)
for _ in range(2):
    check(lib.gprx_factorize_batch(h, cells, ptr(units), ptr(thetas), 7, ptr(losses), ptr(status)), h)
t = time.perf_counter()
for _ in range(steps):
    check(lib.gprx_factorize_batch(h, cells, ptr(units), ptr(thetas), 7, ptr(losses), ptr(status)), h)
dt = (time.perf_counter() - t) / steps
print(f"N={n} cells={cells}: {dt*1e3:.2f} ms per step = {cells/dt:.0f} fits/s = {cells*n**3/3/dt/1e12:.1f} TF/s")
lib.gprx_destroy(h)
