"""512 cells of N = 1024 (or argv: N cells) factorised by batched calls in a loop, for rocprofv3 (development aid).
GPRX_CELL_KERNEL=1 selects the one-workgroup-per-cell kernel."""
import ctypes as C, sys, time
import numpy as np
sys.path.insert(0, ".")
from gpras_amd import _lib
from gpras_amd._lib import check, ptr
from gpras_amd.model import NOISE_LOWER, softplus_inv
from gpras_amd.synth import make_regression
import os
lib = _lib.load()
if os.environ.get("GPRX_IGNORE_STATUS"):  # (timing experiments with builds whose results are wrong on purpose)
    check = lambda *a, **k: None  # noqa: E731
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
cells = int(sys.argv[2]) if len(sys.argv) > 2 else 512
x, y, _ = make_regression(n, 8, n_outputs=cells, n_test=0, config=2, unit=500)
theta = np.ascontiguousarray([softplus_inv(1.0), softplus_inv(np.mean(np.abs(x))), softplus_inv(1.0 - NOISE_LOWER)], dtype=np.float64)
thetas = np.ascontiguousarray(theta[None, :] + np.random.default_rng(7).uniform(-0.15, 0.15, size=(cells, 3)))
units = np.arange(cells, dtype=np.int32)
h = C.c_void_p()
check(lib.gprx_create(0, n, 8, 0, 0, 0, C.byref(h)))
check(lib.gprx_set_data(h, ptr(x), ptr(y), cells), h)
losses, status = np.zeros(cells), np.zeros(cells, dtype=np.int32)
for _ in range(2):
    check(lib.gprx_factorize_batch(h, cells, ptr(units), ptr(thetas), 7, ptr(losses), ptr(status)), h)
t0 = time.perf_counter()
for _ in range(5):
    check(lib.gprx_factorize_batch(h, cells, ptr(units), ptr(thetas), 7, ptr(losses), ptr(status)), h)
dt = (time.perf_counter() - t0) / 5
print(f"N={n} cells={cells}: {1e3*dt:.2f} ms per batch = {cells/dt:.0f} fits/s = {cells*n**3/3/dt/1e12:.1f} TFLOP/s", flush=True)
lib.gprx_destroy(h)
