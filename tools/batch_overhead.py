"""Host-side overhead of gprx_factorize_batch: wall time per call against the device time between the call's first and last event
(gprx_last_batch_ms) -- development aid.  argv: N cells"""
import ctypes as C, sys, time
import numpy as np
sys.path.insert(0, ".")
from gpras_amd import _lib
from gpras_amd._lib import check, ptr
from gpras_amd.model import NOISE_LOWER, softplus_inv
from gpras_amd.synth import make_regression
lib = _lib.load()
n, cells = int(sys.argv[1]), int(sys.argv[2])
x, y, _ = make_regression(n, 8, n_outputs=cells, n_test=0, config=2, unit=500)
theta = np.ascontiguousarray([softplus_inv(1.0), softplus_inv(np.mean(np.abs(x))), softplus_inv(1.0 - NOISE_LOWER)], dtype=np.float64)
thetas = np.ascontiguousarray(theta[None, :] + np.random.default_rng(7).uniform(-0.15, 0.15, size=(cells, 3)))
units = np.arange(cells, dtype=np.int32)
h = C.c_void_p()
check(lib.gprx_create(0, n, 8, 0, 0, 0, C.byref(h)))
check(lib.gprx_set_data(h, ptr(x), ptr(y), cells), h)
losses, status = np.zeros(cells), np.zeros(cells, dtype=np.int32)
for _ in range(3):
    check(lib.gprx_factorize_batch(h, cells, ptr(units), ptr(thetas), 7, ptr(losses), ptr(status)), h)
wall, dev = [], []
ms = C.c_double()
for _ in range(20):
    t0 = time.perf_counter()
    check(lib.gprx_factorize_batch(h, cells, ptr(units), ptr(thetas), 7, ptr(losses), ptr(status)), h)
    wall.append(time.perf_counter() - t0)
    check(lib.gprx_last_batch_ms(h, C.byref(ms)), h)
    dev.append(ms.value * 1e-3)
print(f"N={n} cells={cells}: wall {1e3*np.median(wall):.3f} ms, device {1e3*np.median(dev):.3f} ms, host overhead {1e6*(np.median(wall)-np.median(dev)):.0f} us per call", flush=True)
lib.gprx_destroy(h)
