import ctypes as C, sys, time
import numpy as np
sys.path.insert(0, ".")
from gpras_amd import _lib
from gpras_amd._lib import check, ptr
from gpras_amd.model import NOISE_LOWER, softplus_inv
from gpras_amd.synth import make_regression
lib = _lib.load()
n, d = int(sys.argv[1]), int(sys.argv[2])
x, y, _ = make_regression(n, d, n_outputs=1, n_test=0, config=5, unit=0)
h = C.c_void_p()
check(lib.gprx_create(0, n, d, 0, 0, 0, C.byref(h)))
check(lib.gprx_set_data(h, ptr(x), ptr(y), 1), h)
th = np.ascontiguousarray([softplus_inv(1.0), softplus_inv(np.mean(np.abs(x))), softplus_inv(1.0 - NOISE_LOWER)], dtype=np.float64)
loss = C.c_double()
res = []
for _ in range(8):
    t = time.perf_counter(); check(lib.gprx_factorize(h, 0, ptr(th), None, 7, C.byref(loss)), h); dt = time.perf_counter() - t
    ms = (C.c_double * 4)(); lib.gprx_last_timings(h, ms); res.append((dt * 1e3, ms[0], ms[1], ms[2]))
r = np.array(res[2:]); m = np.median(r, axis=0)
print(f"N={n}: wall {m[0]:.3f} ms | kernel build {m[1]:.3f} | cholesky {m[2]:.3f} | solves {m[3]:.3f}")
