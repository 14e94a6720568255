"""Single large exact fit: time of gprx_factorize at N (development aid).  argv: N d [outer_block]"""
import ctypes as C, sys, time
import numpy as np
sys.path.insert(0, ".")
from gpras_amd import _lib
from gpras_amd._lib import check, ptr
from gpras_amd.model import NOISE_LOWER, softplus_inv
from gpras_amd.synth import make_regression
lib = _lib.load()
n, d = int(sys.argv[1]), int(sys.argv[2])
if len(sys.argv) > 3:
    check(lib.gprx_set_tuning(b"outer_block", int(sys.argv[3])))
x, y, _ = make_regression(n, d, n_outputs=1, n_test=0, config=5, unit=0)
h = C.c_void_p()
check(lib.gprx_create(0, n, d, 0, 0, 0, C.byref(h)))
check(lib.gprx_set_data(h, ptr(x), ptr(y), 1), h)
th = np.ascontiguousarray([softplus_inv(1.0), softplus_inv(np.mean(np.abs(x))), softplus_inv(1.0 - NOISE_LOWER)], dtype=np.float64)
loss = C.c_double()
check(lib.gprx_factorize(h, 0, ptr(th), None, 7, C.byref(loss)), h)
ts = []
for _ in range(4):
    t = time.perf_counter(); check(lib.gprx_factorize(h, 0, ptr(th), None, 7, C.byref(loss)), h); ts.append(time.perf_counter() - t)
ms = (C.c_double * 4)(); lib.gprx_last_timings(h, ms)
best = min(ts)
print(f"N={n} d={d}: fit {best*1e3:.2f} ms, cholesky {ms[1]:.2f} ms = {n**3/3/(ms[1]*1e-3)/1e12:.1f} TF/s ({n**3/3/(ms[1]*1e-3)/1e12/78.6:.3f} of 78.6), loss {loss.value:.10f}", flush=True)
lib.gprx_destroy(h)
