"""Predict latency against the number of test points (N = 4096): which of the two paths should small batches take?"""
import ctypes as C, sys, time
import numpy as np
sys.path.insert(0, ".")
from gpras_amd import _lib
from gpras_amd._lib import DeviceBuffer, check, ptr
from gpras_amd.synth import make_regression
lib = _lib.load()
n, d = 4096, 8
x, y, xs = make_regression(n, d, n_outputs=1, n_test=40000, config=2, unit=0)
h = C.c_void_p()
check(lib.gprx_create(0, n, d, 0, 0, 0, C.byref(h)))
check(lib.gprx_set_data(h, ptr(x), ptr(y), 1), h)
theta = np.array([0.5413, 0.37, 0.5413])
loss = C.c_double()
dxs = DeviceBuffer.from_array(xs)
dm, dv = DeviceBuffer(8 * 40000), DeviceBuffer(8 * 40000)
for ns in (100, 1000, 4000, 8000, 8192, 12000, 20000, 40000):
    best = 1e9
    for rep in range(3):
        check(lib.gprx_factorize(h, 0, ptr(theta), None, 7, C.byref(loss)), h)  # fresh factorisation: L^-1 not cached
        t0 = time.perf_counter()
        check(lib.gprx_predict_dev(h, dxs.ptr, ns, dm.ptr, dv.ptr, 1), h)
        check(lib.gprx_synchronize(h), h)
        best = min(best, time.perf_counter() - t0)
    print(f"N*={ns:6d}: {best*1e3:8.3f} ms  {ns/best/1e6:6.3f} M pts/s", flush=True)
