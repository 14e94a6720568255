"""Predict (mean + variance, device-resident) of ONE exact model at N* test points, for rocprofv3 (development aid).
    python3 tools/predict_probe.py N d [N*]      e.g. 1024 8, 4096 8, 16384 12 -- the sizes of bench.py's extra.other_sizes"""
import ctypes as C, sys, time
import numpy as np
sys.path.insert(0, ".")
from gpras_amd import _lib
from gpras_amd._lib import DeviceBuffer, check, ptr
from gpras_amd.model import NOISE_LOWER, softplus_inv
from gpras_amd.synth import make_regression
lib = _lib.load()
n, d = int(sys.argv[1]), int(sys.argv[2])
ns = int(sys.argv[3]) if len(sys.argv) > 3 else 100_000
x, y, xs = make_regression(n, d, n_outputs=1, n_test=ns, config=5, unit=0)
h = C.c_void_p()
check(lib.gprx_create(0, n, d, 0, 0, 0, C.byref(h)))
check(lib.gprx_set_data(h, ptr(x), ptr(y), 1), h)
theta = np.ascontiguousarray([softplus_inv(1.0), softplus_inv(np.mean(np.abs(x))), softplus_inv(1.0 - NOISE_LOWER)], dtype=np.float64)
loss = C.c_double()
check(lib.gprx_factorize(h, 0, ptr(theta), None, 7, C.byref(loss)), h)
dxs = DeviceBuffer.from_array(xs)
dm, dv = DeviceBuffer(8 * ns), DeviceBuffer(8 * ns)
best = 1e9
for rep in range(4):
    t0 = time.perf_counter()
    check(lib.gprx_predict_dev(h, dxs.ptr, ns, dm.ptr, dv.ptr, 1), h)
    check(lib.gprx_synchronize(h), h)
    if rep:
        best = min(best, time.perf_counter() - t0)
tf = (float(n) ** 2 * ns + 2.0 * n * ns) / best / 1e12
print(f"N={n} d={d} N*={ns}: {best*1e3:.3f} ms  {ns/best/1e6:.3f} M pts/s  {tf:.1f} TFLOP/s = {tf/78.6:.3f} of peak", flush=True)
lib.gprx_destroy(h)
