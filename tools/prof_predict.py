"""rocprofv3 workload: one N=4096 factorisation, then predict at 100k resident points (3 times)."""
import ctypes as C
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from gpras_amd import _lib  # noqa: E402
from gpras_amd._lib import DeviceBuffer, check, ptr  # noqa: E402
from gpras_amd.synth import make_regression  # noqa: E402

lib = _lib.load()
n, d, ns = 4096, 8, 100_000
x, y, xs = make_regression(n, d, n_outputs=1, n_test=ns, config=2, unit=0)
h = C.c_void_p()
check(lib.gprx_create(0, n, d, 0, 0, 0, C.byref(h)))
check(lib.gprx_set_data(h, ptr(x), ptr(y), 1), h)
theta = np.array([0.5413, 0.37, 0.5413])
loss = C.c_double()
check(lib.gprx_factorize(h, 0, ptr(theta), None, 7, C.byref(loss)), h)
dxs = DeviceBuffer.from_array(xs)
dm, dv = DeviceBuffer(8 * ns), DeviceBuffer(8 * ns)
for rep in range(4):
    t0 = time.perf_counter()
    check(lib.gprx_predict_dev(h, dxs.ptr, ns, dm.ptr, dv.ptr, 1), h)
    check(lib.gprx_synchronize(h), h)
    dt = time.perf_counter() - t0
    print(f"predict {ns} pts: {dt*1e3:.2f} ms  {ns/dt/1e6:.3f} M pts/s  {(n*n*ns + 2.0*n*ns)/dt/1e12:.2f} TFLOP/s", flush=True)
