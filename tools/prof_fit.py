"""Workload for rocprofv3: a few exact-GP evaluations at one size (development aid)."""
import ctypes as C
import sys

import numpy as np

sys.path.insert(0, ".")
from gpras_amd import _lib  # noqa: E402
from gpras_amd._lib import check, ptr  # noqa: E402
from gpras_amd.synth import make_regression  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
mode = sys.argv[2] if len(sys.argv) > 2 else "grad"
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
lib = _lib.load()
x, y, xs = make_regression(n, 8, n_outputs=1, n_test=8192, config=2, unit=0)
h = C.c_void_p()
check(lib.gprx_create(0, n, 8, 0, 0, 0, C.byref(h)))
check(lib.gprx_set_data(h, ptr(x), ptr(y), 1), h)
theta = np.array([0.5413, 0.37, 0.5413])
loss = C.c_double()
grad = np.zeros(3)
for _ in range(reps):
    if mode == "grad":
        check(lib.gprx_objective(h, 0, ptr(theta), None, 7, C.byref(loss), ptr(grad)), h)
    else:
        check(lib.gprx_factorize(h, 0, ptr(theta), None, 7, C.byref(loss)), h)
if mode == "predict":
    mean = np.zeros(xs.shape[0]); var = np.zeros(xs.shape[0])
    check(lib.gprx_predict(h, ptr(xs), xs.shape[0], ptr(mean), ptr(var), 1), h)
print("loss", loss.value)
lib.gprx_destroy(h)
