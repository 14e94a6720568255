"""In-kernel stamps of the tile-DAG factorisation (GPRX_DAG_STAMPS=1): chain phases per step and the wait / work split of the
workers.  s_memrealtime ticks are 10 ns."""
import ctypes as C
import os
import sys

import numpy as np

os.environ["GPRX_DAG_STAMPS"] = "1"
sys.path.insert(0, ".")
from gpras_amd import _build, _lib  # noqa: E402
from gpras_amd._lib import check, ptr  # noqa: E402
from gpras_amd.model import NOISE_LOWER, softplus_inv  # noqa: E402
from gpras_amd.synth import make_regression  # noqa: E402

_build.build()
lib = _lib.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
x, y, _ = make_regression(n, 8, n_outputs=1, n_test=0, config=2, unit=0)
theta = np.ascontiguousarray([softplus_inv(1.0), softplus_inv(np.mean(np.abs(x))), softplus_inv(1.0 - NOISE_LOWER)], dtype=np.float64)
check(lib.gprx_set_tuning(b"dag", 1))
h = C.c_void_p()
check(lib.gprx_create(0, n, 8, 0, _lib.KERNEL_IDS["RBF"], 0, C.byref(h)))
check(lib.gprx_set_data(h, ptr(x), ptr(y), 1), h)
loss = C.c_double()
for _ in range(3):
    check(lib.gprx_factorize(h, 0, ptr(theta), None, 7, C.byref(loss)), h)
lib.gprx_dag_stamps.restype = C.c_int
buf = (C.c_ulonglong * 20000)()
T, grid = C.c_int(), C.c_int()
nw = lib.gprx_dag_stamps(h, buf, 20000, C.byref(T), C.byref(grid))
a = np.array(buf[:nw], dtype=np.int64)
T, grid = T.value, grid.value
ch = a[: 4 * T].reshape(T, 4).astype(float)
t0 = ch[0, 0]
print("T", T, "grid", grid, "chain total us", (ch[-1, 2] - t0) / 100)
sub = ch[1:, 0] - ch[:-1, 3]   # substeps of step k (from the flags of the previous step)
p1 = ch[:, 1] - ch[:, 0]
p2 = ch[:, 2] - ch[:, 1]
p3 = ch[:-1, 3] - ch[:-1, 2]
for name, v in (("substeps(+publish)", sub), ("stores/late waits", p1), ("trsm", p2), ("syrk", p3)):
    v = v / 100
    print(f"{name:20s} mean {v.mean():7.2f} us  median {np.median(v):7.2f}  min {v.min():7.2f}  max {v.max():7.2f}  first8 {np.round(v[:8],1)}  last4 {np.round(v[-4:],1)}")
w = a[4 * T : 4 * T + 4 * grid].reshape(grid, 4).astype(float)
act = w[w[:, 2] > 0]
print("workers with tasks", len(act), "tasks", int(act[:, 2].sum()))
print("per task: wait us", act[:, 0].sum() / act[:, 2].sum() / 100, "work us", act[:, 1].sum() / act[:, 2].sum() / 100)
print("per worker: wait ms", act[:, 0].mean() / 1e5, "work ms", act[:, 1].mean() / 1e5, "tasks", act[:, 2].mean())
lib.gprx_destroy(h)
