"""Phase durations of workgroup 0 of the column-pair cell kernel summed over launches (needs the -DGPRX_CELL_ACC build, tools/cell_acc.sh).
argv: N cells"""
import ctypes as C, sys
import numpy as np
sys.path.insert(0, ".")
from gpras_amd import _lib
from gpras_amd._lib import check, ptr
from gpras_amd.model import NOISE_LOWER, softplus_inv
from gpras_amd.synth import make_regression
lib = C.CDLL(sys.argv[3]) if len(sys.argv) > 3 else None
base = _lib.load() if lib is None else None
n, cells = int(sys.argv[1]), int(sys.argv[2])
L = lib or base
for name in ("gprx_create", "gprx_set_data", "gprx_factorize_batch", "gprx_destroy"):
    getattr(L, name).restype = C.c_int
x, y, _ = make_regression(n, 8, n_outputs=cells, n_test=0, config=2, unit=500)
theta = np.ascontiguousarray([softplus_inv(1.0), softplus_inv(np.mean(np.abs(x))), softplus_inv(1.0 - NOISE_LOWER)], dtype=np.float64)
thetas = np.ascontiguousarray(theta[None, :] + np.random.default_rng(7).uniform(-0.15, 0.15, size=(cells, 3)))
units = np.arange(cells, dtype=np.int32)
h = C.c_void_p()
vp = C.c_void_p
assert L.gprx_create(0, C.c_int64(n), 8, C.c_int64(0), 0, 0, C.byref(h)) == 0
assert L.gprx_set_data(h, x.ctypes.data_as(vp), y.ctypes.data_as(vp), cells) == 0
losses, status = np.zeros(cells), np.zeros(cells, dtype=np.int32)
acc = (C.c_ulonglong * 16)()
for rep in range(3):
    if rep == 1:
        L.gprx_cell_acc(acc, 1)
    assert L.gprx_factorize_batch(h, cells, units.ctypes.data_as(vp), thetas.ctypes.data_as(vp), 7, losses.ctypes.data_as(vp), status.ctypes.data_as(vp)) == 0
L.gprx_cell_acc(acc, 0)
a = np.array(acc[:8], dtype=np.float64)
us = a[1:] / a[0] / 100.0
names = ["pair stream", "pair chains+solves", "beta", "rows stream", "E1", "E2", "E3"]
print(f"N={n} cells={cells}: per launch of workgroup 0 (us): " + " | ".join(f"{k} {v:.0f}" for k, v in zip(names, us)) + f" | sum {us.sum():.0f}", flush=True)
L.gprx_destroy(h)
