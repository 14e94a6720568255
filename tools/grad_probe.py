import ctypes as C, sys, time
import numpy as np
sys.path.insert(0, ".")
from gpras_amd import _lib
from gpras_amd._lib import check, ptr
from gpras_amd.synth import make_regression
lib = _lib.load()
for n in (4096, 8192):
    x, y, _ = make_regression(n, 8, 1, 0, config=2, unit=0)
    h = C.c_void_p()
    check(lib.gprx_create(0, n, 8, 0, 0, 0, C.byref(h)))
    check(lib.gprx_set_data(h, ptr(x), ptr(y), 1), h)
    theta = np.array([0.5413, 0.37, 0.5413]); loss = C.c_double(); grad = np.zeros(3); ms = (C.c_double * 4)()
    for tile in (64, 128):
        lib.gprx_set_tuning(b"update_tile", tile)
        best = 1e9
        for _ in range(4):
            check(lib.gprx_objective(h, 0, ptr(theta), None, 7, C.byref(loss), ptr(grad)), h)
            lib.gprx_last_timings(h, ms); best = min(best, ms[3])
        print(f"N={n} tile {tile}: grad stage {best:.3f} ms grad {grad}", flush=True)
    lib.gprx_set_tuning(b"update_tile", 0)
    lib.gprx_destroy(h)
