import ctypes as C, sys
import numpy as np
sys.path.insert(0, ".")
from gpras_amd import _lib
from gpras_amd._lib import check, ptr
from gpras_amd.synth import make_regression
lib = _lib.load()
for n in [int(a) for a in sys.argv[1:]]:
    x, y, _ = make_regression(n, 8, 1, 0, config=2, unit=0)
    h = C.c_void_p()
    check(lib.gprx_create(0, n, 8, 0, 0, 0, C.byref(h)))
    check(lib.gprx_set_data(h, ptr(x), ptr(y), 1), h)
    theta = np.array([0.5413, 0.37, 0.5413]); loss = C.c_double(); ms = (C.c_double * 4)()
    for ob in (256, 512, 1024, 2048):
        for tile in (64, 128):
            lib.gprx_set_tuning(b"outer_block", ob); lib.gprx_set_tuning(b"update_tile", tile)
            best = 1e9
            for _ in range(4):
                check(lib.gprx_factorize(h, 0, ptr(theta), None, 7, C.byref(loss)), h)
                lib.gprx_last_timings(h, ms); best = min(best, ms[1])
            print(f"N={n} ob {ob:4d} tile {tile:3d}: chol {best:7.3f} ms  {n**3/3/best/1e9:6.2f} TF/s", flush=True)
    lib.gprx_destroy(h)
