"""Timing of the sparse (SGPR) path at the reference's realistic sizes (development aid)."""
import ctypes as C, sys, time
import numpy as np
sys.path.insert(0, ".")
from gpras_amd import _lib
from gpras_amd._lib import check, ptr
from gpras_amd.synth import make_regression
lib = _lib.load()
for n, d, m in [(2000, 10, 50), (4000, 10, 50), (4000, 10, 100), (4000, 10, 300), (10000, 10, 300)]:
    x, y, xs = make_regression(n, d, n_outputs=1, n_test=2000, config=6, unit=0)
    z = np.ascontiguousarray(x[:m] + 0.01)
    h = C.c_void_p()
    check(lib.gprx_create(0, n, d, m, 0, 0, C.byref(h)))
    check(lib.gprx_set_data(h, ptr(x), ptr(y), 1), h)
    theta = np.array([0.5413, 0.37, 0.5413]); loss = C.c_double(); grad = np.zeros(3 + m * d)
    for _ in range(3):
        check(lib.gprx_objective(h, 0, ptr(theta), ptr(z), 15, C.byref(loss), ptr(grad)), h)
    t0 = time.perf_counter(); reps = 20
    for _ in range(reps):
        check(lib.gprx_objective(h, 0, ptr(theta), ptr(z), 15, C.byref(loss), ptr(grad)), h)
    t_obj = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps):
        check(lib.gprx_factorize(h, 0, ptr(theta), ptr(z), 15, C.byref(loss)), h)
    t_fac = (time.perf_counter() - t0) / reps
    mean = np.zeros(2000); var = np.zeros(2000)
    t0 = time.perf_counter()
    for _ in range(5):
        check(lib.gprx_predict(h, ptr(xs), 2000, ptr(mean), ptr(var), 1), h)
    t_pred = (time.perf_counter() - t0) / 5
    print(f"N={n} d={d} M={m}: loss+grad {t_obj*1e3:.3f} ms, loss only {t_fac*1e3:.3f} ms, predict 2000 pts {t_pred*1e3:.3f} ms", flush=True)
    lib.gprx_destroy(h)
