"""Quick stage timings on one MI355X (development aid; bench.py is the judged measurement)."""

import ctypes as C
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from gpras_amd import _lib  # noqa: E402
from gpras_amd._lib import DeviceBuffer, check, ptr  # noqa: E402
from gpras_amd.synth import make_regression  # noqa: E402

lib = _lib.load()
tf = C.c_double()
check(lib.gprx_mfma_f64_peak(0, C.byref(tf)))
print(f"mfma_f64 16x16x4 back-to-back: {tf.value:.2f} TFLOP/s", flush=True)


def time_call(fn, reps=5):
    fn()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        best = min(best, time.perf_counter() - t0)
    return best


# GEMM: syrk-shaped (K = 64) and square
for (m, n, k, flags, tile, name) in [
    (4096, 4096, 64, 1, 128, "syrk K=64 t128"),
    (4096, 4096, 64, 1, 64, "syrk K=64 t64"),
    (4096, 4096, 256, 1, 128, "syrk K=256 t128"),
    (4096, 4096, 4096, 0, 128, "gemm NT 4096^3 t128"),
    (8192, 8192, 512, 1, 128, "syrk 8192 K=512"),
]:
    rng = np.random.default_rng(0)
    a = rng.standard_normal((m, k))
    dA = DeviceBuffer.from_array(a)
    dC = DeviceBuffer(m * n * 8)
    t = time_call(lambda: check(lib.gprx_gemm(0, 0, 1, m, n, k, -1.0, dA.ptr, k, dA.ptr, k, 0.0, dC.ptr, n, flags, tile)))
    fl = 2.0 * m * n * k * (0.5 if flags & 1 else 1.0)
    print(f"{name:24s} {t*1e3:8.3f} ms  {fl/t/1e12:6.2f} TFLOP/s", flush=True)
    dA.free()
    dC.free()

for n, d in [(1024, 8), (4096, 8), (8192, 8), (16384, 12)]:
    x, y, xs = make_regression(n, d, n_outputs=1, n_test=20000, config=2, unit=0)
    h = C.c_void_p()
    check(lib.gprx_create(0, n, d, 0, 0, 0, C.byref(h)))
    check(lib.gprx_set_data(h, ptr(x), ptr(y), 1), h)
    theta = np.array([0.5413, 0.37, 0.5413])
    loss = C.c_double()
    ms = (C.c_double * 4)()
    best = None
    for _ in range(4):
        t0 = time.perf_counter()
        check(lib.gprx_factorize(h, 0, ptr(theta), None, 7, C.byref(loss)), h)
        wall = time.perf_counter() - t0
        lib.gprx_last_timings(h, ms)
        cur = (wall, ms[0], ms[1], ms[2])
        best = cur if best is None or cur[0] < best[0] else best
    print(f"N={n:6d} factorize wall {best[0]*1e3:8.3f} ms | kmat {best[1]:.3f} chol {best[2]:.3f} solve {best[3]:.3f} ms"
          f" | chol {n**3/3/best[2]/1e9:.2f} TFLOP/s", flush=True)
    if n <= 8192:
        grad = np.zeros(3)
        t0 = time.perf_counter()
        check(lib.gprx_objective(h, 0, ptr(theta), None, 7, C.byref(loss), ptr(grad)), h)
        wall = time.perf_counter() - t0
        lib.gprx_last_timings(h, ms)
        print(f"          objective+grad wall {wall*1e3:8.3f} ms | grad stage {ms[3]:.3f} ms", flush=True)
    mean = np.zeros(xs.shape[0])
    var = np.zeros(xs.shape[0])
    check(lib.gprx_factorize(h, 0, ptr(theta), None, 7, C.byref(loss)), h)
    t = time_call(lambda: check(lib.gprx_predict(h, ptr(xs), xs.shape[0], ptr(mean), ptr(var), 1), h), reps=2)
    print(f"          predict {xs.shape[0]} pts: {t*1e3:.2f} ms -> {xs.shape[0]/t:,.0f} pts/s, {n*n*xs.shape[0]/t/1e12:.2f} TFLOP/s", flush=True)
    lib.gprx_destroy(h)
