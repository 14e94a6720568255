"""A/B of two builds of the GEMM kernel in one process (interleaved rounds)."""
import ctypes as C, sys, time
import numpy as np
libs = {"base": C.CDLL("tools/libgprx_base.so"), "variant": C.CDLL(sys.argv[1])}
vp = C.c_void_p
for l in libs.values():
    l.gprx_dev_malloc.argtypes = [C.c_int, C.c_int64, C.POINTER(vp)]
    l.gprx_memcpy_h2d.argtypes = [C.c_int, vp, vp, C.c_int64]
    l.gprx_gemm.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_double, vp, C.c_int64, vp, C.c_int64, C.c_double, vp, C.c_int64, C.c_int, C.c_int]
base = libs["base"]
cases = [(4096, 4096, 4096, 0, 128, "dense 4096^3 t128"), (4096, 4096, 1024, 1, 64, "syrk K=1024 t64"), (4096, 4096, 256, 1, 64, "syrk K=256 t64"), (8192, 8192, 512, 1, 128, "syrk 8192 K=512 t128")]
rng = np.random.default_rng(0)
for m, n, k, flags, tile, name in cases:
    a = rng.standard_normal((m, k))
    dA = vp(); dC = vp()
    base.gprx_dev_malloc(0, a.nbytes, C.byref(dA)); base.gprx_dev_malloc(0, m * n * 8, C.byref(dC))
    base.gprx_memcpy_h2d(0, dA, a.ctypes.data_as(vp), a.nbytes)
    res = {k_: [] for k_ in libs}
    for rnd in range(6):
        for key, l in libs.items():
            t0 = time.perf_counter()
            l.gprx_gemm(0, 0, 1, m, n, k, -1.0, dA, k, dA, k, 0.0, dC, n, flags, tile)
            res[key].append(time.perf_counter() - t0)
    fl = 2.0 * m * n * k * (0.5 if flags & 1 else 1.0)
    print(name, {key: f"{fl/min(v)/1e12:.2f} TF/s" for key, v in res.items()}, flush=True)
