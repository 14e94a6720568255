"""A/B of the register-staged and the LDS-DMA operand staging of the NT GEMM in one process (interleaved rounds): the same
library loaded twice (two copies of the file), one with GPRX_GEMM_DMA=1 read at its first launch."""
import ctypes as C, os, shutil, sys, time
import numpy as np
os.environ["GPRX_GEMM_DMA"] = "0"
base = C.CDLL("gpras_amd/libgprx.so")
shutil.copy("gpras_amd/libgprx.so", "/tmp/libgprx_dma.so")
vp = C.c_void_p
def proto(l):
    l.gprx_dev_malloc.argtypes = [C.c_int, C.c_int64, C.POINTER(vp)]
    l.gprx_memcpy_h2d.argtypes = [C.c_int, vp, vp, C.c_int64]
    l.gprx_memcpy_d2h.argtypes = [C.c_int, vp, vp, C.c_int64]
    l.gprx_gemm.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_double, vp, C.c_int64, vp, C.c_int64, C.c_double, vp, C.c_int64, C.c_int, C.c_int]
proto(base)
rng = np.random.default_rng(0)
a0 = rng.standard_normal((128, 64))
dA = vp(); dC = vp()
base.gprx_dev_malloc(0, a0.nbytes, C.byref(dA)); base.gprx_dev_malloc(0, 128 * 128 * 8, C.byref(dC))
base.gprx_memcpy_h2d(0, dA, a0.ctypes.data_as(vp), a0.nbytes)
base.gprx_gemm(0, 0, 1, 128, 128, 64, 1.0, dA, 64, dA, 64, 0.0, dC, 128, 0, 64)   # first launch: reads GPRX_GEMM_DMA=0
os.environ["GPRX_GEMM_DMA"] = "1"
var = C.CDLL("/tmp/libgprx_dma.so"); proto(var)
var.gprx_gemm(0, 0, 1, 128, 128, 64, 1.0, dA, 64, dA, 64, 0.0, dC, 128, 0, 64)    # first launch of the copy: reads 1
libs = {"regs": base, "dma": var}
cases = [(4096, 4096, 4096, 0, 128, "dense 4096^3 t128"), (4096, 4096, 4096, 0, 64, "dense 4096^3 t64"), (4096, 4096, 1024, 1, 64, "syrk K=1024 t64"),
         (4096, 4096, 1024, 1, 128, "syrk K=1024 t128"), (4096, 4096, 256, 1, 64, "syrk K=256 t64"), (8192, 8192, 512, 1, 128, "syrk 8192 K=512 t128"), (8192, 8192, 512, 1, 64, "syrk 8192 K=512 t64")]
for m, n, k, flags, tile, name in cases:
    a = rng.standard_normal((m, k))
    dA = vp(); dC = vp()
    base.gprx_dev_malloc(0, a.nbytes, C.byref(dA)); base.gprx_dev_malloc(0, m * n * 8, C.byref(dC))
    base.gprx_memcpy_h2d(0, dA, a.ctypes.data_as(vp), a.nbytes)
    res = {k_: [] for k_ in libs}
    outs = {}
    for rnd in range(6):
        for key, l in libs.items():
            t0 = time.perf_counter()
            l.gprx_gemm(0, 0, 1, m, n, k, -1.0, dA, k, dA, k, 0.0, dC, n, flags, tile)
            res[key].append(time.perf_counter() - t0)
            if rnd == 0:
                out = np.empty((m, n)); base.gprx_memcpy_d2h(0, out.ctypes.data_as(vp), dC, out.nbytes); outs[key] = np.tril(out) if flags & 1 else out
    same = bool(np.array_equal(outs["regs"], outs["dma"]))
    ref = -(a[:256] @ a[:256].T)
    err = float(np.max(np.abs(np.tril(outs["dma"][:256, :256]) - np.tril(ref))))
    fl = 2.0 * m * n * k * (0.5 if flags & 1 else 1.0)
    print(name, {key: f"{fl/min(v)/1e12:.2f} TF/s" for key, v in res.items()}, "bit-identical" if same else "DIFFERENT", f"err vs numpy {err:.1e}", flush=True)
