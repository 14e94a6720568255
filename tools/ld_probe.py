"""Does a power-of-two leading dimension cost HBM bandwidth?  gprx_potrf and gprx_gemm on the same matrix stored with
different leading dimensions (development aid).  argv: N [pads...]"""
import ctypes as C, sys, time
import numpy as np
sys.path.insert(0, ".")
from gpras_amd import _lib
from gpras_amd._lib import DeviceBuffer, check, ptr
lib = _lib.load()
n = int(sys.argv[1])
pads = [int(a) for a in sys.argv[2:]] or [0, 16, 32, 64, 128]
rng = np.random.default_rng(0)
a = rng.normal(size=(n, 64))
spd = a @ a.T + n * np.eye(n)
for pad in pads:
    ld = n + pad
    host = np.zeros((n + 64, ld))
    host[:n, :n] = spd
    dA = DeviceBuffer.from_array(host, 0)
    dA0 = DeviceBuffer.from_array(host, 0)
    dinv = DeviceBuffer(8 * n * 64, 0)
    info = C.c_int()
    best = 1e9
    for _ in range(4):
        check(lib.gprx_memcpy_h2d(0, dA.ptr, ptr(host), host.nbytes))
        t = time.perf_counter()
        check(lib.gprx_potrf(0, dA.ptr, ld, n, 0, dinv.ptr, C.byref(info)))
        best = min(best, time.perf_counter() - t)
    L = np.tril(dA.to_array((n + 64, ld))[:n, :n])
    err = np.abs(L @ L.T - spd).max() / n
    # trailing-update shaped GEMM: C (n,n) lower -= A (n,1024) A^T, and the K = 64 update
    out = []
    for k in (1024, 128, 64):
        if k > n:
            continue
        bg = 1e9
        for _ in range(4):
            t = time.perf_counter()
            check(lib.gprx_gemm(0, 0, 1, n, n, k, -1.0, dA0.ptr, ld, dA0.ptr, ld, 1.0, dA.ptr, ld, 1, 64))
            bg = min(bg, time.perf_counter() - t)
        out.append(f"K={k}: {bg*1e3:.3f} ms ({n*n*k/bg/1e12:.1f} TF/s)")
    print(f"N={n} ld={ld}: potrf {best*1e3:.3f} ms = {n**3/3/best/1e12:.1f} TF/s (info {info.value}, err {err:.1e}); gemm lower " + "; ".join(out), flush=True)
    for b in (dA, dA0, dinv):
        b.free()
