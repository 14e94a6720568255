import ctypes as C, sys, numpy as np
sys.path.insert(0, ".")
lib = C.CDLL("tools/libgprx_stamps.so")
n = 4096
NP = int(sys.argv[1]) if len(sys.argv) > 1 else 64
rng = np.random.default_rng(0)
g = rng.standard_normal((n, 64))
a = np.ascontiguousarray(g @ g.T / 64 + np.eye(n))
vp = C.c_void_p
lib.gprx_dev_malloc.argtypes = [C.c_int, C.c_int64, C.POINTER(vp)]
dA = vp(); dI = vp()
lib.gprx_dev_malloc(0, a.nbytes, C.byref(dA)); lib.gprx_dev_malloc(0, n * 64 * 8, C.byref(dI))
lib.gprx_memcpy_h2d.argtypes = [C.c_int, vp, vp, C.c_int64]
lib.gprx_potrf.argtypes = [C.c_int, vp, C.c_int64, C.c_int64, C.c_int64, vp, C.POINTER(C.c_int)]
info = C.c_int(0)
for rep in range(3):
    lib.gprx_memcpy_h2d(0, dA, a.ctypes.data_as(vp), a.nbytes)
    rc = lib.gprx_potrf(0, dA, n, NP, n - NP, dI, C.byref(info))
    out = (C.c_ulonglong * 64)()
    lib.gprx_panel_stamps(out)
    s = np.array(out[:16], dtype=np.int64)
    print("rep", rep, "rc", rc, "| load %d | p0 %d | p1-3 %d | p4-7+sync %d | store %d | total %d" % (s[1]-s[0], s[2]-s[1], s[3]-s[2], s[4]-s[3], s[5]-s[4], s[5]-s[0]))
    print("    P=1: A-write+barrier1 %d | factor %d | solve+write %d | barrier2 %d | mfma %d" % (s[11]-s[10], s[12]-s[11], s[13]-s[12], s[14]-s[13], s[15]-s[14]))
