"""Cycle stamps of panel workgroup 0 (a -DGPRX_PANEL_STAMPS build: tools/libgprx_stamps.so, or the library named by GPRX_STAMPS_LIB).
argv[1] = rows below the 64 x 64 diagonal block (one panel launch of rows/64 + 1 workgroups)."""
import ctypes as C, os, sys, numpy as np
sys.path.insert(0, ".")
lib = C.CDLL(os.environ.get("GPRX_STAMPS_LIB", "tools/libgprx_stamps.so"))
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
lda = 4096
rng = np.random.default_rng(0)
a = rng.standard_normal((rows + 64, lda)) * 0.01
g = rng.standard_normal((64, 64))
a[:64, :64] = g @ g.T / 64 + np.eye(64)
vp = C.c_void_p
lib.gprx_dev_malloc.argtypes = [C.c_int, C.c_int64, C.POINTER(vp)]
dA = vp(); dI = vp()
lib.gprx_dev_malloc(0, a.nbytes, C.byref(dA)); lib.gprx_dev_malloc(0, 64 * 64 * 8, C.byref(dI))
lib.gprx_memcpy_h2d.argtypes = [C.c_int, vp, vp, C.c_int64]
lib.gprx_potrf.argtypes = [C.c_int, vp, C.c_int64, C.c_int64, C.c_int64, vp, C.POINTER(C.c_int)]
info = C.c_int(0)
for rep in range(3):
    lib.gprx_memcpy_h2d(0, dA, a.ctypes.data_as(vp), a.nbytes)
    rc = lib.gprx_potrf(0, dA, lda, 64, rows, dI, C.byref(info))
    out = (C.c_ulonglong * 64)()
    lib.gprx_panel_stamps(out)
    s = np.array(out[:20], dtype=np.int64)
    print("    solve phase: read+compute %d | sX writes %d | global stores %d" % (s[16]-s[12], s[17]-s[16], s[13]-s[17]))
    m = np.array(out[32:38], dtype=np.int64) - s[0]
    l = np.array(out[48:54], dtype=np.int64) - s[0]
    print("    middle workgroup: start %d end %d | last-but-one: start %d end %d (relative to workgroup 0's start)" % (m[0], m[5], l[0], l[5]))
    print("rows", rows, "rep", rep, "rc", rc, "| load %d | p0 %d | p1-3 %d | p4-7+sync %d | store %d | total %d" % (s[1]-s[0], s[2]-s[1], s[3]-s[2], s[4]-s[3], s[5]-s[4], s[5]-s[0]))
    print("    P=1: A-write+barrier1 %d | factor %d | solve+write %d | barrier2 %d | mfma %d" % (s[11]-s[10], s[12]-s[11], s[13]-s[12], s[14]-s[13], s[15]-s[14]), flush=True)
