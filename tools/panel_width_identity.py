"""Do 64-column panels (+ K = 64 syrk) and the 128-column fused panel kernel give bit-identical factors?"""
import ctypes as C, sys
import numpy as np
sys.path.insert(0, ".")
from gpras_amd import _lib
from gpras_amd._lib import DeviceBuffer, check
lib = _lib.load()
n, extra = 1024, 64
rng = np.random.default_rng(2)
g = rng.standard_normal((n, 48))
full = np.vstack([g @ g.T / 48 + np.eye(n), rng.standard_normal((extra, n))])
outs = []
for pw, ob in ((64, 128), (128, 128), (64, 1024), (128, 1024)):
    check(lib.gprx_set_tuning(b"panel_width", pw)); check(lib.gprx_set_tuning(b"outer_block", ob)); check(lib.gprx_set_tuning(b"split_panel", -1))
    dA, dI = DeviceBuffer.from_array(full), DeviceBuffer(n * 64 * 8)
    info = C.c_int(0)
    check(lib.gprx_potrf(0, dA.ptr, n, n, extra, dI.ptr, C.byref(info)))
    out = dA.to_array((n + extra, n))
    outs.append((np.tril(out[:n]), out[n:], dI.to_array((n // 64, 64, 64))))
    dA.free(); dI.free()
for i in range(1, 4):
    print("config", i, "vs 0:", [bool(np.array_equal(a, b)) for a, b in zip(outs[0], outs[i])], "max diff L", np.max(np.abs(outs[0][0] - outs[i][0])))
