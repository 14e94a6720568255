"""Phase durations of panel workgroup 0 summed over one factorisation (needs the -DGPRX_PANEL_ACC build, tools/panel_acc.sh).
argv: N d no_lookahead"""
import ctypes as C, sys, time
import numpy as np
sys.path.insert(0, ".")
from gpras_amd import _lib
from gpras_amd._lib import check, ptr
from gpras_amd.model import NOISE_LOWER, softplus_inv
from gpras_amd.synth import make_regression
lib = _lib.load()
n, d, nla = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
check(lib.gprx_set_tuning(b"no_lookahead", nla))
x, y, _ = make_regression(n, d, n_outputs=1, n_test=0, config=5, unit=0)
h = C.c_void_p()
check(lib.gprx_create(0, n, d, 0, 0, 0, C.byref(h)))
check(lib.gprx_set_data(h, ptr(x), ptr(y), 1), h)
th = np.ascontiguousarray([softplus_inv(1.0), softplus_inv(np.mean(np.abs(x))), softplus_inv(1.0 - NOISE_LOWER)], dtype=np.float64)
loss = C.c_double()
check(lib.gprx_factorize(h, 0, ptr(th), None, 7, C.byref(loss)), h)
acc = (C.c_ulonglong * 8)()
lib.gprx_panel_acc(acc, 1)
check(lib.gprx_factorize(h, 0, ptr(th), None, 7, C.byref(loss)), h)
ms = (C.c_double * 4)(); lib.gprx_last_timings(h, ms)
lib.gprx_panel_acc(acc, 0)
a = np.array(acc[:6], dtype=np.float64)
k = a[0]
us = a[1:] / k / 2100.0  # s_memtime counts shader clocks: ~2.1 GHz under this load (approximate)
print(f"N={n} no_lookahead={nla}: cholesky {ms[1]:.2f} ms, {int(k)} panel launches; per launch (us): loads {us[0]:.2f} | sub-panel 0 {us[1]:.2f} | 1-3 {us[2]:.2f} | 4-7 {us[3]:.2f} | stores {us[4]:.2f} | sum {us.sum():.2f}", flush=True)
lib.gprx_destroy(h)
