"""Throughput of B cells per call through gprx_factorize_many (single host thread)."""
import ctypes as C, sys, time
import numpy as np
sys.path.insert(0, ".")
from gpras_amd import _lib
from gpras_amd._lib import check, ptr
from gpras_amd.synth import make_regression
lib = _lib.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = 12
if len(sys.argv) > 2: lib.gprx_set_tuning(b"no_lookahead", int(sys.argv[2]))
for B in (1, 2, 3, 4, 6, 8, 12):
    hs = (C.c_void_p * B)()
    for b in range(B):
        x, y, _ = make_regression(n, 8, 1, 0, config=2, unit=b)
        h = C.c_void_p()
        check(lib.gprx_create(0, n, 8, 0, 0, 0, C.byref(h)))
        check(lib.gprx_set_data(h, ptr(x), ptr(y), 1), h)
        hs[b] = h
    units = np.zeros(B, dtype=np.int32)
    thetas = np.tile(np.array([0.5413, 0.37, 0.5413]), (B, 1))
    losses = np.zeros(B)
    for _ in range(2):
        check(lib.gprx_factorize_many(B, hs, ptr(units), ptr(thetas), 7, ptr(losses)))
    t0 = time.perf_counter()
    for _ in range(steps):
        check(lib.gprx_factorize_many(B, hs, ptr(units), ptr(thetas), 7, ptr(losses)))
    el = time.perf_counter() - t0
    print(f"N={n} B={B}: {B*steps/el:8.1f} fits/s  ({el/steps*1e3:.2f} ms per step of {B} cells)  loss0 {losses[0]:.6f} lossB {losses[-1]:.6f}", flush=True)
    for b in range(B): lib.gprx_destroy(hs[b])
