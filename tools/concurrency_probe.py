"""Throughput of B concurrent cells on one GPU (one handle + stream per host thread)."""
import ctypes as C, sys, time, threading
import numpy as np
sys.path.insert(0, ".")
from gpras_amd import _lib
from gpras_amd._lib import check, ptr
from gpras_amd.synth import make_regression
lib = _lib.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = 12
theta = np.array([0.5413, 0.37, 0.5413])
for B in (1, 2, 3, 4, 6, 8):
    handles = []
    for b in range(B):
        x, y, _ = make_regression(n, 8, 1, 0, config=2, unit=b)
        h = C.c_void_p()
        check(lib.gprx_create(0, n, 8, 0, 0, 0, C.byref(h)))
        check(lib.gprx_set_data(h, ptr(x), ptr(y), 1), h)
        handles.append(h)
    losses = [C.c_double() for _ in range(B)]
    def work(b, k):
        for _ in range(k):
            check(lib.gprx_factorize(handles[b], 0, ptr(theta), None, 7, C.byref(losses[b])), handles[b])
    for b in range(B): work(b, 2)
    ts = [threading.Thread(target=work, args=(b, steps)) for b in range(B)]
    t0 = time.perf_counter()
    for t in ts: t.start()
    for t in ts: t.join()
    el = time.perf_counter() - t0
    print(f"N={n} B={B}: {B*steps/el:8.1f} fits/s  ({el/steps*1e3:.2f} ms per step of {B} cells)  loss0 {losses[0].value:.6f}", flush=True)
    for h in handles: lib.gprx_destroy(h)
