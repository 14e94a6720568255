"""Would two halves of a sparse batch on two streams overlap?  One engine with all cells against two engines (own streams, own graphs) with
half the cells each, driven from two host threads.  argv: M [cells]"""
import sys, time, threading
import numpy as np
sys.path.insert(0, ".")
from gpras_amd.gpr import GPRAS
from gpras_amd.synth import make_regression

m = int(sys.argv[1]) if len(sys.argv) > 1 else 128
cells = int(sys.argv[2]) if len(sys.argv) > 2 else 16
n, d = 4096, 10
x, y, _ = make_regression(n, d, n_outputs=cells, n_test=0, config=6, unit=1)
x, y = x.astype(np.float64), y.astype(np.float64)

def make(cols):
    g = GPRAS("RBF")
    g._init_models(x, y[:, cols], m, "grid")
    u = np.arange(len(cols), dtype=np.int32)
    th = np.stack([mm.theta() for mm in g.models])
    z = np.stack([mm.Z for mm in g.models])
    for _ in range(4):
        g.engine.objective_batch(u, th, 15, True, zs=z)
    return g, u, th, z

reps = 30
g, u, th, z = make(list(range(cells)))
t0 = time.perf_counter()
for _ in range(reps):
    g.engine.objective_batch(u, th, 15, True, zs=z)
one = (time.perf_counter() - t0) / reps
half = cells // 2
a, b = make(list(range(half))), make(list(range(half, cells)))
def loop(t):
    for _ in range(reps):
        t[0].engine.objective_batch(t[1], t[2], 15, True, zs=t[3])
t0 = time.perf_counter()
ths = [threading.Thread(target=loop, args=(t,)) for t in (a, b)]
for t in ths: t.start()
for t in ths: t.join()
two = (time.perf_counter() - t0) / reps
t0 = time.perf_counter()
loop(a)
alone = (time.perf_counter() - t0) / reps
print(f"M={m} cells={cells}: one engine {one*1e6:.0f} us per evaluation of all cells; two engines x {half} cells concurrently {two*1e6:.0f} us; one engine with {half} cells {alone*1e6:.0f} us")
