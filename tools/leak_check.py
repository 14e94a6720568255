"""Create / use / destroy handles repeatedly and watch free device memory (hipMemGetInfo)."""
import ctypes as C, sys
import numpy as np
sys.path.insert(0, ".")
from gpras_amd.gpr import GPRAS
from gpras_amd.preprocess import EOFProjector
from gpras_amd import metrics
from gpras_amd.synth import make_regression, make_eof_state
hip = C.CDLL("libamdhip64.so")
def free_mb():
    f, t = C.c_size_t(), C.c_size_t()
    hip.hipMemGetInfo(C.byref(f), C.byref(t))
    return f.value / 2**20
x, y, xs = make_regression(600, 5, n_outputs=4, n_test=50, config=1, unit=0)
st = make_eof_state(3000, 4, 20, 1)
base = None
for rep in range(8):
    for nind in (None, 20):
        g = GPRAS("RBF"); g.fit(x, y, nind, "grid", "adam", max_iter=3); g.predict(xs)
        for e in g.engines: e.close()
    p = EOFProjector(st["dry"], st["elevations"], st["x"][:, ~st["dry"]].mean(axis=0), st["weights"], st["eofs"], st["x_mean"], st["x_std"], "wse")
    z = p.transform(st["x"]); p.reverse_transform(z, np.abs(z)); p.close()
    metrics.FieldMetrics(st["x"], st["x"] + 0.1, t_tol=1, v_tol=0.2)
    f = free_mb()
    if rep == 1: base = f
    print(f"round {rep}: free {f:.1f} MiB", flush=True)
assert base - f < 64, "device memory keeps shrinking"
print("no leak")
