"""GPRAS.predict of 50 fitted sparse modes at 100 000 points, host arrays both ways (bench.py sparse_section's predict leg alone)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from gpras_amd.gpr import GPRAS
from gpras_amd.synth import make_regression
x, y, xt = make_regression(4096, 10, n_outputs=50, n_test=100000, config=6, unit=1)
g = GPRAS("RBF")
g.fit(x.astype(np.float64), y.astype(np.float64), 50, "kmeans", "adam", max_iter=5)
g.predict(xt[:4096])
best = 1e9
for _ in range(4):
    t0 = time.perf_counter(); m, v = g.predict(xt); best = min(best, time.perf_counter() - t0)
print(f"50 modes x {xt.shape[0]} points: {best*1e3:.1f} ms = {50*xt.shape[0]/best/1e6:.0f} M point-modes/s; finite {bool(np.isfinite(m).all() and (v > 0).all())}")
