"""Default sparse fit (k-means Z, two-stage Adam 100 + 100, lock-step) for 10 / 16 / 50 modes at N = 4096, d = 10, M = 50 (development aid)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from gpras_amd.gpr import GPRAS
from gpras_amd.synth import make_regression
n, d, m = 4096, 10, 50
for k in (10, 16, 50):
    x, y, xt = make_regression(n, d, n_outputs=k, n_test=2000, config=6, unit=1)
    best = 1e9
    for rep in range(3):
        g = GPRAS("RBF")
        t0 = time.perf_counter()
        g.fit(x, y, m, "kmeans", "two-stage")
        best = min(best, time.perf_counter() - t0)
    bp = 1e9
    for rep in range(3):
        t0 = time.perf_counter(); g.predict(xt); bp = min(bp, time.perf_counter() - t0)
    print(f"{k} modes: fit {best:.3f} s = {k/best:.0f} units/s; predict 2000 points {bp*1e3:.2f} ms", flush=True)
