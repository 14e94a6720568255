"""Single-cell objective + gradient at N = 4096 in a loop, for rocprofv3 (development aid)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from gpras_amd.engine import Engine
from gpras_amd.synth import make_regression
n, d = 4096, 8
x, y, _ = make_regression(n, d, n_outputs=1, n_test=8, config=2, unit=0)
eng = Engine("RBF", x, y)
theta = np.array([0.5413, 0.37, 0.5413])
eng.objective(0, theta, None, 7, True)
t0 = time.perf_counter()
for _ in range(10):
    eng.objective(0, theta, None, 7, True)
dt = (time.perf_counter() - t0) / 10
print(f"single: {dt*1e3:.2f} ms per evaluation = {1/dt:.0f} evaluations/s", flush=True)
