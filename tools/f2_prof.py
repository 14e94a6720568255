"""Batched objective + gradient at N = 4096, 128 cells, in a loop for rocprofv3 (development aid)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from gpras_amd.engine import Engine
from gpras_amd.synth import make_regression
n, d, count = 4096, 8, int(sys.argv[1]) if len(sys.argv) > 1 else 128
x, y, _ = make_regression(n, d, n_outputs=4, n_test=8, config=2, unit=0)
eng = Engine("RBF", x, y)
rng = np.random.default_rng(0)
units = np.arange(count, dtype=np.int32) % 4
thetas = np.tile([0.5413, 0.37, 0.5413], (count, 1)) + 0.01 * rng.standard_normal((count, 3))
eng.objective_batch(units, thetas, 7)
t0 = time.perf_counter()
for _ in range(2):
    eng.objective_batch(units, thetas, 7)
dt = (time.perf_counter() - t0) / 2
print(f"cells={count}: {dt*1e3:.1f} ms per batch = {count/dt:.0f} evaluations/s = {count*n**3/dt/1e12:.1f} TFLOP/s")
