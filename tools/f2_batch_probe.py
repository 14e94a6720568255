"""Batched objective + gradient (gprx_objective_batch) against single calls: evaluations/s at N = 4096, d = 8."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from gpras_amd.engine import Engine  # noqa: E402
from gpras_amd.synth import make_regression  # noqa: E402

n, d = 4096, 8
x, y, _ = make_regression(n, d, n_outputs=4, n_test=8, config=2, unit=0)
eng = Engine("RBF", x, y)
rng = np.random.default_rng(0)
for count in (1, 4, 8, 16, 32):
    units = np.arange(count, dtype=np.int32) % 4
    best = 1e9
    for rep in range(3):
        thetas = np.tile([0.5413, 0.37, 0.5413], (count, 1)) + 0.01 * rng.standard_normal((count, 3))
        t0 = time.perf_counter()
        if count == 1:
            eng.objective(0, thetas[0], None, 7, True)
        else:
            losses, grads, ok = eng.objective_batch(units, thetas, 7)
        best = min(best, time.perf_counter() - t0)
    print(f"cells={count:3d}: {best*1e3:8.3f} ms  {count/best:8.1f} evaluations/s  {count*n**3/best/1e12:6.2f} TFLOP/s", flush=True)
