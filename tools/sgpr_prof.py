"""Sparse batch evaluation in a loop for rocprofv3 (development aid).  argv: cells [reps]"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from gpras_amd.gpr import GPRAS
from gpras_amd.synth import make_regression
cells = int(sys.argv[1]); reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
n, d, m = 4096, 10, (int(sys.argv[3]) if len(sys.argv) > 3 else 50)
x, y, _ = make_regression(n, d, n_outputs=cells, n_test=0, config=6, unit=1)
g = GPRAS("RBF")
g._init_models(x.astype(np.float64), y.astype(np.float64), m, "grid")
eng = g.engine
units = np.arange(cells, dtype=np.int32)
thetas = np.stack([g.models[i].theta() for i in range(cells)])
zs = np.stack([g.models[i].Z for i in range(cells)])
for _ in range(3):
    eng.objective_batch(units, thetas, 15, True, zs=zs)
t0 = time.perf_counter()
for _ in range(reps):
    eng.objective_batch(units, thetas, 15, True, zs=zs)
print(f"cells={cells}: {(time.perf_counter()-t0)/reps*1e3:.3f} ms per evaluation")
