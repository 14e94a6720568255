"""Sparse models: batched objective + gradient against single calls, and the default two-stage fit over the modes."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from gpras_amd.gpr import GPRAS  # noqa: E402
from gpras_amd.synth import make_regression  # noqa: E402

n, d, m, k = 4096, 10, 50, 16
x, y, _ = make_regression(n, d, n_outputs=k, n_test=0, config=6, unit=1)
g = GPRAS("RBF")
g._init_models(x.astype(np.float64), y.astype(np.float64), m, "kmeans")
eng = g.engine
for cells in (1, 4, 8, 16):
    units = np.arange(cells, dtype=np.int32)
    thetas = np.stack([g.models[i].theta() for i in range(cells)])
    zs = np.stack([g.models[i].Z for i in range(cells)])
    best = 1e9
    for rep in range(5):
        t0 = time.perf_counter()
        if cells == 1:
            eng.objective(0, thetas[0], zs[0], 15, True)
        else:
            eng.objective_batch(units, thetas, 15, True, zs=zs)
        best = min(best, time.perf_counter() - t0)
    print(f"cells={cells:3d}: {best*1e3:7.3f} ms per call  {cells/best:9.0f} evaluations/s", flush=True)
for lockstep in (True, False):
    gg = GPRAS("RBF")
    t0 = time.perf_counter()
    gg.fit(x, y, m, "kmeans", "two-stage", lockstep=lockstep)
    dt = time.perf_counter() - t0
    ev = sum(mm.n_evals for mm in gg.models)
    print(f"two-stage fit of {k} modes, lockstep={lockstep}: {dt:.3f} s ({ev} evaluations, {ev/dt:.0f}/s)", flush=True)
