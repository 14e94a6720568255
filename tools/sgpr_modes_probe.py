"""The reference's sweep couples the input dimension to the number of modes (d = K = spatial_mode_count): lock-step Adam step time and
host-driven evaluation rate at (modes = d) points.  argv: mode counts"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from gpras_amd.gpr import GPRAS
from gpras_amd.synth import make_regression

n, m = 4096, 50
for k in [int(a) for a in sys.argv[1:]] or [10, 16, 20, 28, 40, 50]:
    x, y, _ = make_regression(n, k, n_outputs=k, n_test=0, config=6, unit=1)
    g = GPRAS("RBF")
    g._init_models(x.astype(np.float64), y.astype(np.float64), m, "grid")
    eng = g.engine
    units = np.arange(k, dtype=np.int32)
    thetas = np.stack([mm.theta() for mm in g.models])
    zs = np.stack([mm.Z for mm in g.models])
    for _ in range(3):
        eng.objective_batch(units, thetas, 15, True, zs=zs)
    t0 = time.perf_counter()
    for _ in range(30):
        eng.objective_batch(units, thetas, 15, True, zs=zs)
    dt = (time.perf_counter() - t0) / 30
    eng.adam_batch(units, thetas, 15, 20, zs=zs)
    steps = 300
    t0 = time.perf_counter()
    eng.adam_batch(units, thetas, 15, steps, zs=zs)
    ta = (time.perf_counter() - t0) / steps
    print(f"modes = d = {k}: host-driven {dt*1e6:.0f} us per call = {k/dt:.0f} evaluations/s; resident Adam {ta*1e6:.1f} us per step = {k/ta:.0f} evaluations/s", flush=True)
    del g
