"""Sparse loss + gradient evaluations per second against the input dimension d (the reference sweeps 1..50 spatial modes = d): fused against the launch sequence."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from gpras_amd import _lib
from gpras_amd._lib import check
from gpras_amd.gpr import GPRAS
from gpras_amd.synth import make_regression

lib = _lib.load()
n, m, cells = 4096, 50, 16
for d in [int(a) for a in sys.argv[1:]] or [4, 10, 16, 20, 32, 50]:
    x, y, _ = make_regression(n, d, n_outputs=cells, n_test=0, config=6, unit=1)
    for ard in (False, True):
        for fused in (1, 0):
            check(lib.gprx_set_tuning(b"sgpr_fused", fused))
            g = GPRAS("RBF")
            g._init_models(x.astype(np.float64), y.astype(np.float64), m, "grid", ard)
            eng = g.engine
            units = np.arange(cells, dtype=np.int32)
            thetas = np.stack([g.models[i].theta() for i in range(cells)])
            zs = np.stack([g.models[i].Z for i in range(cells)])
            for _ in range(4):
                eng.objective_batch(units, thetas, 15, True, zs=zs)
            reps = 50
            t0 = time.perf_counter()
            for _ in range(reps):
                eng.objective_batch(units, thetas, 15, True, zs=zs)
            dt = (time.perf_counter() - t0) / reps
            print(f"d={d} ard={int(ard)} fused={fused}: {dt*1e6:.1f} us per 16-cell evaluation = {cells/dt:.0f} evaluations/s", flush=True)
            del g
check(lib.gprx_set_tuning(b"sgpr_fused", 1))
