"""Large sparse batches as one group of cells or as two groups on two streams ("sgpr_groups_from"): evaluations/s of the batched call and
microseconds per lock-step Adam step.  argv: cell counts (default 16 24 32 50)"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from gpras_amd import _lib
from gpras_amd._lib import check
from gpras_amd.gpr import GPRAS
from gpras_amd.synth import make_regression

lib = _lib.load()
import os
n, d, m = int(os.environ.get("PROBE_N", "4096")), 10, 50
for cells in [int(a) for a in sys.argv[1:]] or [16, 24, 32, 50]:
    x, y, _ = make_regression(n, d, n_outputs=cells, n_test=0, config=6, unit=1)
    ref = None
    for groups_from in (0, 1):  # 0: one group; 1: two groups whatever the batch
        check(lib.gprx_set_tuning(b"sgpr_groups_from", groups_from))
        g = GPRAS("RBF")
        g._init_models(x.astype(np.float64), y.astype(np.float64), m, "grid")
        eng = g.engine
        units = np.arange(cells, dtype=np.int32)
        thetas = np.stack([g.models[i].theta() for i in range(cells)])
        zs = np.stack([g.models[i].Z for i in range(cells)])
        for _ in range(4):
            out = eng.objective_batch(units, thetas, 15, True, zs=zs)
        reps = 100
        t0 = time.perf_counter()
        for _ in range(reps):
            out = eng.objective_batch(units, thetas, 15, True, zs=zs)
        dt = (time.perf_counter() - t0) / reps
        steps = 300
        eng.adam_batch(units, thetas, 15, 20, zs=zs)
        t0 = time.perf_counter()
        th, zz, ev, _ = eng.adam_batch(units, thetas, 15, steps, zs=zs)
        ta = time.perf_counter() - t0
        same = ""
        if ref is None:
            ref = (out[0].copy(), out[1].copy(), th.copy(), zz.copy())
        else:
            same = f"  same bits as one group: {np.array_equal(out[0], ref[0]) and np.array_equal(out[1], ref[1]) and np.array_equal(th, ref[2]) and np.array_equal(zz, ref[3])}"
        print(f"N={n} cells={cells} groups={'yes' if groups_from else 'no'}: {dt*1e6:.1f} us per call = {cells/dt:.0f} evaluations/s; Adam {ta/steps*1e6:.1f} us per step "
              f"({int(ev.min())}-{int(ev.max())} evaluations per cell){same}", flush=True)
        del g
check(lib.gprx_set_tuning(b"sgpr_groups_from", 17))
