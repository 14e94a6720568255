"""GPRAS.fit on exact models with K modes: lock-step (batched evaluations) against the serial per-mode loop."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from gpras_amd.gpr import GPRAS  # noqa: E402
from gpras_amd.synth import make_regression  # noqa: E402

n, d, k = 4096, 8, int(sys.argv[1]) if len(sys.argv) > 1 else 16
x, y, xs = make_regression(n, d, n_outputs=k, n_test=1000, config=3, unit=0)
for lockstep in (True, False):
    g = GPRAS("Matern52")
    t0 = time.perf_counter()
    g.fit(x, y, None, optimization_method="L-BFGS-B", ard=True, max_iter=20, lockstep=lockstep)
    dt = time.perf_counter() - t0
    evals = sum(m.n_evals for m in g.models)
    t1 = time.perf_counter()
    mean, var = g.predict(xs)
    tp = time.perf_counter() - t1
    print(f"lockstep={lockstep}: fit {k} modes {dt:.3f} s ({evals} evaluations, {evals/dt:.0f}/s), predict {tp*1e3:.1f} ms, stats {getattr(g, 'lockstep_stats', None)}", flush=True)
    params = [(m.variance, m.noise, tuple(np.atleast_1d(m.lengthscales))) for m in g.models]
    if lockstep:
        first = params
    else:
        print("bit-identical parameters:", first == params)
