"""Default sparse fit (k-means Z, two-stage Adam 100 + 100) of `modes` modes, N = 4096, d = 10, M = 50: seconds, resident loop against the host-stepped one."""
import os, sys, time
import numpy as np
sys.path.insert(0, ".")
from gpras_amd.gpr import GPRAS
from gpras_amd.synth import make_regression

modes = int(sys.argv[1]) if len(sys.argv) > 1 else 16
n, d, m = 4096, 10, 50
x, y, _ = make_regression(n, d, n_outputs=modes, n_test=0, config=6, unit=1)
for rep in range(3):
    g = GPRAS("RBF")
    t0 = time.perf_counter()
    g.fit(x, y, m, "kmeans", "two-stage")
    dt = time.perf_counter() - t0
    ev = sum(mm.n_evals for mm in g.models)
    print(f"modes={modes} GPRX_ADAM_HOST={os.environ.get('GPRX_ADAM_HOST', '0')}: fit {dt:.4f} s, {ev} evaluations, {ev/dt:.0f} evaluations/s", flush=True)
    del g
