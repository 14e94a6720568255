#!/bin/bash
for m in 50 100 300; do timeout -k 10 200 python tools/sgpr_prof.py 16 10 $m; done
timeout -k 10 200 python tools/sgpr_prof.py 1 10 300
python - <<'PY'
import sys, time
import numpy as np
sys.path.insert(0, ".")
from gpras_amd.gpr import GPRAS
from gpras_amd.synth import make_regression
from gpras_amd.kmeans import kmeans_centers
x, y, xt = make_regression(4096, 10, n_outputs=16, n_test=2000, config=6, unit=1)
for m in (100, 300):
    t0 = time.perf_counter(); kmeans_centers(x, m); tk = time.perf_counter() - t0
    g = GPRAS("RBF"); t0 = time.perf_counter(); g.fit(x, y, m, "kmeans", "two-stage"); tf = time.perf_counter() - t0
    g = GPRAS("RBF"); t0 = time.perf_counter(); g.fit(x, y, m, "kmeans", "two-stage"); tf = time.perf_counter() - t0
    t0 = time.perf_counter(); g.predict(xt); tp = time.perf_counter() - t0
    print(f"M={m}: kmeans {tk*1e3:.0f} ms, 16-mode default fit {tf:.3f} s, predict 2000 pts {tp*1e3:.1f} ms", flush=True)
PY
