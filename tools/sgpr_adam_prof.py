"""Resident Adam loop of `modes` sparse modes (N = 4096, d = 10, M = 50) for rocprofv3.  argv: modes [steps]"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from gpras_amd.gpr import GPRAS
from gpras_amd.synth import make_regression
modes = int(sys.argv[1]); steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
x, y, _ = make_regression(4096, 10, n_outputs=modes, n_test=0, config=6, unit=1)
g = GPRAS("RBF")
t0 = time.perf_counter()
g.fit(x, y, 50, "kmeans", "adam", max_iter=steps)
dt = time.perf_counter() - t0
print(f"modes={modes}: {steps} Adam steps in {dt:.4f} s = {dt/steps*1e6:.1f} us per step")
