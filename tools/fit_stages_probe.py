"""Where a default sparse fit spends its host time: engine + k-means initialisation, stage 1, stage 2, final evaluation."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from gpras_amd import optimizers
from gpras_amd.gpr import GPRAS
from gpras_amd.synth import make_regression
modes = int(sys.argv[1]) if len(sys.argv) > 1 else 16
x, y, _ = make_regression(4096, 10, n_outputs=modes, n_test=0, config=6, unit=1)
for rep in range(3):
    g = GPRAS("RBF")
    t0 = time.perf_counter()
    g._init_models(x, y, 50, "kmeans")
    t1 = time.perf_counter()
    ms = g.models
    for m in ms:
        m.set_all_trainable(False); m.set_trainable(Z=True)
    optimizers._optimize_adam_many(ms, 100)
    t2 = time.perf_counter()
    for m in ms:
        m.set_all_trainable(True); m.set_trainable(Z=False)
    optimizers._optimize_adam_many(ms, 100)
    t3 = time.perf_counter()
    for m in ms:
        m.set_trainable(Z=True)
    optimizers._evaluate_many(ms, want_grad=False)
    t4 = time.perf_counter()
    print(f"modes={modes}: init {1e3*(t1-t0):.2f} ms, stage 1 {1e3*(t2-t1):.2f} ms, stage 2 {1e3*(t3-t2):.2f} ms, final loss {1e3*(t4-t3):.2f} ms; total {1e3*(t4-t0):.2f}")
