"""PMC workload: one batched factorisation step with a given libgprx build (argv[1])."""
import ctypes as C, sys, numpy as np
sys.path.insert(0, ".")
from gpras_amd import _lib
from pathlib import Path
_lib.LIB_PATH = Path(sys.argv[1]).resolve()
from gpras_amd.engine import Engine
from gpras_amd.synth import make_regression
x, y, _ = make_regression(4096, 8, n_outputs=4, n_test=8, config=2, unit=0)
eng = Engine("RBF", x, y)
units = np.arange(64, dtype=np.int32) % 4
thetas = np.tile([0.5413, 0.37, 0.5413], (64, 1))
import time
for rep in range(3):
    t0 = time.perf_counter(); eng.factorize_batch(units, thetas, 7); print(time.perf_counter() - t0, flush=True)
