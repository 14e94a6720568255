"""cProfile of the default sparse fit (lock-step two-stage Adam over 16 modes): where the host time goes (development aid)."""
import cProfile, pstats, sys, time
import numpy as np
sys.path.insert(0, ".")
from gpras_amd.gpr import GPRAS
from gpras_amd.synth import make_regression
n, d, m, k = 4096, 10, 50, int(sys.argv[1]) if len(sys.argv) > 1 else 16
x, y, _ = make_regression(n, d, n_outputs=k, n_test=0, config=6, unit=1)
g = GPRAS("RBF"); g.fit(x, y, m, "grid", "two-stage")  # warm
g = GPRAS("RBF")
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable(); g.fit(x, y, m, "grid", "two-stage"); pr.disable()
print(f"fit {time.perf_counter()-t0:.3f} s")
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
