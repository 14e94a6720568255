"""Batched factorisation (gprx_factorize_batch) against the many-handles schedule: fits/s at N = 4096, d = 8."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from gpras_amd import _lib  # noqa: E402
from gpras_amd.engine import Engine  # noqa: E402
from gpras_amd.synth import make_regression  # noqa: E402

n, d = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4096, 8)
counts = [int(c) for c in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1, 2, 4, 8, 16, 32]
x, y, _ = make_regression(n, d, n_outputs=4, n_test=8, config=2, unit=0)
eng = Engine("RBF", x, y)
rng = np.random.default_rng(0)
for count in counts:
    units = np.arange(count, dtype=np.int32) % 4
    best = 1e9
    dev = 0.0
    for rep in range(4):
        thetas = np.tile([0.5413, 0.37, 0.5413], (count, 1)) + 0.01 * rng.standard_normal((count, 3))
        t0 = time.perf_counter()
        losses, ok = eng.factorize_batch(units, thetas, 7)
        dt = time.perf_counter() - t0
        if dt < best:
            best, dev = dt, eng.last_batch_ms()
    assert ok.all()
    fl = count * n**3 / 3
    print(f"N={n} cells={count:3d}: wall {best*1e3:8.3f} ms  device {dev:8.3f} ms  {count/best:8.1f} fits/s  {fl/best/1e12:6.2f} TFLOP/s", flush=True)
