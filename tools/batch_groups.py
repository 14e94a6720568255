"""Batched cells split into concurrent groups (one handle + stream + host thread per group)."""
import sys
import threading
import time

import numpy as np

sys.path.insert(0, ".")
from gpras_amd.engine import Engine  # noqa: E402
from gpras_amd.synth import make_regression  # noqa: E402

n, d = 4096, 8
total = int(sys.argv[1]) if len(sys.argv) > 1 else 16
x, y, _ = make_regression(n, d, n_outputs=4, n_test=8, config=2, unit=0)
for groups in (1, 2, 4):
    per = total // groups
    engs = [Engine("RBF", x, y) for _ in range(groups)]
    units = np.arange(per, dtype=np.int32) % 4
    thetas = np.tile([0.5413, 0.37, 0.5413], (per, 1))
    for e in engs:
        e.factorize_batch(units, thetas, 7)
    best = 1e9
    for rep in range(5):
        bar = threading.Barrier(groups + 1)
        def work(e):
            bar.wait()
            e.factorize_batch(units, thetas, 7)
        th = [threading.Thread(target=work, args=(e,)) for e in engs]
        for t in th:
            t.start()
        bar.wait()
        t0 = time.perf_counter()
        for t in th:
            t.join()
        best = min(best, time.perf_counter() - t0)
    print(f"total {total} cells in {groups} group(s) of {per}: {best*1e3:8.3f} ms  {total/best:8.1f} fits/s", flush=True)
    for e in engs:
        e.close()
