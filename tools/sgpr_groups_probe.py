"""Sparse lock-step Adam: one engine with all cells against G engines (own handle and stream each) with cells / G cells, driven
from G host threads (development aid).  argv: cells"""
import sys, time, threading
import numpy as np
sys.path.insert(0, ".")
from gpras_amd.engine import Engine
from gpras_amd.gpr import GPRAS
from gpras_amd.synth import make_regression
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 16
n, d, m = 4096, 10, 50
x, y, _ = make_regression(n, d, n_outputs=cells, n_test=0, config=6, unit=1)
g = GPRAS("RBF"); g._init_models(x.astype(np.float64), y.astype(np.float64), m, "grid")
thetas = np.stack([mod.theta() for mod in g.models]); zs = np.stack([mod.Z for mod in g.models])
for groups in (1, 2, 4):
    per = cells // groups
    engines = [Engine("RBF", x, y, m) for _ in range(groups)]
    def run(k):
        sl = slice(k * per, (k + 1) * per)
        engines[k].adam_batch(np.arange(cells, dtype=np.int32)[sl], thetas[sl], 15, 100, zs=zs[sl])
    for rep in range(2):
        t0 = time.perf_counter()
        ths = [threading.Thread(target=run, args=(k,)) for k in range(groups)]
        [t.start() for t in ths]; [t.join() for t in ths]
        dt = time.perf_counter() - t0
    print(f"cells={cells} groups={groups}: 100 Adam steps in {dt*1e3:.1f} ms = {cells*100/dt:.0f} evaluations/s", flush=True)
    for e in engines: e.close()
