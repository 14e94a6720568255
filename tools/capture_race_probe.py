"""Two threads, two handles, batched sparse evaluations at the same time, repeated from fresh handles: looks for the capture race
(development aid)."""
import sys, threading, traceback
import numpy as np
sys.path.insert(0, ".")
from gpras_amd.engine import Engine
from gpras_amd.synth import make_regression
n, d, m, cells = 700, 5, 40, 6
x, y, _ = make_regression(n, d, n_outputs=cells, n_test=0, config=14, unit=31)
rng = np.random.default_rng(12)
thetas = np.ascontiguousarray(rng.normal(0.2, 0.3, size=(cells, 3)))
zs = np.ascontiguousarray(np.stack([x[rng.choice(n, size=m, replace=False)] for _ in range(cells)]))
units = np.arange(cells, dtype=np.int32)
ref = Engine("Matern32", x, y, m); want = ref.objective_batch(units, thetas, 15, zs=zs); ref.close()
bad = 0
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 30):
    engines = [Engine("Matern32", x, y, m) for _ in range(2)]
    errors = []
    def run(k):
        try:
            for it in range(6):
                losses, grads, ok = engines[k].objective_batch(units, thetas, 15, zs=zs)
                if not (ok.all() and np.array_equal(losses, want[0]) and np.array_equal(grads, want[1])):
                    errors.append(f"mismatch thread {k} call {it}: max loss diff {np.nanmax(np.abs(losses - want[0]))}, grad diff {np.nanmax(np.abs(grads - want[1]))}, ok {ok}")
        except Exception as exc:
            errors.append(f"{type(exc).__name__}: {exc}")
    ts = [threading.Thread(target=run, args=(k,)) for k in range(2)]
    [t.start() for t in ts]; [t.join() for t in ts]
    for e in engines: e.close()
    if errors:
        bad += 1
        print("rep", rep, errors, flush=True)
print("failures:", bad)
