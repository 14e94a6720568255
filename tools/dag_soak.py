"""Soak test of the tile-DAG hand-offs (MI355X_MICROARCH.md: "test every hand-off under UNEVEN load, consumer L1-warm, checking every
word"): the factorisation is repeated many times -- both schedules, several sizes, three panel sizes -- while a second stream
streams through a 1 GB buffer in bursts of varying length (uneven load on the memory system and on the CUs), and every run's factor,
right-hand-side rows and inverse blocks are compared WORD FOR WORD with the first run (the version counters fix the order of the
updates of a tile, so any difference is a stale or torn hand-off)."""
import ctypes as C
import json
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, ".")
from gpras_amd import _build, _lib  # noqa: E402
from gpras_amd._lib import DeviceBuffer, check  # noqa: E402

_build.build()
lib = _lib.load()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
stop = False


def hog():
    """a second host thread with its own handle: batched factorisations of 48 cells of N = 1024 in bursts, pauses of 0 .. 3 ms in
    between -- its kernels take CUs and HBM bandwidth away at irregular times (the DAG's persistent workgroups then are NOT all
    resident from the start, and its hand-offs see a loaded memory system)"""
    from gpras_amd._lib import ptr
    from gpras_amd.model import NOISE_LOWER, softplus_inv
    from gpras_amd.synth import make_regression

    rng = np.random.default_rng(5)
    cells = 48
    x, y, _ = make_regression(1024, 8, n_outputs=cells, n_test=0, config=2, unit=77)
    theta = np.array([softplus_inv(1.0), softplus_inv(np.mean(np.abs(x))), softplus_inv(1.0 - NOISE_LOWER)])
    thetas = np.ascontiguousarray(np.tile(theta, (cells, 1)))
    units = np.arange(cells, dtype=np.int32)
    h = C.c_void_p()
    check(lib.gprx_create(0, 1024, 8, 0, 0, 0, C.byref(h)))
    check(lib.gprx_set_data(h, ptr(x), ptr(y), cells), h)
    losses, status = np.zeros(cells), np.zeros(cells, dtype=np.int32)
    while not stop:
        for _ in range(int(rng.integers(1, 4))):
            check(lib.gprx_factorize_batch(h, cells, ptr(units), ptr(thetas), 7, ptr(losses), ptr(status)), h)
        time.sleep(float(rng.uniform(0.0, 0.003)))
    lib.gprx_destroy(h)


has_d2d = True
out = {"reps": reps, "hog": has_d2d, "cases": []}
check(lib.gprx_set_tuning(b"dag", 1))
for sched in ("0", "1"):
    for ni in ("4", "1"):
        os.environ["GPRX_DAG_SCHED"], os.environ["GPRX_DAG_NI"] = sched, ni
        for n in (448, 1024, 2112, 4096):
            rng = np.random.default_rng(n)
            g = rng.standard_normal((n, 96))
            full = np.vstack([g @ g.T / 96 + np.eye(n), rng.standard_normal((64, n))])
            stop = False
            th = threading.Thread(target=hog) if has_d2d else None
            if th:
                th.start()
            first, bad = None, 0
            t0 = time.perf_counter()
            for rep in range(reps):
                dA, dI = DeviceBuffer.from_array(full), DeviceBuffer(n * 64 * 8)
                info = C.c_int(0)
                check(lib.gprx_potrf(0, dA.ptr, n, n, 64, dI.ptr, C.byref(info)))
                cur = (np.tril(dA.to_array((n + 64, n))[:n]), dA.to_array((n + 64, n))[n:], dI.to_array((n // 64, 64, 64)))
                dA.free()
                dI.free()
                if first is None:
                    first = cur
                elif not all(np.array_equal(x, y) for x, y in zip(first, cur)):
                    bad += 1
            stop = True
            if th:
                th.join()
            out["cases"].append({"schedule": "lazy" if sched == "1" else "eager", "panel": int(ni), "n": n, "runs": reps, "differing_runs": bad,
                                 "seconds": time.perf_counter() - t0})
            print(out["cases"][-1], flush=True)
check(lib.gprx_set_tuning(b"dag", 0))
out["all_identical"] = all(c["differing_runs"] == 0 for c in out["cases"])
json.dump(out, open("gpurun_out/dag_soak.json", "w"), indent=1)
print("ALL IDENTICAL" if out["all_identical"] else "DIFFERENCES FOUND")
