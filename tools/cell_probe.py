"""One-workgroup-per-cell Cholesky (potrf_cell.h) against the batched launch sequence: losses, predictions of a selected slot,
and fits/s at N = 1024 (512 cells) and smaller sizes."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from gpras_amd import _build, _lib  # noqa: E402
from gpras_amd._lib import check, ptr  # noqa: E402
from gpras_amd.model import NOISE_LOWER, softplus_inv  # noqa: E402
from gpras_amd.synth import make_regression  # noqa: E402

_build.build()
lib = _lib.load()
out = {}
for n, cells in ((1024, 512), (512, 512), (200, 300), (1000, 256)):
    x, y, xs = make_regression(n, 8, n_outputs=cells, n_test=50, config=2, unit=500)
    theta = np.ascontiguousarray([softplus_inv(1.0), softplus_inv(np.mean(np.abs(x))), softplus_inv(1.0 - NOISE_LOWER)], dtype=np.float64)
    spread = np.random.default_rng(7).uniform(-0.15, 0.15, size=(cells, 3))
    thetas = np.ascontiguousarray(theta[None, :] + spread)
    units = np.arange(cells, dtype=np.int32)
    res = {}
    for mode in (-1, 1):
        h = C.c_void_p()
        check(lib.gprx_create(0, n, 8, 0, _lib.KERNEL_IDS["RBF"], 0, C.byref(h)))
        check(lib.gprx_set_handle_tuning(h, b"cell_kernel", mode), h)  # (per handle: nothing process-wide to restore)
        check(lib.gprx_set_data(h, ptr(x), ptr(y), cells), h)
        losses, status = np.zeros(cells), np.zeros(cells, dtype=np.int32)
        for _ in range(2):
            check(lib.gprx_factorize_batch(h, cells, ptr(units), ptr(thetas), 7, ptr(losses), ptr(status)), h)
        t0 = time.perf_counter()
        reps = 10
        for _ in range(reps):
            check(lib.gprx_factorize_batch(h, cells, ptr(units), ptr(thetas), 7, ptr(losses), ptr(status)), h)
        dt = (time.perf_counter() - t0) / reps
        check(lib.gprx_select_slot(h, cells // 3), h)
        mean, var = np.empty(50), np.empty(50)
        check(lib.gprx_predict(h, ptr(xs), 50, ptr(mean), ptr(var), 1), h)
        res[mode] = (losses.copy(), mean, var, dt, status.copy())
        lib.gprx_destroy(h)
    a, b = res[-1], res[1]
    out[f"n{n}_c{cells}"] = {
        "launches_ms": 1e3 * a[3], "cell_kernel_ms": 1e3 * b[3], "fits_per_s_launches": cells / a[3], "fits_per_s_cell_kernel": cells / b[3],
        "tflops_cell_kernel": cells * n**3 / 3 / b[3] / 1e12,
        "loss_rel_gap_max": float(np.max(np.abs(a[0] - b[0]) / np.abs(a[0]))), "mean_gap": float(np.max(np.abs(a[1] - b[1])) / np.max(np.abs(a[1]))),
        "var_gap": float(np.max(np.abs(a[2] - b[2]) / a[2])), "status_ok": bool(np.all(a[4] == 0) and np.all(b[4] == 0)),
    }
    print(n, cells, out[f"n{n}_c{cells}"], flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/cell_probe.json", "w"), indent=1)
