"""Tile-DAG Cholesky (potrf_dag.h) on the GPU box: factor against scipy for several sizes (forced through the "dag" knob),
repeated runs compared bit for bit (a stale hand-off would show as a differing factor), and stage timings of gprx_factorize
with the DAG on and off."""
import ctypes as C
import json
import sys
import time

import numpy as np
from scipy.linalg import cholesky, solve_triangular

sys.path.insert(0, ".")
from gpras_amd import _build, _lib  # noqa: E402
from gpras_amd._lib import DeviceBuffer, check, ptr  # noqa: E402
from gpras_amd.model import NOISE_LOWER, softplus_inv  # noqa: E402
from gpras_amd.synth import make_regression  # noqa: E402

_build.build()
lib = _lib.load()
out = {}


def factor(n, extra, seed):
    rng = np.random.default_rng(seed)
    g = rng.standard_normal((n, n + 8))
    spd = g @ g.T / n + 0.5 * np.eye(n)
    rhs = rng.standard_normal((extra, n))
    full = np.vstack([spd, rhs])
    dA = DeviceBuffer.from_array(full)
    dI = DeviceBuffer(n * 64 * 8)
    info = C.c_int(-1)
    t0 = time.perf_counter()
    check(lib.gprx_potrf(0, dA.ptr, n, n, extra, dI.ptr, C.byref(info)))
    dt = time.perf_counter() - t0
    got = dA.to_array((n + extra, n))
    inv = dI.to_array((n // 64, 64, 64))
    dA.free()
    dI.free()
    return spd, rhs, got, inv, info.value, dt


sizes = [int(a) for a in sys.argv[1:]] or [64, 128, 448, 1024, 2048, 4096]
for n in sizes:
    extra = 64
    check(lib.gprx_set_tuning(b"dag", 1))
    spd, rhs, got, inv, info, dt = factor(n, extra, n)
    L_ref = cholesky(spd, lower=True)
    L = np.tril(got[:n])
    err = float(np.max(np.abs(L - L_ref)) / np.max(np.abs(L_ref)))
    beta_ref = solve_triangular(L_ref, rhs.T, lower=True).T
    berr = float(np.max(np.abs(got[n:] - beta_ref)) / np.max(np.abs(beta_ref)))
    ierr = max(float(np.max(np.abs(inv[i] - np.linalg.inv(L_ref[64 * i : 64 * i + 64, 64 * i : 64 * i + 64])))) for i in range(n // 64))
    same = True
    for rep in range(3):
        _, _, again, _, _, _ = factor(n, extra, n)
        same = same and np.array_equal(np.tril(again[:n]), L) and np.array_equal(again[n:], got[n:])
    check(lib.gprx_set_tuning(b"dag", -1))
    _, _, base, _, _, _ = factor(n, extra, n)
    vs_base = float(np.max(np.abs(np.tril(base[:n]) - L)) / np.max(np.abs(L)))
    out[f"n{n}"] = {"info": info, "L_rel_err": err, "beta_rel_err": berr, "inv_abs_err": ierr, "repeat_bitwise": bool(same), "vs_launch_schedule": vs_base}
    print(n, out[f"n{n}"], flush=True)

# stage timings through the handle (kernel build, Cholesky, solves)
for n in (4096, 2048, 8192):
    if n > max(sizes):
        continue
    x, y, _ = make_regression(n, 8, n_outputs=1, n_test=0, config=2, unit=0)
    theta = np.ascontiguousarray([softplus_inv(1.0), softplus_inv(np.mean(np.abs(x))), softplus_inv(1.0 - NOISE_LOWER)], dtype=np.float64)
    res = {}
    for dag in (1, -1):
        check(lib.gprx_set_tuning(b"dag", dag))
        h = C.c_void_p()
        check(lib.gprx_create(0, n, 8, 0, _lib.KERNEL_IDS["RBF"], 0, C.byref(h)))
        check(lib.gprx_set_data(h, ptr(x), ptr(y), 1), h)
        loss = C.c_double()
        for _ in range(3):
            check(lib.gprx_factorize(h, 0, ptr(theta), None, 7, C.byref(loss)), h)
        t0 = time.perf_counter()
        for _ in range(20):
            check(lib.gprx_factorize(h, 0, ptr(theta), None, 7, C.byref(loss)), h)
        dt = (time.perf_counter() - t0) / 20
        ms = (C.c_double * 4)()
        lib.gprx_last_timings(h, ms)
        res["dag" if dag > 0 else "launches"] = {"fit_ms": 1e3 * dt, "kmat_ms": ms[0], "chol_ms": ms[1], "solves_ms": ms[2], "loss": loss.value}
        lib.gprx_destroy(h)
    res["loss_rel_gap"] = abs(res["dag"]["loss"] - res["launches"]["loss"]) / abs(res["launches"]["loss"])
    out[f"fit_n{n}"] = res
    print(n, res, flush=True)
check(lib.gprx_set_tuning(b"dag", 0))
import os
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/dag_probe.json", "w"), indent=1)
