"""Does the column-pair cell kernel give the single-column kernel's losses bit for bit at FULL load (N = 1024 x 512 cells)?  Each kernel runs in
a child process (the selection is read from the environment once per process); prints how many cells differ (development aid)."""
import json, os, subprocess, sys
CHILD = r"""
import ctypes as C, sys, json
import numpy as np
sys.path.insert(0, ".")
from gpras_amd import _lib
from gpras_amd._lib import check, ptr
from gpras_amd.model import NOISE_LOWER, softplus_inv
from gpras_amd.synth import make_regression
try:
    lib = _lib.load()
except _lib.GprxLibraryError:  # (an older build of the library: without the newest export)
    _lib.PROTOTYPES.pop("gprx_exp_probe", None)
    lib = _lib.load()
n, cells, reps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
x, y, _ = make_regression(n, 8, n_outputs=cells, n_test=0, config=2, unit=500)
theta = np.ascontiguousarray([softplus_inv(1.0), softplus_inv(np.mean(np.abs(x))), softplus_inv(1.0 - NOISE_LOWER)], dtype=np.float64)
thetas = np.ascontiguousarray(theta[None, :] + np.random.default_rng(7).uniform(-0.15, 0.15, size=(cells, 3)))
units = np.arange(cells, dtype=np.int32)
h = C.c_void_p()
check(lib.gprx_create(0, n, 8, 0, 0, 0, C.byref(h)))
check(lib.gprx_set_handle_tuning(h, b"cell_kernel", int(__import__("os").environ.get("GPRX_CELL_KERNEL", "1"))), h)
check(lib.gprx_set_data(h, ptr(x), ptr(y), cells), h)
out = []
for r in range(reps):
    losses, status = np.zeros(cells), np.zeros(cells, dtype=np.int32)
    lib.gprx_factorize_batch(h, cells, ptr(units), ptr(thetas), 7, ptr(losses), ptr(status))
    out.append([float.hex(v) for v in losses])
print(json.dumps(out))
"""
n, cells, reps = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1024, 512, 4)
def run(env):
    base = {k: v for k, v in os.environ.items() if not k.startswith("GPRX_CELL")}
    res = subprocess.run([sys.executable, "-c", CHILD, str(n), str(cells), str(reps)], capture_output=True, text=True, env=dict(base, **env))
    if res.returncode:
        print(res.stderr[-2000:]); sys.exit(1)
    return json.loads(res.stdout.strip().splitlines()[-1])
ref = run({"GPRX_CELL_SINGLE_COLUMN": "1"})
print("single column vs its own first repetition", [sum(a != b for a, b in zip(g, ref[0])) for g in ref], "cells differ")
seq = [float.fromhex(v) for v in run({"GPRX_CELL_KERNEL": "-1"})[0]]
def gap(g):
    d = [abs(float.fromhex(a) - b) / abs(b) for a, b in zip(g, seq)]
    return f"max rel gap to the launch sequence {max(d):.2e}, cells beyond 1e-12: {sum(x > 1e-12 for x in d)}"
print("single column:", gap(ref[0]))
for name, env in (("launch sequence (vs its own first repetition)", {"GPRX_CELL_KERNEL": "-1"}), ("pair, K from memory", {}), ("pair, K built in the kernel", {"GPRX_CELL_BUILD_K": "1"})):
    got = run(env)
    base = got[0] if "own first" in name else ref[0]
    print(name, [sum(a != b for a, b in zip(g, base)) for g in got], "cells differ per repetition;", gap(got[-1]))
