"""Phase durations inside the five fused sparse kernels (workgroup (0, 0) of each): development aid.  argv: cells [m] [d]"""
import ctypes as C
import sys
import numpy as np
sys.path.insert(0, ".")
from gpras_amd import _lib
from gpras_amd._lib import check
from gpras_amd.gpr import GPRAS
from gpras_amd.synth import make_regression

lib = _lib.load()
lib.gprx_sf_stamps.restype = C.c_int
lib.gprx_sf_stamps.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 16
m = int(sys.argv[2]) if len(sys.argv) > 2 else 50
d = int(sys.argv[3]) if len(sys.argv) > 3 else 10
n = 4096
x, y, _ = make_regression(n, d, n_outputs=cells, n_test=0, config=6, unit=1)
g = GPRAS("RBF")
g._init_models(x.astype(np.float64), y.astype(np.float64), m, "grid")
eng = g.engine
units = np.arange(cells, dtype=np.int32)
thetas = np.stack([g.models[i].theta() for i in range(cells)])
zs = np.stack([g.models[i].Z for i in range(cells)])
check(lib.gprx_sf_stamps(eng._h, 1, None))
for _ in range(5):
    eng.objective_batch(units, thetas, 15, True, zs=zs)
out = (C.c_ulonglong * 160)()
check(lib.gprx_sf_stamps(eng._h, 1, out))
w = np.array(list(out), dtype=np.uint64).astype(np.int64)
# kernel -> (name, [(phase, first stamp, last stamp)]): the stamp indices of SF_STAMP in the kernels
names = {0: ("prep", [("par / Z in", 0, 1), ("Kuu", 1, 2), ("chain", 2, 3), ("L, L^-1 out", 3, 4)]),
         32: ("pass1", [("fragments, Z staged", 0, 1), ("tile 0 staged", 1, 2), ("r2", 2, 3), ("exp + P", 3, 4), ("mfma A'", 4, 5), ("A' store", 5, 6),
                        ("mfma S + u", 6, 7), ("tiles 1..3", 7, 8), ("slab out", 8, 9)]),
         64: ("mid", [("slab sum + L^-1 in", 0, 1), ("chain", 1, 2), ("LB out, c, logdet", 2, 3), ("trsv m", 3, 4), ("5 products + W, GQ out", 4, 5)]),
         96: ("pass2", [("fragments", 0, 1), ("tile 0 staged", 1, 2), ("r2", 2, 3), ("g, h + P", 3, 4), ("mfma WP + P^T m + sum", 4, 5), ("WP store, G, w v h", 5, 6),
                        ("dZ loop", 6, 7), ("tiles 1..3, Kuu slice, partials out", 7, 9)]),
         128: ("final", [("all", 0, 1)])}
print("prep chain steps (clocks):", [int(w[9 + i] - w[8 + i]) for i in range(8)])
real = {b: w[b + 31] for b in names}
t0 = min(real.values())
for base, (kname, phases) in names.items():
    first, last = phases[0][1], phases[-1][2]
    print(f"{kname}: entry at +{(real[base] - t0) * 10} ns; total {int(w[base + last] - w[base + first])} clocks")
    for ph, a, b in phases:
        print(f"    {ph:36s} {int(w[base + b] - w[base + a]):8d}")
