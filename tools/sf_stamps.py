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
names = {0: ("prep", ["par/Z in", "Kuu", "chain", "L, L^-1 out"]),
         32: ("pass1", ["fragments", "-> tile 0", "stage", "r2", "exp + P", "mfma A'", "A' store", "mfma S + u", "tiles 1..3", "slab out"]),
         64: ("mid", ["slab sum + L^-1 in", "chain", "LB out, c, logdet", "trsv m", "5 products + W, GQ out"]),
         96: ("pass2", ["fragments", "-> tile 0", "stage", "r2", "g, h + P", "mfma WP + P^T m + sum", "WP store, G, w v h", "dZ loop", "tiles 1..3", "partials out"]),
         128: ("final", ["all"])}
print("prep chain steps (clocks):", [int(w[9 + i] - w[8 + i]) for i in range(8)])
real = {b: w[b + 31] for b in names}
t0 = min(real.values())
for base, (kname, phases) in names.items():
    s = w[base:base + len(phases) + 1]
    print(f"{kname}: entry at +{(real[base] - t0) * 10} ns; total {int(s[len(phases)] - s[0])} clocks")
    for i, ph in enumerate(phases):
        print(f"    {ph:32s} {int(s[i + 1] - s[i]):8d}")
