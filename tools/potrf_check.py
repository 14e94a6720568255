"""Large-N correctness of the blocked Cholesky under every schedule (development aid)."""
import ctypes as C, sys
import numpy as np
from scipy.linalg import cholesky
sys.path.insert(0, ".")
from gpras_amd import _lib
from gpras_amd._lib import DeviceBuffer, check
lib = _lib.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
extra = 64
rng = np.random.default_rng(1)
g = rng.standard_normal((n, 96))
spd = g @ g.T / 96 + np.eye(n)
rhs = rng.standard_normal((extra, n))
L_ref = cholesky(spd, lower=True)
full = np.vstack([spd, rhs])
nola = int(sys.argv[2]) if len(sys.argv) > 2 else 0
lib.gprx_set_tuning(b"no_lookahead", nola)
for pw in [int(a) for a in (sys.argv[3].split(',') if len(sys.argv) > 3 else ['64','128'])]:
    for ob in (128, 256, 512):
        for tile in (64, 128):
            lib.gprx_set_tuning(b"panel_width", pw); lib.gprx_set_tuning(b"outer_block", ob); lib.gprx_set_tuning(b"update_tile", tile)
            errs = []
            for rep in range(3):
                dA = DeviceBuffer.from_array(full); dI = DeviceBuffer(n * 64 * 8)
                info = C.c_int(0)
                rc = lib.gprx_potrf(0, dA.ptr, n, n, extra, dI.ptr, C.byref(info))
                out = dA.to_array((n + extra, n))
                L = np.tril(out[:n])
                err = np.abs(L - L_ref)
                bad_rows = np.where(err.max(axis=1) > 1e-9)[0]
                bad_cols = np.where(err.max(axis=0) > 1e-9)[0]
                errs.append((float(err.max()), (bad_rows.min(), bad_rows.max(), bad_cols.min(), bad_cols.max()) if bad_rows.size else None))
                dA.free(); dI.free()
            print(f"n={n} panel {pw} ob {ob} tile {tile}: ", errs, flush=True)
