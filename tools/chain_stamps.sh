#!/bin/bash
# cycle stamps (s_memtime) inside sub-panel step 3 of the tile-DAG chain (a -DGPRX_CHAIN_STAMPS build): where a sub-panel step spends its time
cd $GRAFT_REPO_ROOT
cp gpras_amd/libgprx.so /tmp/libgprx_keep.so
export GPRX_EXTRA_FLAGS=-DGPRX_CHAIN_STAMPS   # (every unit is rebuilt with the flag: ~1 min on the box; the library is restored below)
python3 -m gpras_amd._build --stale > /dev/null || exit 1
python3 - <<'PY'
import ctypes as C, sys
import numpy as np
sys.path.insert(0, ".")
from gpras_amd import _lib
from gpras_amd._lib import DeviceBuffer, check
lib = _lib.load()
n = 256   # four chain steps, no workers worth mentioning: the chain alone on the chip
rng = np.random.default_rng(1)
g = rng.standard_normal((n, n + 8)); spd = g @ g.T / n + 0.5 * np.eye(n)
check(lib.gprx_set_tuning(b"dag", 1))
names = ["acc->LDS", "barrier 1", "read 8x8 block", "factor 8x8", "solve own row + write", "barrier 2", "fragments + MFMA issue", "solved values back"]
rows = []
for rep in range(5):
    dA, dI = DeviceBuffer.from_array(spd), DeviceBuffer(n * 64 * 8)
    info = C.c_int(0)
    check(lib.gprx_potrf(0, dA.ptr, n, n, 0, dI.ptr, C.byref(info)))
    buf = (C.c_ulonglong * 16)()
    lib.gprx_chain_stamps(buf)
    st = np.array(buf[:9], dtype=np.int64)
    rows.append(np.diff(st))
    dA.free(); dI.free()
rows = np.array(rows)
print("cycles per phase of sub-panel step 3 (median of 5 factorisations; s_memtime ticks):")
for nm, v in zip(names, np.median(rows, axis=0)):
    print(f"  {nm:28s} {v:7.0f}")
print(f"  {'total':28s} {np.median(rows.sum(axis=1)):7.0f}")
PY
cp /tmp/libgprx_keep.so gpras_amd/libgprx.so
