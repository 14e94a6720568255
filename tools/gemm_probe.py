"""GEMM-only workload for rocprofv3 counter runs (development aid)."""
import sys
import numpy as np
sys.path.insert(0, ".")
from gpras_amd import _lib
from gpras_amd._lib import DeviceBuffer, check
lib = _lib.load()
m = int(sys.argv[1]); k = int(sys.argv[2]); flags = int(sys.argv[3]); tile = int(sys.argv[4]); lda = int(sys.argv[5]) if len(sys.argv) > 5 else k
reps = 5
rng = np.random.default_rng(0)
a = rng.standard_normal((m, lda))
dA = DeviceBuffer.from_array(a); dC = DeviceBuffer(m * m * 8)
for _ in range(reps):
    check(lib.gprx_gemm(0, 0, 1, m, m, k, -1.0, dA.ptr, lda, dA.ptr, lda, 1.0, dC.ptr, m, flags, tile))
print("done")
