cd $GRAFT_REPO_ROOT
cat > /tmp/commdiag.py <<'PY'
import sys, os
mode = sys.argv[1]
if mode == "sk":
    import sklearn.cluster, scipy.linalg
if mode == "pytestmod":
    import pytest
from gpras_amd import _lib
lib = _lib.load()
from gpras_amd.comm import Communicator
import numpy as np
try:
    c = Communicator.bootstrap(0, rank=0, world=1)
    print(mode, "OK gathered", c.all_gather(np.arange(3.0)))
    c.close()
except Exception as e:
    print(mode, "FAILED", e)
maps = sorted({l.split()[-1] for l in open("/proc/self/maps") if any(k in l for k in ("hsa", "amdhip", "rccl", "libdrm"))})
print(mode, "loaded:", maps)
PY
for mode in plain; do PYTHONPATH=$GRAFT_REPO_ROOT GPRX_COMM_DEBUG=1 NCCL_DEBUG=WARN timeout -k 10 120 python /tmp/commdiag.py $mode 2>&1 | grep -v "alt_rsmi\|iommu" | tail -6 | cut -c1-600; done
echo "--- env"; env | grep -i "LD_LIBRARY\|ROCM\|HSA\|HIP" | head
echo "--- pytest with INFO"
NCCL_DEBUG=INFO timeout -k 10 200 python -m pytest tests/test_gpu_comm.py -q -s -k world_of_one 2>&1 | grep -v "alt_rsmi\|iommu\|Channel" | grep -i "librccl\|hsa\|warn\|passed\|failed\|ROCr\|LOADED" | head -30 | cut -c1-600
