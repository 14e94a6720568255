cd $GRAFT_REPO_ROOT
echo "--- comm debug"
cat > /tmp/commdiag.py <<'PY'
import sys
from gpras_amd import _lib
lib = _lib.load()
from gpras_amd.comm import Communicator
import numpy as np
try:
    c = Communicator.bootstrap(0, rank=0, world=1)
    print("OK gathered", c.all_gather(np.arange(3.0)))
    c.close()
except Exception as e:
    print("FAILED", e)
print("loaded:", sorted({l.split()[-1] for l in open("/proc/self/maps") if any(k in l for k in ("hsa", "amdhip", "rccl"))}))
PY
PYTHONPATH=$GRAFT_REPO_ROOT GPRX_COMM_DEBUG=1 NCCL_DEBUG=WARN timeout -k 10 120 python /tmp/commdiag.py 2>&1 | grep -v "alt_rsmi\|iommu" | tail -12 | cut -c1-500
echo "--- bench variants (batched only)"
for v in "0 0" "1 0" "1 1" "1 2"; do set -- $v; echo "DMA=$1 PFC=$2"; GPRX_GEMM_DMA=$1 GPRX_GEMM_PFC=$2 timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-extras --batched-only 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('  fits/s %.0f  ms/step %.2f  main gemm: %.1f TF/s (%d launches, %.0f us avg)  short-K: %s TF/s  panel avg %.1f us' % (d['value'], d['ms_per_step'], r['achieved'], r['launches_per_step'], r['avg_launch_us'], r['short_k_inblock_updates']['tflops'], r['panel_kernel']['avg_launch_us']))"; done
