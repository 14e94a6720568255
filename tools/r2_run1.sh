# round 2, run 1: communicator diagnostics, the distributed bench path with one rank, baseline bench line
cd $GRAFT_REPO_ROOT
rocminfo 2>/dev/null | grep -c gfx950
echo "--- comm (no torch in the process)"
NCCL_DEBUG=INFO timeout -k 10 120 python -c "
from gpras_amd.comm import Communicator
import numpy as np
c = Communicator.bootstrap(0, rank=0, world=1)
print('gathered', c.all_gather(np.arange(4.0)))
c.close()
" > gpurun_out/r2_comm_diag.log 2>&1; echo rc=$?; grep -i "warn\|error\|fail\|gathered" gpurun_out/r2_comm_diag.log | head -20
echo "--- comm (torch imported first)"
NCCL_DEBUG=INFO timeout -k 10 120 python -c "
import torch
torch.cuda.init()
from gpras_amd.comm import Communicator
import numpy as np
c = Communicator.bootstrap(0, rank=0, world=1)
print('gathered', c.all_gather(np.arange(4.0)))
c.close()
" > gpurun_out/r2_comm_diag2.log 2>&1; echo rc=$?; grep -i "warn\|error\|fail\|gathered" gpurun_out/r2_comm_diag2.log | head -20
echo "--- distributed bench, one rank"
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 5 --warmup 2 --no-extras > gpurun_out/r2_bench_dist1.json 2> gpurun_out/r2_bench_dist1.err; echo rc=$?; tail -5 gpurun_out/r2_bench_dist1.err; tail -c 400 gpurun_out/r2_bench_dist1.json
echo "--- bench"
timeout -k 10 500 python bench.py > gpurun_out/r2_bench_a.json 2> gpurun_out/r2_bench_a.err || { tail -30 gpurun_out/r2_bench_a.err; exit 1; }
python - <<'PY'
import json
d=json.load(open('gpurun_out/r2_bench_a.json'))
print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['frac'], d.get('single_cell_ms_per_fit'), d.get('cpu_baseline'), d.get('extra_error'))
e=d.get('extra',{})
print({k:v for k,v in e.items() if not isinstance(v,dict)})
print(e.get('other_sizes'))
PY
