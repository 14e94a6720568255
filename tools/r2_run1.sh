# round 2, run 1: communicator tests, the distributed bench path with one rank (RCCL through gprx_comm_*), baseline bench line
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_comm.py tests/test_gpu_distance_form.py tests/test_gpu_gpras.py -q > gpurun_out/r2_run1_tests.log 2>&1 || { tail -30 gpurun_out/r2_run1_tests.log; exit 1; }
tail -3 gpurun_out/r2_run1_tests.log
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 5 --warmup 2 --no-extras > gpurun_out/r2_bench_dist1.json 2> gpurun_out/r2_bench_dist1.err || { tail -30 gpurun_out/r2_bench_dist1.err; exit 1; }
tail -c 600 gpurun_out/r2_bench_dist1.json
timeout -k 10 500 python bench.py > gpurun_out/r2_bench_a.json 2> gpurun_out/r2_bench_a.err || { tail -30 gpurun_out/r2_bench_a.err; exit 1; }
python - <<'PY'
import json
d=json.load(open('gpurun_out/r2_bench_a.json'))
print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['frac'], d.get('single_cell_ms_per_fit'), d.get('cpu_baseline'), d.get('extra_error'))
e=d.get('extra',{})
print({k:v for k,v in e.items() if not isinstance(v,dict)})
print(e.get('other_sizes'))
PY
