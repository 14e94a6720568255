cd $GRAFT_REPO_ROOT
echo "--- comm debug"
PYTHONPATH=$GRAFT_REPO_ROOT GPRX_COMM_DEBUG=1 NCCL_DEBUG=WARN timeout -k 10 120 python tools/commdiag.py 2>&1 | grep -v "alt_rsmi\|iommu" | tail -8 | cut -c1-500
echo "--- comm pytest"
timeout -k 10 200 python -m pytest tests/test_gpu_comm.py -q 2>&1 | tail -3
echo "--- distributed bench, one rank"
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 5 --warmup 2 --no-extras --batched-only > gpurun_out/r2_bench_dist1.json 2> gpurun_out/r2_bench_dist1.err; echo rc=$?; tail -3 gpurun_out/r2_bench_dist1.err | cut -c1-300; head -c 300 gpurun_out/r2_bench_dist1.json; echo
echo "--- bench variants (batched only)"
run() { echo "$*"; env "$@" timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-extras --batched-only 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('  fits/s %.0f  ms/step %.2f  main gemm: %.1f TF/s (%d launches, %.0f us avg)  short-K: %.1f TF/s (%d launches, %.0f us avg)  panel avg %.1f us' % (d['value'], d['ms_per_step'], r['achieved'], r['launches_per_step'], r['avg_launch_us'], r['short_k_inblock_updates']['tflops'], r['short_k_inblock_updates']['launches_per_step'], r['short_k_inblock_updates']['avg_launch_us'], r['panel_kernel']['avg_launch_us']))"; }
run GPRX_K64_GEMM=0
run GPRX_K64_GEMM=1
run GPRX_BATCH_GROUPS=2
run GPRX_BATCH_GROUPS=2 GPRX_K64_GEMM=1
