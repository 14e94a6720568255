cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_blocks.py tests/test_gpu_exact.py -q > gpurun_out/r2_run12_tests.log 2>&1; echo rc=$?; tail -6 gpurun_out/r2_run12_tests.log | cut -c1-300
for n in 16384 8192 4096; do for sl in 0 2 3 4; do echo "--- N=$n TAIL slots $sl"; GPRX_TAIL_SLOTS=$sl timeout -k 10 120 python tools/large_probe.py $n 12; done; done
echo "--- N=16384 slots 3 outer 1024"; GPRX_TAIL_SLOTS=3 timeout -k 10 120 python tools/large_probe.py 16384 12 1024
echo "--- N=16384 slots 2 outer 1024"; GPRX_TAIL_SLOTS=2 timeout -k 10 120 python tools/large_probe.py 16384 12 1024
echo "--- N=2048"; for sl in 0 3; do GPRX_TAIL_SLOTS=$sl timeout -k 10 120 python tools/large_probe.py 2048 8; done
