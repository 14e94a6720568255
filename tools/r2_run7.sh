cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_blocks.py -q -k "block_column or potrf" > gpurun_out/r2_run7_tests.log 2>&1; echo rc=$?; tail -12 gpurun_out/r2_run7_tests.log | cut -c1-300
for n in 16384 8192; do
echo "--- N=$n old schedule"; GPRX_LARGE_MIN=-1 timeout -k 10 120 python tools/large_probe.py $n 12
for r in 16 0 8 32; do for ob in 1024 512; do echo "--- N=$n block-column schedule, reserved CUs $r, outer block $ob"; GPRX_LARGE_RESERVED_CUS=$r timeout -k 10 120 python tools/large_probe.py $n 12 $ob; done; done
done
timeout -k 10 600 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_edges.py -q > gpurun_out/r2_run7_tests2.log 2>&1; echo rc=$?; tail -8 gpurun_out/r2_run7_tests2.log | cut -c1-300
