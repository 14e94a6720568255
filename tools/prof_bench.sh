# rocprofv3 passes over the judged bench command (kernel stats, then the two PMC passes separately)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bench6 -o b -- python3 bench.py --steps 5 --warmup 2 --no-extras --batched-only > gpurun_out/prof_bench6.log 2>&1 || exit 1
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch6 -o f -- python3 bench.py --steps 3 --warmup 1 --no-extras --batched-only > gpurun_out/pmc_fetch6.log 2>&1 || exit 1
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write6 -o w -- python3 bench.py --steps 3 --warmup 1 --no-extras --batched-only > gpurun_out/pmc_write6.log 2>&1 || exit 1
ls gpurun_out/prof_bench6 gpurun_out/pmc_fetch6 gpurun_out/pmc_write6
