#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cfg in "4096 8" "16384 12"; do
  set -- $cfg
  rm -rf gpurun_out/st
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/st -o s -- python3 tools/large_probe.py $1 $2 > gpurun_out/st.log 2>&1
  tail -1 gpurun_out/st.log
  cp $(find gpurun_out/st -name "*kernel_stats.csv" | head -1) gpurun_out/single_n$1_stats.csv
  head -8 gpurun_out/single_n$1_stats.csv | cut -c1-150
done
rm -rf gpurun_out/st
timeout -k 10 500 python3 bench.py > gpurun_out/r2_bench_c.json 2> gpurun_out/r2_bench_c.err; echo bench rc=$?
