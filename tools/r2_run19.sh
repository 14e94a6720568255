#!/bin/bash
# kmat kernel at N = 1024 / 2048 / 4096 with different cell counts: time per tile
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cfg in "1024 512" "1024 128" "2048 256" "4096 128" "512 2048"; do
  set -- $cfg
  rm -rf gpurun_out/kmat_trace
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kmat_trace -o s -- python3 tools/batch_prof.py $1 $2 3 > gpurun_out/kmat_trace.log 2>&1
  f=$(find gpurun_out/kmat_trace -name "*kernel_stats.csv" | head -1)
  echo "N=$1 cells=$2: $(tail -1 gpurun_out/kmat_trace.log)"
  grep -E "kmat_kernel|potrf_rows|trsv_bwd|potrf_panel" $f | cut -d, -f1-4 | cut -c1-140
done
rm -rf gpurun_out/kmat_trace
