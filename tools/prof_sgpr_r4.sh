#!/bin/bash
# Round 4: rocprofv3 kernel stats of the batched sparse loss + gradient (16 cells and 1 cell of N = 4096, M = 50, d = 10).  bash tools/prof_sgpr_r4.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
#  the library is built BEFORE any profiler line; under rocprofv3 a stale library is an error, not a fork + exec of hipcc
python3 -m gpras_amd._build --stale > /dev/null || exit 1
export GPRX_NO_BUILD=1
for c in 16 1; do
  tag=r04_sgpr_${c}cells
  rm -rf gpurun_out/prof_$tag
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -o b -- python3 tools/sgpr_prof.py $c 50 > gpurun_out/${tag}.log 2>&1 || { echo "$tag failed"; exit 1; }
  cp $(find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1) gpurun_out/${tag}_kernel_stats.csv
  echo "== $tag: $(grep cells= gpurun_out/${tag}.log | tail -1)"; cut -c1-120 gpurun_out/${tag}_kernel_stats.csv | head -8
done
