#!/bin/bash
# Round 4: rocprofv3 kernel stats of the predict at the three sizes of the target.  On the GPU box: bash tools/prof_predict_r4.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
#  the library is built BEFORE any profiler line; under rocprofv3 a stale library is an error, not a fork + exec of hipcc
python3 -m gpras_amd._build --stale > /dev/null || exit 1
export GPRX_NO_BUILD=1
for cfg in "1024 8" "4096 8" "16384 12"; do
  set -- $cfg
  tag=${PREFIX:-r04}_predict_n$1
  rm -rf gpurun_out/prof_$tag
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -o b -- python3 tools/predict_probe.py $1 $2 > gpurun_out/${tag}.log 2>&1 || { echo "$tag failed"; tail -5 gpurun_out/${tag}.log; exit 1; }
  cp $(find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1) gpurun_out/${tag}_kernel_stats.csv
  echo "== $tag: $(tail -1 gpurun_out/${tag}.log)"; cut -c1-150 gpurun_out/${tag}_kernel_stats.csv | head -8
done
