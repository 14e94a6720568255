#!/bin/bash
# Round 5: rocprofv3 kernel stats of the resident Adam loop (16 modes and 1 mode).  bash tools/prof_sgpr_adam_r5.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 -m gpras_amd._build --stale > /dev/null || exit 1
export GPRX_NO_BUILD=1
for c in 16 1; do
  tag=r05_sgpr_adam_${c}modes
  rm -rf gpurun_out/prof_$tag
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -o b -- python3 tools/sgpr_adam_prof.py $c 400 > gpurun_out/${tag}.log 2>&1 || { echo "$tag failed"; tail -5 gpurun_out/${tag}.log; exit 1; }
  cp $(find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1) gpurun_out/${tag}_kernel_stats.csv
  echo "== $tag: $(grep modes= gpurun_out/${tag}.log | tail -1)"; cut -c1-150 gpurun_out/${tag}_kernel_stats.csv | head -9
done
