#!/bin/bash
# Round 5: the fused sparse evaluation -- timing against the launch sequence, then rocprofv3 kernel stats (16 cells and 1 cell of
# N = 4096, M = 50, d = 10).  bash tools/prof_sgpr_r5.sh   (the library must be built: no build under the profiler)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 -m gpras_amd._build --stale > /dev/null || exit 1
export GPRX_NO_BUILD=1
python3 tools/sgpr_fused_probe.py 16 1 50 > gpurun_out/r05_sgpr_probe.log 2>&1 || { cat gpurun_out/r05_sgpr_probe.log; exit 1; }
cat gpurun_out/r05_sgpr_probe.log
for c in 16 1; do
  tag=r05_sgpr_${c}cells
  rm -rf gpurun_out/prof_$tag
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -o b -- python3 tools/sgpr_prof.py $c 200 > gpurun_out/${tag}.log 2>&1 || { echo "$tag failed"; tail -5 gpurun_out/${tag}.log; exit 1; }
  cp $(find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1) gpurun_out/${tag}_kernel_stats.csv
  echo "== $tag: $(grep cells= gpurun_out/${tag}.log | tail -1)"; cut -c1-150 gpurun_out/${tag}_kernel_stats.csv | head -9
done
