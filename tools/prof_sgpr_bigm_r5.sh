#!/bin/bash
# Round 5: kernel stats of the sparse evaluation beyond one 64-block of inducing points (M = 128 and 300, 16 cells, N = 4096, d = 10):
# the general launch sequence.  bash tools/prof_sgpr_bigm_r5.sh   (the library must be built: no build under the profiler)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 -m gpras_amd._build --stale > /dev/null || exit 1
export GPRX_NO_BUILD=1
for m in 128 300; do
  tag=r05_sgpr_16cells_M${m}
  rm -rf gpurun_out/prof_$tag
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -o b -- python3 tools/sgpr_prof.py 16 100 $m > gpurun_out/${tag}.log 2>&1 || { echo "$tag failed"; tail -5 gpurun_out/${tag}.log; exit 1; }
  cp $(find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1) gpurun_out/${tag}_kernel_stats.csv
  echo "== $tag: $(grep cells= gpurun_out/${tag}.log | tail -1)"; cut -c1-170 gpurun_out/${tag}_kernel_stats.csv | head -30
done
