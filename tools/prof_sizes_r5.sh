#!/bin/bash
# Round 5: rocprofv3 kernel stats of ONE matrix (N = 4096, 8192, 16384) and of the lone objective + gradient (F2 single, the inner step of
# configs[2]).  On the GPU box: bash tools/prof_sizes_r5.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 -m gpras_amd._build --stale > /dev/null || exit 1
export GPRX_NO_BUILD=1
run() {  # tag, command...
  tag=$1; shift
  rm -rf gpurun_out/prof_$tag
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -o b -- "$@" > gpurun_out/${tag}.log 2>&1 || { echo "$tag failed"; tail -5 gpurun_out/${tag}.log; return 1; }
  cp $(find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1) gpurun_out/${tag}_kernel_stats.csv
  echo "== $tag: $(grep -v rocprofv3 gpurun_out/${tag}.log | tail -1)"; cut -c1-150 gpurun_out/${tag}_kernel_stats.csv | head -9
}
run r05_n4096_single python3 tools/large_probe.py 4096 8 &&
run r05_n8192_single python3 tools/large_probe.py 8192 8 &&
run r05_n16384_single python3 tools/large_probe.py 16384 12 &&
run r05_f2_single python3 tools/f2_single_prof.py
