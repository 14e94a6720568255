#!/bin/bash
# Round 4: rocprofv3 kernel stats of the other sizes of the target (one matrix at N = 4096 with both schedules, N = 16384; many
# cells at N = 1024 and at N = 512 with both batched schedules).  On the GPU box: bash tools/prof_sizes_r4.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
#  the library is built BEFORE any profiler line; under rocprofv3 a stale library is an error, not a fork + exec of hipcc
python3 -m gpras_amd._build --stale > /dev/null || exit 1
export GPRX_NO_BUILD=1
run() {  # tag, env assignment, command...
  tag=$1; shift; envs=$1; shift
  rm -rf gpurun_out/prof_$tag
  ( export $envs; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -o b -- "$@" > gpurun_out/${tag}.log 2>&1 ) || { echo "$tag failed"; tail -5 gpurun_out/${tag}.log; return; }
  cp $(find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1) gpurun_out/${tag}_kernel_stats.csv
  echo "== $tag: $(tail -1 gpurun_out/${tag}.log)"; cut -c1-150 gpurun_out/${tag}_kernel_stats.csv | head -7
}
run r04_n4096_single GPRX_X=0 python3 tools/large_probe.py 4096 8
run r04_n8192_single GPRX_X=0 python3 tools/large_probe.py 8192 8
run r04_n1024_batched GPRX_CELL_KERNEL=-1 python3 tools/batch_n1024.py 1024 512
run r04_n512_batched_cell_kernel GPRX_X=0 python3 tools/batch_n1024.py 512 512
run r04_f2_batched GPRX_X=0 python3 tools/f2_prof.py 128
