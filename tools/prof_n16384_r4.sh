#!/bin/bash
# Round 4: rocprofv3 kernel stats of ONE N = 16384 factorisation under schedule variants (env knobs).  bash tools/prof_n16384_r4.sh <tag> [VAR=val ...]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
#  the library is built BEFORE any profiler line; under rocprofv3 a stale library is an error, not a fork + exec of hipcc
python3 -m gpras_amd._build --stale > /dev/null || exit 1
export GPRX_NO_BUILD=1
tag=$1; shift
for kv in "$@"; do export "$kv"; done
rm -rf gpurun_out/prof_$tag
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -o b -- python3 tools/large_probe.py 16384 12 > gpurun_out/${tag}.log 2>&1 || { echo "$tag failed"; tail -5 gpurun_out/${tag}.log; exit 1; }
cp $(find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1) gpurun_out/${tag}_kernel_stats.csv
echo "== $tag: $(tail -1 gpurun_out/${tag}.log)"; cut -c1-170 gpurun_out/${tag}_kernel_stats.csv | head -12
