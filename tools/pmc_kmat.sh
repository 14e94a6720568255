#!/bin/bash
# Round 4: LDS-conflict and instruction counters of the kernel build inside the batched bench step (one --pmc pass, no tracing).  bash tools/pmc_kmat.sh <tag>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
#  the library is built BEFORE any profiler line; under rocprofv3 a stale library is an error, not a fork + exec of hipcc
python3 -m gpras_amd._build --stale > /dev/null || exit 1
export GPRX_NO_BUILD=1
tag=${1:-r04_kmat}
B3="python3 bench.py --steps 3 --warmup 1 --no-extras --batched-only"
timeout -k 10 400 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU --output-format csv -d gpurun_out/pmc_${tag}_lds -o l -- $B3 > gpurun_out/${tag}_pmc_lds.log 2>&1 || exit 1
python3 tools/pmc_summary.py gpurun_out/${tag}_pmc_summary.json lds=gpurun_out/pmc_${tag}_lds | grep kmat
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -o b -- $B3 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_prof.log || exit 1
grep kmat $(find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1) | cut -c1-140
