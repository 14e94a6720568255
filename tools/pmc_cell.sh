#!/bin/bash
# PMC passes over the one-workgroup-per-cell kernel at N = 1024 x 512 cells (tools/batch_n1024.py): ONE counter group per run, no tracing
# with --pmc, and FETCH_SIZE / WRITE_SIZE each in a pass of its own (together they exceed the TCC's 4 slots: FETCH_SIZE costs 3,
# WRITE_SIZE 2 -- MI355X_MICROARCH.md, "rocprofv3 PMC slots"; round 3 asked for both in one pass and rocprofv3 aborted with error 38).
#     bash tools/pmc_cell.sh [tag]        -> gpurun_out/<tag>_pmc_cell_kernel.json (+ kernel stats of an un-instrumented run)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
#  the library is built BEFORE any profiler line; under rocprofv3 a stale library is an error, not a fork + exec of hipcc
python3 -m gpras_amd._build --stale > /dev/null || exit 1
export GPRX_NO_BUILD=1
tag=${1:-r04}
export GPRX_CELL_KERNEL=1
P="python3 tools/batch_n1024.py 1024 512"
D=gpurun_out/pmc_cell_$tag
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats -o s -- $P > gpurun_out/${tag}_cell_stats.log 2>&1 || { echo "kernel-trace pass failed"; tail -5 gpurun_out/${tag}_cell_stats.log; exit 1; }
cp $(find $D/stats -name "*kernel_stats.csv" | head -1) gpurun_out/${tag}_n1024_batched_cell_kernel_kernel_stats.csv
args=""
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $D/p$i -o c -- $P > gpurun_out/${tag}_pmc_cell_p$i.log 2>&1 || { echo "failed: $set"; tail -5 gpurun_out/${tag}_pmc_cell_p$i.log; exit 1; }
  args="$args p$i=$D/p$i"
done
python3 tools/pmc_summary.py gpurun_out/${tag}_pmc_cell_kernel.json $args
