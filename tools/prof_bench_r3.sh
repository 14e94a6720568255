#!/bin/bash
# Round 3: rocprofv3 passes over the judged bench command (batched step only: every launch in the trace belongs to a step).
# Kernel stats first, then each PMC group in a pass of its own (no tracing together with --pmc).  On the GPU box:
#     bash tools/prof_bench_r3.sh <tag> [pmc]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
#  the library is built BEFORE any profiler line; under rocprofv3 a stale library is an error, not a fork + exec of hipcc
python3 -m gpras_amd._build --stale > /dev/null || exit 1
export GPRX_NO_BUILD=1
tag=${1:-r03_bench}
B="python3 bench.py --steps 10 --warmup 2 --no-extras --batched-only"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -o b -- $B > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_prof.log || exit 1
cp $(find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1) gpurun_out/${tag}_kernel_stats.csv
cut -c1-180 gpurun_out/${tag}_kernel_stats.csv | head -14
if [ "$2" = "pmc" ]; then
  B3="python3 bench.py --steps 3 --warmup 1 --no-extras --batched-only"
  timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_${tag}_fetch -o f -- $B3 > gpurun_out/${tag}_pmc_fetch.log 2>&1 || exit 1
  timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_${tag}_write -o w -- $B3 > gpurun_out/${tag}_pmc_write.log 2>&1 || exit 1
  timeout -k 10 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d gpurun_out/pmc_${tag}_mfma -o m -- $B3 > gpurun_out/${tag}_pmc_mfma.log 2>&1 || exit 1
  timeout -k 10 400 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU --output-format csv -d gpurun_out/pmc_${tag}_lds -o l -- $B3 > gpurun_out/${tag}_pmc_lds.log 2>&1 || exit 1
  python3 tools/pmc_summary.py gpurun_out/${tag}_pmc_summary.json fetch=gpurun_out/pmc_${tag}_fetch write=gpurun_out/pmc_${tag}_write mfma=gpurun_out/pmc_${tag}_mfma lds=gpurun_out/pmc_${tag}_lds
fi
