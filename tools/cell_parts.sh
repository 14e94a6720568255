#!/bin/bash
# where does the column-pair cell kernel's time go: builds without the operand DMA / without the MFMAs of the rows phase / without the
# rows phase / at one workgroup per CU (tools/_lib, built by hand with -DGPRX_CELL2_*), N = 1024 x 512 cells
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export GPRX_CELL_KERNEL=1 GPRX_IGNORE_STATUS=1
for v in "" NODMA NOMMA NOROWS OCC1; do
  lib=${v:+$GRAFT_REPO_ROOT/tools/_lib/libgprx_$v.so}
  rm -rf gpurun_out/cp_$v
  GPRX_LIBRARY=$lib timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/cp_$v -o s -- python3 tools/batch_n1024.py ${1:-1024} ${2:-512} > gpurun_out/cp_$v.log 2>&1
  echo "${v:-BASE}: $(grep potrf_cell $(find gpurun_out/cp_$v -name '*kernel_stats.csv' | head -1) | cut -d, -f2-4)"
done
