#!/bin/bash
set -e
for v in 0 1 0 1; do
  if [ $v = 1 ]; then export GPRX_ROWS_WAVE=1; else unset GPRX_ROWS_WAVE; fi
  echo "== GPRX_ROWS_WAVE=$v"
  timeout -k 10 120 python tools/batch_prof.py 4096 128 10
  timeout -k 10 120 python tools/batch_prof.py 1024 512 10
done
GPRX_ROWS_WAVE=1 timeout -k 10 300 python -m pytest tests/test_gpu_blocks.py tests/test_gpu_exact.py -x -q 2>&1 | tail -3
