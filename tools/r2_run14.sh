#!/bin/bash
# rows-kernel variants: 0 = 128 rows x 3 WG/CU (default), 1 = 256 rows x 2 WG/CU, 2 = 128 rows x 2 WG/CU
set -e
for v in 0 1 2 0 1; do
  echo "== GPRX_ROWS_VARIANT=$v"
  GPRX_ROWS_VARIANT=$v timeout -k 10 120 python tools/batch_prof.py 4096 128 10
  GPRX_ROWS_VARIANT=$v timeout -k 10 120 python tools/batch_prof.py 1024 512 10
done
