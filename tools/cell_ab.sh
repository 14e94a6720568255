#!/bin/bash
# A/B of the one-workgroup-per-cell kernels: round 3's single-column kernel (GPRX_CELL_SINGLE_COLUMN=1) against the column-pair kernel
cd $GRAFT_REPO_ROOT
export GPRX_CELL_KERNEL=1
for cfg in "1024 512" "1000 256" "512 512" "256 512" "192 512" "640 512"; do
  for v in 1 0; do
    echo -n "single_column=$v: "; GPRX_CELL_SINGLE_COLUMN=$v timeout -k 5 120 python3 tools/batch_n1024.py $cfg || exit 1
  done
done
