#!/bin/bash
# one-workgroup-per-cell kernel: row tiles per fused pass (GPRX_CELL_NI)
cd $GRAFT_REPO_ROOT
cp gpras_amd/libgprx.so /tmp/libgprx_keep.so
for v in "4 0" "3 0" "2 0" "6 0"; do
  set -- $v
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value -DGPRX_CELL_NI=$1 -o gpras_amd/libgprx.so gpras_amd/csrc/gprx.hip 2>/dev/null || exit 1
  for a in "1024 512" "512 512"; do GPRX_CELL_KERNEL=1 timeout -k 10 100 python3 tools/batch_n1024.py $a | sed "s/^/NI=$1 deep=$2 /"; done
done
GPRX_CELL_KERNEL=1 timeout -k 10 200 python -m pytest tests/test_gpu_cells.py -x -q -m gpu 2>&1 | tail -2
cp /tmp/libgprx_keep.so gpras_amd/libgprx.so
