#!/bin/bash
# N = 16384 (one matrix) and N = 4096 with and without s_setprio(3) in the panel kernel
cd $GRAFT_REPO_ROOT
cp gpras_amd/libgprx.so /tmp/libgprx_keep.so
for v in GPRX_PANEL_NO_SETPRIO NONE; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value -D$v -o gpras_amd/libgprx.so gpras_amd/csrc/gprx.hip || exit 1
  echo variant $v
  for a in "16384 12" "8192 8" "4096 8"; do timeout -k 10 200 python3 tools/large_probe.py $a 2>&1 | tail -1; done
done
cp /tmp/libgprx_keep.so gpras_amd/libgprx.so
