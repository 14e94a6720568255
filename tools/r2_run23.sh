#!/bin/bash
set -e
timeout -k 10 600 python -m pytest tests/test_gpu_blocks.py tests/test_gpu_exact.py tests/test_gpu_edges.py -x -q 2>&1 | tail -4
for v in 1 0 1 0; do
  echo "== GPRX_FUSE_K64=$v"
  GPRX_FUSE_K64=$v timeout -k 10 120 python tools/large_probe.py 4096 8 | tail -1
  GPRX_FUSE_K64=$v timeout -k 10 120 python tools/large_probe.py 16384 12 | tail -1
  GPRX_FUSE_K64=$v timeout -k 10 120 python tools/large_probe.py 1024 8 | tail -1
done
