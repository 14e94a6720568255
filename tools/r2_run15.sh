#!/bin/bash
# sparse batch evaluation: hipGraph replay vs eager launches
set -e
echo "== graph replay"; timeout -k 10 300 python tools/sgpr_batch_probe.py
echo "== eager (GPRX_NO_GRAPH=1)"; GPRX_NO_GRAPH=1 timeout -k 10 300 python tools/sgpr_batch_probe.py
timeout -k 10 600 python -m pytest tests/test_gpu_sgpr.py tests/test_gpu_gpras.py tests/test_gpu_pipeline.py -x -q 2>&1 | tail -5
