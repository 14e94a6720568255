#!/bin/bash
set -e
timeout -k 10 600 python -m pytest tests/test_gpu_sgpr.py tests/test_gpu_gpras.py tests/test_gpu_pipeline.py tests/test_gpu_distance_form.py -x -q 2>&1 | tail -15
timeout -k 10 300 python tools/sgpr_batch_probe.py
