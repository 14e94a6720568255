#!/bin/bash
tools/r2_run16.sh > gpurun_out/r2_run16.log 2>&1; grep -E "trace_pair|dz_kernel" gpurun_out/r2_run16.log
timeout -k 10 200 python tools/sgpr_prof.py 16 20
timeout -k 10 200 python tools/f2_prof.py 128
