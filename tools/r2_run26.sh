#!/bin/bash
for ob in 1024 2048 512 1536; do echo "outer_block=$ob"; timeout -k 10 200 python tools/large_probe.py 16384 12 $ob | tail -1; done
for ob in 1024 2048; do echo "N=8192 outer_block=$ob"; timeout -k 10 200 python tools/large_probe.py 8192 8 $ob | tail -1; done
