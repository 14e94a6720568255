#!/bin/bash
for cfg in "0 0" "3 1" "3 0" "0 1" "2 1"; do
  set -- $cfg
  echo "== TAIL_SLOTS=$1 SPLIT_PANEL=$2"
  GPRX_TAIL_SLOTS=$1 GPRX_SPLIT_PANEL=$2 timeout -k 10 200 python tools/large_probe.py 16384 12 | tail -1
  GPRX_TAIL_SLOTS=$1 GPRX_SPLIT_PANEL=$2 timeout -k 10 200 python tools/large_probe.py 4096 8 | tail -1
done
