#!/bin/bash
set -e
for w in 1024 512 2048 1024; do
  echo "== outer_block=$w"
  timeout -k 10 120 python tools/batch_prof.py 4096 128 10 outer_block=$w
done
for w in 1024 512 256; do
  echo "== N=1024 outer_block=$w"
  timeout -k 10 120 python tools/batch_prof.py 1024 512 10 outer_block=$w
done
for w in 1024 2048 512; do
  echo "== N=2048 outer_block=$w"
  timeout -k 10 120 python tools/batch_prof.py 2048 256 10 outer_block=$w
done
