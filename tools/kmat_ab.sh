#!/bin/bash
# A/B of the kernel-build variants built by hand into tools/_kb (see tools/kmat_bench.hip); on the GPU box: bash tools/kmat_ab.sh
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for b in tools/_kb/kb_*; do
  echo -n "$(basename $b): "; timeout -k 5 120 $b 128 4096 8 || exit 1
done
done
