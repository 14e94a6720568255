#!/bin/bash
# Round 4: A/B of the split panel's rows kernel -- substitution (potrf_rows_kernel) against one MFMA tile product with L11^-1
# (potrf_rows_inv_kernel, "rows_inv" = 1) -- on a lone large matrix and on the batched headline step.  On the GPU box: bash tools/rows_inv_ab.sh
cd $GRAFT_REPO_ROOT
for n in "16384 12" "8192 8" "4096 8"; do
  echo "== lone N=$n"
  echo -n "default            : "; python3 tools/large_probe.py $n | tail -1
  echo -n "split_panel        : "; GPRX_SPLIT_PANEL=1 python3 tools/large_probe.py $n | tail -1
  echo -n "rows_inv rt=1      : "; GPRX_ROWS_INV=1 GPRX_ROWS_INV_LONE=1 python3 tools/large_probe.py $n | tail -1
  echo -n "rows_inv rt=2      : "; GPRX_ROWS_INV=1 GPRX_ROWS_INV_LONE=1 GPRX_ROWS_INV_RT=2 python3 tools/large_probe.py $n | tail -1
done
B="python3 bench.py --steps 10 --warmup 2 --no-extras --batched-only"
for v in 0 1 0 1; do
  echo -n "batched rows_inv=$v rt=1: "; GPRX_ROWS_INV=$v $B 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['panel_kernel'])"
done
echo -n "batched rows_inv=1 rt=2: "; GPRX_ROWS_INV=1 GPRX_ROWS_INV_RT=2 $B 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['panel_kernel'])"
