#!/bin/bash
# Round 5: s_memtime phase tables of the 64 x 64 diagonal-block chain -- the panel kernel's sub-panel steps
# (potrf.h panel_step: the diagonal block alone, and the first panel of an N = 4096 matrix) and the chain workgroup of chain64.h
# (chain_step: tile-DAG, cell kernel and the fused sparse evaluation).  A development build with both stamp sets; the library is restored.
#   bash tools/phase_tables_r5.sh > gpurun_out/r05_diag_block_phases.txt
cd $GRAFT_REPO_ROOT
cp gpras_amd/libgprx.so /tmp/libgprx_keep.so
export GPRX_EXTRA_FLAGS="-DGPRX_PANEL_STAMPS -DGPRX_CHAIN_STAMPS"
python3 -m gpras_amd._build --stale > /dev/null || exit 1
export GPRX_STAMPS_LIB=gpras_amd/libgprx.so
echo "== potrf_panel_kernel<2,2>, the diagonal block alone (no rows below): cycles (s_memtime) =="
timeout -k 10 120 python3 tools/panel_stamps.py 0 || exit 1
echo "== potrf_panel_kernel<2,2>, 4032 rows below (first panel of N = 4096) =="
timeout -k 10 120 python3 tools/panel_stamps.py 4032 || exit 1
echo "== chain_step<3> of chain64.h =="
sed -n '/^python3 - <<.PY.$/,/^PY$/p' tools/chain_stamps.sh | sed '1d;$d' > /tmp/chain_stamps_body.py
timeout -k 10 120 python3 /tmp/chain_stamps_body.py || exit 1
cp /tmp/libgprx_keep.so gpras_amd/libgprx.so
