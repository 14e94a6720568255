#!/bin/bash
# Which phase of a panel workgroup grows beside the bulk TAIL update?  Development build (-DGPRX_PANEL_ACC) swapped in for the
# run: phase durations of rows-workgroup 0 summed over every panel launch of one N = 16384 factorisation, with the look-ahead
# (TAIL on its own stream, concurrent) and without (no_lookahead: everything in stream order).
cd $GRAFT_REPO_ROOT
cp gpras_amd/libgprx.so /tmp/libgprx_keep.so
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value -DGPRX_PANEL_ACC -o gpras_amd/libgprx.so gpras_amd/csrc/gprx.hip || exit 1
for la in 0 1; do timeout -k 10 200 python3 tools/panel_acc.py 16384 12 $la || break; done
# (the CU-mask variants recorded in profiles/r03_panel_phases_beside_tail.txt need tools/patches/r03_chain_cus_cu_mask_streams.patch)
cp /tmp/libgprx_keep.so gpras_amd/libgprx.so
