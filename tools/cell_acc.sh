#!/bin/bash
# Where does a workgroup of the column-pair cell kernel spend its time?  Development build (-DGPRX_CELL_ACC) beside the product library.
cd $GRAFT_REPO_ROOT
mkdir -p tools/_lib
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value -DGPRX_CELL_ACC -o tools/_lib/libgprx_cellacc.so gpras_amd/csrc/gprx.hip || exit 1
for cfg in "1024 512" "512 512"; do python3 tools/cell_acc.py $cfg tools/_lib/libgprx_cellacc.so; done
