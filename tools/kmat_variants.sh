#!/bin/bash
# where does kmat_kernel's time go: rebuild the library with the exp or the stores of the interior tiles removed and read the
# kernel's average duration from rocprofv3 (the factorisation then fails or not -- irrelevant: only the kmat launches are read)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
#  the library is built BEFORE any profiler line; under rocprofv3 a stale library is an error, not a fork + exec of hipcc
python3 -m gpras_amd._build --stale > /dev/null || exit 1
export GPRX_NO_BUILD=1
cp gpras_amd/libgprx.so /tmp/libgprx_keep.so
for v in NONE GPRX_KMAT_NOEXP GPRX_KMAT_NOSTORE; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value -D$v -o gpras_amd/libgprx.so gpras_amd/csrc/gprx.hip || exit 1
  touch gpras_amd/libgprx.so
  rm -rf gpurun_out/kv_$v
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kv_$v -o b -- python3 bench.py --steps 3 --warmup 1 --no-extras --batched-only > /dev/null 2> gpurun_out/kv_$v.log
  echo $v $(grep kmat_kernel $(find gpurun_out/kv_$v -name "*kernel_stats.csv" | head -1) | cut -d, -f1-4 | tail -1)
done
cp /tmp/libgprx_keep.so gpras_amd/libgprx.so
