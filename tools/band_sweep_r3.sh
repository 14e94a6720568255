#!/bin/bash
# tile-band height of the C_LOWER decode (gemm_f64.h GPRX_BAND) under the cell -> XCD mapping: fits/s and FETCH/WRITE traffic of the main kernel
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
#  the library is built BEFORE any profiler line; under rocprofv3 a stale library is an error, not a fork + exec of hipcc
python3 -m gpras_amd._build --stale > /dev/null || exit 1
export GPRX_NO_BUILD=1
cp gpras_amd/libgprx.so /tmp/libgprx_keep.so
for b in 2 4 8 16; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value -DGPRX_BAND=$b -o gpras_amd/libgprx.so gpras_amd/csrc/gprx.hip || exit 1
  v=$(timeout -k 10 150 python3 bench.py --steps 10 --warmup 2 --no-extras --batched-only 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['roofline']['frac'],4))")
  rm -rf gpurun_out/band_f gpurun_out/band_w
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/band_f -o f -- python3 bench.py --steps 3 --warmup 1 --no-extras --batched-only > /dev/null 2>&1
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/band_w -o w -- python3 bench.py --steps 3 --warmup 1 --no-extras --batched-only > /dev/null 2>&1
  t=$(python3 tools/pmc_summary.py /tmp/band.json fetch=gpurun_out/band_f write=gpurun_out/band_w | grep "gemm_f64_kernel<0,1,64,64,0,0,1>" | sed "s/.*corrected.: .\([0-9.e+]*\).*/\1/")
  echo "band=$b fits/s,frac=$v traffic_per_launch=$t"
done
cp /tmp/libgprx_keep.so gpras_amd/libgprx.so
