cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for lib in tools/libgprx_band2.so tools/libgprx_band3.so tools/libgprx_band6.so; do
  tag=$(basename $lib .so)
  timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/band/$tag -o f -- python3 tools/band_probe.py $lib > gpurun_out/band_$tag.log 2>&1 || exit 1
  tail -1 gpurun_out/band_$tag.log
done
