cd $GRAFT_REPO_ROOT
for n in 16384 8192; do
echo "--- N=$n old schedule, TAIL tile 64"; GPRX_UPDATE_TILE=64 GPRX_LARGE_MIN=-1 timeout -k 10 120 python tools/large_probe.py $n 12
echo "--- N=$n old schedule, TAIL tile 64, outer 1024"; GPRX_UPDATE_TILE=64 GPRX_LARGE_MIN=-1 timeout -k 10 120 python tools/large_probe.py $n 12 1024
echo "--- N=$n block-column, TAIL tile 64, ob 1024"; GPRX_UPDATE_TILE=64 GPRX_LARGE_RESERVED_CUS=0 timeout -k 10 120 python tools/large_probe.py $n 12 1024
done
