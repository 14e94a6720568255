run() { echo "== $*"; env "$@" timeout -k 10 200 python tools/perf_probe.py 2>&1 | grep -E "N=  8192 fact|N= 16384 fact" || exit 1; }
run GPRX_UPDATE_TILE=64
run GPRX_UPDATE_TILE=64 GPRX_OUTER_BLOCK=1024
run GPRX_NO_LOOKAHEAD_X=1 GPRX_OUTER_BLOCK=256
run GPRX_SPLIT_PANEL=1
