for ob in 512 1024 2048; do echo "== ob=$ob"; GPRX_OUTER_BLOCK=$ob timeout -k 10 200 python tools/perf_probe.py 2>&1 | grep "factorize wall" || exit 1; done
