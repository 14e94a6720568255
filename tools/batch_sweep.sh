export GPRX_OUTER_BLOCK=512 GPRX_UPDATE_TILE=64
echo "== base"; timeout -k 10 100 python tools/batch_probe.py 4096 8 1,16,32 || exit 1
echo "== panel_rows=256"; GPRX_PANEL_ROWS=256 timeout -k 10 100 python tools/batch_probe.py 4096 8 1,16,32 || exit 1
echo "== panel_width=128"; GPRX_PANEL_WIDTH=128 timeout -k 10 100 python tools/batch_probe.py 4096 8 1,16,32 || exit 1
