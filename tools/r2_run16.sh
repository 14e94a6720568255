#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/sgpr_prof
GPRX_NO_GRAPH=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sgpr_prof -o s -- python3 tools/sgpr_prof.py 16 20 > gpurun_out/sgpr_prof.log 2>&1
cat gpurun_out/sgpr_prof.log | tail -3
f=$(find gpurun_out/sgpr_prof -name "*kernel_stats.csv" | head -1)
cp $f gpurun_out/sgpr_prof_kernel_stats.csv
t=$(find gpurun_out/sgpr_prof -name "*kernel_trace.csv" | head -1)
python3 - "$t" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last evaluation: take the final 60 kernels, print name, duration, gap to previous end
tail = rows[-20:]
prev = None
for r in tail:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev) / 1e3 if prev else 0.0
    print(f"{r['Kernel_Name'][:70]:70s} dur {(e-s)/1e3:7.2f} us gap {gap:6.2f} us")
    prev = e
PY
rm -rf gpurun_out/sgpr_prof
