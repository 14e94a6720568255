#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/f2_trace
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/f2_trace -o s -- python3 tools/f2_prof.py 128 > gpurun_out/f2_trace.log 2>&1
tail -1 gpurun_out/f2_trace.log
f=$(find gpurun_out/f2_trace -name "*kernel_stats.csv" | head -1)
cp $f gpurun_out/f2_stats.csv
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/f2_stats.csv')))
tot=sum(int(r['TotalDurationNs']) for r in rows)
print('total ms per batch', tot/1e6/3)
for r in rows[:14]:
    print(r['Name'][:84], r['Calls'], round(int(r['TotalDurationNs'])/1e6/3,2), 'ms/batch', round(float(r['AverageNs'])/1e3,1), r['Percentage'])
PY
t=$(find gpurun_out/f2_trace -name "*kernel_trace.csv" | head -1)
python3 - "$t" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = max(i for i, r in enumerate(rows) if "kmat_kernel" in r["Kernel_Name"])
seen_trsv = False
for r in rows[idx:]:
    name = r["Kernel_Name"].replace("void gprx::", "").replace("gprx::", "").split("(")[0]
    if "trsv" in name: seen_trsv = True
    if seen_trsv and "trsv" not in name:
        print(f"{name[:46]:46s} grid {int(r['Grid_Size_X'])//256:6d} x{r['Grid_Size_Y']:>4s} x{r['Grid_Size_Z']:>3s} dur {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:9.1f} us")
PY
rm -rf gpurun_out/f2_trace
