#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/kmat_trace
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kmat_trace -o s -- python3 tools/batch_prof.py 1024 512 5 > gpurun_out/kmat_trace.log 2>&1
f=$(find gpurun_out/kmat_trace -name "*kernel_stats.csv" | head -1)
cp $f gpurun_out/n1024_stats.csv
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/n1024_stats.csv')))
tot=sum(int(r['TotalDurationNs']) for r in rows)
print('total ms per step', tot/1e6/7)
for r in rows[:10]:
    print(r['Name'][:70], r['Calls'], round(int(r['TotalDurationNs'])/1e3/7,1), 'us/step', round(float(r['AverageNs'])/1e3,1), r['Percentage'])
PY
t=$(find gpurun_out/kmat_trace -name "*kernel_trace.csv" | head -1)
python3 - "$t" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = max(i for i, r in enumerate(rows) if "kmat_kernel" in r["Kernel_Name"])
for r in rows[idx:]:
    name = r["Kernel_Name"].replace("void gprx::", "").replace("gprx::", "").split("(")[0]
    if "gemm" in name:
        print(f"{name[:40]:40s} grid {int(r['Grid_Size_X'])//256:6d} dur {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:8.1f} us")
PY
rm -rf gpurun_out/kmat_trace
