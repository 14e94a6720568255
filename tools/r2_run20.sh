#!/bin/bash
# single-cell N = 4096 fit: kernel timeline of one graph replay (durations and gaps)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/single_trace
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/single_trace -o s -- python3 tools/large_probe.py 4096 8 > gpurun_out/single_trace.log 2>&1
tail -1 gpurun_out/single_trace.log
t=$(find gpurun_out/single_trace -name "*kernel_trace.csv" | head -1)
python3 - "$t" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = max(i for i, r in enumerate(rows) if "kmat_kernel" in r["Kernel_Name"])
step = rows[idx:]
t0 = int(step[0]["Start_Timestamp"]); t1 = int(step[-1]["End_Timestamp"])
print("launches", len(step), "span us", (t1 - t0) / 1e3)
busy = 0; prev_end = None; gaps = []; by = collections.Counter(); byn = collections.Counter()
ends = []
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("void gprx::", "").replace("gprx::", "").split("(")[0][:40]
    by[name] += (e - s) / 1e3; byn[name] += 1
    if prev_end is not None: gaps.append((s - prev_end) / 1e3)
    prev_end = max(prev_end or 0, e)
print("sum of kernel durations us", sum(by.values()))
pos = [g for g in gaps if g > 0]
print("positive gaps: n", len(pos), "sum us", sum(pos), "mean", sum(pos) / max(len(pos), 1))
neg = [g for g in gaps if g <= 0]
print("overlapping starts:", len(neg))
for k, v in by.most_common(8): print(f"  {k:40s} n={byn[k]:4d} total {v:8.1f} us avg {v/byn[k]:6.1f}")
# first 40 kernels timeline
for r in step[:44]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("void gprx::", "").replace("gprx::", "").split("(")[0][:36]
    print(f"  +{(s-t0)/1e3:8.1f} {name:36s} {(e-s)/1e3:6.1f} us grid {int(r['Grid_Size_X'])//256}")
PY
rm -rf gpurun_out/single_trace
