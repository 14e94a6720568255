#!/bin/bash
# per-launch durations of the in-block kernels in one batched step (128 cells, N = 4096)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/step_trace
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/step_trace -o s -- python3 tools/batch_prof.py 4096 128 2 > gpurun_out/step_trace.log 2>&1
tail -1 gpurun_out/step_trace.log
t=$(find gpurun_out/step_trace -name "*kernel_trace.csv" | head -1)
python3 - "$t" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last step: find the last kmat launch
idx = max(i for i, r in enumerate(rows) if "kmat_kernel" in r["Kernel_Name"])
step = rows[idx:]
t0 = int(step[0]["Start_Timestamp"])
print("launches in the step:", len(step), "span ms:", (int(step[-1]["End_Timestamp"]) - t0) / 1e6)
agg = collections.OrderedDict()
for r in step:
    name = r["Kernel_Name"].replace("void gprx::", "").replace("gprx::", "").split("(")[0]
    key = (name[:44], r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    agg.setdefault(key, []).append(d)
for k, v in agg.items():
    print(f"{k[0]:44s} grid {k[1]:>8s} {k[2]:>5s} {k[3]:>4s}  n={len(v):3d}  avg {sum(v)/len(v):8.1f} us  total {sum(v)/1e3:7.3f} ms")
PY
rm -rf gpurun_out/step_trace
