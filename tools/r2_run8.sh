cd $GRAFT_REPO_ROOT
for n in 16384 8192; do
echo "--- N=$n old schedule"; GPRX_LARGE_MIN=-1 timeout -k 10 120 python tools/large_probe.py $n 12
for r in 16 0; do for ob in 1024 512; do echo "--- N=$n block-column schedule, reserved CUs $r, outer block $ob"; GPRX_LARGE_RESERVED_CUS=$r timeout -k 10 120 python tools/large_probe.py $n 12 $ob; done; done
done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_large -o l -- python3 tools/large_probe.py 16384 12 > gpurun_out/prof_large.log 2>&1; echo rc=$?
python - <<'PY'
import csv, glob
f=glob.glob('gpurun_out/prof_large/**/*kernel_stats.csv', recursive=True)
for row in list(csv.DictReader(open(f[0])))[:14]:
    print(row['Name'][:90], row['Calls'], row['TotalDurationNs'], row['AverageNs'][:9], row['Percentage'])
PY
