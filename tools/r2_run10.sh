cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
stats() { python - "$1" <<'PY'
import csv, glob, sys
f=glob.glob(sys.argv[1]+'/**/*kernel_stats.csv', recursive=True)
for row in list(csv.DictReader(open(f[0])))[:12]:
    print('   ', row['Name'][:86], row['Calls'], '%.1f us avg' % (float(row['AverageNs'])/1e3), row['Percentage']+'%')
PY
}
echo "--- N=1024 x 512 cells"; timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_n1024 -o p -- python3 tools/batch_prof.py 1024 512 5 2>&1 | grep "fits/s"; stats gpurun_out/prof_n1024
echo "--- N=4096 single cell"; timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_n4096s -o p -- python3 tools/large_probe.py 4096 8 2>&1 | grep "fit "; stats gpurun_out/prof_n4096s
echo "--- N=16384 single"; timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_n16384 -o p -- python3 tools/large_probe.py 16384 12 2>&1 | grep "fit "; stats gpurun_out/prof_n16384
