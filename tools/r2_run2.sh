cd $GRAFT_REPO_ROOT
echo "--- kmeans + modelfile-related gpu tests"
timeout -k 10 400 python -m pytest tests/test_kmeans.py tests/test_gpu_gpras.py tests/test_gpu_sgpr.py -q -m gpu > gpurun_out/r2_run2_tests.log 2>&1; echo rc=$?; tail -15 gpurun_out/r2_run2_tests.log
echo "--- comm pytest (debug)"
NCCL_DEBUG=WARN timeout -k 10 200 python -m pytest tests/test_gpu_comm.py -q -s > gpurun_out/r2_run2_comm.log 2>&1; echo rc=$?; grep -v "alt_rsmi\|iommu" gpurun_out/r2_run2_comm.log | tail -30 | cut -c1-400
echo "--- comm pytest (no debug env)"
timeout -k 10 200 python -m pytest tests/test_gpu_comm.py -q > gpurun_out/r2_run2_comm2.log 2>&1; echo rc=$?; tail -5 gpurun_out/r2_run2_comm2.log | cut -c1-400
echo "--- kmeans timing"
python - <<'PY'
import time, numpy as np
from gpras_amd.kmeans import kmeans_centers
from gpras_amd.synth import make_regression
from sklearn.cluster import KMeans
for n,d,m in ((4096,10,50),(4096,10,300),(16384,10,50),(16384,10,300)):
    x,_,_=make_regression(n,d,1,0,config=6,unit=0)
    kmeans_centers(x,m)
    t=time.perf_counter(); c,info=kmeans_centers(x,m,return_info=True); td=time.perf_counter()-t
    t=time.perf_counter(); km=KMeans(n_clusters=m, random_state=0, n_init="auto").fit(x); ts=time.perf_counter()-t
    print(n,d,m,"device path %.1f ms (device=%s, %d iters)  sklearn %.1f ms  max|dc| %.2e" % (td*1e3, info["device"], info["n_iter"], ts*1e3, np.max(np.abs(c-km.cluster_centers_))))
PY
