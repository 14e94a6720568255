cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -m pytest tests/test_gpu_comm.py tests/test_gpu_sgpr.py tests/test_gpu_gpras.py tests/test_metrics_ref.py -q -m gpu > gpurun_out/r2_run3_tests.log 2>&1; echo rc=$?; tail -25 gpurun_out/r2_run3_tests.log | cut -c1-300
python - <<'PY'
import time, numpy as np
from gpras_amd.gpr import GPRAS
from gpras_amd.synth import make_regression
# sparse predict: batched against the per-mode loop (reference-shaped), 16 and 50 modes, M = 50, N = 4096, N* = 2000
for k in (10, 16, 50):
    x, y, xs = make_regression(4096, 10, n_outputs=k, n_test=2000, config=6, unit=k)
    g = GPRAS("RBF"); g._init_models(x.astype(float), y.astype(float), 50, "kmeans")
    g.predict(xs)
    t=time.perf_counter(); a=g.predict(xs); tb=time.perf_counter()-t
    t=time.perf_counter(); loop=[m.predict_y(xs) for m in g.models]; tl=time.perf_counter()-t
    print(f"sparse predict {k} modes x 2000 points: batched {tb*1e3:.2f} ms, per-mode loop {tl*1e3:.2f} ms")
    t=time.perf_counter(); g.fit(x, y, 50, "kmeans", "two-stage"); tf=time.perf_counter()-t
    print(f"default fit (kmeans Z, two-stage Adam 100+100) of {k} modes: {tf:.3f} s = {k/tf:.1f} units/s")
PY
