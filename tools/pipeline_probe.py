"""Tail of the analysis pipeline (predict -> reverse projection -> depths -> metric tables): host chain against DevicePipeline
(development aid).  argv: cells [t_star]"""
import sys, time, tempfile, os
import numpy as np, pandas as pd
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from test_gpu_pipeline import _setup, _host_chain
from gpras_amd.metrics import export_metric_summary
from gpras_amd.pipeline import DevicePipeline
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
t_star = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
rng = np.random.default_rng(5)
gpr, proj, x_test, truth_df, elev = _setup("wse", 50, rng, n=4096, d=10, k=10, cells=cells, t_star=t_star)
pipe = DevicePipeline(gpr, proj)
tmp = tempfile.mkdtemp()
for rep in range(2):
    t0 = time.perf_counter()
    d, p, c = _host_chain(gpr, proj, x_test, truth_df.values.copy(), elev, "wse")
    t1 = time.perf_counter()
    frame = lambda a: pd.DataFrame(a, index=truth_df.index, columns=truth_df.columns)
    export_metric_summary(frame(d), frame(p), frame(c), os.path.join(tmp, "h.db"))
    t2 = time.perf_counter()
    f = pipe.predict_fields(x_test); proj._lib.gprx_pca_synchronize(proj.handle)
    t3 = time.perf_counter()
    f.close()
    f = pipe.export_metric_summary(x_test, truth_df, os.path.join(tmp, "d.db")); f.close()
    t4 = time.perf_counter()
    print(f"cells={cells} T*={t_star}: host chain fields {1e3*(t1-t0):.1f} ms + metrics export {1e3*(t2-t1):.1f} ms; "
          f"device fields {1e3*(t3-t2):.1f} ms; device predict+fields+metrics export {1e3*(t4-t3):.1f} ms", flush=True)
