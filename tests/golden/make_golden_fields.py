"""Generate tests/golden/fields_golden.npz: inputs and expected outputs of the EOF projection (N1) and the field metrics
(N3), computed by the CPU oracle (oracle/pca.py, oracle/metrics.py) -- they pin the restatement and the HIP path against
regressions, not against the reference (which has no fixtures for these functions).

    python tests/golden/make_golden_fields.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gpras_amd.synth import make_eof_state  # noqa: E402
from oracle import metrics as om  # noqa: E402
from oracle import pca as opca  # noqa: E402

st = make_eof_state(257, 5, 12, seed=42)
out = {"dry": st["dry"], "elevations": st["elevations"], "weights": st["weights"], "eofs": st["eofs"], "x_mean": st["x_mean"], "x_std": st["x_std"],
       "x": st["x"]}
for mode in ("wse", "depth"):
    field = opca.wse_2_depth(st["x"].copy(), st["elevations"]) if mode == "depth" else st["x"]
    mu = field[:, ~st["dry"]].mean(axis=0)
    args = (st["dry"], st["elevations"], mu, st["weights"], st["eofs"], st["x_mean"], st["x_std"], mode)
    z = opca.transform(st["x"], *args)
    full, vfull = opca.reverse_transform(z, 0.01 + 0.1 * np.abs(z), *args)
    out[f"{mode}_input_mean"], out[f"{mode}_z"], out[f"{mode}_full"], out[f"{mode}_vfull"] = mu, z, full, vfull
x = out["wse_full"]
y = x + 0.05 * np.random.default_rng(7).standard_normal(x.shape)
out["met_y"] = y
out["met_scalars"] = np.array([om.rmse_aoi_toi(x, y), om.mae_aoi_toi(x, y), om.err_aoi_toi(x, y), om.rmse_aoi_mts(x, y), om.nse_aoi_mts(x, y),
                               om.err_aoi_mts(x, y), om.fi_aoi_toi(x, y, 2, 0.04), om.pod_mts(x, y, 101.0), om.rfa_mts(x, y, 101.0),
                               om.csi_mts(x, y, 101.0), om.f2_mts(x, y, 101.0), om.f3_mts(x, y, 101.0)])
out["met_rmse_ts"], out["met_rmse_cell"], out["met_err_cell_mts"] = om.rmse_aoi_ts(x, y), om.rmse_cell_toi(x, y), om.err_cell_mts(x, y)
out["met_x_mts"], out["met_y_mts"] = np.argmax(x, axis=0), np.argmax(y, axis=0)
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "fields_golden.npz"), **out)
print("written", len(out), "arrays")
