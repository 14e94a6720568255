"""CPU restatement (test infrastructure) of the EOF projection either side of the GP path -- SURVEY.md section 8(f)
row N1: ``PreProcessor.transform`` (/root/reference/gpras/preprocess.py:1009-1038), ``wse_2_depth`` (:1040-1044),
``reverse_transform`` (:1052-1085) and ``_linear_transform_for_var`` (:1087-1094).

PINNED by outputs of the reference itself (round 3): ``tests/golden/make_golden_pca_ref.py`` imports
``/root/reference/gpras/preprocess.py`` in the build container (inert recording modules stand in for the absent third-party
imports of that file, and the script asserts none of them was touched while the recorded calls ran), builds ``PreProcessor``
objects through the reference's constructor and records ``transform`` / ``reverse_transform`` / ``_linear_transform_for_var``
on 12 seeded states (wse / depth / velocity x weighted / unweighted, dry cells) into ``tests/golden/pca_ref_golden.npz``;
``tests/test_pca.py`` holds this file to those outputs BIT FOR BIT.  These functions are plain numpy in the reference, and
this file follows them operation by operation (same order of subtract / weight / dot / divide).

State = what a fitted reference PreProcessor holds (preprocess.py:868-927): ``dry`` (bool per cell), ``elevations``
(per cell), and over the wet cells ``input_mean``, ``weights`` (or None), ``eofs`` (k, n_wet); ``x_mean``, ``x_std`` (k);
``hydraulic_parameter`` in {"wse", "depth", "velocity"}.
"""

from __future__ import annotations

import numpy as np


def wse_2_depth(x, elevations):
    """preprocess.py:1040-1044: depths, negative values set to zero."""
    d = x - elevations
    d[d < 0] = 0
    return d


def transform(x, dry, elevations, input_mean, weights, eofs, x_mean, x_std, hydraulic_parameter="wse"):
    """preprocess.py:1009-1038."""
    x = np.asarray(x, dtype=np.float64)
    if hydraulic_parameter == "depth":
        x = wse_2_depth(x, elevations)
    x = x[:, ~dry].copy()
    x = x - input_mean
    if weights is not None:
        x *= weights
    x = np.dot(x, eofs.T)
    return (x - x_mean) / x_std


def linear_transform_for_var(weights, eofs, x_std):
    """preprocess.py:1087-1094: squared linear map used for the variance."""
    a = np.diag(x_std)
    a = a.dot(eofs)
    if weights is not None:
        a = a / weights.reshape(1, -1)
    return a**2


def reverse_transform(mean, var, dry, elevations, input_mean, weights, eofs, x_mean, x_std, hydraulic_parameter="wse"):
    """preprocess.py:1052-1085.  Returns ``full`` or ``(full, var_full)``."""
    mean = (np.asarray(mean, dtype=np.float64) * x_std) + x_mean
    mean = np.dot(mean, eofs)
    if weights is not None:
        mean = mean / weights
    mean = mean + input_mean
    full = np.empty((mean.shape[0], dry.shape[0]))
    if hydraulic_parameter == "depth":
        full[:, dry] = 0
    else:
        full[:, dry] = elevations[dry]
    full[:, ~dry] = mean
    if var is None:
        return full
    var_prop = np.asarray(var, dtype=np.float64).dot(linear_transform_for_var(weights, eofs, x_std))
    var_full = np.empty((var_prop.shape[0], dry.shape[0]))
    var_full[:, dry] = 0
    var_full[:, ~dry] = var_prop
    return full, var_full
