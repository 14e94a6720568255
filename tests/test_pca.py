"""EOF projection: CPU pins of the oracle restatement, GPU parity of the HIP path through
the C ABI.  Tolerances: the projection sums n_wet products per output, so results are compared at 1e-11 relative to the
largest output (summation order differs between numpy's dot, a hand loop and the split-K MFMA GEMM); the reverse
transform has k <= 64 terms per output and is held to 1e-13."""

import numpy as np
import pytest

from gpras_amd.synth import make_eof_state
from oracle import pca as opca


def explicit_transform(st, mode):
    x = st["x"].astype(float)
    wet = np.flatnonzero(~st["dry"])
    out = np.zeros((x.shape[0], st["eofs"].shape[0]))
    for t in range(x.shape[0]):
        for k in range(out.shape[1]):
            s = 0.0
            for j, c in enumerate(wet):
                v = x[t, c]
                if mode == "depth":
                    v = max(v - st["elevations"][c], 0.0)
                v = v - st["input_mean_mode"][j]
                if st["weights"] is not None:
                    v *= st["weights"][j]
                s += v * st["eofs"][k, j]
            out[t, k] = (s - st["x_mean"][k]) / st["x_std"][k]
    return out


def state(n_cells, k, t, seed, mode, weighted=True):
    st = make_eof_state(n_cells, k, t, seed, weighted=weighted)
    # the stored mean belongs to the field the projector sees (depths in depth mode)
    field = opca.wse_2_depth(st["x"].copy(), st["elevations"]) if mode == "depth" else st["x"]
    st["input_mean_mode"] = field[:, ~st["dry"]].mean(axis=0)
    return st


@pytest.mark.parametrize("mode", ["wse", "depth", "velocity"])
@pytest.mark.parametrize("weighted", [True, False])
def test_oracle_against_explicit_loops_and_round_trip(mode, weighted):
    st = state(37, 4, 5, 11, mode, weighted)
    args = (st["dry"], st["elevations"], st["input_mean_mode"], st["weights"], st["eofs"], st["x_mean"], st["x_std"], mode)
    z = opca.transform(st["x"], *args)
    assert np.allclose(z, explicit_transform(st, mode), rtol=0, atol=1e-12 * np.abs(z).max())
    full, vfull = opca.reverse_transform(z, np.abs(z) * 0.1, *args)
    assert full.shape == st["x"].shape and vfull.shape == st["x"].shape
    assert np.all(vfull[:, st["dry"]] == 0) and np.all(vfull[:, ~st["dry"]] >= 0)
    if mode == "depth":
        assert np.all(full[:, st["dry"]] == 0)
    else:
        assert np.array_equal(full[:, st["dry"]], np.tile(st["elevations"][st["dry"]], (5, 1)))
    # projecting the reconstruction again reproduces the mode amplitudes (EOF rows are orthonormal); not in depth mode,
    # where the clamp at zero is not invertible
    if mode != "depth":
        z2 = opca.transform(full, *args)
        assert np.allclose(z2, z, rtol=0, atol=1e-9 * np.abs(z).max())
    a2 = opca.linear_transform_for_var(st["weights"], st["eofs"], st["x_std"])
    assert np.allclose(vfull[:, ~st["dry"]], (np.abs(z) * 0.1) @ a2)


@pytest.mark.gpu
@pytest.mark.parametrize("mode,weighted,n_cells,k,t", [("wse", True, 5000, 6, 70), ("depth", True, 3001, 20, 33), ("velocity", False, 777, 3, 150), ("wse", False, 64, 1, 2)])
def test_gpu_projection_matches_oracle(mode, weighted, n_cells, k, t):
    from gpras_amd.preprocess import EOFProjector

    st = state(n_cells, k, t, 100 + k, mode, weighted)
    proj = EOFProjector(st["dry"], st["elevations"], st["input_mean_mode"], st["weights"], st["eofs"], st["x_mean"], st["x_std"], mode)
    args = (st["dry"], st["elevations"], st["input_mean_mode"], st["weights"], st["eofs"], st["x_mean"], st["x_std"], mode)
    x_in = st["x"].copy()
    z = proj.transform(x_in)
    assert np.array_equal(x_in, st["x"])  # the caller's array is not modified
    zr = opca.transform(st["x"], *args)
    assert z.shape == (t, k) and np.max(np.abs(z - zr)) <= 1e-11 * np.max(np.abs(zr))
    var = 0.05 + 0.01 * np.arange(t * k).reshape(t, k) / (t * k)
    full, vfull = proj.reverse_transform(zr, var)
    fr, vr = opca.reverse_transform(zr, var, *args)
    assert np.max(np.abs(full - fr)) <= 1e-13 * np.max(np.abs(fr))
    assert np.max(np.abs(vfull - vr)) <= 1e-13 * np.max(np.abs(vr))
    assert np.array_equal(full[:, st["dry"]], fr[:, st["dry"]]) and np.all(vfull[:, st["dry"]] == 0)
    only_mean = proj.reverse_transform(zr)
    assert np.array_equal(only_mean, full)
    proj.close()


@pytest.mark.gpu
def test_gpu_projection_chunked_rows_and_errors(monkeypatch):
    """Rows are staged through the device in bounded passes; results do not depend on the pass size."""
    import subprocess
    import sys

    code = (
        "import numpy as np, sys; sys.path.insert(0, '.');"
        "from gpras_amd.preprocess import EOFProjector; from gpras_amd.synth import make_eof_state;"
        "st = make_eof_state(1000, 5, 300, 3); m = st['x'][:, ~st['dry']].mean(axis=0);"
        "p = EOFProjector(st['dry'], st['elevations'], m, st['weights'], st['eofs'], st['x_mean'], st['x_std'], 'wse');"
        "z = p.transform(st['x']); f, v = p.reverse_transform(z, np.abs(z)); np.save(sys.argv[1], np.concatenate([z.ravel(), f.ravel(), v.ravel()]))"
    )
    import os
    import tempfile

    outs = []
    for chunk in ("65536", "134217728"):  # 64 rows per pass (the minimum) vs one pass
        with tempfile.TemporaryDirectory() as d:
            path = os.path.join(d, "o.npy")
            env = dict(os.environ, GPRX_PCA_CHUNK_DOUBLES=chunk)
            subprocess.run([sys.executable, "-c", code, path], check=True, env=env, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
            outs.append(np.load(path))
    assert np.array_equal(outs[0], outs[1])
    from gpras_amd.preprocess import EOFProjector

    st = make_eof_state(50, 2, 4, 1)
    m = st["x"][:, ~st["dry"]].mean(axis=0)
    with pytest.raises(ValueError):
        EOFProjector(st["dry"], st["elevations"], m[:-1], st["weights"], st["eofs"], st["x_mean"], st["x_std"], "wse")
    with pytest.raises(ValueError):
        EOFProjector(st["dry"], None, m, st["weights"], st["eofs"], st["x_mean"], st["x_std"], "depth")
    p = EOFProjector(st["dry"], st["elevations"], m, st["weights"], st["eofs"], st["x_mean"], st["x_std"], "wse")
    with pytest.raises(ValueError):
        p.transform(st["x"][:, :-1])


def _golden():
    import os

    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fields_golden.npz"))


@pytest.mark.parametrize("mode", ["wse", "depth"])
def test_oracle_reproduces_golden_fields(mode):
    g = _golden()
    args = (g["dry"], g["elevations"], g[f"{mode}_input_mean"], g["weights"], g["eofs"], g["x_mean"], g["x_std"], mode)
    z = opca.transform(g["x"], *args)
    np.testing.assert_allclose(z, g[f"{mode}_z"], rtol=0, atol=1e-13 * np.abs(g[f"{mode}_z"]).max())
    full, vfull = opca.reverse_transform(g[f"{mode}_z"], 0.01 + 0.1 * np.abs(g[f"{mode}_z"]), *args)
    np.testing.assert_allclose(full, g[f"{mode}_full"], rtol=1e-14)
    np.testing.assert_allclose(vfull, g[f"{mode}_vfull"], rtol=1e-13, atol=1e-300)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["wse", "depth"])
def test_gpu_projection_reproduces_golden_fields(mode):
    from gpras_amd.preprocess import EOFProjector

    g = _golden()
    p = EOFProjector(g["dry"], g["elevations"], g[f"{mode}_input_mean"], g["weights"], g["eofs"], g["x_mean"], g["x_std"], mode)
    z = p.transform(g["x"])
    assert np.max(np.abs(z - g[f"{mode}_z"])) <= 1e-11 * np.max(np.abs(g[f"{mode}_z"]))
    full, vfull = p.reverse_transform(g[f"{mode}_z"], 0.01 + 0.1 * np.abs(g[f"{mode}_z"]))
    assert np.max(np.abs(full - g[f"{mode}_full"])) <= 1e-13 * np.max(np.abs(g[f"{mode}_full"]))
    assert np.max(np.abs(vfull - g[f"{mode}_vfull"])) <= 1e-13 * np.max(np.abs(g[f"{mode}_vfull"]))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(16))
def test_gpu_projection_random_shapes(seed):
    """Random shapes around the edges: very few cells, k up to the 64-mode limit, one row, no dry cells, no weights."""
    from gpras_amd.preprocess import EOFProjector

    rng = np.random.default_rng(300 + seed)
    n_cells = int(rng.choice([rng.integers(3, 40), rng.integers(40, 600), rng.integers(600, 5000)]))
    mode = ["wse", "depth", "velocity"][seed % 3]
    weighted = bool(rng.integers(2))
    st = make_eof_state(n_cells, 1, 3, 400 + seed, dry_fraction=float(rng.choice([0.0, 0.2, 0.6])), weighted=weighted)
    n_wet = int((~st["dry"]).sum())
    if n_wet == 0:
        st["dry"][0] = False
        n_wet = 1
    k = int(min(n_wet, rng.choice([1, 2, 16, 17, 33, 64])))
    t = int(rng.choice([1, 2, 31, 64, 65]))
    q, _ = np.linalg.qr(rng.standard_normal((n_wet, k)))
    eofs = np.ascontiguousarray(q.T)
    x = st["elevations"] + rng.standard_normal((t, n_cells))
    field = opca.wse_2_depth(x.copy(), st["elevations"]) if mode == "depth" else x
    mu = field[:, ~st["dry"]].mean(axis=0)
    w = (0.5 + rng.random(n_wet)) if weighted else None
    xm, xs_ = rng.standard_normal(k), 0.5 + rng.random(k)
    args = (st["dry"], st["elevations"], mu, w, eofs, xm, xs_, mode)
    p = EOFProjector(st["dry"], st["elevations"], mu, w, eofs, xm, xs_, mode)
    z = p.transform(x)
    zr = opca.transform(x, *args)
    assert np.max(np.abs(z - zr)) <= 1e-11 * max(np.max(np.abs(zr)), 1e-3), (n_cells, k, t, mode)
    var = rng.random((t, k))
    full, vfull = p.reverse_transform(zr, var)
    fr, vr = opca.reverse_transform(zr, var, *args)
    assert np.max(np.abs(full - fr)) <= 1e-13 * np.max(np.abs(fr)), (n_cells, k, t, mode)
    assert np.max(np.abs(vfull - vr)) <= 1e-13 * max(np.max(np.abs(vr)), 1e-300), (n_cells, k, t, mode)
    p.close()


# ---- row N1 pinned by the REFERENCE's own outputs -----------------------------------------------------
# tests/golden/pca_ref_golden.npz was written by tests/golden/make_golden_pca_ref.py, which imports
# /root/reference/gpras/preprocess.py in the build container and runs PreProcessor.transform / reverse_transform /
# _linear_transform_for_var (:1009-1094) on seeded fitted states; pca_ref_cases() regenerates the inputs here.
import json  # noqa: E402
import os  # noqa: E402
import sys  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden_pca_ref as pca_gen  # noqa: E402  (it reads the reference only in main())

PCA_GOLD = np.load(os.path.join(HERE, "golden", "pca_ref_golden.npz"))
PCA_CASES = pca_gen.pca_ref_cases()


def _oracle_args(c):
    kw = c["kwargs"]
    return (c["dry"], kw["elevations"], kw["input_mean"], kw["weights"], kw["eofs"], kw["x_mean"], kw["x_std"], c["mode"])


def test_reference_fixture_is_what_it_says():
    meta = json.loads(str(PCA_GOLD["meta_json"]))
    assert meta["reference_file"] == "gpras/preprocess.py" and len(PCA_CASES) == 12
    # the unweighted constructor state of the reference cannot transform (np.empty(0) weights, preprocess.py:916, :1030)
    assert set(meta["unweighted_constructor_state_raises"].values()) == {"ValueError"}
    assert {"geopandas", "gpflow", "tensorflow", "hecdss"} <= set(meta["inert_modules"])


@pytest.mark.parametrize("name", sorted(PCA_CASES))
def test_oracle_equals_reference_outputs_bit_for_bit(name):
    c = PCA_CASES[name]
    args = _oracle_args(c)
    assert np.array_equal(opca.transform(c["x"].copy(), *args), PCA_GOLD[f"{name}/transform"])
    assert np.array_equal(opca.reverse_transform(c["mean"].copy(), None, *args), PCA_GOLD[f"{name}/reverse_mean_only"])
    full, vfull = opca.reverse_transform(c["mean"].copy(), c["var"].copy(), *args)
    assert np.array_equal(full, PCA_GOLD[f"{name}/reverse_full"]) and np.array_equal(vfull, PCA_GOLD[f"{name}/reverse_var"])
    kw = c["kwargs"]
    assert np.array_equal(opca.linear_transform_for_var(kw["weights"], kw["eofs"], kw["x_std"]), PCA_GOLD[f"{name}/linear_transform_for_var"])
    if c["mode"] == "depth":
        assert np.array_equal(opca.wse_2_depth(c["x"].copy(), kw["elevations"]), PCA_GOLD[f"{name}/wse_2_depth"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(PCA_CASES))
def test_gpu_projector_against_reference_outputs(name):
    from gpras_amd.preprocess import EOFProjector

    c = PCA_CASES[name]
    proj = EOFProjector(*_oracle_args(c))
    want = PCA_GOLD[f"{name}/transform"]
    z = proj.transform(c["x"].copy())
    assert z.shape == want.shape and np.max(np.abs(z - want)) <= 1e-11 * np.max(np.abs(want))
    want_full, want_var = PCA_GOLD[f"{name}/reverse_full"], PCA_GOLD[f"{name}/reverse_var"]
    only = proj.reverse_transform(c["mean"].copy())
    assert np.max(np.abs(only - PCA_GOLD[f"{name}/reverse_mean_only"])) <= 1e-13 * np.max(np.abs(want_full))
    full, vfull = proj.reverse_transform(c["mean"].copy(), c["var"].copy())
    assert np.max(np.abs(full - want_full)) <= 1e-13 * np.max(np.abs(want_full))
    assert np.max(np.abs(vfull - want_var)) <= 1e-13 * np.max(np.abs(want_var))
    assert np.array_equal(vfull[:, c["dry"]], want_var[:, c["dry"]])  # exact zeros on the always-dry cells
