"""Distance forms on the device: gpflow evaluates the scaled squared distance in the expanded form
``|a|^2 + |b|^2 - 2 a.b``; the HIP path defaults to the difference form.  ``gprx_set_distance_form``
switches every kernel evaluation of a handle to the expanded form.  Here BOTH device forms are compared with
``oracle(form="expanded")`` -- gpflow's literal arithmetic -- and with each other, for all five kernels, on the sparse model
the reference runs and on the exact specialisation.  Tolerances: the smooth kernels (RBF, Matern32, Matern52) agree to
rounding; Matern12 and "Exponential" are not differentiable at r = 0, where the expanded form leaves r2 ~ 1e-15 instead of
0 on coincident points (Kuu's diagonal), so their outputs carry 1e-9..1e-8 of BLAS-order-dependent noise in ANY expanded
implementation (``tests/test_oracle.py::test_distance_forms_agree`` needs the same envelope between the oracle's two
forms).  The measured gaps are written to ``gpurun_out/distance_form_gaps.json`` and quoted in DESIGN.md section 1."""

import ctypes as C
import json
import os

import numpy as np
import pytest

from gpras_amd import _lib
from gpras_amd._lib import check, ptr
from gpras_amd.synth import make_regression
from oracle import exact as oex
from oracle import gpras_oracle
from oracle import kernels as okn
from oracle import sgpr as osg
from oracle import transforms as otr

pytestmark = pytest.mark.gpu

SMOOTH = ("RBF", "Matern32", "Matern52")
GAPS = {}


def tolerances(kernel, form="difference"):
    """(loss, mean, variance) relative, against oracle(form="expanded").
    Device EXPANDED form: the kernel restates numpy's order (squares rounded and summed in k order, the dot product as an fma
    chain in k order), so it reproduces the oracle's expanded arithmetic INCLUDING its rounding noise on coincident points:
    measured <= 3e-15 on every output of every kernel (gpurun_out/distance_form_gaps.json).
    Device DIFFERENCE form (default): rounding for the smooth kernels; for Matern12 / Exponential the envelope measured
    between the two forms (sparse model: 2e-9 loss, 1e-9 mean, 1e-8 variance; exact N = 256: 3e-9, 1.2e-8, 2.5e-8), asserted
    with a factor 2-4 of head room -- the same size the oracle shows between its own two forms."""
    if form == "expanded" or kernel in SMOOTH:
        return (1e-12, 1e-11, 1e-11)
    return (1e-8, 3e-8, 1e-7)


def device_eval(lib, kernel, x, y, z, theta, xs, form):
    n, d = x.shape
    m = 0 if z is None else z.shape[0]
    h = C.c_void_p()
    check(lib.gprx_create(0, n, d, m, okn.KERNEL_IDS[kernel], 0, C.byref(h)))
    try:
        check(lib.gprx_set_data(h, ptr(x), ptr(y), y.shape[1]), h)
        check(lib.gprx_set_distance_form(h, _lib.DISTANCE_FORMS[form]), h)
        loss = C.c_double()
        grad = np.zeros(theta.size + m * d)
        check(lib.gprx_objective(h, 0, ptr(theta), None if z is None else ptr(z), 15 if m else 7, C.byref(loss), ptr(grad)), h)
        mean, var = np.empty(xs.shape[0]), np.empty(xs.shape[0])
        check(lib.gprx_predict(h, ptr(xs), xs.shape[0], ptr(mean), ptr(var), 1), h)
        return loss.value, grad, mean, var
    finally:
        lib.gprx_destroy(h)


def rel(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / np.max(np.abs(b)))


def record(case, kernel, name, value):
    GAPS.setdefault(case, {}).setdefault(kernel, {})[name] = value


@pytest.mark.parametrize("kernel", okn.KERNEL_NAMES)
def test_sparse_model_both_forms_against_gpflows_arithmetic(lib, kernel):
    x, y, xs = make_regression(200, 4, 1, 40, config=9, unit=3)  # the oracle's own distance-form test case
    z = np.ascontiguousarray(gpras_oracle.create_inducing(x, 24, "kmeans"))
    variance, ls, noise = 1.2, 0.8, 0.1
    wv, wl, wn = otr.unconstrain(variance, ls, noise)
    theta = np.ascontiguousarray([wv, wl, wn], dtype=np.float64)
    ref_loss = osg.loss(kernel, x, y[:, 0], z, float(wv), float(wl), float(wn), form="expanded")
    ref_mean, ref_var = osg.predict(kernel, x, y[:, 0], z, variance, ls, noise, xs, form="expanded")
    out = {}
    for form in ("difference", "expanded"):
        tl, tm, tv = tolerances(kernel, form)
        loss, grad, mean, var = device_eval(lib, kernel, x, y, z, theta, xs, form)
        out[form] = (loss, grad, mean, var)
        gl, gm, gv = abs(loss - ref_loss) / abs(ref_loss), rel(mean, ref_mean), float(np.max(np.abs(var - ref_var) / ref_var))
        for name, val in (("loss", gl), ("mean", gm), ("var", gv)):
            record("sparse_n200_m24", kernel, f"device_{form}_vs_oracle_expanded_{name}", val)
        assert gl <= tl and gm <= tm and gv <= tv, (kernel, form, gl, gm, gv)
    # the two device forms against each other: same envelope; the hyperparameter gradient moves no more than the loss does
    a, b = out["difference"], out["expanded"]
    tl, tm, tv = tolerances(kernel, "difference")
    assert abs(a[0] - b[0]) <= tl * abs(a[0])
    assert rel(a[2], b[2]) <= tm and float(np.max(np.abs(a[3] - b[3]) / b[3])) <= tv
    gtol = 1e-10 if kernel in SMOOTH else 1e-5
    gap = float(np.max(np.abs(a[1][:3] - b[1][:3]) / (np.abs(b[1][:3]) + 1e-3 * np.max(np.abs(b[1][:3])))))
    record("sparse_n200_m24", kernel, "device_forms_theta_gradient", gap)
    assert gap <= gtol, (kernel, gap)


@pytest.mark.parametrize("kernel", okn.KERNEL_NAMES)
def test_exact_model_both_forms(lib, kernel):
    x, y, xs = make_regression(256, 4, 1, 60, config=9, unit=5)
    variance, ls, noise = 0.9, 0.7, 0.15
    wv, wl, wn = otr.unconstrain(variance, ls, noise)
    theta = np.ascontiguousarray([wv, wl, wn], dtype=np.float64)
    ref_loss = oex.loss(kernel, x, y[:, 0], float(wv), float(wl), float(wn), form="expanded")
    ref_mean, ref_var = oex.predict(kernel, x, y[:, 0], variance, ls, noise, xs, form="expanded")
    for form in ("difference", "expanded"):
        tl, tm, tv = tolerances(kernel, form)
        loss, _, mean, var = device_eval(lib, kernel, x, y, None, theta, xs, form)
        gl, gm, gv = abs(loss - ref_loss) / abs(ref_loss), rel(mean, ref_mean), float(np.max(np.abs(var - ref_var) / ref_var))
        for name, val in (("loss", gl), ("mean", gm), ("var", gv)):
            record("exact_n256", kernel, f"device_{form}_vs_oracle_expanded_{name}", val)
        assert gl <= tl and gm <= tm and gv <= tv, (kernel, form, gl, gm, gv)


@pytest.mark.parametrize("kernel", okn.KERNEL_NAMES)
def test_kernel_matrix_expanded_form_elementwise(lib, kernel):
    """gprx_kmat with mode + 4 against oracle kmat(form="expanded"): elementwise, including the diagonal of k(Z, Z) where the
    expanded form differs from the exact v."""
    from gpras_amd._lib import DeviceBuffer

    rng = np.random.default_rng(11)
    a = np.ascontiguousarray(rng.standard_normal((100, 5)) * 2.0)
    ls = np.linspace(0.6, 1.4, 5)
    da, dout = DeviceBuffer.from_array(a), DeviceBuffer(8 * 128 * 128)
    check(lib.gprx_kmat(0, okn.KERNEL_IDS[kernel], da.ptr, 100, da.ptr, 100, 5, ptr(ls), 1.3, 0.0, dout.ptr, 128, 128, 128, 2 + 4))
    got = dout.to_array((128, 128))[:100, :100]
    want = okn.kmat(kernel, a, a, 1.3, ls, form="expanded")
    off = ~np.eye(100, dtype=bool)
    assert np.max(np.abs(got[off] - want[off])) <= 1e-13
    # diagonal: r2 is rounding noise of size eps * |a/l|^2 in both implementations (its value depends on the summation
    # order), so k = v g(sqrt(noise)); bound it by the size of that noise
    r2_noise = 8 * np.finfo(float).eps * np.max(np.sum((a / ls) ** 2, axis=1))
    bound = 1.3 * (2.0 * r2_noise if kernel in SMOOTH else 2.3 * np.sqrt(r2_noise)) + 1e-15
    assert np.max(np.abs(np.diag(got) - np.diag(want))) <= bound
    record("kmat_diag", kernel, "max_abs_diag_gap", float(np.max(np.abs(np.diag(got) - np.diag(want)))))
    check(lib.gprx_kmat(0, okn.KERNEL_IDS[kernel], da.ptr, 100, da.ptr, 100, 5, ptr(ls), 1.3, 0.0, dout.ptr, 128, 128, 128, 2))
    direct = dout.to_array((128, 128))[:100, :100]
    assert np.array_equal(np.diag(direct), np.full(100, 1.3))  # difference form: r2(a, a) == 0 exactly
    da.free()
    dout.free()


def test_gpras_distance_form_option():
    """The option reaches GPRAS: a fit + predict with distance_form="expanded" against the oracle driver in the same form."""
    from gpras_amd.gpr import GPRAS

    x, y, xs = make_regression(256, 4, n_outputs=2, n_test=50, config=1, unit=0)
    g = GPRAS("Matern32", distance_form="expanded")
    g.fit(x, y, 32, "kmeans", "adam", max_iter=3)
    assert g.engine.distance_form == "expanded"
    ref = gpras_oracle.GPRASOracle("Matern32")
    ref.fit(x, y, 32, "kmeans", "adam", max_iter=3)
    mean, var = g.predict(xs)
    rmean, rvar = ref.predict(xs)
    assert rel(mean, rmean) < 1e-8 and float(np.max(np.abs(var - rvar) / rvar)) < 1e-8


@pytest.mark.parametrize("n_inducing", [24, None])
@pytest.mark.parametrize("kernel", okn.KERNEL_NAMES)
def test_default_constructed_gpras_within_1e8_of_gpflows_arithmetic(kernel, n_inducing):
    """VERDICT r2 item 6 / north_star "within 1e-8": ``GPRAS(kernel)`` with NO distance_form argument against the oracle driver in
    gpflow's literal (expanded) arithmetic, all five kernels, the sparse model the reference runs and the exact specialisation.
    Matern12 / Exponential default to the expanded form for this reason (gpr.py DEFAULT_DISTANCE_FORM)."""
    from gpras_amd.gpr import DEFAULT_DISTANCE_FORM, GPRAS

    x, y, xs = make_regression(256, 4, n_outputs=2, n_test=60, config=9, unit=7)
    g = GPRAS(kernel)
    assert g.distance_form == DEFAULT_DISTANCE_FORM.get(kernel, "difference")
    g.fit(x, y, n_inducing, "kmeans", "adam", max_iter=3)
    assert g.engine.distance_form == g.distance_form
    ref = gpras_oracle.GPRASOracle(kernel, form="expanded")
    ref.fit(x, y, n_inducing, "kmeans", "adam", max_iter=3)
    mean, var = g.predict(xs)
    rmean, rvar = ref.predict(xs)
    gm, gv = rel(mean, rmean), float(np.max(np.abs(var - rvar) / rvar))
    record("default_gpras_n256" + ("_exact" if n_inducing is None else "_m24"), kernel, "mean", gm)
    record("default_gpras_n256" + ("_exact" if n_inducing is None else "_m24"), kernel, "var", gv)
    assert gm <= 1e-8 and gv <= 1e-8, (kernel, n_inducing, gm, gv)


def test_zz_write_measured_gaps():
    """(runs last in this module) keep the measured numbers for DESIGN.md"""
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, "distance_form_gaps.json"), "w") as f:
        json.dump(GAPS, f, indent=1, sort_keys=True)
    assert GAPS
