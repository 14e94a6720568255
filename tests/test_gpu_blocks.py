"""GPU parity of the building-block kernels, called through the C ABI, against numpy / the oracle.

Tolerances are for fp64: elementwise kernels 1e-13 relative, factorizations 1e-11 (they are
backward stable, the bound scales with the condition number of the test matrix).
"""

import ctypes as C

import numpy as np
import pytest
from scipy.linalg import cholesky, solve_triangular

from gpras_amd import _lib
from gpras_amd._lib import DeviceBuffer, check, ptr
from oracle import kernels as okn

pytestmark = pytest.mark.gpu


def rel_err(a, b):
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


def test_mfma_f64_layout_asymmetric(lib):
    """A = I with an asymmetric B catches a swapped C/D lane map (guide section 3)."""
    n = 64
    a = np.eye(n)
    b = np.arange(n * n, dtype=np.float64).reshape(n, n) * 0.5 + 1.0  # asymmetric
    dA, dB, dC = DeviceBuffer.from_array(a), DeviceBuffer.from_array(b), DeviceBuffer(n * n * 8)
    check(lib.gprx_gemm(0, 0, 0, n, n, n, 1.0, dA.ptr, n, dB.ptr, n, 0.0, dC.ptr, n, 0, 64))
    assert np.array_equal(dC.to_array((n, n)), b)
    check(lib.gprx_gemm(0, 0, 1, n, n, n, 1.0, dA.ptr, n, dB.ptr, n, 0.0, dC.ptr, n, 0, 64))
    assert np.array_equal(dC.to_array((n, n)), b.T)
    check(lib.gprx_gemm(0, 1, 0, n, n, n, 1.0, dB.ptr, n, dA.ptr, n, 0.0, dC.ptr, n, 0, 64))
    assert np.array_equal(dC.to_array((n, n)), b.T)


@pytest.mark.parametrize("ta,tb", [(0, 1), (0, 0), (1, 0)])
@pytest.mark.parametrize("tile", [64, 128])
@pytest.mark.parametrize("shape", [(256, 192, 64), (200, 136, 48), (64, 1000, 128), (384, 384, 256)])
def test_gemm_vs_numpy(lib, ta, tb, tile, shape):
    m, n, k = shape
    rng = np.random.default_rng(m * 7 + n * 3 + k + ta * 2 + tb)
    a = rng.standard_normal((k, m) if ta else (m, k))
    b = rng.standard_normal((n, k) if tb else (k, n))
    c0 = rng.standard_normal((m, n))
    ref = 0.7 * (a.T if ta else a) @ (b.T if tb else b) - 1.3 * c0
    dA, dB, dC = DeviceBuffer.from_array(a), DeviceBuffer.from_array(b), DeviceBuffer.from_array(c0)
    check(lib.gprx_gemm(0, ta, tb, m, n, k, 0.7, dA.ptr, a.shape[1], dB.ptr, b.shape[1], -1.3, dC.ptr, n, 0, tile))
    got = dC.to_array((m, n))
    assert rel_err(got, ref) < 1e-13


def test_gemm_triangular_flags(lib):
    """Zero-tile skipping must give the same product as the dense call on explicitly triangular operands."""
    n = 320
    rng = np.random.default_rng(5)
    lo = np.tril(rng.standard_normal((n, n)))
    de = rng.standard_normal((n, n))
    dL, dD, dC = DeviceBuffer.from_array(lo), DeviceBuffer.from_array(de), DeviceBuffer(n * n * 8)
    for tile in (64, 128):
        check(lib.gprx_gemm(0, 0, 0, n, n, n, 1.0, dL.ptr, n, dD.ptr, n, 0.0, dC.ptr, n, _lib.GEMM_A_LOWER, tile))
        assert rel_err(dC.to_array((n, n)), lo @ de) < 1e-13
        check(lib.gprx_gemm(0, 0, 0, n, n, n, 1.0, dD.ptr, n, dL.ptr, n, 0.0, dC.ptr, n, _lib.GEMM_B_LOWER, tile))
        assert rel_err(dC.to_array((n, n)), de @ lo) < 1e-13
        # X^T X with X lower: op(A) upper, B lower, lower tiles of C only
        check(lib.gprx_gemm(0, 1, 0, n, n, n, 1.0, dL.ptr, n, dL.ptr, n, 0.0, dC.ptr, n,
                            _lib.GEMM_C_LOWER | _lib.GEMM_A_UPPER | _lib.GEMM_B_LOWER, tile))
        got = np.tril(dC.to_array((n, n)))
        assert rel_err(got, np.tril(lo.T @ lo)) < 1e-13


@pytest.mark.parametrize("kernel", okn.KERNEL_NAMES)
@pytest.mark.parametrize("ard", [False, True])
def test_kmat_vs_oracle(lib, kernel, ard):
    rng = np.random.default_rng(11)
    n1, n2, d = 150, 200, 5
    a, b = rng.standard_normal((n1, d)), rng.standard_normal((n2, d))
    b[:20] = a[:20]  # coincident points: r2 == 0 exactly
    ls = rng.uniform(0.5, 2.0, size=d) if ard else np.full(d, 0.8)
    n1p, n2p = 192, 256
    dA, dB, dO = DeviceBuffer.from_array(a), DeviceBuffer.from_array(b), DeviceBuffer(n1p * n2p * 8)
    inv = np.ascontiguousarray(ls, dtype=np.float64)
    check(lib.gprx_kmat(0, okn.KERNEL_IDS[kernel], dA.ptr, n1, dB.ptr, n2, d, ptr(inv), 1.7, 0.0, dO.ptr, n2p, n1p, n2p, 0))
    got = dO.to_array((n1p, n2p))
    ref = okn.kmat(kernel, a, b, 1.7, ls)
    assert rel_err(got[:n1, :n2], ref) < 1e-13
    assert np.all(got[n1:, :] == 0.0) and np.all(got[:, n2:] == 0.0)
    # symmetric, lower tiles, identity padding, diagonal term
    dS = DeviceBuffer(n1p * n1p * 8)
    check(lib.gprx_kmat(0, okn.KERNEL_IDS[kernel], dA.ptr, n1, dA.ptr, n1, d, ptr(inv), 1.7, 0.25, dS.ptr, n1p, n1p, n1p, 2))
    got = dS.to_array((n1p, n1p))
    ref = okn.kmat(kernel, a, a, 1.7, ls) + 0.25 * np.eye(n1)
    assert rel_err(got[:n1, :n1], ref) < 1e-13
    assert np.array_equal(got[n1:, n1:], np.eye(n1p - n1))
    assert np.all(got[:n1, n1:] == 0.0) and np.all(got[n1:, :n1] == 0.0)
    assert np.array_equal(np.diag(got)[:n1], np.full(n1, 1.7 + 0.25))  # r2(a, a) == 0 exactly


@pytest.mark.parametrize("n", [64, 100, 448, 1000, 2100])
def test_kmat_lower_tiles_only_launch(lib, n):
    """mode 1 (what the Cholesky reads) launches only the T (T + 1) / 2 tiles on or below the diagonal, decoded from the linear workgroup
    index (kmat.h `tri`): every such tile equals the all-tiles build (mode 2) bit for bit, no tile above the diagonal is touched."""
    rng = np.random.default_rng(5)
    d = 3
    a = rng.standard_normal((n, d))
    npad = -(-n // 64) * 64
    inv = np.full(d, 0.9)
    dA = DeviceBuffer.from_array(a)
    full, low = DeviceBuffer(npad * npad * 8), DeviceBuffer.from_array(np.full((npad, npad), -7.0))
    check(lib.gprx_kmat(0, 0, dA.ptr, n, dA.ptr, n, d, ptr(inv), 1.3, 0.5, full.ptr, npad, npad, npad, 2))
    check(lib.gprx_kmat(0, 0, dA.ptr, n, dA.ptr, n, d, ptr(inv), 1.3, 0.5, low.ptr, npad, npad, npad, 1))
    f, g = full.to_array((npad, npad)), low.to_array((npad, npad))
    t = npad // 64
    for ti in range(t):
        for tj in range(t):
            blk = g[64 * ti:64 * ti + 64, 64 * tj:64 * tj + 64]
            if tj <= ti:
                assert np.array_equal(blk, f[64 * ti:64 * ti + 64, 64 * tj:64 * tj + 64]), (ti, tj)
            else:
                assert np.all(blk == -7.0), (ti, tj)


def test_kernel_build_exponential_within_one_ulp_of_numpy(lib):
    """the kernel build's exp (2^(j/64) table + degree-5 polynomial, gprx_common.h exp_nonpos_tab) against the oracle's
    np.exp on 1e7 arguments over the whole range of non-positive kernel arguments: <= 1 ulp everywhere, exact at the ends."""
    rng = np.random.default_rng(2026)
    x = np.concatenate([-rng.uniform(0.0, 40.0, 6_000_000), -rng.uniform(0.0, 1.0, 2_000_000), -np.exp(rng.uniform(-40.0, 6.6, 2_000_000)),
                        [0.0, -0.0, -1e-300, -745.0, -745.2, -746.0, -1000.0, -1e300, -np.inf]])
    got = np.empty_like(x)
    check(lib.gprx_exp_probe(0, 0, ptr(x), x.size, ptr(got)))
    with np.errstate(under="ignore"):
        ref = np.exp(x)
    ulp = np.spacing(ref)
    err = np.abs(got - ref) / ulp
    assert np.max(err) <= 1.0, (float(np.max(err)), float(x[np.argmax(err)]))
    assert got[-9] == 1.0 and got[-8] == 1.0 and got[-7] == 1.0 and np.all(got[-3:] == 0.0)
    assert np.mean(err > 0) < 0.3  # (mostly np.exp's value; measured 0.22 -- the rest one ulp beside it)
    nan = np.array([np.nan])
    out = np.zeros(1)
    check(lib.gprx_exp_probe(0, 0, ptr(nan), 1, ptr(out)))
    assert np.isnan(out[0])
    # the degree-13 form of the gradient passes, for the record (same bar)
    check(lib.gprx_exp_probe(0, 1, ptr(x[:1_000_000]), 1_000_000, ptr(got[:1_000_000])))
    assert np.max(np.abs(got[:1_000_000] - ref[:1_000_000]) / ulp[:1_000_000]) <= 1.0


@pytest.mark.parametrize("n,extra", [(64, 0), (128, 64), (448, 64), (1024, 128)])
def test_potrf_vs_scipy(lib, n, extra):
    rng = np.random.default_rng(n)
    g = rng.standard_normal((n, n + 8))
    spd = g @ g.T / n + 0.5 * np.eye(n)
    rhs = rng.standard_normal((extra, n))
    full = np.vstack([np.tril(spd) + np.triu(np.full((n, n), np.nan), 1), rhs])  # NaN above the diagonal: must never be read
    # the diagonal 64-blocks are read in full by the symmetric update of the GEMM: keep them finite there
    for b0 in range(0, n, 64):
        full[b0 : b0 + 64, b0 : b0 + 64] = spd[b0 : b0 + 64, b0 : b0 + 64]
    dA = DeviceBuffer.from_array(full)
    dI = DeviceBuffer(n * 64 * 8)
    info = C.c_int(-1)
    check(lib.gprx_potrf(0, dA.ptr, n, n, extra, dI.ptr, C.byref(info)))
    assert info.value == 0
    out = dA.to_array((n + extra, n))
    L_ref = cholesky(spd, lower=True)
    L = np.zeros((n, n))
    for b0 in range(0, n, 64):  # compare the lower part block row by block row
        L[b0 : b0 + 64, : b0 + 64] = out[b0 : b0 + 64, : b0 + 64]
    assert np.all(np.triu(L, 1) == 0.0)
    assert rel_err(L, L_ref) < 1e-11
    if extra:
        beta_ref = solve_triangular(L_ref, rhs.T, lower=True).T
        assert rel_err(out[n:], beta_ref) < 1e-11
    inv = dI.to_array((n // 64, 64, 64))
    for i in range(n // 64):
        blk = L_ref[64 * i : 64 * i + 64, 64 * i : 64 * i + 64]
        assert rel_err(inv[i], np.linalg.inv(blk)) < 1e-10


def test_potrf_reports_failing_pivot(lib):
    n = 128
    a = np.eye(n)
    a[70, 70] = -1.0
    dA, dI = DeviceBuffer.from_array(a), DeviceBuffer(n * 64 * 8)
    info = C.c_int(0)
    rc = lib.gprx_potrf(0, dA.ptr, n, n, 0, dI.ptr, C.byref(info))
    assert rc == _lib.GPRX_ENOTPD and info.value == 71
    with pytest.raises(np.linalg.LinAlgError):
        check(rc)


@pytest.mark.parametrize(
    "panel,outer,tile,split,inblock",
    [(64, 256, 0, 0, 0), (128, 128, 128, 0, 0), (64, 512, 64, 0, 1), (128, 256, 64, 0, 0), (64, 1024, 0, 1, 0), (64, 512, 64, 1, 1)],
)
def test_potrf_large_every_schedule(lib, panel, outer, tile, split, inblock):
    """N = 2048 under several schedules, repeated: catches ordering races that small grids hide (every
    workgroup of a panel launch re-reads the diagonal block, so it must not be overwritten in place
    during that launch; the bulk trailing update runs on a second stream)."""
    n, extra = 2048, 64
    rng = np.random.default_rng(1)
    g = rng.standard_normal((n, 96))
    spd = g @ g.T / 96 + np.eye(n)
    rhs = rng.standard_normal((extra, n))
    L_ref = cholesky(spd, lower=True)
    full = np.vstack([spd, rhs])
    try:
        for key, val in ((b"panel_width", panel), (b"outer_block", outer), (b"update_tile", tile), (b"split_panel", split), (b"inblock", inblock)):
            check(lib.gprx_set_tuning(key, val))
        first = None
        for _ in range(3):
            dA, dI = DeviceBuffer.from_array(full), DeviceBuffer(n * 64 * 8)
            info = C.c_int(0)
            check(lib.gprx_potrf(0, dA.ptr, n, n, extra, dI.ptr, C.byref(info)))
            out = dA.to_array((n + extra, n))
            assert rel_err(np.tril(out[:n]), L_ref) < 1e-11
            assert rel_err(out[n:], solve_triangular(L_ref, rhs.T, lower=True).T) < 1e-11
            if first is None:
                first = out
            else:
                assert np.array_equal(np.tril(out[:n]), np.tril(first[:n]))  # run-to-run reproducible
            dA.free()
            dI.free()
    finally:
        for key in (b"panel_width", b"outer_block", b"update_tile", b"split_panel", b"inblock"):
            lib.gprx_set_tuning(key, 0)


def test_split_panel_is_bit_identical_to_the_fused_panel(lib):
    """The rows-only kernels of the split panel (64 and 128 columns) repeat the fused kernel's arithmetic with L11 read
    from the staging area: fused 64 == split 64 and fused 128 == split 128 bit for bit."""
    n, extra = 1024, 64
    rng = np.random.default_rng(2)
    g = rng.standard_normal((n, 48))
    full = np.vstack([g @ g.T / 48 + np.eye(n), rng.standard_normal((extra, n))])
    outs = []
    try:
        for split, width in ((-1, 64), (1, 64), (-1, 128), (1, 128)):
            check(lib.gprx_set_tuning(b"split_panel", split))
            check(lib.gprx_set_tuning(b"panel_width", width))
            dA, dI = DeviceBuffer.from_array(full), DeviceBuffer(n * 64 * 8)
            info = C.c_int(0)
            check(lib.gprx_potrf(0, dA.ptr, n, n, extra, dI.ptr, C.byref(info)))
            outs.append((np.tril(dA.to_array((n + extra, n))[:n]), dA.to_array((n + extra, n))[n:], dI.to_array((n // 64, 64, 64))))
            dA.free()
            dI.free()
    finally:
        lib.gprx_set_tuning(b"split_panel", 0)
        lib.gprx_set_tuning(b"panel_width", 0)
    # fused and split panels of one width are bit-identical; the two widths agree to rounding (the K = 64 update between two
    # 64-column panels runs in the general GEMM kernel, whose k order differs from the 128-column kernels' sub-panel updates)
    for a, b in zip(outs[0], outs[1]):
        assert np.array_equal(a, b)
    for a, b in zip(outs[2], outs[3]):
        assert np.array_equal(a, b)
    for a, b in zip(outs[0], outs[2]):
        assert rel_err(a, b) < 1e-13


@pytest.mark.parametrize("n,extra,rt", [(128, 0, 1), (1024, 64, 1), (1088, 40, 2), (2048, 64, 2), (2048 + 192, 17, 1)])
def test_rows_by_inverse_product_equals_the_substitution_to_rounding(lib, n, extra, rt):
    """potrf_rows_inv_kernel (split panel, rows = A21 L11^-T as one MFMA tile product against the diagonal block's inverse,
    "rows_inv" = 1): against scipy and against the substitution schedule -- equal to rounding, not bit for bit -- for both
    tile counts per wave, the fused K = 64 update (odd panels) and ragged row counts; repeated runs are bit-identical."""
    rng = np.random.default_rng(n + rt)
    g = rng.standard_normal((n, 80))
    spd = g @ g.T / 80 + np.eye(n)
    rhs = rng.standard_normal((extra, n))
    full = np.vstack([spd, rhs]) if extra else spd
    L_ref = cholesky(spd, lower=True)
    outs = []
    try:
        for inv in (0, 1, 1):
            check(lib.gprx_set_tuning(b"split_panel", 1))
            check(lib.gprx_set_tuning(b"rows_inv", inv))
            check(lib.gprx_set_tuning(b"rows_inv_rt", rt))
            dA, dI = DeviceBuffer.from_array(full), DeviceBuffer(n * 64 * 8)
            info = C.c_int(0)
            check(lib.gprx_potrf(0, dA.ptr, n, n, extra, dI.ptr, C.byref(info)))
            assert info.value == 0
            outs.append(dA.to_array((n + extra, n)))
            dA.free()
            dI.free()
    finally:
        for key in (b"split_panel", b"rows_inv", b"rows_inv_rt"):
            lib.gprx_set_tuning(key, 0)
    for out in outs:
        assert rel_err(np.tril(out[:n]), L_ref) < 1e-11
        if extra:
            assert rel_err(out[n:], solve_triangular(L_ref, rhs.T, lower=True).T) < 1e-11
    assert rel_err(np.tril(outs[1][:n]), np.tril(outs[0][:n])) < 1e-13
    assert not np.array_equal(np.tril(outs[1][:n]), np.tril(outs[0][:n]))  # (it IS another arithmetic: the knob reached the kernel)
    assert np.array_equal(outs[1], outs[2])


@pytest.mark.parametrize("n,extra,ni", [(64, 64, 4), (192, 0, 4), (2048, 64, 4), (3136, 0, 2), (2112, 128, 1), (4096, 64, 4)])
def test_potrf_tile_dag(lib, n, extra, ni, monkeypatch):
    """The tile-DAG factorisation of a lone matrix (potrf_dag.h: one persistent launch, chain workgroup + task-queue workers,
    per-tile version counters), selected through the "dag" knob: factor, right-hand-side rows and the 64 x 64 inverse blocks
    against scipy; BIT-IDENTICAL run to run (a stale hand-off between workgroups would show as a differing factor -- the order
    of the updates of a tile is fixed by its version counter, so timing cannot change the result); within rounding of the
    launch-per-panel schedule; failing pivot reported with its global index."""
    monkeypatch.setenv("GPRX_DAG_NI", str(ni))  # tiles per panel task (read when the plan is built: every call here builds one)
    rng = np.random.default_rng(n)
    g = rng.standard_normal((n, 80))
    spd = g @ g.T / 80 + np.eye(n)
    rhs = rng.standard_normal((extra, n))
    L_ref = cholesky(spd, lower=True)
    full = np.vstack([spd, rhs])

    def run(dag):
        check(lib.gprx_set_tuning(b"dag", dag))
        dA, dI = DeviceBuffer.from_array(full), DeviceBuffer(n * 64 * 8)
        info = C.c_int(0)
        check(lib.gprx_potrf(0, dA.ptr, n, n, extra, dI.ptr, C.byref(info)))
        out, inv = dA.to_array((n + extra, n)), dI.to_array((n // 64, 64, 64))
        dA.free()
        dI.free()
        return out, inv

    try:
        first = None
        for _ in range(3):
            out, inv = run(1)
            assert rel_err(np.tril(out[:n]), L_ref) < 1e-11
            if extra:
                assert rel_err(out[n:], solve_triangular(L_ref, rhs.T, lower=True).T) < 1e-11
            for b in sorted({0, n // 128, n // 64 - 1}):
                blk = L_ref[64 * b : 64 * b + 64, 64 * b : 64 * b + 64]
                assert rel_err(inv[b], np.linalg.inv(blk)) < 1e-10
            for b0 in range(0, n, 64):  # explicit zeros above the diagonal inside the diagonal tiles (the triangular products rely on them)
                assert np.all(np.triu(out[b0 : b0 + 64, b0 : b0 + 64], 1) == 0.0)
            if first is None:
                first = out
            else:
                assert np.array_equal(np.tril(out[:n]), np.tril(first[:n])) and np.array_equal(out[n:], first[n:])
        base, _ = run(0)
        assert rel_err(np.tril(first[:n]), np.tril(base[:n])) < 1e-12
        if n > 1500:
            bad = spd.copy()
            bad[1500, 1500] = -1.0  # not positive definite: the pivot index is global, not relative to its diagonal block
            check(lib.gprx_set_tuning(b"dag", 1))
            dA, dI = DeviceBuffer.from_array(np.vstack([bad, rhs])), DeviceBuffer(n * 64 * 8)
            info = C.c_int(0)
            rc = lib.gprx_potrf(0, dA.ptr, n, n, extra, dI.ptr, C.byref(info))
            assert rc == _lib.GPRX_ENOTPD and info.value == 1501
            dA.free()
            dI.free()
    finally:
        lib.gprx_set_tuning(b"dag", 0)


def test_fit_through_the_tile_dag_equals_the_launch_schedule(lib):
    """gprx_factorize + predict with the handle's "dag" knob on: loss and predictions within rounding of the default schedule."""
    from gpras_amd.model import NOISE_LOWER, softplus_inv
    from gpras_amd.synth import make_regression

    x, y, xs = make_regression(1500, 6, n_outputs=1, n_test=300, config=2, unit=4)
    theta = np.ascontiguousarray([softplus_inv(1.1), softplus_inv(0.9), softplus_inv(0.2 - NOISE_LOWER)], dtype=np.float64)
    res = {}
    for dag in (0, 1):
        h = C.c_void_p()
        check(lib.gprx_create(0, 1500, 6, 0, _lib.KERNEL_IDS["Matern32"], 0, C.byref(h)))
        check(lib.gprx_set_data(h, ptr(x), ptr(y), 1), h)
        check(lib.gprx_set_handle_tuning(h, b"dag", dag), h)
        loss = C.c_double()
        check(lib.gprx_factorize(h, 0, ptr(theta), None, 7, C.byref(loss)), h)
        mean, var = np.empty(300), np.empty(300)
        check(lib.gprx_predict(h, ptr(xs), 300, ptr(mean), ptr(var), 1), h)
        res[dag] = (loss.value, mean, var)
        lib.gprx_destroy(h)
    assert abs(res[1][0] - res[0][0]) <= 1e-12 * abs(res[0][0])
    assert rel_err(res[1][1], res[0][1]) < 1e-11 and rel_err(res[1][2], res[0][2]) < 1e-11


@pytest.mark.parametrize("n,extra", [(2112, 64), (4096, 64)])
def test_potrf_lookahead_split_is_bit_identical(lib, n, extra):
    """"split_updates" = 1 (the K >= 256 updates of a lone matrix split by columns over a side stream, potrf.h): the same tiles
    with the same K ranges, so the factor, the right-hand-side rows and the inverse blocks equal the default schedule bit for bit."""
    rng = np.random.default_rng(n + 1)
    g = rng.standard_normal((n, 96))
    full = np.vstack([g @ g.T / 96 + np.eye(n), rng.standard_normal((extra, n))])
    outs = []
    try:
        for knob in (0, 1, 1):
            check(lib.gprx_set_tuning(b"split_updates", knob))
            dA, dI = DeviceBuffer.from_array(full), DeviceBuffer(n * 64 * 8)
            info = C.c_int(0)
            check(lib.gprx_potrf(0, dA.ptr, n, n, extra, dI.ptr, C.byref(info)))
            outs.append((np.tril(dA.to_array((n + extra, n))[:n]), dA.to_array((n + extra, n))[n:], dI.to_array((n // 64, 64, 64))))
            dA.free()
            dI.free()
    finally:
        lib.gprx_set_tuning(b"split_updates", 0)
    for other in outs[1:]:
        assert all(np.array_equal(a, b) for a, b in zip(outs[0], other))
