"""Finite-difference refresh fused into the J^T J kernel (k_jtj2<., false, true>, mir_lsq_gpu_options.fbRowMajor).

Kernel level (mir_lsq_fd_jtj_d): J is bit-exact against the reference's column arithmetic copy / axpy(-1) / scal(1/twh)
(LS:1041-1047) done in numpy; J^T J and J^T y are bit-exact on exact-integer inputs and within rounding otherwise.
Whole path: solves through the row-major batched callback against the same solves through the point-major callback +
k_fd_fill, and the user-side row-major residual kernel against its point-major twin."""
import ctypes as C

import numpy as np
import pytest

import mir_optim_amd as M
from mir_optim_amd import api, workloads as W
import problems as P
from test_gpu_broyden_lr import assert_same_trajectory

pytestmark = pytest.mark.gpu


def ref_fill(Yrm, twh):
    yp, ym = Yrm[:, 0::2], Yrm[:, 1::2]
    d = yp.copy()
    d += -1.0 * ym
    with np.errstate(divide="ignore"):
        inv = 1.0 / twh
    J = d * inv
    J[:, twh == 0] = 0.0
    return J


@pytest.mark.parametrize("m,n", [(4096, 128), (5000, 16), (30, 32), (2, 48), (10002, 64), (7778, 80), (12344, 96),
                                 (6, 112), (100000, 128), (4097, 128), (10001, 64), (1, 16), (33, 32), (7777, 112),
                                 (5000, 100), (3001, 7), (2048, 17), (999, 33), (6000, 127), (40, 1), (12345, 90),
                                 # 128 < n <= 256, n % 32 == 0: k_jtj_fdp8 (eight producer + consumer waves, jtj_fdp8.h)
                                 (4096, 256), (5001, 160), (33, 192), (10003, 224), (50000, 256), (1, 256), (17, 160),
                                 # ... and any other n there: the kernel of n rounded up to a multiple of 32, zero padding columns
                                 (4096, 200), (5001, 250), (33, 129), (10003, 161), (20000, 255), (1, 193), (777, 240)])
def test_fd_jtj_exact_integers(m, n):
    rng = np.random.default_rng(m + n)
    Yrm = rng.integers(-8, 9, size=(m, 2 * n)).astype(np.float64)
    twh = np.full(n, 2.0 ** -25)
    twh[rng.integers(0, n)] = 2.0 ** -26                       # a clipped interval
    if n > 3:
        twh[3] = 0.0                                           # a collapsed one
    y = rng.integers(-4, 5, size=m).astype(np.float64)
    J, JJ, Jy, ms = M.fd_jtj(Yrm, twh, y)
    Jr = ref_fill(Yrm, twh)
    assert np.array_equal(J, Jr)
    # entries are integers times powers of two: sums are exact as long as they stay below 2^53 ulps of the scale
    assert np.array_equal(JJ, Jr.T @ Jr) or np.allclose(JJ, Jr.T @ Jr, rtol=1e-15, atol=0)
    assert np.array_equal(Jy, Jr.T @ y) or np.allclose(Jy, Jr.T @ y, rtol=1e-15, atol=0)
    assert np.array_equal(JJ, JJ.T)


@pytest.mark.parametrize("m,n", [(4096, 128), (5000, 16), (30, 32), (10002, 64), (7778, 80), (12344, 96), (6, 112), (50000, 128),
                                 (4097, 127), (5000, 99), (2000, 9), (12345, 65), (3, 1)])
def test_plain_and_fused_producer_consumer_kernels_agree_on_exact_integers(m, n):
    """The plain J^T J (k_jtj_fdp<., false>) of the J the fused finite-difference kernel wrote: bit-exact on exact-integer
    inputs, like the fused kernel itself."""
    rng = np.random.default_rng(3 * m + n)
    Yrm = rng.integers(-8, 9, size=(m, 2 * n)).astype(np.float64)
    twh = np.full(n, 2.0 ** -25)
    y = rng.integers(-4, 5, size=m).astype(np.float64)
    J0, JJ0, Jy0, _ = M.fd_jtj(Yrm, twh, y)
    P0 = M.jtj(J0, y)
    Jr = ref_fill(Yrm, twh)
    assert np.array_equal(J0, Jr)
    for JJ, Jy in ((JJ0, Jy0), (P0[0], P0[1])):
        assert np.array_equal(JJ, Jr.T @ Jr) or np.allclose(JJ, Jr.T @ Jr, rtol=1e-15, atol=0)
        assert np.array_equal(Jy, Jr.T @ y) or np.allclose(Jy, Jr.T @ y, rtol=1e-15, atol=0)


@pytest.mark.parametrize("m,n", [(20000, 128), (9998, 32), (50, 16), (33334, 112), (5000, 100), (7001, 9), (3000, 126),
                                 (20000, 256), (7001, 192), (2049, 160), (30001, 224)])
def test_fd_jtj_random(m, n):
    rng = np.random.default_rng(7 * m + n)
    base = rng.standard_normal((m, 1))
    Jtrue = rng.standard_normal((m, n))
    h = 2.0 ** -26
    Yrm = np.empty((m, 2 * n))
    Yrm[:, 0::2] = base + h * Jtrue
    Yrm[:, 1::2] = base - h * Jtrue
    twh = np.full(n, 2 * h)
    y = rng.standard_normal(m)
    J, JJ, Jy, ms = M.fd_jtj(Yrm, twh, y)
    Jr = ref_fill(Yrm, twh)
    assert np.array_equal(J, Jr)
    assert np.allclose(JJ, Jr.T @ Jr, rtol=1e-12, atol=1e-9 * m)
    assert np.allclose(Jy, Jr.T @ y, rtol=1e-12, atol=1e-9 * m)
    # the same J through the plain kernel: the two J^T J differ by summation order only
    JJ2, Jy2, _, _ = M.jtj(Jr, y)
    assert np.allclose(JJ, JJ2, rtol=1e-13, atol=1e-10 * m) and np.allclose(Jy, Jy2, rtol=1e-13, atol=1e-10 * m)


@pytest.mark.parametrize("m,n", [(5000, 128), (4098, 64), (3000, 32), (2500, 16), (2222, 48), (40, 128), (31, 32)])
def test_row_major_batched_residual_equals_point_major(m, n):
    w = P.tanh_linear(m, n)
    prob = W.TanhLinear(w["A"], w["b"])
    rng = np.random.default_rng(n)
    p = 2 * n
    X = w["x0"][None, :] + 1e-3 * rng.standard_normal((p, n))
    dX = api.DeviceBuffer(np.ascontiguousarray(X))
    dY1 = api.DeviceBuffer(nbytes=p * m * 8, dtype=np.float64, shape=(p, m))
    dY2 = api.DeviceBuffer(nbytes=p * m * 8, dtype=np.float64, shape=(m, p))
    FB = C.CFUNCTYPE(None, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p)
    FB(prob.fb)(C.addressof(prob.ctx), m, n, p, dX.ptr, dY1.ptr)
    FB(prob.fbr)(C.addressof(prob.ctx), m, n, p, dX.ptr, dY2.ptr)
    prob.stream.synchronize()
    Y1, Y2 = dY1.download(), dY2.download()
    expect = np.tanh(w["A"] @ X.T) - w["b"][:, None]
    assert np.allclose(Y2, expect, rtol=0, atol=1e-13)
    if n in (32, 64, 128) and m >= 32:
        assert np.array_equal(Y2, Y1.T)                          # same kernel, same arithmetic, other store pattern
    else:
        assert np.allclose(Y2, Y1.T, rtol=0, atol=1e-14)
    for b in (dX, dY1, dY2):
        b.free()


@pytest.mark.parametrize("m,p", [(5000, 512), (4113, 144), (33, 16), (2048, 400), (777, 272), (16 * 256 * 3 + 5, 128)])
def test_batched_residual_n256_any_point_count(m, p):
    """workloads_gemm.hip at n = 256: eight compute waves that issue their own DMA. Point counts that leave waves without points in
    the last sweep over A (they keep loading their share and keep the barrier count), row counts that end inside a 16-row stage,
    more stages than workgroups: all three layouts against numpy, and against each other bit for bit."""
    n = 256
    w = P.tanh_linear(m, n)
    prob = W.TanhLinear(w["A"], w["b"])
    rng = np.random.default_rng(p)
    X = w["x0"][None, :] + 1e-3 * rng.standard_normal((p, n))
    dX = api.DeviceBuffer(np.ascontiguousarray(X))
    dY1 = api.DeviceBuffer(np.zeros((p, m)))
    dY2 = api.DeviceBuffer(np.zeros((m, p)))
    dD = api.DeviceBuffer(np.zeros((m, p // 2)))
    WL = api.workloads_lib()
    ctx = C.c_void_p(C.addressof(prob.ctx))
    args = (ctx, C.c_size_t(m), C.c_size_t(n), C.c_size_t(p), C.c_void_p(dX.ptr))
    WL.wl_tanh_linear_fb_d(*args, C.c_void_p(dY1.ptr))
    WL.wl_tanh_linear_fbr_d(*args, C.c_void_p(dY2.ptr))
    WL.wl_tanh_linear_fbd_d(*args, C.c_void_p(dD.ptr))
    prob.stream.synchronize()
    Y1, Y2, D = dY1.download(), dY2.download(), dD.download()
    expect = np.tanh(w["A"] @ X.T) - w["b"][:, None]
    assert np.allclose(Y2, expect, rtol=0, atol=1e-13)
    assert np.array_equal(Y2, Y1.T)
    assert np.array_equal(D, Y2[:, 0::2] - Y2[:, 1::2])
    for b in (dX, dY1, dY2, dD):
        b.free()


@pytest.mark.parametrize("m,n,bounded", [(20000, 32, False), (50000, 128, False), (4096, 16, False), (30000, 64, False),
                                         (7000, 48, False), (3000, 16, True), (10000, 96, True), (20001, 32, False),
                                         (9999, 128, False), (5001, 80, True), (9973, 100, False), (6000, 7, False),
                                         (8000, 33, True), (12000, 90, False), (30000, 256, False), (9001, 192, True),
                                         (12000, 160, False)])
def test_fused_fd_solve_matches_fill_pass(m, n, bounded):
    w = P.tanh_linear(m, n)
    prob = W.TanhLinear(w["A"], w["b"])
    s = M.LeastSquaresSettings(); s.absTolerance = 1e-9
    lo = up = None
    if bounded:
        lo = np.where(np.arange(n) % 3 == 0, w["xstar"] + 0.02, -np.inf)     # the minimiser violates a third of the bounds
        up = np.full(n, np.inf)
    x0 = w["x0"] if not bounded else np.maximum(w["x0"], np.where(np.isfinite(lo), lo, -np.inf))
    out = {}
    for mode in (True, "pointmajor"):
        tr, st = M.Trace(4096), M.Stats()
        res, x = prob.solve(x0, l=lo, u=up, settings=s, batched=mode, trace=tr, stats=st, flags=M.TIME_KERNELS)
        out[mode] = (res, x, tr.records(), st)
    (rf, xf, tf, sf), (rp, xp, tp, sp) = out[True], out["pointmajor"]
    assert sf.jtj_fd_launches == sf.jacobian_full >= 1 and sp.jtj_fd_launches == 0
    assert (rf.fCalls, rf.gCalls) == (rp.fCalls, rp.gCalls) or first_differs_late(tf, tp)
    assert int(rf.status) >= 0 and int(rp.status) >= 0
    if [a[0] for a in tf] == [b[0] for b in tp]:
        assert np.allclose(xf, xp, rtol=1e-6, atol=1e-9), np.abs(xf - xp).max()
    else:
        # two end games (the two J differ in the last bits, a noise-decided acceptance branches, one run makes a pass more): each
        # stops within the tolerance of the minimiser, so they agree NORM-wise to 1e-6 -- a component of 0.15 need not to 1e-7
        assert np.abs(xf - xp).max() <= 1e-6 * np.abs(xp).max(), np.abs(xf - xp).max()
    assert np.isclose(rf.residual, rp.residual, rtol=1e-9)
    assert_same_trajectory(tf, tp, (m, n, bounded))


def first_differs_late(ta, tb):
    k = next((i for i, (a, b) in enumerate(zip(ta, tb)) if a[0] != b[0]), min(len(ta), len(tb)))
    return k >= 5


def test_collapsed_interval_gives_zero_column_through_the_fused_path():
    """x_j pinned by l_j = u_j: xph == xmh, twh = 0, the column of J is zero (LS:1033, 1046) and x_j never moves."""
    m, n = 6000, 32
    w = P.tanh_linear(m, n)
    prob = W.TanhLinear(w["A"], w["b"])
    lo, up = np.full(n, -np.inf), np.full(n, np.inf)
    lo[5] = up[5] = w["x0"][5]
    s = M.LeastSquaresSettings(); s.absTolerance = 1e-9
    r1, x1 = prob.solve(w["x0"], l=lo, u=up, settings=s, batched=True)
    r2, x2 = prob.solve(w["x0"], l=lo, u=up, settings=s, batched="pointmajor")
    assert x1[5] == w["x0"][5] == x2[5]
    assert int(r1.status) >= 0 and np.allclose(x1, x2, rtol=1e-6, atol=1e-9) and np.isclose(r1.residual, r2.residual, rtol=1e-9)


# ---- the m x n DIFFERENCE panel (mir_lsq_gpu_options.fbRowMajorDiff, k_jtj_fdp<., false, true>) --------------------------------

@pytest.mark.parametrize("m,n", [(4096, 128), (5000, 16), (30, 32), (2, 48), (10002, 64), (7778, 80), (6, 112), (100000, 128),
                                 (4097, 128), (1, 16), (12345, 90), (5000, 100), (33, 2),
                                 (4096, 256), (5001, 192), (50000, 256), (1, 256), (17, 192),
                                 (4097, 127), (5000, 99), (333, 1), (2000, 9), (12345, 65), (50001, 127), (7, 33)])   # odd n: 8-byte loads
def test_fd_diff_panel_gives_the_pair_panel_jacobian_bit_for_bit(m, n):
    """D = Y+ + (-1) Y- formed by the caller (LS:1041, 1045), scal(1 / twh) by the kernel (LS:1047): the same J as from the pair
    panel, bit for bit, incl. a clipped and a collapsed interval; J^T J / J^T y exact on exact-integer inputs."""
    rng = np.random.default_rng(11 * m + n)
    for integers in (True, False):
        Yrm = rng.integers(-8, 9, size=(m, 2 * n)).astype(np.float64) if integers else rng.standard_normal((m, 2 * n))
        twh = np.full(n, 2.0 ** -25)
        twh[rng.integers(0, n)] = 2.0 ** -26
        if n > 3:
            twh[3] = 0.0
        y = rng.integers(-4, 5, size=m).astype(np.float64) if integers else rng.standard_normal(m)
        D = Yrm[:, 0::2].copy()
        D += -1.0 * Yrm[:, 1::2]
        Jp, JJp, Jyp, _ = M.fd_jtj(Yrm, twh, y)
        Jd, JJd, Jyd, _ = M.fd_jtj(D, twh, y, diff=True)
        assert np.array_equal(Jd, Jp) and np.array_equal(Jd, ref_fill(Yrm, twh))
        assert np.array_equal(JJd, JJd.T)
        if integers:
            assert np.array_equal(JJd, Jd.T @ Jd) or np.allclose(JJd, Jd.T @ Jd, rtol=1e-15, atol=0)
            assert np.array_equal(Jyd, Jd.T @ y) or np.allclose(Jyd, Jd.T @ y, rtol=1e-15, atol=0)
        else:
            scale = np.sqrt(np.outer(np.diag(JJp), np.diag(JJp))) + 1e-300
            assert np.max(np.abs(JJd - JJp) / scale) < 1e-13 and np.allclose(Jyd, Jyp, rtol=1e-10, atol=1e-13 * np.abs(Jyp).max())


@pytest.mark.parametrize("m,n", [(40000, 128), (10002, 64), (5001, 32), (3000, 16), (2049, 100), (20000, 256), (4001, 192)])
def test_user_side_difference_kernel_is_the_pair_kernel_minus(m, n):
    """workloads.hip: wl_tanh_linear_fbd_d writes exactly column 2j minus column 2j + 1 of what wl_tanh_linear_fbr_d writes."""
    w = P.tanh_linear(m, n)
    prob = W.TanhLinear(w["A"], w["b"])
    rng = np.random.default_rng(n)
    p = 2 * n
    X = np.repeat(w["x0"][None, :], p, axis=0)
    h = 2.0 ** -26
    X[np.arange(p), np.arange(p) // 2] += h * (1 - 2 * (np.arange(p) % 2))
    X += 1e-3 * rng.standard_normal((1, n))
    dX = api.DeviceBuffer(X)
    dY = api.DeviceBuffer(np.zeros((m, p)))
    dD = api.DeviceBuffer(np.zeros((m, n)))
    WL = api.workloads_lib()
    ctx = C.c_void_p(C.addressof(prob.ctx))
    WL.wl_tanh_linear_fbr_d(ctx, C.c_size_t(m), C.c_size_t(n), C.c_size_t(p), C.c_void_p(dX.ptr), C.c_void_p(dY.ptr))
    prob.stream.synchronize()
    Y = dY.download()
    for once in (0, 1):
        # read_a_once = 1: the stage-outer variant (one sweep over A, operands of the k-steps selected from the base point's
        # fragments and the one perturbed coordinate per point): the same operands, the same bits
        prob.ctx.read_a_once = once
        dD.upload(np.zeros((m, n)))
        WL.wl_tanh_linear_fbd_d(ctx, C.c_size_t(m), C.c_size_t(n), C.c_size_t(p), C.c_void_p(dX.ptr), C.c_void_p(dD.ptr))
        prob.stream.synchronize()
        D = dD.download()
        assert np.array_equal(D, Y[:, 0::2] - Y[:, 1::2]), once
        assert np.abs(D).max() > 0


@pytest.mark.parametrize("m,n", [(60000, 128), (30000, 64), (20001, 32), (9000, 100), (30000, 256), (10001, 192)])
def test_solve_through_the_difference_panel_equals_the_pair_panel_solve(oracle, m, n):
    w = P.tanh_linear(m, n)
    prob = W.TanhLinear(w["A"], w["b"])
    s = M.LeastSquaresSettings(); s.absTolerance = 1e-7
    sd, sp = M.Stats(), M.Stats()
    rd, xd = prob.solve(w["x0"], settings=s, batched=True, stats=sd)            # difference panel (fbRowMajorDiff)
    rp, xp = prob.solve(w["x0"], settings=s, batched="rowmajor", stats=sp)      # pair panel (fbRowMajor)
    assert rd.status >= 0 and rd.status == rp.status and rd.fCalls == rp.fCalls
    if n % 64 == 0:          # same stage partition in both kernels: the same bits all the way
        assert np.array_equal(xd, xp) and rd.residual == rp.residual and rd.iterations == rp.iterations
    else:
        assert np.allclose(xd, xp, rtol=1e-9, atol=1e-12) and np.isclose(rd.residual, rp.residual, rtol=1e-12)
    assert sd.jacobian_full == sp.jacobian_full >= 1
