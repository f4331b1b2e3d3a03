"""fitSpline (the caller of the LM path at /root/reference/source/mir/optim/fit_splie.d:26-85).

CPU part: the spline restatement against scipy's not-a-knot CubicSpline, and the ORACLE driven by the library's
residual function against the reference's own unittest (FS:88-141) -- two more known answers that pin the
oracle, including the reference's quirks (first-derivative penalty, last point replaced when lambda != 0).
GPU part: mir_fit_spline_d end to end against the same known answers and against the oracle."""
import numpy as np
import pytest
from scipy.interpolate import CubicSpline

import mir_optim_amd as M

# FS:96-117 and FS:125-138: the reference's unittest data and expected spline values (data, not code)
X = np.array([-1.0, 2, 4, 5, 8, 10, 12, 15, 19, 22])
Y0 = np.array([17.0, 0, 16, 4, 10, 15, 19, 5, 18, 6])
POINTS = np.column_stack([X + 0.5, [-0.68361541, 7.28568719, 10.490694, 0.36192032, 11.91572713, 16.44546433,
                                   17.66699525, 4.52730869, 19.22825394, -2.3242592]])
Y_LAMBDA = np.array([15.898984945597563, 0.44978154774119194, 15.579636654078188, 4.028312405287987, 9.945895290402778,
                     15.07778815727665, 18.877926155854535, 5.348699237978274, 16.898507797404278, 22.024920998359942])
APPROX = 2.0 ** -20         # mir.test shouldApprox: maxRelDiff = maxAbsDiff = 0x1p-20


def approx(a, b):
    return np.all(np.abs(a - b) <= APPROX * np.maximum(np.abs(a), np.abs(b)) + APPROX)


@pytest.mark.parametrize("n", [2, 3, 4, 5, 10, 37])
def test_spline_matches_scipy_not_a_knot(n):
    rng = np.random.default_rng(n)
    x = np.cumsum(rng.uniform(0.3, 2.0, n))
    y = rng.standard_normal(n)
    s = M.Spline(x, y)
    if n >= 4:
        cs = CubicSpline(x, y, bc_type="not-a-knot")
    else:
        cs = np.poly1d(np.polyfit(x, y, n - 1))
    t = np.concatenate([x, np.linspace(x[0] - 1.0, x[-1] + 1.0, 41)])
    for tt in t:
        v = s.withTwoDerivatives(tt)
        if n >= 4:
            ref = [cs(tt), cs(tt, 1), cs(tt, 2)]
        else:
            ref = [cs(tt), cs.deriv(1)(tt), cs.deriv(2)(tt) if n > 2 else 0.0]
        assert np.allclose(v, ref, rtol=1e-9, atol=1e-9), (tt, v, ref)


def test_spline_reproduces_the_reference_points():
    """FS:103-114: the unittest's points are the not-a-knot spline through (X, Y0) at X + 0.5 (8 printed digits)."""
    assert np.max(np.abs(M.Spline(X, Y0)(POINTS[:, 0]) - POINTS[:, 1])) < 1e-8


def test_residual_function_quirks():
    v = Y0 + 0.1
    r0 = M.fit_spline_residuals(POINTS, X, 0.0, v)
    assert r0.size == 11 and r0[-1] == 0.0                                     # FS:62, FS:83: m = k + 1, penalty 0
    assert np.allclose(r0[:10], M.Spline(X, v)(POINTS[:, 0]) - POINTS[:, 1])
    r1 = M.fit_spline_residuals(POINTS, X, 1e-3, v)
    assert r1.size == 10 and np.array_equal(r1[:9], r0[:9])                    # last point replaced by the penalty
    d = M.Spline(X, v).derivatives                                             # FIRST derivatives at the knots
    integral = sum((d[i] ** 2 + d[i] * d[i - 1] + d[i - 1] ** 2) * (X[i] - X[i - 1]) for i in range(1, 10))
    assert np.isclose(r1[-1], np.sqrt(integral * 1e-3 * 10 / 30), rtol=1e-14)


@pytest.mark.parametrize("lam,expect", [(0.0, Y0), (1e-3, Y_LAMBDA)])
def test_oracle_reproduces_reference_fit_spline_unittest(oracle, lam, expect):
    """The oracle (LM + BOXCQP restatement) minimising the library's fitSpline residuals lands on the reference's
    asserted spline values (FS:119-120, FS:140-141)."""
    def f(v, y):
        y[:] = M.fit_spline_residuals(POINTS, X, lam, v)
    m = 10 + (1 if lam == 0 else 0)
    res, v = oracle.optimize(f, m, np.zeros(10))
    assert res.status >= 0
    assert approx(v, expect), np.abs(v - expect).max()


@pytest.mark.gpu
@pytest.mark.parametrize("lam,expect", [(0.0, Y0), (1e-3, Y_LAMBDA)])
def test_fit_spline_reference_unittest_on_gpu(oracle, lam, expect):
    inf = np.full(10, np.inf)
    r = M.fitSpline(M.LeastSquaresSettings(), POINTS, X, -inf, inf, lam)
    got = np.array([r.spline(t) for t in X])
    assert approx(got, expect), np.abs(got - expect).max()                     # FS:119-120, FS:140-141

    def f(v, y):
        y[:] = M.fit_spline_residuals(POINTS, X, lam, v)
    ro, vo = oracle.optimize(f, 10 + (1 if lam == 0 else 0), np.zeros(10))
    lr = r.leastSquaresResult
    assert int(lr.status) == ro.status and np.allclose(r.spline.values, vo, rtol=1e-6, atol=1e-8)
    assert np.isclose(lr.residual, ro.residual, rtol=1e-9, atol=1e-12)
    if lam == 0:                    # FS:122: "this case sensetive for numeric noise" -- the tail of the lambda fit is
        assert abs(lr.iterations - ro.iterations) <= 2


@pytest.mark.gpu
def test_fit_spline_bounds_and_errors(oracle):
    inf = np.full(10, np.inf)
    lo = np.full(10, 0.0)           # the start y = 0 (FS:56-57) has to be inside the box, else badBounds
    up = np.full(10, 15.0)          # forces BOXCQP: five unconstrained values are above 15
    r = M.fitSpline(None, POINTS, X, lo, up, 1e-3)
    v = r.spline.values
    assert np.all(v >= 0.0) and np.all(v <= 15.0) and np.sum(v == 15.0) >= 3

    def f(vv, y):
        y[:] = M.fit_spline_residuals(POINTS, X, 1e-3, vv)
    ro, vo = oracle.optimize(f, 10, np.zeros(10), lower=lo, upper=up)
    assert np.allclose(v, vo, rtol=1e-6, atol=1e-8)
    with pytest.raises(Exception, match="greater or equal"):                   # FS:47-51
        M.fitSpline(None, POINTS[:5], X, -inf, inf, 0.0)
    # fewer points are fine with lambda > 0 (FS:47 only fires for lambda == 0). The fit itself is rank deficient
    # (m = 5 < n = 10) and its LM trajectory chaotic: the oracle wanders for maxIterations = 1000 passes and ends at
    # residual ~1.137; the GPU path ends in the same basin, with `maxIterations` (LS:175-179 turns it into the
    # exception) or, when a 1e-14 difference takes it elsewhere, with a non-negative status
    def f5(vv, y):
        y[:] = M.fit_spline_residuals(POINTS[:5], X, 1e-2, vv)
    ro5, _ = oracle.optimize(f5, 5, np.zeros(10))
    try:
        res5 = M.fitSpline(None, POINTS[:5], X, -inf, inf, 1e-2).leastSquaresResult
    except M.LeastSquaresException as e:
        res5 = e.result
    assert ro5.status == M.LeastSquaresStatus.maxIterations and ro5.iterations == 1000
    assert int(res5.status) >= int(M.LeastSquaresStatus.maxIterations)          # maxIterations (-1) or a converged status
    assert 1.0 < res5.residual < 1.4 and 1.0 < ro5.residual < 1.4
    with pytest.raises(M.LeastSquaresException, match="[Bb]ound"):             # y = 0 outside [1, 15]: badBounds (LS:175-179)
        M.fitSpline(None, POINTS, X, lo + 1.0, up, 0.0)
    rf = M.fitSpline(M.LeastSquaresSettings(np.float32), POINTS, X, -inf, inf, 0.0, dtype=np.float32)
    assert np.allclose(rf.spline.values, Y0, atol=0.05)
