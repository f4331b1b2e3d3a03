"""Edge cases of the path, GPU vs oracle through the reference C entry point (host callbacks):
every exit status of LS:20-46 that the loop can produce, collapsed finite-difference intervals (LS:1033-1046),
degenerate shapes, NaN handling (LS:990, 1087, 1117), the step-size guard (LS:1101), maxAge handling (LS:945)."""
import numpy as np
import pytest

import mir_optim_amd as M

pytestmark = pytest.mark.gpu
S = M.LeastSquaresStatus


def both(oracle, f, m, x0, l=None, u=None, g=None, tweak=None):
    s, so = M.LeastSquaresSettings(), oracle.default_settings()
    if tweak:
        for k, v in tweak.items():
            setattr(s, k, v); setattr(so, k, v)
    # x is updated IN PLACE by the API (like the reference): give each solver its own copy of the start
    res, x = M.optimizeLeastSquares(f, m, np.array(x0, dtype=np.float64), l, u, g=g, settings=s)
    ro, xo = oracle.optimize(f, m, np.array(x0, dtype=np.float64), lower=l, upper=u, g=g, settings=so)
    return res, x, ro, xo


def lin(A, b):
    def f(x, y):
        y[:] = A @ x - b

    def g(x, J):
        J[:] = A
    return f, g


def test_fConverged_and_gConverged(oracle):
    rng = np.random.default_rng(0)
    A = rng.standard_normal((12, 3)); xs = np.array([1.0, -2.0, 0.5])
    f, g = lin(A, A @ xs)                                   # zero residual at xs -> fConverged
    res, x, ro, xo = both(oracle, f, 12, [0.0, 0.0, 0.0], g=g)
    assert res.status == S.fConverged == ro.status and np.allclose(x, xs, atol=1e-9)
    assert (res.iterations, res.fCalls, res.gCalls) == (ro.iterations, ro.fCalls, ro.gCalls)
    b = A @ xs + rng.standard_normal(12)                    # start AT the least-squares solution -> gradient ~ 0
    xls = np.linalg.lstsq(A, b, rcond=None)[0]
    f, g = lin(A, b)
    res, x, ro, xo = both(oracle, f, 12, xls, g=g, tweak=dict(gradTolerance=1e-10))
    assert res.status == S.gConverged == ro.status and res.iterations == ro.iterations == 0
    assert res.lambda_ == ro.lambda_ == 0.0                 # the gradient test precedes lambda_0 (LS:1053 vs 1067)


def test_xConverged_and_maxIterations(oracle):
    rng = np.random.default_rng(1)
    A = rng.standard_normal((30, 4)); b = rng.standard_normal(30)
    f, g = lin(A, b)
    res, x, ro, xo = both(oracle, f, 30, np.zeros(4), g=g, tweak=dict(absTolerance=1e-7))
    assert res.status >= 0 and ro.status >= 0 and np.allclose(x, xo, rtol=1e-6, atol=1e-9)
    res, x, ro, xo = both(oracle, f, 30, np.zeros(4), g=g, tweak=dict(maxIterations=1))
    assert res.status == S.maxIterations == ro.status and res.iterations == ro.iterations == 1
    assert np.allclose(x, xo, rtol=1e-9)


def test_nan_residual_is_numeric_error(oracle):
    def f(x, y):
        y[0] = x[0] - 1
        y[1] = np.nan if x[0] < 5 else x[1]
    res, x, ro, xo = both(oracle, f, 2, [10.0, 1.0])
    assert res.status == ro.status
    assert res.status in (S.numericError, S.furtherImprovement)
    # NaN already at the first evaluation: the loop sees residual = NaN (LS:955) and lambda <= maxLambda etc.
    def f2(x, y):
        y[:] = np.nan
    res, x, ro, xo = both(oracle, f2, 2, [1.0, 1.0])
    assert int(res.status) == ro.status


def test_collapsed_fd_interval_and_fixed_parameter(oracle):
    """lower == upper for one parameter: twh == 0 -> zero Jacobian column, no residual calls for it (LS:1033-1046)."""
    rng = np.random.default_rng(2)
    t = np.linspace(0, 1, 40); data = 2.0 * np.exp(-1.5 * t) + 0.3 + 0.01 * rng.standard_normal(40)
    calls = []

    def f(p, y):
        calls.append(p.copy())
        y[:] = p[0] * np.exp(-p[1] * t) + p[2] - data
    l, u = [-np.inf, 1.5, -np.inf], [np.inf, 1.5, np.inf]
    res, x, ro, xo = both(oracle, f, 40, [1.0, 1.5, 0.0], l, u)
    assert x[1] == 1.5 and res.status >= 0 and ro.status >= 0
    assert np.allclose(x, xo, rtol=1e-6, atol=1e-9) and np.isclose(res.residual, ro.residual, rtol=1e-9)
    assert all(c[1] == 1.5 for c in calls)            # the collapsed column is never perturbed (LS:1033: no call when twh == 0)
    # fCalls: the fit ends in the reference's rejection tail (quirk Q3) at a residual that moves in its 16th digit; whether the
    # last candidate step is accepted (one more refresh + a second tail) is a rounding-level decision (scripts/dbg_collapsed.py
    # prints both traces: identical events until that pass) -- the two known traces: the same counters, or exactly ONE such episode
    # more on one side (a refresh: n = 3 calls, LS:1049, + a tail of at most 16 rejected trials)
    d = abs(int(res.fCalls) - int(ro.fCalls))
    assert d == 0 or 3 <= d <= 3 + 16, (res.fCalls, ro.fCalls)


def test_degenerate_shapes(oracle):
    def f11(x, y):
        y[0] = x[0] * x[0] - 2.0
    res, x, ro, xo = both(oracle, f11, 1, [1.0])             # m = n = 1
    assert res.status >= 0 and abs(x[0] - np.sqrt(2.0)) < 1e-7 and np.allclose(x, xo, rtol=1e-9)

    def f13(x, y):                                           # m = 1 < n = 3
        y[0] = x[0] + 2 * x[1] - 3 * x[2] - 1.0
    res, x, ro, xo = both(oracle, f13, 1, [0.0, 0.0, 0.0])
    assert res.status >= 0 and abs(x[0] + 2 * x[1] - 3 * x[2] - 1.0) < 1e-7 and np.allclose(x, xo, rtol=1e-6, atol=1e-9)

    rng = np.random.default_rng(3)                           # n = 17 (not a multiple of 16), m = 33 odd
    A = rng.standard_normal((33, 17)); b = rng.standard_normal(33)
    f, _ = lin(A, b)
    res, x, ro, xo = both(oracle, f, 33, np.zeros(17))
    assert res.status >= 0 and np.allclose(x, np.linalg.lstsq(A, b, rcond=None)[0], rtol=1e-6, atol=1e-8)
    assert np.allclose(x, xo, rtol=1e-6, atol=1e-9)


def test_step_guard_and_lambda_schedule(oracle):
    """maxStep tiny: every step trips LS:1101 until lambda has grown enough; lambda/mu follow LS:1103-1104."""
    rng = np.random.default_rng(4)
    A = rng.standard_normal((20, 3)); b = 10 * rng.standard_normal(20)
    f, g = lin(A, b)
    res, x, ro, xo = both(oracle, f, 20, np.zeros(3), g=g, tweak=dict(maxStep=1e-3))
    assert int(res.status) == ro.status
    assert (res.iterations, res.fCalls, res.gCalls) == (ro.iterations, ro.fCalls, ro.gCalls)
    assert np.allclose(x, xo, rtol=1e-9, atol=1e-12) and np.isclose(res.lambda_, ro.lambda_, rtol=1e-9)


def test_max_age_settings(oracle):
    """maxAge = 1: every second Jacobian is a full refresh (LS:945, 999-1010); counters must agree."""
    from problems import rosenbrock_f
    for age in (1, 2, 5):
        res, x, ro, xo = both(oracle, rosenbrock_f, 2, [-1.2, 1.0], tweak=dict(maxAge=age))
        assert np.linalg.norm(x - [1.0, 1.0]) < 1e-6 and np.linalg.norm(xo - [1.0, 1.0]) < 1e-6
        assert res.status >= 0 and ro.status >= 0
        assert abs(int(res.fCalls) - int(ro.fCalls)) <= 8


def test_x0_on_bounds_and_mixed_infinite_bounds(oracle):
    rng = np.random.default_rng(5)
    A = rng.standard_normal((25, 5)); b = rng.standard_normal(25)
    f, g = lin(A, b)
    l = np.array([-np.inf, 0.0, -0.1, -np.inf, 0.2]); u = np.array([0.05, np.inf, 0.1, np.inf, 0.2])
    x0 = np.array([0.05, 0.0, 0.1, 3.0, 0.2])                # several components start exactly on a bound
    res, x, ro, xo = both(oracle, f, 25, x0, l, u, g=g)
    from scipy.optimize import lsq_linear
    free = u > l                                             # x[4] is pinned (l == u): solve the rest independently
    ref = np.array(l, dtype=float)
    ref[free] = lsq_linear(A[:, free], b - A[:, ~free] @ l[~free], bounds=(l[free], u[free]), tol=1e-14).x
    assert res.status >= 0 and np.all(x >= l) and np.all(x <= u) and x[4] == 0.2
    assert np.allclose(x, xo, rtol=1e-6, atol=1e-9) and np.allclose(x, ref, rtol=1e-5, atol=1e-7)


def test_python_callback_exception_is_reraised_not_swallowed():
    """an exception inside a ctypes callback used to be printed and swallowed, the solve continued on
    garbage. Now the output is poisoned with NaN (the solver stops with numericError at its next check) and the exception
    is re-raised when the C call returns."""
    calls = {"n": 0}

    def f(x, y):
        calls["n"] += 1
        if calls["n"] == 4:
            raise ValueError("boom in the residual callback")
        y[0] = 10 * (x[1] - x[0] ** 2); y[1] = 1 - x[0]
    with pytest.raises(ValueError, match="boom"):
        M.optimizeLeastSquares(f, 2, np.array([-1.2, 1.0]))
    assert calls["n"] < 40                                     # it did not run the whole solve on garbage
