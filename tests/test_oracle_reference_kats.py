"""ORACLE PIN: the 8 known-answer unittests the reference holds for this path
(/root/reference/source/mir/optim/least_squares.d:217-434 T1-T6, boxcqp.d:382-402 TQ).
The assertions are the reference's own (same tolerances)."""
import numpy as np

import problems as P


def run(oracle, p, **kw):
    return oracle.optimize(p["f"], p["m"], p["x0"], lower=p["lower"], upper=p["upper"], g=p["g"], **kw)


def test_T1_with_jacobian(oracle):
    p = P.t1()
    res, x = run(oracle, p)
    assert np.linalg.norm(x - p["expect"]) < 1e-8              # LS:244
    assert oracle.STATUS[res.status] == "fConverged"
    assert (res.iterations, res.fCalls, res.gCalls) == (5, 6, 2)


def test_T2_rosenbrock_finite_difference(oracle):
    p = P.t2()
    res, x = run(oracle, p)
    assert np.linalg.norm(x - p["expect"]) < 1e-6              # LS:272
    assert res.status >= 0
    assert (res.iterations, res.fCalls, res.gCalls) == (19, 38, 0)


def test_T3a_rosenbrock_analytic(oracle):
    p = P.t3a()
    res, x = run(oracle, p)
    assert np.linalg.norm(x - p["expect"]) < 1e-8              # LS:317
    assert (res.iterations, res.fCalls, res.gCalls) == (18, 29, 5)


def test_T3b_rosenbrock_bounded(oracle):
    p = P.t3b()
    res, x = run(oracle, p)
    assert np.linalg.norm(x - p["expect"]) < 1e-5              # LS:329
    assert np.all(x >= 10)                                      # LS:330
    assert res.status >= 0 and res.iterations == 20 and abs(res.residual - 81.0) < 1e-9


def test_T4_exp_fit(oracle):
    p = P.t4()
    res, x = run(oracle, p)
    assert np.linalg.norm(x - p["expect"]) < 0.05              # LS:362
    assert res.status >= 0


def test_T5_bounded_exp_fit(oracle):
    a, b = P.t5()
    res, x = run(oracle, a)
    assert np.all(x >= np.array(a["lower"])) and res.status >= 0    # LS:393
    res, x = run(oracle, b)
    assert np.all(x <= np.array(b["upper"])) and res.status >= 0    # LS:407


def test_T6_underdetermined_bounded(oracle):
    p = P.t6()
    res, x = run(oracle, p)
    assert np.linalg.norm(x - p["upper"]) < 1e-8               # LS:433
    assert res.iterations == 1 and abs(res.residual - 0.5) < 1e-12


def test_TQ_boxcqp(oracle):
    p = P.tq()
    st, x, iters = oracle.solve_box_qp(p["P"], p["q"], p["l"], p["u"])
    assert st == 0
    assert np.allclose(x, p["expect"], rtol=1e-2, atol=1e-5)     # approxEqual, QP:401
    assert np.allclose(x, p["expect"], rtol=1e-14) and iters == 1


def test_validation_codes(oracle):
    """LS:930-943 (quirk Q9): order and codes of the argument checks."""
    f = P.rosenbrock_f
    st = lambda **kw: oracle.STATUS[oracle.optimize(f, kw.pop("m", 2), kw.pop("x0", [0.0, 0.0]), **kw)[0].status]
    assert st(x0=[np.nan, 0.0]) == "badGuess"
    assert st(x0=[np.inf, 0.0]) == "badGuess"
    assert st(m=0) == "badGuess"
    assert st(lower=[1.0, -1.0], upper=[2.0, 2.0]) == "badBounds"
    for field, val, code in [("minStepQuality", 1.0, "badMinStepQuality"), ("minStepQuality", -0.1, "badMinStepQuality"),
                             ("goodStepQuality", 1.5, "badGoodStepQuality"), ("goodStepQuality", 0.05, "badStepQuality"),
                             ("lambdaIncrease", 0.5, "badLambdaParams"), ("lambdaDecrease", 2.0, "badLambdaParams")]:
        s = oracle.default_settings()
        setattr(s, field, val)
        assert st(settings=s) == code, field
    res, _ = oracle.optimize(f, 2, [np.nan, 0.0])
    assert res.residual == np.inf and res.lambda_ == 0 and res.iterations == 0   # LS:132-142 defaults


def test_defaults_and_lengths(oracle):
    """LS:93-122, QP:62-70 (quirk Q10), LS:642-656, QP:36-50."""
    s = oracle.default_settings()
    eps = np.finfo(np.float64).eps
    assert s.maxIterations == 1000 and s.maxAge == 0
    assert s.jacobianEpsilon == 2.0 ** -26
    assert s.absTolerance == eps and s.relTolerance == 0 and s.gradTolerance == eps
    assert s.maxGoodResidual == eps ** 2
    assert s.maxStep == np.sqrt(np.finfo(np.float64).max) / 16
    assert s.maxLambda == np.finfo(np.float64).max / 16
    assert s.minLambda == np.finfo(np.float64).tiny * 16
    assert (s.minStepQuality, s.goodStepQuality, s.lambdaIncrease) == (0.1, 0.5, 2.0)
    assert s.lambdaDecrease == 0.30901699437494745
    assert s.qpSettings.relTolerance == 16 * eps and s.qpSettings.absTolerance == 16 * eps and s.qpSettings.maxIterations == 0
    sf = oracle.default_settings(np.float32)
    assert sf.jacobianEpsilon == 2.0 ** -11 and sf.absTolerance == np.finfo(np.float32).eps
    L = oracle.lib()
    for m, n in [(2, 2), (100, 3), (1000000, 128), (5, 7)]:
        assert L.lmo_box_qp_work_length(n) == 2 * n * n + 8 * n
        assert L.lmo_box_qp_iwork_length(n) == n + (n + 3) // 4
        assert L.lmo_work_length(m, n) == 2 * n * n + 8 * n + 5 * n + n * n + n * m + 2 * m
        assert L.lmo_iwork_length(m, n) == max(n + (n + 3) // 4, n)
    assert L.lmo_status_string(-26) == b"Numeric Error"
    assert L.lmo_status_string(3) == b"Residual is small enough"
