"""GPU parity of the whole LM path through the C ABI against the oracle and the reference's own
known answers (T1-T6, LS:217-434).

  * host-callback entry mir_optimize_least_squares_d (the reference signature, LS:705-724):
    the reference unittests T1-T6, incl. a thread manager (T2), analytic Jacobians, bounds.
  * device-callback entry mir_optimize_least_squares_gpu_d: synthetic families of SURVEY 8d at
    sizes the oracle finishes in seconds.
Tolerances: x rtol 1e-6 (north star), residual rtol 1e-9, same status class.
Counters are compared on the well-conditioned reference cases only."""
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

import mir_optim_amd as M
from mir_optim_amd import workloads as W
import problems as P

pytestmark = pytest.mark.gpu


def run_host(p, **kw):
    return M.optimize(p["f"], p["m"], p["x0"], p["lower"], p["upper"], g=p["g"], **kw)


def run_oracle(oracle, p, **kw):
    return oracle.optimize(p["f"], p["m"], p["x0"], lower=p["lower"], upper=p["upper"], g=p["g"], **kw)


def same_class(a, b):
    return (int(a) >= 0) == (int(b) >= 0)


def first_noisy_pass(recs):
    """Index of the first pass whose trial residual differs from the current one by less than ~1e-10 relative: from
    there on `improvement > 0` (LS:1125) and the step quality rho (LS:1150) compare rounding noise of two different
    summation orders and two runs may legitimately branch differently. len(recs) if there is none."""
    prev = None
    for k, r in enumerate(recs):
        if r[0] in (2, 3):
            before = r[3] if r[0] == 2 else prev
            if before is not None and abs(before - r[4]) <= 1e-10 * abs(before):
                return k
        prev = r[3]
    return len(recs)


def test_T1_with_jacobian(oracle):
    p = P.t1()
    res, x = run_host(p)
    ro, xo = run_oracle(oracle, p)
    assert np.linalg.norm(x - p["expect"]) < 1e-8                    # LS:244
    assert int(res.status) == ro.status == M.LeastSquaresStatus.fConverged
    assert (res.iterations, res.fCalls, res.gCalls) == (ro.iterations, ro.fCalls, ro.gCalls) == (5, 6, 2)


def test_T2_rosenbrock_fd_with_task_pool(oracle):
    p = P.t2()
    with ThreadPoolExecutor(max_workers=3) as pool:                  # LS:266: taskPool overload
        res, x = M.optimize(p["f"], 2, p["x0"], taskPool=pool)
    assert np.linalg.norm(x - p["expect"]) < 1e-6                    # LS:272
    res2, x2 = run_host(p)                                           # serial thread manager LS:947-951
    ro, xo = run_oracle(oracle, p)
    assert np.linalg.norm(x2 - p["expect"]) < 1e-6
    assert (res2.iterations, res2.fCalls) == (ro.iterations, ro.fCalls) == (19, 38)
    assert (res.iterations, res.fCalls) == (19, 38)


def test_T3a_rosenbrock_analytic(oracle):
    p = P.t3a()
    res, x = run_host(p)
    ro, _ = run_oracle(oracle, p)
    assert np.linalg.norm(x - p["expect"]) < 1e-8                    # LS:317
    assert (res.iterations, res.fCalls, res.gCalls) == (ro.iterations, ro.fCalls, ro.gCalls) == (18, 29, 5)


def test_T3b_rosenbrock_bounded(oracle):
    p = P.t3b()
    res, x = run_host(p)
    ro, xo = run_oracle(oracle, p)
    assert np.linalg.norm(x - p["expect"]) < 1e-5                    # LS:329
    assert np.all(x >= 10)                                           # LS:330
    assert same_class(res.status, ro.status) and abs(res.residual - 81.0) < 1e-9
    assert np.allclose(x, xo, rtol=1e-6)


def test_T4_exp_fit(oracle):
    p = P.t4()
    res, x = run_host(p)
    ro, xo = run_oracle(oracle, p)
    assert np.linalg.norm(x - p["expect"]) < 0.05                    # LS:362
    assert np.allclose(x, xo, rtol=1e-6) and np.isclose(res.residual, ro.residual, rtol=1e-9)
    assert same_class(res.status, ro.status)


def test_T5_bounded_exp_fit(oracle):
    a, b = P.t5()
    for p, key in ((a, "lower"), (b, "upper")):
        res, x = run_host(p)
        ro, xo = run_oracle(oracle, p)
        if key == "lower":
            assert np.all(x >= np.array(p["lower"]))                 # LS:393
        else:
            assert np.all(x <= np.array(p["upper"]))                 # LS:407
        assert np.allclose(x, xo, rtol=1e-6) and np.isclose(res.residual, ro.residual, rtol=1e-9)


def test_T6_underdetermined_bounded(oracle):
    p = P.t6()
    res, x = run_host(p)
    assert np.linalg.norm(x - p["upper"]) < 1e-8                     # LS:433
    assert res.iterations == 1 and abs(res.residual - 0.5) < 1e-12


@pytest.mark.parametrize("name", ["t1", "t2", "t3a", "t3b", "t4", "t6"])
def test_reference_unittests_trace_equals_oracle_trace(oracle, name):
    """Host-callback mode on the reference's own unittest problems (m <= 20, n <= 3): here the GPU path and the oracle
    run the same few flops per reduction, so the complete trace -- every event, counter and lambda; 51 rejected passes
    for T3b -- must be the same. Values agree to ~1e-14 until the first finite-difference refresh after a step and to
    ~1e-8 after it: with the reference's absolute step 2^-26 (Q10) the FD Jacobian carries eps |f| / h ~ 1e-8 of
    rounding noise that any 1e-14 change of x re-draws."""
    p = getattr(P, name)()
    tr = M.Trace()
    opt = M.GpuOptions()
    import ctypes as C
    opt.trace = C.pointer(tr.header)
    res, x = M.optimizeLeastSquares(p["f"], p["m"], np.array(p["x0"], dtype=float), p["lower"], p["upper"], g=p["g"], options=opt)
    ev = []
    ro, xo = run_oracle(oracle, p, trace=lambda *a: ev.append(a))
    got = tr.records()
    assert tr.count == len(got)
    if name in ("t1", "t2", "t3a", "t3b", "t6"):
        assert int(res.status) == ro.status and (res.iterations, res.fCalls, res.gCalls) == (ro.iterations, ro.fCalls, ro.gCalls)
        assert len(got) == len(ev)
    else:       # noisy data, nonzero residual: the tail of the fit decides on rounding noise (see first_noisy_pass)
        K = min(first_noisy_pass(got), first_noisy_pass(ev), len(got), len(ev))
        assert K >= 5 or K == len(ev) == len(got), (K, len(got), len(ev))
        got, ev = got[:K], ev[:K]
        assert res.status >= 0 and ro.status >= 0 and np.allclose(x, xo, rtol=1e-6, atol=1e-9)
    assert [g[:2] for g in got] == [e[:2] for e in ev]
    for g, e in zip(got, ev):
        assert np.isclose(g[2], e[2], rtol=1e-12), (g, e)                # lambda: products of exact constants
        # residual and trial residual: 1e-7 (measured <= 6e-9, scripts/trace_diff.py; round 1 allowed 2e-5, which hid a staging
        # defect of 1.5e-8 -- tests/test_gpu_host_fd_staging.py); atol: noise floor of a zero-residual fit.
        assert np.allclose(g[3:5], e[3:5], rtol=1e-7, atol=1e-10 * (1 + ev[0][3])), (g, e)
        # dx.dx of the tiny steps at the end of a noisy fit is itself a finite-difference-noise quantity (measured 5e-5 on T4)
        assert np.isclose(g[5], e[5], rtol=1e-3, atol=1e-10 * (1 + ev[0][3])), (g, e)


def test_nothrow_tier_and_exception(oracle):
    """optimize throws for status < 0 (LS:175-179), optimizeLeastSquares returns the status."""
    s = M.LeastSquaresSettings()
    s.maxIterations = 3
    p = P.t3a()
    res, _ = M.optimizeLeastSquares(p["f"], 2, p["x0"], g=p["g"], settings=s)
    so = oracle.default_settings(); so.maxIterations = 3
    ro, _ = run_oracle(oracle, p, settings=so)
    assert res.status == M.LeastSquaresStatus.maxIterations == ro.status and res.iterations == 3
    with pytest.raises(M.LeastSquaresException) as ei:
        M.optimize(p["f"], 2, p["x0"], g=p["g"], settings=s)
    assert "Maximum number of iterations reached" in str(ei.value)


def test_float_entry_uses_callers_m(oracle):
    """The reference's float entry passes the literal 2 for m (LS:629, quirk Q7); here m is honoured."""
    p = P.t4()
    t, yd = p["t"].astype(np.float32), p["data"].astype(np.float32)

    def f(q, y):
        y[:] = q[0] * np.exp(-t * q[1]) - yd
    res, x = M.optimize(f, 20, [0.5, 0.5], dtype=np.float32)
    ro, xo = oracle.optimize(f, 20, [0.5, 0.5], dtype=np.float32)
    assert np.linalg.norm(x - [1.0, 2.0]) < 0.05
    assert np.allclose(x, xo, rtol=2e-3) and same_class(res.status, ro.status)


# ---------------------------------------------------------------- device callbacks, synthetic families
def oracle_tanh(oracle, w, settings, lower=None, upper=None, analytic=False):
    import ctypes as C
    ctx = oracle.TanhLinearCtx(w["A"].ctypes.data, w["b"].ctypes.data)
    return oracle.optimize(oracle.native_fn("wlc_tanh_linear_f"), w["m"], w["x0"], lower=lower, upper=upper,
                           settings=settings, fctx=C.addressof(ctx),
                           g=oracle.native_fn("wlc_tanh_linear_g") if analytic else None, gctx=C.addressof(ctx))


@pytest.mark.parametrize("m,n", [(64, 4), (512, 8), (4096, 16), (20000, 32), (30000, 64), (50000, 128), (9973, 100),
                                 (20000, 160), (30000, 256)])
def test_tanh_linear_fd_matches_oracle(oracle, m, n):
    w = P.tanh_linear(m, n)
    prob = W.TanhLinear(w["A"], w["b"])
    s = M.LeastSquaresSettings(); s.absTolerance = 1e-9
    so = oracle.default_settings(); so.absTolerance = 1e-9
    res, x = prob.solve(w["x0"], settings=s)
    ro, xo = oracle_tanh(oracle, w, so)
    assert res.status >= 0 and ro.status >= 0
    assert np.allclose(x, xo, rtol=1e-6, atol=1e-9)
    assert np.isclose(res.residual, ro.residual, rtol=1e-9)
    assert res.iterations > 2 and res.fCalls >= n


def _oracle_trace(oracle, w, so, lower=None, upper=None, analytic=False):
    ev = []
    import ctypes as C
    ctx = oracle.TanhLinearCtx(w["A"].ctypes.data, w["b"].ctypes.data)
    ro, xo = oracle.optimize(oracle.native_fn("wlc_tanh_linear_f"), w["m"], w["x0"], lower=lower, upper=upper,
                             settings=so, fctx=C.addressof(ctx), trace=lambda *a: ev.append(a),
                             g=oracle.native_fn("wlc_tanh_linear_g") if analytic else None, gctx=C.addressof(ctx))
    return ro, xo, ev


@pytest.mark.parametrize("m,n,bounded,mode", [(4096, 16, False, "fd"), (20000, 32, False, "batched"), (3000, 16, True, "fd"),
                                              (6000, 24, False, "analytic"), (50000, 128, False, "batched")])
def test_pass_by_pass_trajectory_matches_oracle(oracle, m, n, bounded, mode):
    """The reference pins no intermediate quantity; this pins the whole trajectory on the oracle: the
    SAME sequence of events (Jacobian refresh / Broyden update / step guard / rejection / acceptance), the same
    iteration counter and, pass by pass, the same damping, residuals and step lengths -- speculative lambda-ladder
    rounds included (mode "batched"), whose discarded trials must leave no record."""
    w = P.tanh_linear(m, n)
    lo = up = None
    if bounded:
        lo = w["xstar"] - 0.02
        lo[::3] = w["xstar"][::3] + 0.01
        up = w["xstar"] + 0.5
        w["x0"] = np.clip(w["x0"], lo, up)
    prob = W.TanhLinear(w["A"], w["b"])
    s = M.LeastSquaresSettings(); s.absTolerance = 1e-9
    so = oracle.default_settings(); so.absTolerance = 1e-9
    tr = M.Trace()
    res, x = prob.solve(w["x0"].copy(), lo, up, settings=s, trace=tr, batched=(mode == "batched"), analytic=(mode == "analytic"))
    ro, xo, ev = _oracle_trace(oracle, w, so, lo, up, analytic=(mode == "analytic"))
    got = tr.records()
    assert tr.count == len(got)

    # everything before the first noise-level decision must agree; after it only the end result is compared
    K = min(first_noisy_pass(got), first_noisy_pass(ev), len(got), len(ev))
    assert K >= 9, (K, len(got), len(ev))
    assert [g[0] for g in got[:K]] == [e[0] for e in ev[:K]]            # event kinds
    assert [g[1] for g in got[:K]] == [e[1] for e in ev[:K]]            # iteration counter
    kinds = {g[0] for g in got[:K]}
    assert {0, 1, 3} <= kinds
    for k, (g, e) in enumerate(zip(got[:K], ev[:K])):
        assert np.isclose(g[2], e[2], rtol=1e-8), ("lambda", k, g, e)          # measured <= 5e-10 (scripts/trace_diff.py)
        assert np.isclose(g[3], e[3], rtol=1e-7), ("residual", k, g, e)
        assert np.isclose(g[4], e[4], rtol=1e-7, atol=1e-300), ("trial residual", k, g, e)
        assert np.isclose(g[5], e[5], rtol=1e-3, atol=1e-22), ("dx_dot", k, g, e)
    assert res.status >= 0 and ro.status >= 0
    assert np.allclose(x, xo, rtol=1e-6, atol=1e-9) and np.isclose(res.residual, ro.residual, rtol=1e-9)


def test_tanh_linear_analytic_and_batched(oracle):
    w = P.tanh_linear(6000, 24)
    prob = W.TanhLinear(w["A"], w["b"])
    s = M.LeastSquaresSettings(); s.absTolerance = 1e-9
    so = oracle.default_settings(); so.absTolerance = 1e-9
    res, x = prob.solve(w["x0"], settings=s, analytic=True)
    ro, xo = oracle_tanh(oracle, w, so, analytic=True)
    assert res.gCalls > 0 and res.status >= 0
    assert np.allclose(x, xo, rtol=1e-6, atol=1e-9) and np.isclose(res.residual, ro.residual, rtol=1e-9)
    # batched FD (one MFMA sweep for the 2n perturbed points) vs one residual call per point: same
    # minimiser; the two differ only by the summation order inside the user's residual kernel
    resb, xb = prob.solve(w["x0"], settings=s, batched=True)
    res1, x1 = prob.solve(w["x0"], settings=s)
    ro1, xo1 = oracle_tanh(oracle, w, so)
    assert resb.status >= 0 and res1.status >= 0
    assert np.allclose(xb, x1, rtol=1e-8, atol=1e-11) and np.allclose(xb, xo1, rtol=1e-6, atol=1e-9)
    assert np.isclose(resb.residual, ro1.residual, rtol=1e-9)


def test_tanh_linear_bounded_hits_boxcqp(oracle):
    w = P.tanh_linear(3000, 16)
    lo = w["xstar"] - 0.02
    lo[::3] = w["xstar"][::3] + 0.01          # optimum of every third parameter sits ON its lower bound
    up = w["xstar"] + 0.5
    w = dict(w, x0=np.clip(w["x0"], lo, up))
    prob = W.TanhLinear(w["A"], w["b"])
    s = M.LeastSquaresSettings(); s.absTolerance = 1e-9
    so = oracle.default_settings(); so.absTolerance = 1e-9
    st = M.Stats()
    res, x = prob.solve(w["x0"], lo, up, settings=s, stats=st)
    ro, xo = oracle_tanh(oracle, w, so, lower=lo, upper=up)
    assert st.qp_active_set_passes > 0                                # the active-set loop really ran on the device
    assert np.all(x >= lo) and np.all(x <= up) and np.sum(x == lo) >= len(lo) // 3
    assert np.allclose(x, xo, rtol=1e-6, atol=1e-9) and np.isclose(res.residual, ro.residual, rtol=1e-9)
    assert same_class(res.status, ro.status)


def test_gauss_sum_cfg2_family(oracle):
    import ctypes as C
    g = P.gauss_sum(20000, K=3)
    prob = W.Curve("gauss_sum", g["t"], g["data"])
    res, x = prob.solve(g["x0"], g["lower"], g["upper"])
    ctx = oracle.GaussSumCtx(g["t"].ctypes.data, g["data"].ctypes.data)
    ro, xo = oracle.optimize(oracle.native_fn("wlc_gauss_sum_f"), g["m"], g["x0"], lower=g["lower"], upper=g["upper"],
                             fctx=C.addressof(ctx))
    assert res.status >= 0 and ro.status >= 0
    assert np.allclose(x, xo, rtol=1e-6, atol=1e-9) and np.isclose(res.residual, ro.residual, rtol=1e-8)
    assert np.allclose(x, g["truth"], rtol=5e-3, atol=1e-3)


@pytest.mark.parametrize("m,n", [(1000, 4), (5000, 16), (3001, 30), (20000, 64), (40000, 128), (777, 100),
                                 (4000, 32), (40001, 128), (50, 64), (300001, 128), (70000, 32), (20000, 256), (4099, 256)])
def test_batched_residual_callback_matches_pointwise(m, n):
    """workloads.hip: the batched MFMA residual kernel == the per-point kernel (user-code side)."""
    import ctypes as C
    from mir_optim_amd import api
    w = P.tanh_linear(m, n)
    prob = W.TanhLinear(w["A"], w["b"])
    rng = np.random.default_rng(n)
    for p in (1, 2 * n, 2 * n + 3):
        X = w["xstar"][None, :] + 0.1 * rng.standard_normal((p, n))
        dX = api.DeviceBuffer(X)
        dYb = api.DeviceBuffer(np.zeros((p, m)))
        dY1 = api.DeviceBuffer(np.zeros((p, m)))
        WL = api.workloads_lib()
        WL.wl_tanh_linear_fb_d(C.c_void_p(C.addressof(prob.ctx)), C.c_size_t(m), C.c_size_t(n), C.c_size_t(p),
                               C.c_void_p(dX.ptr), C.c_void_p(dYb.ptr))
        for k in range(p):
            WL.wl_tanh_linear_f_d(C.c_void_p(C.addressof(prob.ctx)), C.c_size_t(m), C.c_size_t(n),
                                  C.c_void_p(dX.ptr + k * n * 8), C.c_void_p(dY1.ptr + k * m * 8))
        prob.stream.synchronize()
        Yb, Y1 = dYb.download(), dY1.download()
        ref = np.tanh(X @ w["A"].T) - w["b"][None, :]
        assert np.max(np.abs(Yb - ref)) < 1e-13 and np.max(np.abs(Y1 - ref)) < 1e-13


def assert_traces_agree_until_noise(got, ref, min_passes):
    """Pass-by-pass comparison of mir_lsq_trace with the oracle's trace up to the first pass whose accept / reject decision
    compares rounding noise (first_noisy_pass): same events and iteration counters, lambda to 1e-6, residuals to 1e-9."""
    k_end = min(first_noisy_pass(got), first_noisy_pass(ref), len(got), len(ref))
    assert k_end >= min_passes, (k_end, len(got), len(ref))
    for k in range(k_end):
        g, e = got[k], ref[k]
        assert (int(g[0]), int(g[1])) == (int(e[0]), int(e[1])), (k, g, e)
        assert np.isclose(g[2], e[2], rtol=1e-6), (k, g, e)
        assert np.allclose(g[3:5], e[3:5], rtol=1e-9, atol=1e-300), (k, g, e)
    return k_end


def test_cfg2_gauss_sum_full_size(oracle):
    """BASELINE cfg 2: Gaussian-sum curve fit, m = 1e5 residuals x n = 16 parameters, fp64, width bounds, FD Jacobian.
    The two sides end in different ways (the GPU: furtherImprovement after the reference's lambda ladder, the oracle:
    xConverged -- the last acceptance compares rounding noise, quirk Q3 / DESIGN.md section 5), so besides the minimiser
    (x rtol 1e-6, residual rtol 1e-9: BASELINE.md section 2) the TRAJECTORIES are compared pass by pass up to the first
    noise-decided pass (round-2 verdict, "what's weak" 1)."""
    import ctypes as C
    g = P.gauss_sum(100000, K=5)
    assert g["n"] == 16
    prob = W.Curve("gauss_sum", g["t"], g["data"])
    tr = M.Trace(4096)
    res, x = prob.solve(g["x0"], g["lower"], g["upper"], trace=tr)
    ctx = oracle.GaussSumCtx(g["t"].ctypes.data, g["data"].ctypes.data)
    ev = []
    ro, xo = oracle.optimize(oracle.native_fn("wlc_gauss_sum_f"), g["m"], g["x0"], lower=g["lower"], upper=g["upper"],
                             fctx=C.addressof(ctx), trace=lambda *a: ev.append(a))
    assert res.status >= 0 and ro.status >= 0
    assert np.allclose(x, xo, rtol=1e-6, atol=1e-9) and np.isclose(res.residual, ro.residual, rtol=1e-9)
    assert np.allclose(x, g["truth"], rtol=5e-3, atol=1e-3)
    k_end = assert_traces_agree_until_noise(tr.records(), ev, min_passes=30)
    accepted = sum(1 for r in tr.records()[:k_end] if r[0] == 3)
    assert accepted >= 12                                              # most of the 22-23 accepted steps are compared one by one


def test_cfg2_gauss_sum_full_size_with_binding_width_bounds(oracle):
    """cfg 2's width bounds w_k >= 1e-3 never bind on the SURVEY inputs (qp_active_set_passes = 0 in round 2's bench line), so
    "bounded cfg 2" did not exercise BOXCQP. Here two widths are boxed in ABOVE their true value and the amplitudes from above:
    the minimiser sits on those bounds, the active-set loop runs (QP:234-376) at m = 1e5 x n = 16, and the result is the
    oracle's: same active set, x rtol 1e-6, residual rtol 1e-9, traces equal up to the first noise-decided pass."""
    import ctypes as C
    g = P.gauss_sum(100000, K=5)
    K = g["K"]
    lower, upper = g["lower"].copy(), g["upper"].copy()
    lower[2 * K] = 0.045; lower[2 * K + 3] = 0.05                      # true widths are 0.04
    upper[0] = 0.95; upper[3] = 0.85                                   # true amplitudes 1.0 and 0.9
    x0 = np.clip(g["x0"], lower, upper)
    prob = W.Curve("gauss_sum", g["t"], g["data"])
    st, tr = M.Stats(), M.Trace(4096)
    res, x = prob.solve(x0, lower, upper, stats=st, trace=tr)
    ctx = oracle.GaussSumCtx(g["t"].ctypes.data, g["data"].ctypes.data)
    ev = []
    ro, xo = oracle.optimize(oracle.native_fn("wlc_gauss_sum_f"), g["m"], x0, lower=lower, upper=upper, fctx=C.addressof(ctx),
                             trace=lambda *a: ev.append(a))
    assert res.status >= 0 and ro.status >= 0
    assert st.qp_active_set_passes >= 3                                # BOXCQP's active-set loop really ran on the device
    on_gpu = (x == lower) | (x == upper)
    on_cpu = (xo == lower) | (xo == upper)
    assert on_gpu.sum() >= 4 and np.array_equal(on_gpu, on_cpu)
    assert np.allclose(x, xo, rtol=1e-6, atol=1e-9) and np.isclose(res.residual, ro.residual, rtol=1e-9)
    assert_traces_agree_until_noise(tr.records(), ev, min_passes=8)
    # The run above passes trace=, which turns the pipelining of rounds off. The DEFAULT path of this small problem enqueues
    # rounds ahead of time behind the device-side guard -- with BOXCQP active-set passes inside the speculative rounds: it must
    # give the same bits and the same counters as VARIANT_NO_PIPELINE.
    outs = []
    for variant in (0, M.VARIANT_NO_PIPELINE):
        s2 = M.Stats()
        r2, x2 = prob.solve(x0, lower, upper, stats=s2, batched=True, variant=variant)
        outs.append((x2.tobytes(), int(r2.status), r2.iterations, r2.fCalls, r2.residual, r2.lambda_, s2.passes, s2.accepted, s2.rejected,
                     s2.qp_active_set_passes, s2.jacobian_full, s2.jacobian_broyden))
    assert outs[0] == outs[1] and outs[0][9] >= 3


def test_gauss_sum_batched_callback_equals_pointwise(oracle):
    """workloads.hip: the batched Gaussian-sum callbacks (point-major and row-major) produce the per-point kernel's values,
    and the solve through them ends where the per-point solve and the oracle end."""
    import ctypes as C
    from mir_optim_amd import api
    g = P.gauss_sum(30000, K=5)
    m, n = g["m"], g["n"]
    prob = W.Curve("gauss_sum", g["t"], g["data"])
    rng = np.random.default_rng(5)
    p = 2 * n
    X = np.asarray(g["x0"])[None, :] * (1 + 0.01 * rng.standard_normal((p, n)))
    dX = api.DeviceBuffer(X)
    dYb, dYr, dY1 = (api.DeviceBuffer(np.zeros((p, m))) for _ in range(3))
    WL = api.workloads_lib()
    ctx = C.c_void_p(C.addressof(prob.ctx))
    WL.wl_gauss_sum_fb_d(ctx, C.c_size_t(m), C.c_size_t(n), C.c_size_t(p), C.c_void_p(dX.ptr), C.c_void_p(dYb.ptr))
    WL.wl_gauss_sum_fbr_d(ctx, C.c_size_t(m), C.c_size_t(n), C.c_size_t(p), C.c_void_p(dX.ptr), C.c_void_p(dYr.ptr))
    for k in range(p):
        WL.wl_gauss_sum_f_d(ctx, C.c_size_t(m), C.c_size_t(n), C.c_void_p(dX.ptr + k * n * 8), C.c_void_p(dY1.ptr + k * m * 8))
    prob.stream.synchronize()
    Y1 = dY1.download()
    assert np.array_equal(dYb.download(), Y1)
    assert np.array_equal(dYr.download().reshape(m, p).T, Y1)
    r1, x1 = prob.solve(g["x0"], g["lower"], g["upper"])
    rb, xb = prob.solve(g["x0"], g["lower"], g["upper"], batched=True)
    rp, xp = prob.solve(g["x0"], g["lower"], g["upper"], batched="pointmajor")
    octx = oracle.GaussSumCtx(g["t"].ctypes.data, g["data"].ctypes.data)
    ro, xo = oracle.optimize(oracle.native_fn("wlc_gauss_sum_f"), m, g["x0"], lower=g["lower"], upper=g["upper"], fctx=C.addressof(octx))
    for r, x in ((r1, x1), (rb, xb), (rp, xp)):
        assert r.status >= 0 and np.allclose(x, xo, rtol=1e-6, atol=1e-9) and np.isclose(r.residual, ro.residual, rtol=1e-8)


@pytest.mark.parametrize("m,n", [(60000, 128), (40000, 256), (50000, 64), (30000, 208)])
def test_repeated_solve_is_bit_reproducible(m, n):
    """The whole path is deterministic (fixed-order reductions, no float atomics): repeated solves are
    bitwise equal. m even and n % 16 == 0 select the LDS-DMA ring kernels (k_jtj2 for n <= 128, the eight-wave
    k_jtj8 above; the batched residual callback is a DMA ring kernel too); an earlier version of k_jtj2 mixed
    stores into a counted vmcnt wait and was caught by exactly this check."""
    w = P.tanh_linear(m, n)
    prob = W.TanhLinear(w["A"], w["b"])
    s = M.LeastSquaresSettings(); s.absTolerance = 1e-9
    ref = None
    for rep in range(6):
        res, x = prob.solve(w["x0"], settings=s, batched=(rep % 2 == 0))
        key = (res.iterations, res.fCalls, res.residual, x.tobytes())
        if rep < 2:
            ref = ref or {}
            ref[rep % 2] = key
        else:
            assert key == ref[rep % 2], f"solve {rep} differs from solve {rep % 2}"


@pytest.mark.parametrize("m,n", [(3000, 17), (9000, 33), (20001, 9)])
def test_speculative_lambda_ladder_is_bitwise_equivalent(m, n):
    """After a rejection the solver evaluates a ladder of up to 8 lambdas at once (batched residual callback) and
    walks them in the reference's order. With a batched callback that is numerically the same function as the
    single-point one (odd n: workloads.hip falls back to one sweep per point) the result must be bit-identical
    to the one-trial-per-pass execution: same x, residual, lambda, iterations, fCalls, pass count."""
    w = P.tanh_linear(m, n)
    prob = W.TanhLinear(w["A"], w["b"])
    s = M.LeastSquaresSettings(); s.absTolerance = 1e-12          # force the long noisy rejection tail (quirk Q3)
    st1, st0 = M.Stats(), M.Stats()
    r1, x1 = prob.solve(w["x0"], settings=s, batched=True, stats=st1, flags=M.TIME_KERNELS)
    r0, x0 = prob.solve(w["x0"], settings=s, batched=True, stats=st0, flags=M.TIME_KERNELS, variant=M.VARIANT_NO_SPECULATION)
    assert st0.rejected >= 5                                       # the ladder really had something to do
    assert np.array_equal(x1, x0) and r1.residual == r0.residual and r1.lambda_ == r0.lambda_
    assert (r1.status, r1.iterations, r1.fCalls) == (r0.status, r0.iterations, r0.fCalls)
    assert (st1.passes, st1.accepted, st1.rejected, st1.step_guard_rejects) == (st0.passes, st0.accepted, st0.rejected, st0.step_guard_rejects)
    assert st1.solve_launches < st0.solve_launches                 # fewer rounds than passes


def test_stats_and_reentrancy(oracle):
    """two different problems interleaved on their own streams/workspaces give the same answers as alone."""
    w1, w2 = P.tanh_linear(5000, 16), P.tanh_linear(7000, 32)
    p1, p2 = W.TanhLinear(w1["A"], w1["b"]), W.TanhLinear(w2["A"], w2["b"])
    st = M.Stats()
    r1, x1 = p1.solve(w1["x0"], stats=st, flags=M.TIME_KERNELS)
    r2, x2 = p2.solve(w2["x0"])
    r1b, x1b = p1.solve(w1["x0"])
    assert np.array_equal(x1, x1b) and r1.iterations == r1b.iterations
    assert st.accepted == r1.iterations and st.jtj_launches == st.jacobian_full + st.jacobian_broyden + st.jtj_resyncs
    assert st.jtj_ms > 0 and st.passes >= st.accepted + st.rejected


@pytest.mark.parametrize("m,n", [(3000, 17), (9000, 32), (20000, 64)])
def test_null_step_elision_is_bitwise_equivalent(m, n):
    """At the end of a noisy solve lambda grows until the rounded step is exactly zero (trial == x bit for bit) long before
    lambda > maxLambda ends the loop (quirk Q3). The callbacks are pure (LS:73-80), so those evaluations are elided: same
    x, residual, lambda, counters (fCalls counts them like the reference) and the same trace, with fewer callback launches."""
    w = P.tanh_linear(m, n)
    prob = W.TanhLinear(w["A"], w["b"])
    s = M.LeastSquaresSettings(); s.absTolerance = 0.0            # an accepted step has dx != 0: never x-converged, always the tail
    out = []
    for skip in (True, False):
        st, tr = M.Stats(), M.Trace(4096)
        r, x = prob.solve(w["x0"], settings=s, batched=True, stats=st, trace=tr, variant=0 if skip else M.VARIANT_NO_NULL_SKIP)
        out.append((r, x, st, tr.records()))
    (r1, x1, st1, t1), (r0, x0, st0, t0) = out
    assert r1.status == M.LeastSquaresStatus.furtherImprovement == r0.status      # lambda > maxLambda, LS:979
    assert st1.elided_evaluations >= 8 and st0.elided_evaluations == 0
    assert np.array_equal(x1, x0) and r1.residual == r0.residual and r1.lambda_ == r0.lambda_
    assert (r1.iterations, r1.fCalls) == (r0.iterations, r0.fCalls)
    assert (st1.passes, st1.accepted, st1.rejected) == (st0.passes, st0.accepted, st0.rejected)
    assert t1 == t0


def test_null_step_elision_host_callbacks(oracle):
    """Reference ABI (host callbacks): the trial point reaches the host anyway, f is simply not called when it equals x.
    T3b ends with 51 rejections at the bound; the counters still match the oracle's."""
    p = P.t3b()
    calls = [0]

    def f(x, y):
        calls[0] += 1
        p["f"](x, y)
    res, x = M.optimize(f, p["m"], p["x0"], p["lower"], p["upper"], g=p["g"])
    ro, xo = run_oracle(oracle, p)
    assert (res.iterations, res.fCalls, res.gCalls) == (ro.iterations, ro.fCalls, ro.gCalls)
    assert np.array_equal(x, xo) or np.allclose(x, xo, rtol=1e-12)
    assert calls[0] < res.fCalls                                                  # some evaluations were elided


@pytest.mark.parametrize("m,n,bounded", [(20000, 32, False), (9000, 17, False), (30000, 128, False), (12000, 24, True), (15000, 256, False)])
def test_rounds_enqueued_ahead_of_time_change_nothing(m, n, bounded):
    """Small problems (J up to 32 MB: every shape here) pipeline by default: while the GPU runs a round, the host enqueues the
    library part of the next one behind a device-side guard (pipelined rounds). Guard closed -> the kernels return
    at once; guard open -> they are exactly the kernels the host would have launched. So x, residual, lambda, status and
    every counter are bit-identical to VARIANT_NO_PIPELINE."""
    w = P.tanh_linear(m, n)
    prob = W.TanhLinear(w["A"], w["b"])
    lo = up = None
    x0 = w["x0"]
    if bounded:
        lo = w["xstar"] - 0.5; up = w["xstar"] + 0.5
        lo[::4] = w["xstar"][::4] + 0.02
        x0 = np.clip(x0, lo, up)
    for tol in (1e-5, 1e-12):
        s = M.LeastSquaresSettings(); s.absTolerance = tol
        out = []
        for variant in (0, M.VARIANT_NO_PIPELINE):
            st = M.Stats()
            r, x = prob.solve(x0, l=lo, u=up, settings=s, batched=True, stats=st, flags=M.TIME_KERNELS, variant=variant)
            out.append((r, x, st))
        (r1, x1, s1), (r0, x0_, s0) = out
        assert np.array_equal(x1, x0_) and r1.residual == r0.residual and r1.lambda_ == r0.lambda_
        assert (r1.status, r1.iterations, r1.fCalls) == (r0.status, r0.iterations, r0.fCalls)
        for k in ("passes", "accepted", "rejected", "jacobian_full", "jacobian_broyden", "broyden_lr_columns", "broyden_flushes",
                  "jtj_resyncs", "jtj_launches", "jtj_broyden_launches", "solve_launches", "trial_callback_points", "elided_evaluations"):
            assert getattr(s1, k) == getattr(s0, k), (k, getattr(s1, k), getattr(s0, k))
