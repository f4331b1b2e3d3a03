"""The resident-J solver (include/mir_optim_amd_resident.hpp, csrc/resident_kernel.h): the whole loop of
least_squares.d:972-1175 in ONE cooperative launch, J sharded over the CUs' LDS. Compared with the oracle -- minimiser,
residual, status class AND the per-pass trace, event by event up to the first noise-decided pass -- and with the launch-chain
path (mir_optimize_least_squares_gpu_d with device callbacks), which minimises the same residual expression.
BASELINE cfg 2 (Gaussian sum, m = 1e5 x n = 16, width bounds) at full size, with and without binding bounds."""
import ctypes as C

import numpy as np
import pytest

import mir_optim_amd as M
from mir_optim_amd import workloads as W
import problems as P
from test_gpu_lm import assert_traces_agree_until_noise, first_noisy_pass

pytestmark = pytest.mark.gpu


def oracle_gauss(oracle, g, x0, lower, upper, settings=None):
    ctx = oracle.GaussSumCtx(g["t"].ctypes.data, g["data"].ctypes.data)
    ev = []
    ro, xo = oracle.optimize(oracle.native_fn("wlc_gauss_sum_f"), g["m"], x0, lower=lower, upper=upper, fctx=C.addressof(ctx),
                             settings=settings, trace=lambda *a: ev.append(a))
    return ro, xo, ev


def test_cfg2_gauss_sum_full_size_resident(oracle):
    """BASELINE cfg 2 through the resident path: same minimiser as the oracle (x rtol 1e-6, residual rtol 1e-9), traces equal
    pass by pass up to the first noise-decided one, every counter consistent with the trace."""
    g = P.gauss_sum(100000, K=5)
    r = W.Resident.gauss_sum(g["t"], g["data"], K=5)
    assert r.plan_rc == 0 and r.n == 16 and r.grid == 256
    tr = M.Trace(4096)
    res, x, st = r.solve(g["x0"], g["lower"], g["upper"], trace=tr)
    ro, xo, ev = oracle_gauss(oracle, g, g["x0"], g["lower"], g["upper"])
    assert res.status >= 0 and ro.status >= 0 and st["abort_code"] == 0
    assert np.allclose(x, xo, rtol=1e-6, atol=1e-9) and np.isclose(res.residual, ro.residual, rtol=1e-9)
    assert np.allclose(x, g["truth"], rtol=5e-3, atol=1e-3)
    recs = tr.records()
    k_end = assert_traces_agree_until_noise(recs, ev, min_passes=30)
    assert sum(1 for q in recs[:k_end] if q[0] == 3) >= 12
    # the counters are those of the trace
    assert res.iterations == sum(1 for q in recs if q[0] == 3) == st["accepted"]
    assert st["rejected"] == sum(1 for q in recs if q[0] == 2) and st["jacobian_full"] == sum(1 for q in recs if q[0] == 0)
    assert st["jacobian_broyden"] == sum(1 for q in recs if q[0] == 1)
    assert res.fCalls == 1 + 16 * st["jacobian_full"] + st["accepted"] + st["rejected"]         # LS:953, 1049 (Q5), 1112
    # one round per executed pass that needs a residual, one more per refresh: elided null steps cost none, and neither do
    # rejections decided from a sum of squares evaluated along an earlier round
    assert st["rounds"] == 1 + st["jacobian_full"] + st["accepted"] + st["rejected"] - st["elided_evaluations"] - st["lookahead_rejections"]
    assert st["lookahead_rejections"] >= 3


def test_cfg2_binding_width_bounds_resident(oracle):
    """cfg 2 with two widths boxed in above their true value and two amplitudes from above: BOXCQP's active-set loop
    (boxcqp.d:234-376) runs inside the launch in nearly every pass; same active set and minimiser as the oracle."""
    g = P.gauss_sum(100000, K=5)
    K = g["K"]
    lower, upper = g["lower"].copy(), g["upper"].copy()
    lower[2 * K] = 0.045; lower[2 * K + 3] = 0.05
    upper[0] = 0.95; upper[3] = 0.85
    x0 = np.clip(g["x0"], lower, upper)
    r = W.Resident.gauss_sum(g["t"], g["data"], K=5)
    tr = M.Trace(4096)
    res, x, st = r.solve(x0, lower, upper, trace=tr)
    ro, xo, ev = oracle_gauss(oracle, g, x0, lower, upper)
    assert res.status >= 0 and ro.status >= 0
    assert st["qp_active_set_passes"] >= 3
    on_gpu = (x == lower) | (x == upper)
    on_cpu = (xo == lower) | (xo == upper)
    assert on_gpu.sum() >= 4 and np.array_equal(on_gpu, on_cpu)
    assert np.allclose(x, xo, rtol=1e-6, atol=1e-9) and np.isclose(res.residual, ro.residual, rtol=1e-9)
    assert_traces_agree_until_noise(tr.records(), ev, min_passes=8)
    # and the launch-chain path lands on the same point
    prob = W.Curve("gauss_sum", g["t"], g["data"])
    res2, x2 = prob.solve(x0, lower, upper)
    assert np.allclose(x, x2, rtol=1e-6, atol=1e-9) and np.isclose(res.residual, res2.residual, rtol=1e-9)


def test_tanh_linear_20000x32_resident(oracle):
    """A second model (two 16-column blocks: three accumulator blocks a wave, the NB = 2 solve): cfg 3's family at an
    LDS-resident size, unbounded (the kernel without the active-set loop)."""
    w = P.tanh_linear(20000, 32)
    r = W.Resident.tanh_linear(w["A"], w["b"])
    assert r.plan_rc == 0 and r.n == 32
    s = M.LeastSquaresSettings(); s.absTolerance = 1e-9
    so = oracle.default_settings(); so.absTolerance = 1e-9
    ctx = oracle.TanhLinearCtx(w["A"].ctypes.data, w["b"].ctypes.data)
    ev = []
    ro, xo = oracle.optimize(oracle.native_fn("wlc_tanh_linear_f"), w["m"], w["x0"], settings=so, fctx=C.addressof(ctx),
                             trace=lambda *a: ev.append(a))
    outs = []
    for variant in (W.RESIDENT_UNBOUNDED, 0):
        tr = M.Trace(1024)
        res, x, st = r.solve(w["x0"], settings=s, trace=tr, variant=variant)
        assert res.status >= 0 and ro.status >= 0
        assert np.allclose(x, xo, rtol=1e-6, atol=1e-9) and np.isclose(res.residual, ro.residual, rtol=1e-9)
        # pass by pass as tests/test_gpu_lm.py::test_pass_by_pass_trajectory_matches_oracle does for this family (the device tanh
        # and libm's differ by 2e-16, which the finite differences amplify to 1e-8 in J: lambda to 1e-8, residuals to 1e-7)
        got = tr.records()
        K = min(first_noisy_pass(got), first_noisy_pass(ev), len(got), len(ev))
        assert K >= 9, (K, len(got), len(ev))
        assert [q[:2] for q in got[:K]] == [(e[0], e[1]) for e in ev[:K]] and {0, 1, 3} <= {q[0] for q in got[:K]}
        for k, (q, e) in enumerate(zip(got[:K], ev[:K])):
            assert np.isclose(q[2], e[2], rtol=1e-8) and np.isclose(q[3], e[3], rtol=1e-7), (k, q, e)
            assert np.isclose(q[4], e[4], rtol=1e-7, atol=1e-300) and np.isclose(q[5], e[5], rtol=1e-3, atol=1e-22), (k, q, e)
        outs.append((x.tobytes(), int(res.status), res.iterations, res.fCalls, res.residual, res.lambda_, tuple(got)))
    assert outs[0] == outs[1]          # with or without the active-set loop compiled in: the same bits when no bound exists


def test_resident_equals_launch_chain_on_the_small_family(oracle):
    g = P.gauss_sum(20000, K=3)
    r = W.Resident.gauss_sum(g["t"], g["data"], K=3)
    res, x, st = r.solve(g["x0"], g["lower"], g["upper"])
    prob = W.Curve("gauss_sum", g["t"], g["data"])
    res2, x2 = prob.solve(g["x0"], g["lower"], g["upper"])
    ro, xo, _ = oracle_gauss(oracle, g, g["x0"], g["lower"], g["upper"])
    assert res.status >= 0 and res2.status >= 0 and ro.status >= 0
    assert np.allclose(x, xo, rtol=1e-6, atol=1e-9) and np.allclose(x, x2, rtol=1e-6, atol=1e-9)
    assert np.isclose(res.residual, ro.residual, rtol=1e-8) and np.isclose(res.residual, res2.residual, rtol=1e-9)


def test_resident_is_reproducible_bit_for_bit():
    """Fixed-order sums, no float atomics: every launch returns the same bits (least_squares.d:73-80: pure callbacks)."""
    g = P.gauss_sum(100000, K=5)
    r = W.Resident.gauss_sum(g["t"], g["data"], K=5)
    outs = set()
    for _ in range(6):
        tr = M.Trace(4096)
        res, x, st = r.solve(g["x0"], g["lower"], g["upper"], trace=tr)
        outs.add((x.tobytes(), int(res.status), res.iterations, res.fCalls, res.residual, res.lambda_, tuple(tr.records())))
    assert len(outs) == 1


@pytest.mark.parametrize("wgs,m", [(1, 800), (3, 2400), (16, 6000), (17, 6001), (100, 20000), (255, 20000)])
def test_resident_any_grid(oracle, wgs, m):
    """The slice / group / leader arithmetic for grids that are not 256: one workgroup, fewer groups than sixteen, ragged groups."""
    g = P.gauss_sum(m, K=3)
    r = W.Resident.gauss_sum(g["t"], g["data"], K=3, max_workgroups=wgs)
    assert r.plan_rc == 0 and r.grid <= wgs and r.grid >= min(wgs, 254)
    tr = M.Trace(2048)
    res, x, st = r.solve(g["x0"], g["lower"], g["upper"], trace=tr)
    ro, xo, ev = oracle_gauss(oracle, g, g["x0"], g["lower"], g["upper"])
    assert res.status >= 0 and st["abort_code"] == 0 and st["grid"] == r.grid
    assert np.allclose(x, xo, rtol=1e-6, atol=1e-9) and np.isclose(res.residual, ro.residual, rtol=1e-8)
    assert_traces_agree_until_noise(tr.records(), ev, min_passes=10)


def test_resident_bounded_three_parameter_fit(oracle):
    """Reference unittest T5's family (least_squares.d:366-411) on two workgroups: the bounds bind, x stays inside."""
    a, b = P.t5()
    for p in (a, b):
        t, yd = p["t"], p["data"]
        lower = np.full(3, -np.inf) if p["lower"] is None else np.array(p["lower"], dtype=float)
        upper = np.full(3, np.inf) if p["upper"] is None else np.array(p["upper"], dtype=float)
        r = W.Resident("exp_decay1", np.stack([t, yd], axis=1))
        res, x, st = r.solve(p["x0"], lower, upper)
        ctx = oracle.ExpDecayCtx(t.ctypes.data, yd.ctypes.data, 1)
        ro, xo = oracle.optimize(oracle.native_fn("wlc_exp_decay_f"), 100, p["x0"], lower=lower, upper=upper, fctx=C.addressof(ctx))
        assert res.status >= 0 and ro.status >= 0
        assert np.all(x >= lower) and np.all(x <= upper)                                           # LS:395, 410
        assert np.allclose(x, xo, rtol=1e-6, atol=1e-9) and np.isclose(res.residual, ro.residual, rtol=1e-9)


def test_resident_validation_and_exits(oracle):
    g = P.gauss_sum(5000, K=3)
    r = W.Resident.gauss_sum(g["t"], g["data"], K=3)
    S = M.LeastSquaresStatus
    # LS:930-932 inside the kernel (x, lower, upper are device data)
    bad = g["x0"].copy(); bad[1] = np.nan
    res, x, _ = r.solve(bad, g["lower"], g["upper"])
    assert res.status == S.badGuess and res.iterations == 0 and res.fCalls == 0 and res.residual == np.inf
    lo = g["lower"].copy(); lo[0] = g["x0"][0] + 1
    res, x, _ = r.solve(g["x0"], lo, g["upper"])
    assert res.status == S.badBounds
    # LS:933-943 on the host
    s = M.LeastSquaresSettings(); s.minStepQuality = 1.5
    res, x, _ = r.solve(g["x0"], g["lower"], g["upper"], settings=s)
    assert res.status == S.badMinStepQuality
    s = M.LeastSquaresSettings(); s.lambdaIncrease = 0.5
    res, x, _ = r.solve(g["x0"], g["lower"], g["upper"], settings=s)
    assert res.status == S.badLambdaParams
    # maxIterations, LS:1175; and the counters of the oracle at that point
    s = M.LeastSquaresSettings(); s.maxIterations = 3
    so = oracle.default_settings(); so.maxIterations = 3
    res, x, _ = r.solve(g["x0"], g["lower"], g["upper"], settings=s)
    ro, xo, _ = oracle_gauss(oracle, g, g["x0"], g["lower"], g["upper"], settings=so)
    assert res.status == S.maxIterations == ro.status and res.iterations == ro.iterations == 3 and res.fCalls == ro.fCalls
    assert np.allclose(x, xo, rtol=1e-9)
    # fConverged, LS:974: data that the model reproduces exactly
    t = g["t"]
    truth = g["truth"]
    K = 3
    clean = sum(truth[k] * np.exp(-(t - truth[K + k]) ** 2 / (2 * truth[2 * K + k] ** 2)) for k in range(K)) + truth[3 * K]
    r2 = W.Resident.gauss_sum(t, clean, K=3)
    s = M.LeastSquaresSettings(); s.maxGoodResidual = 1e-12
    res, x, _ = r2.solve(g["x0"], g["lower"], g["upper"], settings=s)
    assert res.status == S.fConverged and res.residual <= 1e-12 and np.allclose(x, truth, rtol=1e-5)


def test_resident_null_step_elision_changes_nothing():
    """Trials equal to x bit for bit are not evaluated (the callbacks are pure, least_squares.d:73-80): the same bits, counters
    and trace with the elision switched off."""
    g = P.gauss_sum(100000, K=5)
    r = W.Resident.gauss_sum(g["t"], g["data"], K=5)
    outs = []
    for variant in (0, W.RESIDENT_NO_NULL_SKIP):
        tr = M.Trace(4096)
        res, x, st = r.solve(g["x0"], g["lower"], g["upper"], trace=tr, variant=variant)
        outs.append((x.tobytes(), int(res.status), res.iterations, res.fCalls, res.residual, res.lambda_, tuple(tr.records())))
        if variant:
            assert st["elided_evaluations"] == 0
        else:
            assert st["elided_evaluations"] > 0
    assert outs[0] == outs[1]


def test_resident_lookahead_changes_nothing():
    """The sums of squares of the next damping levels, evaluated along a round and used to book rejected passes without a round
    of their own, are the numbers those rounds would have produced: the same bits, counters and trace with the look-ahead
    switched off -- unbounded, with binding bounds (levels behind an infeasible first one are not offered), on ragged grids."""
    for m, K, wgs, bind in ((100000, 5, 0, False), (100000, 5, 0, True), (20011, 3, 37, False), (777, 3, 3, False), (600, 3, 1, False)):
        g = P.gauss_sum(m, K=K)
        lower, upper, x0 = g["lower"].copy(), g["upper"].copy(), g["x0"]
        if bind:
            lower[2 * K] = 0.045; lower[2 * K + 3] = 0.05
            upper[0] = 0.95; upper[3] = 0.85
            x0 = np.clip(x0, lower, upper)
        r = W.Resident.gauss_sum(g["t"], g["data"], K=K, max_workgroups=wgs)
        outs, looks, rounds = [], [], []
        for variant in (0, W.RESIDENT_NO_LOOKAHEAD):
            tr = M.Trace(4096)
            res, x, st = r.solve(x0, lower, upper, trace=tr, variant=variant)
            assert st["abort_code"] == 0
            outs.append((x.tobytes(), int(res.status), res.iterations, res.fCalls, res.residual, res.lambda_, tuple(tr.records())))
            looks.append(st["lookahead_rejections"]); rounds.append(st["rounds"])
        assert outs[0] == outs[1], (m, K, wgs, bind)
        assert looks[1] == 0 and rounds[1] - rounds[0] == looks[0]
    # (cfg 2 itself: at least a handful of rounds saved)
    g = P.gauss_sum(100000, K=5)
    _, _, st = W.Resident.gauss_sum(g["t"], g["data"], K=5).solve(g["x0"], g["lower"], g["upper"])
    assert st["lookahead_rejections"] >= 3


def test_resident_without_clock_stamps():
    """MIR_LSQ_RESIDENT_NO_STAMPS: the statistics carry the counters only (every t_* but t_total is 0); the fit is the same."""
    g = P.gauss_sum(20000, K=3)
    r = W.Resident.gauss_sum(g["t"], g["data"], K=3)
    res, x, st = r.solve(g["x0"], g["lower"], g["upper"])
    res2, x2, st2 = r.solve(g["x0"], g["lower"], g["upper"], variant=W.RESIDENT_NO_STAMPS)
    assert x.tobytes() == x2.tobytes() and res.residual == res2.residual and res.fCalls == res2.fCalls
    for k in ("rounds", "passes", "accepted", "rejected", "jacobian_full", "jacobian_broyden", "elided_evaluations", "lookahead_rejections"):
        assert st[k] == st2[k]
    assert st["t_solver"] > 0 and st["t_worker"] > 0 and st2["t_total"] > 0
    assert all(st2[k] == 0 for k in st2 if k.startswith("t_") and k != "t_total")


def test_resident_analytic_jacobian_matches_oracle_with_g(oracle):
    """The reference's optional g callback (least_squares.d:80, 1010-1014) on the resident path: a model with `jac`
    (tanh-linear: (1 - tanh^2(a.x)) a) and MIR_LSQ_RESIDENT_ANALYTIC_JACOBIAN -- refreshes count in gCalls, not in fCalls, the
    default age limit is 3 (LS:945); against the oracle with the same analytic Jacobian: same counters, trace equal up to the
    first noise-decided pass, same minimiser. A model without `jac` is refused (-1)."""
    w = P.tanh_linear(20000, 32)
    r = W.Resident.tanh_linear(w["A"], w["b"])
    s = M.LeastSquaresSettings(); s.absTolerance = 1e-9
    so = oracle.default_settings(); so.absTolerance = 1e-9
    ctx = oracle.TanhLinearCtx(w["A"].ctypes.data, w["b"].ctypes.data)
    ev = []
    ro, xo = oracle.optimize(oracle.native_fn("wlc_tanh_linear_f"), w["m"], w["x0"], g=oracle.native_fn("wlc_tanh_linear_g"), settings=so,
                             fctx=C.addressof(ctx), gctx=C.addressof(ctx), trace=lambda *a: ev.append(a))
    tr = M.Trace(1024)
    res, x, st = r.solve(w["x0"], settings=s, trace=tr, variant=W.RESIDENT_ANALYTIC_JACOBIAN)
    assert res.status >= 0 and ro.status >= 0 and st["abort_code"] == 0
    assert res.gCalls == ro.gCalls >= 2 and res.gCalls == st["jacobian_full"]
    assert np.allclose(x, xo, rtol=1e-6, atol=1e-9) and np.isclose(res.residual, ro.residual, rtol=1e-9)
    k_end = assert_traces_agree_until_noise(tr.records(), ev, min_passes=8)
    assert (res.iterations, res.fCalls) == (ro.iterations, ro.fCalls) or k_end < len(ev)
    # finite differences on the same model: more residual calls, no g calls
    res_fd, x_fd, _ = r.solve(w["x0"], settings=s)
    assert res_fd.gCalls == 0 and res_fd.fCalls > res.fCalls and np.allclose(x_fd, x, rtol=1e-6, atol=1e-9)
    # a model without jac
    g = P.gauss_sum(6000, K=3)
    rg = W.Resident.gauss_sum(g["t"], g["data"], K=3)
    rg.upload_point(g["x0"], g["lower"], g["upper"])
    assert rg.launch(None, 0, W.RESIDENT_ANALYTIC_JACOBIAN) == -1


def test_a_missing_workgroup_ends_in_numeric_error_not_in_a_hang():
    """Every in-launch wait is bounded: with one workgroup leaving before the first round (DEBUG_DROP_WORKGROUP) the others
    give up, raise the abort word and the launch returns numericError with an abort code; the next launch is clean."""
    import time
    g = P.gauss_sum(20000, K=3)
    r = W.Resident.gauss_sum(g["t"], g["data"], K=3)
    t0 = time.perf_counter()
    res, x, st = r.solve(g["x0"], g["lower"], g["upper"], variant=W.RESIDENT_DEBUG_DROP_WORKGROUP)
    dt = time.perf_counter() - t0
    assert res.status == M.LeastSquaresStatus.numericError and 0.5 < dt < 20.0, (res, dt)
    res2, x2, st2 = r.solve(g["x0"], g["lower"], g["upper"])
    assert int(res2.status) >= 0 and st2["abort_code"] == 0 and np.allclose(x2, g["truth"], rtol=5e-3, atol=1e-3)


def test_an_aborted_launch_degrades_to_the_launch_chain():
    """A launch that gave up on a hand-off is a scheduling fact (a device shared with other work), not a numeric failure: with a
    fallback problem the wrapper runs the same fit through the launch chain in a fresh launch of this process; the caller gets
    that result, and the abort code stays in the statistics."""
    g = P.gauss_sum(20000, K=3)
    prob = W.Curve("gauss_sum", g["t"], g["data"])
    r = W.Resident.gauss_sum(g["t"], g["data"], K=3, fallback=prob)
    res, x, st = r.solve(g["x0"], g["lower"], g["upper"], variant=W.RESIDENT_DEBUG_DROP_WORKGROUP, batched=True)
    assert st["abort_code"] != 0 and st["fallback"] == "launch chain"
    ref, xr = prob.solve(g["x0"], g["lower"], g["upper"], batched=True)
    assert int(res.status) >= 0 and (res.status, res.iterations, res.fCalls, res.residual) == (ref.status, ref.iterations, ref.fCalls, ref.residual)
    assert x.tobytes() == xr.tobytes()
    res2, x2, st2 = r.solve(g["x0"], g["lower"], g["upper"])           # the next launch is clean and stays on the resident path
    assert int(res2.status) >= 0 and st2["abort_code"] == 0 and "fallback" not in st2


def test_solves_beside_a_tenant_that_holds_every_cu_return_the_uncontended_bits():
    """Contention (SURVEY 8b "Threading": re-entrant, every failure a status): while a third stream keeps EVERY CU busy with
    50 ms filler kernels that also hold most of each CU's LDS (wl_busy), one host thread runs a wide-n solve (n = 300: helper
    workgroups that wait for one another, solve_coop.h) and another a resident solve (a cooperative launch whose workgroups
    wait for one another). Their workgroups start late and apart; nothing times out, and both return the bits of the same
    solves on an idle device."""
    import threading
    from mir_optim_amd import api
    WL = api.workloads_lib()
    w = P.tanh_linear(1500, 300)
    wide = W.TanhLinear(w["A"], w["b"])
    g = P.gauss_sum(100000, K=5)
    res_p = W.Resident.gauss_sum(g["t"], g["data"], K=5)
    st0 = M.Stats()
    ref_w = wide.solve(w["x0"], batched=True, stats=st0)
    ref_r = res_p.solve(g["x0"], g["lower"], g["upper"])
    assert int(ref_w[0].status) >= 0 and int(ref_r[0].status) >= 0 and st0.coop_timeouts == 0
    filler = api.Stream()
    stop = threading.Event()
    out, err = {}, []

    def tenant():
        try:
            while not stop.is_set():
                for _ in range(4):                              # 4 x 50 ms queued, then wait: the device is never idle for long
                    WL.wl_busy(C.c_void_p(filler.handle), C.c_uint(512), C.c_uint(72 * 1024), C.c_uint(50000))
                filler.synchronize()
        except BaseException as e:      # noqa: BLE001
            err.append(e)

    def wide_solves():
        try:
            got = []
            for _ in range(2):
                st = M.Stats()
                r, x = wide.solve(w["x0"], batched=True, stats=st)
                got.append((x.tobytes(), int(r.status), r.iterations, r.fCalls, r.residual, st.coop_timeouts))
            out["wide"] = got
        except BaseException as e:      # noqa: BLE001
            err.append(e)

    def resident_solves():
        try:
            got = []
            for _ in range(3):
                r, x, st = res_p.solve(g["x0"], g["lower"], g["upper"])
                got.append((x.tobytes(), int(r.status), r.iterations, r.fCalls, r.residual, st["abort_code"]))
            out["resident"] = got
        except BaseException as e:      # noqa: BLE001
            err.append(e)
    tt = threading.Thread(target=tenant)
    tt.start()
    ts = [threading.Thread(target=wide_solves), threading.Thread(target=resident_solves)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(600)
    stop.set()
    tt.join(60)
    assert not any(t.is_alive() for t in ts + [tt]) and not err, err
    want_w = (ref_w[1].tobytes(), int(ref_w[0].status), ref_w[0].iterations, ref_w[0].fCalls, ref_w[0].residual, 0)
    want_r = (ref_r[1].tobytes(), int(ref_r[0].status), ref_r[0].iterations, ref_r[0].fCalls, ref_r[0].residual, 0)
    assert all(q == want_w for q in out["wide"]), [q[1:] for q in out["wide"]]
    assert all(q == want_r for q in out["resident"]), [q[1:] for q in out["resident"]]


def test_resident_does_not_fit_falls_back(oracle):
    """m x (n + nd + 3) doubles beyond the chip's LDS: launch_resident answers -3 and the caller takes the launch-chain path."""
    g = P.gauss_sum(1000000, K=5)
    prob = W.Curve("gauss_sum", g["t"], g["data"])
    r = W.Resident.gauss_sum(g["t"], g["data"], K=5, fallback=prob)
    assert r.plan_rc == -3
    res, x, st = r.solve(g["x0"], g["lower"], g["upper"])
    assert st is None and res.status >= 0
    assert np.allclose(x, g["truth"], rtol=5e-3, atol=1e-3)
    with pytest.raises(W.ResidentDoesNotFit):
        W.Resident.gauss_sum(g["t"], g["data"], K=5).solve(g["x0"], g["lower"], g["upper"])


def test_two_host_threads_run_resident_solves_concurrently(oracle):
    """Re-entrancy (SURVEY 8b "Threading"): two host threads, each with its own problem, stream and workspace, launch resident
    solves at the same time, several times over -- small slices (both grids can be resident together) and cfg-2-sized ones (a CU's
    LDS holds one workgroup: the second cooperative launch waits for the first). Every solve returns the bits of the same solve
    run alone."""
    import threading
    probs = []
    for m, K in ((6000, 3), (100000, 5)):
        g = P.gauss_sum(m, K=K)
        probs.append((g, W.Resident.gauss_sum(g["t"], g["data"], K=K), W.Resident.gauss_sum(g["t"], g["data"], K=K)))
    for g, ra, rb in probs:
        ref_res, ref_x, _ = ra.solve(g["x0"], g["lower"], g["upper"])
        out, err = {}, []

        def work(tag, r):
            try:
                got = []
                for _ in range(4):
                    res, x, st = r.solve(g["x0"], g["lower"], g["upper"])
                    got.append((x.tobytes(), int(res.status), res.iterations, res.fCalls, res.residual, st["abort_code"]))
                out[tag] = got
            except BaseException as e:      # noqa: BLE001
                err.append(e)
        ts = [threading.Thread(target=work, args=(k, r)) for k, r in (("a", ra), ("b", rb))]
        for t in ts:
            t.start()
        for t in ts:
            t.join(300)
        assert not any(t.is_alive() for t in ts) and not err, err
        want = (ref_x.tobytes(), int(ref_res.status), ref_res.iterations, ref_res.fCalls, ref_res.residual, 0)
        assert all(q == want for q in out["a"] + out["b"])
