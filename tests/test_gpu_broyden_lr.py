"""Broyden passes as pending rank-one terms (csrc/broyden_lr.h) against the kernels that rewrite J on every pass
(variant VARIANT_BROYDEN_REWRITE: the literal restatement of LS:1002-1006 + 1052 + 1065) and against the oracle.

The two paths compute the same quantities in different summation orders, so they agree to rounding, not bitwise:
x rtol 1e-6 / residual rtol 1e-9 like every other whole-path parity test, and the per-pass traces are equal event by
event up to the first pass that compares rounding noise (see test_gpu_lm.first_noisy_pass)."""
import ctypes as C

import numpy as np
import pytest

import mir_optim_amd as M
from mir_optim_amd import workloads as W
import problems as P
from test_gpu_lm import first_noisy_pass

pytestmark = pytest.mark.gpu


def solve_with(prob, w, variant, settings, **kw):
    tr = M.Trace(4096)
    st = M.Stats()
    res, x = prob.solve(w["x0"], settings=settings, trace=tr, stats=st, variant=variant, **kw)
    return res, x, tr.records(), st


def assert_same_trajectory(ra, rb, what):
    k = min(first_noisy_pass(ra), first_noisy_pass(rb), len(ra), len(rb))
    assert k >= 3, what
    for a, b in zip(ra[:k], rb[:k]):
        assert a[0] == b[0] and a[1] == b[1], (what, a, b)
        assert np.isclose(a[2], b[2], rtol=1e-6, atol=0), (what, a, b)
        assert np.isclose(a[3], b[3], rtol=1e-7, atol=1e-300), (what, a, b)


@pytest.mark.parametrize("m,n,dtype", [(20000, 32, np.float64), (4096, 16, np.float64), (5001, 24, np.float64),
                                       (9973, 100, np.float64), (50000, 128, np.float64), (30000, 208, np.float64),
                                       (40000, 256, np.float64), (777, 7, np.float64), (6000, 16, np.float32)])
def test_lowrank_broyden_matches_rewriting_kernels(m, n, dtype):
    w = P.tanh_linear(m, n)
    prob = W.TanhLinear(w["A"], w["b"], dtype=dtype)
    s = M.LeastSquaresSettings(dtype=dtype) if dtype == np.float32 else M.LeastSquaresSettings()
    if dtype == np.float64:
        s.absTolerance = 1e-9
    rf, xf, tf, sf = solve_with(prob, w, M.VARIANT_BROYDEN_REWRITE, s)
    rl, xl, tl, sl = solve_with(prob, w, 0, s)
    assert sl.jacobian_broyden >= 2 and sf.jacobian_broyden >= 2
    assert (int(rf.status) >= 0) and (int(rl.status) >= 0)
    if dtype == np.float64:
        assert np.allclose(xl, xf, rtol=1e-6, atol=1e-9), np.abs(xl - xf).max()
        assert np.isclose(rl.residual, rf.residual, rtol=1e-9)
        assert_same_trajectory(tl, tf, (m, n))
    else:
        assert np.allclose(xl, xf, rtol=2e-2, atol=2e-3)
        assert np.isclose(rl.residual, rf.residual, rtol=1e-3)


@pytest.mark.parametrize("cap", [1, 2, 3])
@pytest.mark.parametrize("m,n", [(20000, 32), (9973, 100), (30000, 208)])
def test_lowrank_flush_into_J(m, n, cap):
    """variant_lr_cap bounds the pending terms; beyond it they are folded into J (k_lr_flush), J^T J is recomputed from
    the flushed J and the sweep restarts at k = 0. Any cap gives the same trajectory to rounding."""
    w = P.tanh_linear(m, n)
    prob = W.TanhLinear(w["A"], w["b"])
    s = M.LeastSquaresSettings(); s.absTolerance = 1e-9
    r0, x0, t0, st0 = solve_with(prob, w, 0, s)
    r1, x1, t1, st1 = solve_with(prob, w, M.variant_lr_cap(cap), s)
    assert st1.jacobian_broyden > int(cap) and st1.broyden_flushes >= 1     # the cap really was reached
    assert st1.jtj_resyncs == st1.broyden_flushes
    assert np.allclose(x1, x0, rtol=1e-6, atol=1e-9)
    assert np.isclose(r1.residual, r0.residual, rtol=1e-9)
    assert_same_trajectory(t1, t0, (m, n, cap))


def test_lowrank_long_broyden_run_with_analytic_jacobian_age(oracle):
    """maxAge large and an analytic Jacobian: one full refresh, then only Broyden updates -- more than kLrMax = 16 of
    them on a slowly converging start, so the default cap flushes too. Compared with the oracle."""
    w = P.tanh_linear(8000, 24)
    rng = np.random.default_rng(5)
    x0 = w["xstar"] + 1.5 * rng.standard_normal(24)
    prob = W.TanhLinear(w["A"], w["b"])
    s = M.LeastSquaresSettings(); s.absTolerance = 1e-10; s.maxAge = 200
    st = M.Stats()
    res, x = prob.solve(x0, settings=s, analytic=True, stats=st)
    so = oracle.default_settings(); so.absTolerance = 1e-10; so.maxAge = 200
    ctx = oracle.TanhLinearCtx(w["A"].ctypes.data, w["b"].ctypes.data)
    ro, xo = oracle.optimize(oracle.native_fn("wlc_tanh_linear_f"), w["m"], x0, settings=so, fctx=C.addressof(ctx),
                             g=oracle.native_fn("wlc_tanh_linear_g"), gctx=C.addressof(ctx))
    assert st.jacobian_broyden > 16
    assert int(res.status) >= 0 and ro.status >= 0
    assert np.allclose(x, xo, rtol=1e-6, atol=1e-9), np.abs(x - xo).max()
    assert np.isclose(res.residual, ro.residual, rtol=1e-9)


def test_lowrank_sweep_reads_J_once_and_never_writes_it():
    """The statistic the bench's roofline uses: pending columns read per sweep."""
    w = P.tanh_linear(20000, 32)
    prob = W.TanhLinear(w["A"], w["b"])
    s = M.LeastSquaresSettings(); s.absTolerance = 1e-9
    st = M.Stats()
    prob.solve(w["x0"], settings=s, stats=st, flags=M.TIME_KERNELS)
    assert st.jtj_broyden_launches == st.jacobian_broyden >= 2
    assert st.broyden_lr_columns <= st.jacobian_broyden * 15


@pytest.mark.parametrize("cap", [2, 5])
def test_ill_conditioned_long_broyden_run_resynchronises_at_each_flush(oracle, cap):
    """the recurrence J^T J += v dx^T + dx v^T + uu dx dx^T must not run from one full refresh to the next
    (up to maxAge = 2n passes) on its own rounding errors. Columns scaled over three decades (cond(J^T J) ~ 1e6+), analytic
    Jacobian with maxAge = 200: one full refresh, then dozens of Broyden passes. At every flush J^T J / J^T y are recomputed
    from the flushed J (stats.jtj_resyncs == stats.broyden_flushes), as the reference's syrk does every pass (LS:1065); the
    trajectory follows the oracle's pass by pass and the literal rewriting kernels' to rounding."""
    m, n = 12000, 24
    w = P.tanh_linear(m, n)
    scale = 10.0 ** (-3.0 * np.arange(n) / (n - 1))
    A = np.ascontiguousarray(w["A"] * scale[None, :])
    xs = w["xstar"] / scale
    rng = np.random.default_rng(11)
    x0 = xs + (0.8 * rng.standard_normal(n)) / scale
    prob = W.TanhLinear(A, w["b"])
    s = M.LeastSquaresSettings(); s.absTolerance = 1e-10; s.maxAge = 200
    out = {}
    for name, variant in (("lr", M.variant_lr_cap(cap) if cap else 0), ("rewrite", M.VARIANT_BROYDEN_REWRITE)):
        st, tr = M.Stats(), M.Trace(8192)
        res, x = prob.solve(x0, settings=s, analytic=True, stats=st, trace=tr, variant=variant)
        out[name] = (res, x, st, tr.records())
    res, x, st, recs = out["lr"]
    so = oracle.default_settings(); so.absTolerance = 1e-10; so.maxAge = 200
    ctx = oracle.TanhLinearCtx(A.ctypes.data, w["b"].ctypes.data)
    ev = []
    ro, xo = oracle.optimize(oracle.native_fn("wlc_tanh_linear_f"), m, x0, settings=so, fctx=C.addressof(ctx),
                             g=oracle.native_fn("wlc_tanh_linear_g"), gctx=C.addressof(ctx), trace=lambda *a: ev.append(a))
    assert st.jacobian_broyden > 15 and st.broyden_flushes >= 2 and st.jtj_resyncs == st.broyden_flushes
    assert int(res.status) >= 0 and ro.status >= 0 and int(out["rewrite"][0].status) >= 0
    assert np.allclose(x * scale, xo * scale, rtol=1e-6, atol=1e-9), np.abs((x - xo) * scale).max()
    assert np.isclose(res.residual, ro.residual, rtol=1e-9)
    assert np.allclose(x * scale, out["rewrite"][1] * scale, rtol=1e-6, atol=1e-9)
    K = min(first_noisy_pass(recs), first_noisy_pass(ev), len(recs), len(ev))
    assert K >= 12, (K, len(recs), len(ev))
    for g, e in zip(recs[:K], ev[:K]):
        assert g[:2] == e[:2], (g, e)
        assert np.isclose(g[2], e[2], rtol=1e-5), (g, e)
        assert np.isclose(g[3], e[3], rtol=1e-6, atol=1e-300), (g, e)
