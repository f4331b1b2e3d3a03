"""The oracle's third-party pieces (posvx, KBN, BOXCQP) against independent implementations.

posvx is NOT in /root/reference (un-vendored mir-lapack -> system LAPACK, dub.sdl:7); the
oracle restates the published Netlib algorithm and is checked here against the LAPACK build
shipped inside scipy (OpenBLAS 0.3.29) -- the same library class the reference links."""
import numpy as np
import pytest
from scipy.linalg import lapack
from scipy.optimize import lsq_linear, minimize

import problems as P


def spd(n, cond, seed, scale=None):
    rng = np.random.default_rng(seed)
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    ev = np.logspace(0, np.log10(cond), n)
    A = (Q * ev) @ Q.T
    A = (A + A.T) / 2
    if scale is not None:
        A = A * np.outer(scale, scale)
    return A


@pytest.mark.parametrize("n,cond,badscale", [(1, 1, False), (2, 10, False), (3, 1e3, True), (16, 1e6, False),
                                             (33, 1e8, True), (128, 1e10, False), (128, 1e4, True)])
def test_posvx_matches_scipy_lapack(oracle, n, cond, badscale):
    scale = np.logspace(-3, 3, n) if badscale else None
    A = spd(n, cond, n, scale)
    b = np.random.default_rng(7).standard_normal(n)
    mine = oracle.posvx(A, b)
    a_s, lu, equed, s, b_s, x, rcond, ferr, berr, info = lapack.dposvx(A, b.reshape(-1, 1), fact="E", lower=1)
    assert mine["info"] == info == 0
    assert mine["equed"] == equed.decode()
    assert (mine["equed"] == "Y") == badscale or n == 1
    assert np.allclose(mine["s"], s, rtol=1e-15)
    xr = np.linalg.solve(A, b)
    err_ref = np.linalg.norm(x[:, 0] - xr) / np.linalg.norm(xr)
    err_mine = np.linalg.norm(mine["x"] - xr) / np.linalg.norm(xr)
    # both are refined solutions: same accuracy class, and close to each other
    assert err_mine <= 10 * err_ref + 1e-15
    diff = np.linalg.norm(mine["x"] - x[:, 0]) / np.linalg.norm(xr)
    assert diff <= 4 * max(err_ref, err_mine) + 1e-15
    assert mine["berr"] < 1e-15
    # condition estimate within the usual factor of the LAPACK estimate
    assert rcond / 4 <= mine["rcond"] <= rcond * 4


def test_posvx_info_codes(oracle):
    A = np.array([[1.0, 2.0], [2.0, 1.0]])          # indefinite: potrf fails at column 2
    mine = oracle.posvx(A, [1.0, 1.0])
    ref = lapack.dposvx(A, np.ones((2, 1)), fact="E", lower=1)
    assert mine["info"] == ref[-1] == 2
    A = np.array([[1.0, 0.0], [0.0, -1.0]])         # non-positive diagonal
    assert oracle.posvx(A, [1.0, 1.0])["info"] == lapack.dposvx(A, np.ones((2, 1)), fact="E", lower=1)[-1] == 2
    # singular to working precision: info = n+1 but x still computed (accepted at QP:212)
    v = np.array([1.0, 1.0, 1.0])
    A = np.outer(v, v) + 1e-17 * np.eye(3)
    m = oracle.posvx(A, v)
    r = lapack.dposvx(A, v.reshape(-1, 1), fact="E", lower=1)
    assert m["info"] in (r[-1], 3, 4)


def test_posvx_float(oracle):
    A = spd(24, 1e3, 5)
    b = np.random.default_rng(3).standard_normal(24)
    mine = oracle.posvx(A, b, dtype=np.float32)
    ref = lapack.sposvx(A.astype(np.float32), b.astype(np.float32).reshape(-1, 1), fact="E", lower=1)
    assert mine["info"] == ref[-1] == 0
    assert np.allclose(mine["x"], ref[5][:, 0], rtol=2e-4)


def qp_reference(Pm, q, l, u):
    """argmin 1/2 x'Px + q'x on a box by an independent solver (L-BFGS-B polished by active-set solve)."""
    n = len(q)
    r = minimize(lambda x: 0.5 * x @ Pm @ x + q @ x, np.clip(np.zeros(n), l, u), jac=lambda x: Pm @ x + q,
                 bounds=list(zip(l, u)), method="L-BFGS-B", options=dict(ftol=1e-15, gtol=1e-12, maxiter=10000))
    x = r.x
    free = (x > l + 1e-9) & (x < u - 1e-9)
    xb = np.where(free, 0.0, np.clip(x, l, u))
    if free.any():
        xf = np.linalg.solve(Pm[np.ix_(free, free)], -(q[free] + Pm[np.ix_(free, ~free)] @ xb[~free]))
        x = xb.copy(); x[free] = xf
    return x


@pytest.mark.parametrize("n,seed", [(3, 0), (8, 1), (16, 2), (40, 3), (64, 4)])
def test_boxcqp_matches_independent_qp(oracle, n, seed):
    rng = np.random.default_rng(seed)
    Pm = spd(n, 100.0, seed + 10)
    q = rng.standard_normal(n) * 3
    l = -np.abs(rng.standard_normal(n)) * 0.3
    u = np.abs(rng.standard_normal(n)) * 0.3
    l[::5] = -np.inf
    u[1::7] = np.inf
    Plow = np.tril(Pm) + np.triu(np.full((n, n), np.nan), 1)      # only the lower triangle may be read (QP:109)
    st, x, iters = oracle.solve_box_qp(Plow, q, l, u)
    assert st == 0 and iters >= 1
    xr = qp_reference(Pm, q, l, u)
    assert np.all(x >= l) and np.all(x <= u)
    assert np.allclose(x, xr, rtol=1e-6, atol=1e-8)
    # KKT: gradient sign pattern
    g = Pm @ x + q
    assert np.all(np.abs(g[(x > l) & (x < u)]) < 1e-9)
    assert np.all(g[x == l] >= -1e-9) and np.all(g[x == u] <= 1e-9)


def test_boxcqp_unbounded_is_plain_solve(oracle):
    Pm = spd(12, 50.0, 3)
    q = np.arange(12.0) - 5
    st, x, iters = oracle.solve_box_qp(np.tril(Pm), q, np.full(12, -np.inf), np.full(12, np.inf))
    assert st == 0 and iters == 0
    assert np.allclose(x, np.linalg.solve(Pm, -q), rtol=1e-12)


def test_boxcqp_float(oracle):
    p = P.tq()
    st, x, _ = oracle.solve_box_qp(p["P"], p["q"], p["l"], p["u"], dtype=np.float32)
    assert st == 0 and np.allclose(x, p["expect"], rtol=1e-5)


def test_lm_same_minimiser_as_scipy(oracle):
    """Independent LM/TRF implementation finds the same minimiser on the synthetic families."""
    from scipy.optimize import least_squares
    w = P.tanh_linear(400, 8)
    A, b = w["A"], w["b"]

    def f(x, y):
        y[:] = np.tanh(A @ x) - b
    s = oracle.default_settings(); s.absTolerance = 1e-10
    res, x = oracle.optimize(f, 400, w["x0"], settings=s)
    ref = least_squares(lambda x: np.tanh(A @ x) - b, w["x0"], xtol=1e-15, ftol=1e-15, gtol=1e-15)
    assert res.status >= 0
    assert np.allclose(x, ref.x, rtol=1e-6, atol=1e-9)
    assert np.isclose(res.residual, 2 * ref.cost, rtol=1e-9)
    # bounded variant hits BOXCQP
    lo = w["xstar"] - 0.02; up = w["xstar"] + 0.5
    x0 = np.clip(w["x0"], lo, up)
    res, x = oracle.optimize(f, 400, x0, lower=lo, upper=up, settings=s)
    ref = least_squares(lambda x: np.tanh(A @ x) - b, x0, bounds=(lo, up), xtol=1e-15, ftol=1e-15, gtol=1e-15)
    assert res.status >= 0 and np.all(x >= lo) and np.all(x <= up)
    assert np.allclose(x, ref.x, rtol=1e-5, atol=1e-7)


def test_openblas_backend_equivalent(oracle):
    """cpu_baseline leg: the oracle with syrk/gemv/ger/posvx routed to scipy's OpenBLAS converges to the same point."""
    if not oracle.load_openblas(threads=2):
        pytest.skip("scipy OpenBLAS not found")
    w = P.tanh_linear(2000, 16)
    A, b = w["A"], w["b"]

    def f(x, y):
        y[:] = np.tanh(A @ x) - b
    s = oracle.default_settings(); s.absTolerance = 1e-10
    r0, x0 = oracle.optimize(f, 2000, w["x0"], settings=s)
    r1, x1 = oracle.optimize(f, 2000, w["x0"], settings=s, use_openblas=True)
    assert r0.status >= 0 and r1.status >= 0
    assert np.allclose(x0, x1, rtol=1e-6, atol=1e-9)      # two roundings of the same path: north-star tolerance
    assert np.isclose(r0.residual, r1.residual, rtol=1e-10)


def test_native_workload_callbacks(oracle):
    """oracle/workloads_cpu.c callbacks == the numpy definitions; RNG twin is bit-identical."""
    import ctypes as C
    u = oracle.uniform(10, 1000, offset=37)
    assert np.array_equal(u, P.splitmix64_uniform(10, 1000, 37))
    w = P.tanh_linear(300, 8)
    ctx = oracle.TanhLinearCtx(w["A"].ctypes.data, w["b"].ctypes.data)
    s = oracle.default_settings(); s.absTolerance = 1e-10
    r0, x0 = oracle.optimize(oracle.native_fn("wlc_tanh_linear_f"), 300, w["x0"], settings=s, fctx=C.addressof(ctx))
    A, b = w["A"], w["b"]

    def f(x, y):
        y[:] = np.tanh(A @ x) - b
    r1, x1 = oracle.optimize(f, 300, w["x0"], settings=s)
    assert np.allclose(x0, x1, rtol=1e-9) and r0.status == r1.status
    # analytic jacobian callback
    r2, x2 = oracle.optimize(oracle.native_fn("wlc_tanh_linear_f"), 300, w["x0"], settings=s, fctx=C.addressof(ctx),
                             g=oracle.native_fn("wlc_tanh_linear_g"), gctx=C.addressof(ctx))
    assert np.allclose(x2, x1, rtol=1e-7) and r2.gCalls > 0
    gs = P.gauss_sum(500, K=2)
    gctx = oracle.GaussSumCtx(gs["t"].ctypes.data, gs["data"].ctypes.data)
    r3, x3 = oracle.optimize(oracle.native_fn("wlc_gauss_sum_f"), 500, gs["x0"], lower=gs["lower"], upper=gs["upper"],
                             fctx=C.addressof(gctx))
    assert r3.status >= 0 and np.allclose(x3, gs["truth"], rtol=2e-2, atol=2e-3)


@pytest.mark.parametrize("n,badscale", [(3, False), (8, False), (8, True), (16, True)])
def test_fused_float_posvx_is_the_float_posvx_up_to_rounding(oracle, n, badscale):
    """lmo_posvx_fused_s (the bit-for-bit reference of the device's posvx_rows) against the oracle's float ?posvx: same info,
    same ?laqsy decision, solutions within a few float roundings of each other relative to the refined double solution."""
    rng = np.random.default_rng(5 + n)
    for trial in range(20):
        G = rng.standard_normal((2 * n, n))
        if badscale:
            G = G * np.logspace(-2, 2, n)[None, :]
        A = (G.T @ G + 1e-3 * np.eye(n)).astype(np.float32)
        b = rng.standard_normal(n).astype(np.float32)
        info, x, eq = oracle.posvx_fused_s(A, b)
        o = oracle.posvx(A.astype(np.float64), b, dtype=np.float32)
        assert info == 0 and o["info"] in (0, n + 1) and eq == (o["equed"] == "Y")
        xr = np.linalg.solve(A.astype(np.float64), b.astype(np.float64))
        scale = np.linalg.norm(xr)
        assert np.linalg.norm(x - xr) <= 4 * np.linalg.norm(o["x"] - xr) + 4e-7 * scale
    A = np.diag([1.0, -2.0, 3.0]).astype(np.float32)
    assert oracle.posvx_fused_s(A, np.ones(3, dtype=np.float32))[0] == 2      # the second leading minor is not positive
