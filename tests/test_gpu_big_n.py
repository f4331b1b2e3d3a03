"""n above 256: the reference accepts any n (least_squares.d:911 reads n = x.length; the workspace carve :913-926 is generic),
so the drop-in must too. Above 256 the n x n part of a pass runs k_lm_solve_big (csrc/solve_big.h: loops over n, factor in
global memory, 512 threads), J^T J the tile-pair kernel, Broyden passes rewrite J row by row. Compared with the oracle like
every other whole-path test (x rtol 1e-6, residual rtol 1e-9); the same kernel is also forced on small n
(VARIANT_SOLVE_GENERIC) where the tuned kernels provide a second opinion pass by pass."""
import ctypes as C

import numpy as np
import pytest

import mir_optim_amd as M
from mir_optim_amd import workloads as W
import problems as P
from test_gpu_lm import first_noisy_pass

pytestmark = pytest.mark.gpu


def spd(n, seed, cond=1e3):
    rng = np.random.default_rng(seed)
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    d = np.geomspace(1.0, cond, n)
    return (Q * d) @ Q.T


@pytest.mark.parametrize("n", [257, 300, 512, 700])
def test_boxcqp_any_n_matches_oracle(oracle, n):
    Pm = spd(n, n)
    rng = np.random.default_rng(n + 1)
    q = rng.standard_normal(n) * 3
    xu = np.linalg.solve(Pm, -q)
    l = np.where(rng.random(n) < 0.3, xu + 0.05 * np.abs(xu) + 1e-3, -np.inf)     # ~30 % of the lower bounds cut the minimiser off
    u = np.where(rng.random(n) < 0.2, np.maximum(l, xu) + 0.5, np.inf)
    st, x, it = M.solveBoxQP(Pm, q, l, u)
    so, xo, ito = oracle.solve_box_qp(Pm, q, l, u)
    assert int(st) == so == 0
    assert it == ito >= 1
    assert np.array_equal((x == l) | (x == u), (xo == l) | (xo == u))            # same active set
    assert np.allclose(x, xo, rtol=1e-9, atol=1e-11)
    # unconstrained = ?posvx alone
    st, x, it = M.solveBoxQP(Pm, q, np.full(n, -np.inf), np.full(n, np.inf))
    assert int(st) == 0 and it == 0 and np.allclose(x, xu, rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("m,n", [(5000, 300), (7001, 384), (6000, 512), (4000, 1024)])
def test_whole_path_above_256_matches_oracle(oracle, m, n):
    w = P.tanh_linear(m, n)
    prob = W.TanhLinear(w["A"], w["b"])
    s = M.LeastSquaresSettings(); s.absTolerance = 1e-9
    st = M.Stats()
    res, x = prob.solve(w["x0"], settings=s, stats=st, batched=True)
    so = oracle.default_settings(); so.absTolerance = 1e-9
    ctx = oracle.TanhLinearCtx(w["A"].ctypes.data, w["b"].ctypes.data)
    ro, xo = oracle.optimize(oracle.native_fn("wlc_tanh_linear_f"), m, w["x0"], settings=so, fctx=C.addressof(ctx),
                             use_openblas=oracle.load_openblas(threads=8))
    assert int(res.status) >= 0 and ro.status >= 0
    assert st.jacobian_broyden >= 2
    assert np.allclose(x, xo, rtol=1e-6, atol=1e-9), np.abs(x - xo).max()
    assert np.isclose(res.residual, ro.residual, rtol=1e-9)


def test_bounded_whole_path_above_256(oracle):
    m, n = 4000, 300
    w = P.tanh_linear(m, n)
    lo = w["xstar"] - 0.5
    up = w["xstar"] + 0.5
    lo[::7] = w["xstar"][::7] + 0.02
    x0 = np.clip(w["x0"], lo, up)
    prob = W.TanhLinear(w["A"], w["b"])
    s = M.LeastSquaresSettings(); s.absTolerance = 1e-9
    st = M.Stats()
    res, x = prob.solve(x0, l=lo, u=up, settings=s, stats=st, batched=True)
    so = oracle.default_settings(); so.absTolerance = 1e-9
    ctx = oracle.TanhLinearCtx(w["A"].ctypes.data, w["b"].ctypes.data)
    ro, xo = oracle.optimize(oracle.native_fn("wlc_tanh_linear_f"), m, x0, lower=lo, upper=up, settings=so, fctx=C.addressof(ctx))
    assert st.qp_active_set_passes > 0 and int(res.status) >= 0 and ro.status >= 0
    assert np.all(x >= lo) and np.all(x <= up)
    assert np.allclose(x, xo, rtol=1e-6, atol=1e-9), np.abs(x - xo).max()
    assert np.isclose(res.residual, ro.residual, rtol=1e-9)


@pytest.mark.parametrize("m,n,bounded", [(9000, 100, False), (12000, 200, False), (6000, 48, True), (8000, 160, True)])
def test_generic_solve_kernel_on_small_n_follows_the_tuned_kernels(m, n, bounded):
    w = P.tanh_linear(m, n)
    prob = W.TanhLinear(w["A"], w["b"])
    lo = up = None
    x0 = w["x0"]
    if bounded:
        lo = w["xstar"] - 0.5; up = w["xstar"] + 0.5
        lo[::4] = w["xstar"][::4] + 0.02
        x0 = np.clip(x0, lo, up)
    s = M.LeastSquaresSettings(); s.absTolerance = 1e-9
    out = []
    for variant in (0, M.VARIANT_SOLVE_GENERIC):
        tr, st = M.Trace(4096), M.Stats()
        r, x = prob.solve(x0, l=lo, u=up, settings=s, trace=tr, stats=st, batched=True, variant=variant)
        out.append((r, x, tr.records(), st))
    (r0, x0_, t0, s0), (r1, x1, t1, s1) = out
    assert int(r0.status) >= 0 and int(r1.status) >= 0
    if bounded:
        assert s0.qp_active_set_passes > 0 and s1.qp_active_set_passes > 0
    assert np.allclose(x1, x0_, rtol=1e-6, atol=1e-9) and np.isclose(r1.residual, r0.residual, rtol=1e-9)
    K = min(first_noisy_pass(t0), first_noisy_pass(t1), len(t0), len(t1))
    assert K >= 4
    for a, b in zip(t0[:K], t1[:K]):
        assert a[:2] == b[:2] and np.isclose(a[2], b[2], rtol=1e-6) and np.isclose(a[3], b[3], rtol=1e-7, atol=1e-300), (a, b)


@pytest.mark.parametrize("m,n", [(3000, 320), (2501, 300), (1000, 513)])
def test_jtj_above_256(m, n):
    rng = np.random.default_rng(m + n)
    J = rng.integers(-4, 5, size=(m, n)).astype(np.float64)
    y = rng.integers(-3, 4, size=m).astype(np.float64)
    JJ, Jy, _, _ = M.jtj(J, y)
    assert np.array_equal(JJ, J.T @ J) and np.array_equal(Jy, J.T @ y)
    yo = y + rng.integers(-2, 3, size=m)
    dx = rng.integers(-2, 3, size=n).astype(np.float64) / 4
    JJ2, Jy2, Jn, _ = M.jtj(J, y, yo, dx)
    u = -(1.0 / (dx @ dx)) * ((yo - y) + J @ dx)
    Jr = J + np.outer(u, dx)
    assert np.allclose(Jn, Jr, rtol=1e-14, atol=1e-13)
    assert np.allclose(JJ2, Jr.T @ Jr, rtol=1e-12, atol=1e-9) and np.allclose(Jy2, Jr.T @ y, rtol=1e-12, atol=1e-9)


@pytest.mark.parametrize("n", [257, 320, 512])
def test_boxcqp_any_n_float32(oracle, n):
    """The same entry in single precision (mir_solve_box_qp_s; the reference is a template over T, boxcqp.d:122): k_box_qp_big<float>
    against the float oracle. A well-conditioned matrix (float Cholesky + two refinement steps decide the last digits)."""
    Pm = spd(n, n, cond=30.0)
    rng = np.random.default_rng(n + 3)
    q = rng.standard_normal(n) * 3
    xu = np.linalg.solve(Pm, -q)
    l = np.where(rng.random(n) < 0.3, xu + 0.05 * np.abs(xu) + 1e-2, -np.inf)
    u = np.where(rng.random(n) < 0.2, np.maximum(l, xu) + 0.5, np.inf)
    st, x, it = M.solveBoxQP(Pm, q, l, u, dtype=np.float32)
    so, xo, ito = oracle.solve_box_qp(Pm, q, l, u, dtype=np.float32)
    assert int(st) == so == 0 and it >= 1 and ito >= 1
    l32, u32 = l.astype(np.float32), u.astype(np.float32)
    assert np.array_equal((x == l32) | (x == u32), (xo == l32) | (xo == u32))    # same active set
    assert np.allclose(x, xo, rtol=2e-4, atol=2e-5)
    xq = np.asarray(x, dtype=np.float64)                                          # and it IS the constrained minimiser: KKT to float accuracy
    g = Pm @ xq + q
    free = (x != l32) & (x != u32)
    assert np.abs(g[free]).max() <= 2e-3 * (np.abs(Pm) @ np.abs(xq) + np.abs(q)).max()


@pytest.mark.parametrize("m,n", [(900, 264), (700, 320)])
def test_whole_path_above_256_float32_reference_abi(oracle, m, n):
    """mir_optimize_least_squares_s above n = 256 (host residual callback, finite differences by the library): the any-n solve
    kernel, the tile-pair J^T J and the rewriting Broyden kernels in single precision, against the float oracle on the same
    callback. Float finite differences are noisy: x to 5e-3, the residual to 5e-3 (the tolerances of the float host sweeps)."""
    w = P.tanh_linear(m, n)
    A32, b32 = w["A"].astype(np.float32), w["b"].astype(np.float32)

    def f(x, y):
        y[:] = np.tanh(A32 @ x) - b32

    s = M.LeastSquaresSettings(np.float32); s.maxIterations = 6
    so = oracle.default_settings(np.float32); so.maxIterations = 6
    x0 = w["x0"].astype(np.float32)
    res, x = M.optimizeLeastSquares(f, m, x0.copy(), settings=s, dtype=np.float32)
    ro, xo = oracle.optimize(f, m, x0.copy(), settings=so, dtype=np.float32)
    assert int(res.status) >= -1 and ro.status >= -1
    assert res.residual < 0.05 * float(np.sum((np.tanh(A32 @ x0) - b32) ** 2))   # it did minimise
    assert np.abs(np.asarray(x, dtype=np.float64) - np.asarray(xo, dtype=np.float64)).max() <= 5e-3 * max(1.0, float(np.abs(xo).max()))
    assert abs(res.residual - ro.residual) <= 5e-3 * abs(ro.residual) + 1e-6


@pytest.mark.parametrize("m,n,bounded", [(3000, 300, True), (4000, 520, False), (2500, 1024, True), (3000, 1100, False)])
def test_helper_workgroups_change_nothing_but_time(m, n, bounded):
    """Above n = 256 every damping level's workgroup has helpers (csrc/solve_coop.h): the look-ahead blocks of ?potrf, the
    residual / prediction products and the copies of J^T J are shared out. With them and without
    (VARIANT_SOLVE_ONE_WORKGROUP): the same status, iterations and minimiser (the W partial sums of a product are added in a
    different order than one workgroup's loop: x to 2e-8 absolute, the residual to 1e-11), up to n = 1100 (above 1024 ?potrs is the
    block-step routine again) and with the BOXCQP active-set loop running reduced systems through the same jobs."""
    w = P.tanh_linear(m, n)
    lo = np.full(n, -np.inf); up = np.full(n, np.inf)
    x0 = w["x0"]
    if bounded:
        lo = w["xstar"] - 0.5; up = w["xstar"] + 0.5
        lo[::7] = w["xstar"][::7] + 0.02
        x0 = np.clip(x0, lo, up)
    prob = W.TanhLinear(w["A"], w["b"])
    s = M.LeastSquaresSettings(); s.absTolerance = 1e-7
    outs = []
    for variant in (0, M.VARIANT_SOLVE_ONE_WORKGROUP):
        st = M.Stats()
        res, x = prob.solve(x0, lo, up, settings=s, stats=st, batched=True, variant=variant)
        assert int(res.status) >= 0, res
        outs.append((res, x, st.qp_active_set_passes if hasattr(st, "qp_active_set_passes") else 0))
    (ra, xa, _), (rb, xb, _) = outs
    assert ra.status == rb.status and ra.iterations == rb.iterations
    assert np.allclose(xa, xb, rtol=0, atol=2e-8), np.abs(xa - xb).max()          # (|x| ~ 1; both stopped by absTolerance = 1e-7)
    assert np.isclose(ra.residual, rb.residual, rtol=1e-11)
    if bounded:
        assert np.array_equal((xa == lo) | (xa == up), (xb == lo) | (xb == up)) and ((xa == lo) | (xa == up)).sum() >= n // 8


def test_two_host_threads_solve_above_256_concurrently():
    """Two host threads, each with its own workspace and stream, run n = 520 fits at the same time: the helper workgroups of
    their solve launches share the chip (every launch is main + helpers per damping level); each thread gets the bits of the
    same fit run alone."""
    import threading
    m, n = 3000, 520
    w = P.tanh_linear(m, n)
    s = M.LeastSquaresSettings(); s.absTolerance = 1e-7
    probs = [W.TanhLinear(w["A"], w["b"]) for _ in range(2)]
    ref_res, ref_x = probs[0].solve(w["x0"], settings=s, batched=True)
    out, err = {}, []

    def work(k):
        try:
            for _ in range(3):
                out[k] = probs[k].solve(w["x0"], settings=s, batched=True)
        except Exception as e:       # noqa: BLE001
            err.append(e)
    th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    [t.start() for t in th]; [t.join() for t in th]
    assert not err, err
    for k in range(2):
        res, x = out[k]
        assert res.status == ref_res.status and res.iterations == ref_res.iterations and res.fCalls == ref_res.fCalls
        assert x.tobytes() == ref_x.tobytes() and res.residual == ref_res.residual


def test_absent_helpers_degrade_to_the_one_workgroup_solve_not_to_an_error():
    """The any-n solve waits for its helper workgroups with bounded spins (csrc/solve_coop.h, 5 s). A helper that does not answer is
    a SCHEDULING fact (a GPU shared with other work, a starved queue), not a numeric one: with the helpers NOT launched
    (VARIANT_DEBUG_HELPERS_ABSENT; one damping level per pass) the first job times out, the rescue launch behind the kernel
    boundary solves the pass again on one workgroup, the statistics count the stall, the rest of the solve does not ask for
    helpers again -- and the caller gets the bits of VARIANT_SOLVE_ONE_WORKGROUP, not numericError, not a hang."""
    import time
    w = P.tanh_linear(1500, 300)
    prob = W.TanhLinear(w["A"], w["b"])
    st = M.Stats()
    t0 = time.perf_counter()
    res, x = prob.solve(w["x0"], batched=True, stats=st, variant=M.VARIANT_DEBUG_HELPERS_ABSENT | M.VARIANT_NO_SPECULATION)
    dt = time.perf_counter() - t0
    assert 4.0 < dt < 30.0, dt                                # ONE stall of 5 s, not one per pass
    assert st.coop_timeouts == 1
    st1 = M.Stats()
    ref, xr = prob.solve(w["x0"], batched=True, stats=st1, variant=M.VARIANT_SOLVE_ONE_WORKGROUP | M.VARIANT_NO_SPECULATION)
    assert int(ref.status) >= 0 and st1.coop_timeouts == 0
    assert (res.status, res.iterations, res.fCalls, res.residual, res.lambda_) == (ref.status, ref.iterations, ref.fCalls, ref.residual, ref.lambda_)
    assert x.tobytes() == xr.tobytes()
    # and the workspace is usable afterwards: the stale words of the failed launch satisfy nobody
    res2, x2 = prob.solve(w["x0"], batched=True)
    assert int(res2.status) >= 0
