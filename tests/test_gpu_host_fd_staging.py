"""Host-callback finite differences (the reference ABI, LS:1018-1049) must never touch the live residual vector.

Round-1 defect: the FD task staged f(x + h e_j) / f(x - h e_j) into the fixed physical buffers B.mB / B.ytmp,
but the solver swaps the ROLES of its two m-vectors on every accepted step (the reference swaps contents, LS:1136), so
after an odd number of accepted steps the live residual y lived in B.mB and a full refresh overwrote it with
f(x + h e_last). J^T y was then off by h (J^T J)[:, last] and the step by about -h in the last coordinate: on a
ZERO-RESIDUAL problem -- where finite-difference noise in J^T y vanishes with |y| -- the iteration stalls near
|x - x*| ~ 1e-8 and never reaches fConverged. With maxAge = 1 every accepted step is followed by a full refresh, so both
parities of the swap are exercised many times."""
import ctypes as C

import numpy as np
import pytest

import mir_optim_amd as M
import problems as P

pytestmark = pytest.mark.gpu


def zero_residual_problem(m, n, seed):
    rng = np.random.default_rng(seed)
    A = rng.uniform(-1, 1, size=(m, n)) * np.sqrt(3.0 / n)
    xs = rng.uniform(-1, 1, size=n)
    b = np.tanh(A @ xs)

    def f(x, y):
        y[:] = np.tanh(A @ x) - b
    x0 = xs + 0.2 * rng.uniform(-1, 1, size=n)
    return f, xs, x0


@pytest.mark.parametrize("m,n,seed", [(300, 5, 1), (64, 3, 2), (500, 8, 3)])
@pytest.mark.parametrize("threads", [0, 3])
def test_full_refresh_after_every_accepted_step_converges_to_machine_precision(oracle, m, n, seed, threads):
    f, xs, x0 = zero_residual_problem(m, n, seed)
    s = M.LeastSquaresSettings(); s.maxAge = 1                       # LS:945: every Jacobian is a full FD refresh
    tr = M.Trace()
    opt = M.GpuOptions()
    opt.trace = C.pointer(tr.header)
    tm = None
    if threads:
        from concurrent.futures import ThreadPoolExecutor
        pool = ThreadPoolExecutor(max_workers=threads)
        ids = {}
        import threading
        lock = threading.Lock()

        def tm(count, task):                                          # LS:184-215: FD columns on a pool's threads
            def run(i):
                with lock:
                    tid = ids.setdefault(threading.get_ident(), len(ids))
                task(threads, tid % threads, i)
            list(pool.map(run, range(count)))
    res, x = M.optimizeLeastSquares(f, m, x0.copy(), settings=s, options=opt, tm=tm)
    so = oracle.default_settings(); so.maxAge = 1
    ev = []
    ro, xo = oracle.optimize(f, m, x0.copy(), settings=so, trace=lambda *a: ev.append(a))
    got = tr.records()
    # both reach the exact minimiser (zero residual): fConverged or xConverged depending on the last roundings
    assert ro.status in (1, 3) and int(res.status) in (1, 3), (res, ro.status)
    assert np.abs(x - xs).max() < 1e-12 and np.abs(xo - xs).max() < 1e-12
    assert res.residual < 1e-28 and ro.residual < 1e-28
    # the first accepted passes (both swap parities) agree far below the 1.5e-8 the defect injected; later ones amplify
    # the finite-difference noise of J quadratically and are only compared through the final answer
    acc_g = [r for r in got if r[0] == 3][:3]
    acc_o = [r for r in ev if r[0] == 3][:3]
    assert len(acc_g) == len(acc_o) == 3
    for k, (g, e) in enumerate(zip(acc_g, acc_o)):
        assert g[1] == e[1] and np.isclose(g[3], e[3], rtol=1e-6 if k < 2 else 1e-2), (g, e)
    assert abs(int(res.iterations) - int(ro.iterations)) <= 1


def test_default_ageing_host_callbacks_odd_number_of_accepts_before_refresh(oracle):
    """Default maxAge = 2n with a tiny n: the second full refresh comes after 2n + 1 Jacobian updates, i.e. after an
    odd number of accepted steps for any n -- the parity that used to clobber y. Zero-residual Rosenbrock-like chain."""
    n = 3

    def f(x, y):
        y[0] = 10 * (x[1] - x[0] ** 2); y[1] = 1 - x[0]; y[2] = 10 * (x[2] - x[1] ** 2); y[3] = 1 - x[1]
    x0 = np.array([-1.2, 1.0, 0.8])
    res, x = M.optimize(f, 4, x0.copy())
    ro, xo = oracle.optimize(f, 4, x0.copy())
    assert ro.status == 3 and int(res.status) in (1, 3)
    assert np.abs(x - 1.0).max() < 1e-12 and np.abs(xo - 1.0).max() < 1e-12
    assert abs(int(res.iterations) - int(ro.iterations)) <= 2


# ---------------------------------------------------------------------------------------------------------------------
# Round 3: the reference-ABI finite-difference refresh stages through a pinned POINT-MAJOR panel -- the caller's f writes
# f(x + h e_j), f(x - h e_j) straight into rows 2j, 2j + 1, one asynchronous copy per column pair on a copy stream, no
# lock and no stream synchronisation inside the task (LS:1019-1048 run concurrently on the manager's threads, LS:184-215),
# ONE coalesced conversion at the end. MIR_LSQ_VARIANT_FD_HOST_COLUMNS keeps the round-2 path (per-slot staging vectors,
# strided column write + synchronisation per task under a mutex) as the restatement to compare with.
# ---------------------------------------------------------------------------------------------------------------------
class _HostCtx(C.Structure):
    _fields_ = [("A", C.c_void_p), ("b", C.c_void_p)]


def _native_host_problem(m, n):
    from mir_optim_amd import api, workloads as W
    w = W.tanh_linear_data(m, n)
    ctx = _HostCtx(w["A"].ctypes.data, w["b"].ctypes.data)
    WL = api.workloads_lib()
    f = C.cast(WL.wl_tanh_linear_f_host_serial_d, C.c_void_p).value
    tm = C.cast(WL.wl_omp_thread_manager, C.c_void_p).value
    return w, ctx, f, tm


def _solve_native(w, ctx, f, tm, threads, variant=0, stats=None, tol=1e-9, gpu_entry=True):
    from mir_optim_amd import api
    L = api.lib()
    s = M.LeastSquaresSettings(); s.absTolerance = tol
    n, m = w["n"], w["m"]
    x = w["x0"].copy()
    lo, up = np.full(n, -np.inf), np.full(n, np.inf)
    nthreads = C.c_int(threads)
    o = M.GpuOptions(); o.variant = variant
    if stats is not None:
        o.stats = C.pointer(stats)
    fn = L.mir_optimize_least_squares_gpu_d
    raw = fn(C.byref(s), m, n, x.ctypes.data, lo.ctypes.data, up.ctypes.data, C.byref(o), C.addressof(ctx), C.c_void_p(f),
             None, None, C.cast(C.pointer(nthreads), C.c_void_p) if tm else None, C.c_void_p(tm) if tm else None)
    return api.LeastSquaresResult(raw), x


@pytest.mark.parametrize("m,n,threads", [(20000, 32, 8), (5001, 7, 16), (3000, 64, 3)])
def test_pinned_panel_refresh_equals_column_path_bitwise_and_the_oracle(oracle, m, n, threads):
    """A native OpenMP thread manager (the C counterpart of the D task-pool overload LS:184-215; threads >= n and < n both:
    LS:1022 picks the scratch slot by task index or by thread id) with a single-threaded native host residual."""
    w, ctx, f, tm = _native_host_problem(m, n)
    st1, st0 = M.Stats(), M.Stats()
    r1, x1 = _solve_native(w, ctx, f, tm, threads, stats=st1)
    r0, x0 = _solve_native(w, ctx, f, tm, threads, variant=M.VARIANT_FD_HOST_COLUMNS, stats=st0)
    assert (int(r1.status), r1.iterations, r1.fCalls, r1.residual, r1.lambda_) == (int(r0.status), r0.iterations, r0.fCalls, r0.residual, r0.lambda_)
    assert np.array_equal(x1, x0)
    rs, xs = _solve_native(w, ctx, f, None, 0)                       # no manager: the serial loop LS:947-951
    assert np.array_equal(xs, x1) and rs.fCalls == r1.fCalls
    assert st1.fd_host_columns == n * st1.jacobian_full and st1.fd_host_f_ms > 0
    so = oracle.default_settings(); so.absTolerance = 1e-9
    octx = oracle.TanhLinearCtx(w["A"].ctypes.data, w["b"].ctypes.data)
    ro, xo = oracle.optimize(oracle.native_fn("wlc_tanh_linear_f"), m, w["x0"], settings=so, fctx=C.addressof(octx))
    assert ro.status >= 0 and r1.status >= 0
    assert np.allclose(x1, xo, rtol=1e-6, atol=1e-9) and np.isclose(r1.residual, ro.residual, rtol=1e-9)


def test_reference_entry_point_with_native_thread_manager(oracle):
    """The unmodified reference signature (mir_optimize_least_squares_d, LS:705-724: by-value Slices, sret result) with the
    native manager -- what tests/c_harness does from C, here at a size where the panel path matters."""
    from mir_optim_amd import api
    w, ctx, f, tm = _native_host_problem(30000, 16)
    L = api.lib()
    s = M.LeastSquaresSettings(); s.absTolerance = 1e-9
    m, n = w["m"], w["n"]
    x = w["x0"].copy()
    lo, up = np.full(n, -np.inf), np.full(n, np.inf)
    iwork = np.zeros(L.mir_least_squares_iwork_length(m, n) + 4, dtype=np.int32)
    work = api._SliceD(L.mir_least_squares_work_length(m, n), None)          # never dereferenced by this implementation
    nthreads = C.c_int(4)
    raw = L.mir_optimize_least_squares_d(C.byref(s), m, n, x.ctypes.data, lo.ctypes.data, up.ctypes.data, work,
                                         api._SliceD(len(iwork), iwork.ctypes.data), C.addressof(ctx), C.c_void_p(f), None, None,
                                         C.cast(C.pointer(nthreads), C.c_void_p), C.c_void_p(tm))
    res = api.LeastSquaresResult(raw)
    so = oracle.default_settings(); so.absTolerance = 1e-9
    octx = oracle.TanhLinearCtx(w["A"].ctypes.data, w["b"].ctypes.data)
    ro, xo = oracle.optimize(oracle.native_fn("wlc_tanh_linear_f"), m, w["x0"], settings=so, fctx=C.addressof(octx))
    assert int(res.status) == ro.status and res.fCalls == ro.fCalls and res.iterations == ro.iterations
    assert np.allclose(x, xo, rtol=1e-6, atol=1e-9) and np.isclose(res.residual, ro.residual, rtol=1e-9)


def test_thread_manager_that_skips_tasks_is_a_numeric_error(capfd):
    """The manager must run task(i) for every i in [0, count) (LS:575-578). One that stops early (an exception in a binding,
    a cancelled pool) would leave stale columns in J: the refresh fails loudly instead."""
    def f(x, y):
        y[0] = 10 * (x[1] - x[0] ** 2); y[1] = 1 - x[0]

    def lazy_tm(count, task):
        for i in range(count - 1):
            task(1, 0, i)
    res, x = M.optimizeLeastSquares(f, 2, np.array([-1.2, 1.0]), tm=lazy_tm)
    assert res.status == M.LeastSquaresStatus.numericError
    assert "ran 1 of 2 finite-difference tasks" in capfd.readouterr().err

    def raising_tm(count, task):
        task(1, 0, 0)
        raise RuntimeError("pool died")
    with pytest.raises(RuntimeError, match="pool died"):
        M.optimizeLeastSquares(f, 2, np.array([-1.2, 1.0]), tm=raising_tm)
