"""GPU parity, kernel level, through the C ABI (mir_lsq_jtj_*, mir_solve_box_qp_gpu_*):
the fused [Broyden +] J^T J + J^T y kernel against numpy float128/float64 restatements of
LS:1003-1006, 1052, 1065, and the device BOXCQP/posvx against the oracle."""
import numpy as np
import pytest

import mir_optim_amd as M
import problems as P

pytestmark = pytest.mark.gpu


def ref_products(J, y):
    Jl = J.astype(np.longdouble)
    return np.asarray(Jl.T @ Jl, dtype=np.float64), np.asarray(Jl.T @ y.astype(np.longdouble), dtype=np.float64)


@pytest.mark.parametrize("m,n", [(1, 1), (3, 2), (7, 3), (64, 4), (100, 16), (1000, 17), (513, 31), (4096, 32),
                                 (999, 48), (5000, 64), (777, 100), (20000, 128), (40001, 127), (2, 128)])
def test_jtj_and_jty_match_numpy(m, n):
    rng = np.random.default_rng(m * 131 + n)
    J = rng.standard_normal((m, n))
    J[:, 0] = np.arange(m) % 7 - 3.0          # exact-integer asymmetric column: catches transposed C/D maps
    y = rng.standard_normal(m)
    JJ, Jy, J_after, _ = M.jtj(J, y)
    JJr, Jyr = ref_products(J, y)
    scale = np.sqrt(np.outer(np.diag(JJr), np.diag(JJr)))
    assert np.array_equal(J_after, J)
    assert np.array_equal(JJ, JJ.T)
    assert np.max(np.abs(JJ - JJr) / scale) < 1e-13
    assert np.max(np.abs(Jy - Jyr)) < 1e-13 * np.sqrt(m) * np.linalg.norm(J, axis=0).max() * np.abs(y).max()


@pytest.mark.parametrize("m,n", [(1000, 129), (5000, 160), (3001, 200), (20000, 256), (777, 255), (30, 256), (4098, 144),
                                 (2, 160), (10000, 176), (6006, 192), (5000, 208), (3000, 224), (2048, 240), (30002, 256)])
def test_jtj_wide_n_matches_numpy(m, n):
    """128 < n <= 256 (cfg 4's n = 256). n % 16 == 0 with even m runs the eight-wave LDS-DMA ring (jtj_ring8.h), the
    rest the tiled jobs + separate Broyden pass (jtj_wide.h)."""
    rng = np.random.default_rng(m + n)
    J = rng.standard_normal((m, n))
    J[:, 1] = np.arange(m) % 5 - 2.0
    y = rng.standard_normal(m)
    JJ, Jy, J_after, _ = M.jtj(J, y)
    JJr, Jyr = ref_products(J, y)
    scale = np.sqrt(np.outer(np.diag(JJr), np.diag(JJr)))
    assert np.array_equal(J_after, J) and np.array_equal(JJ, JJ.T)
    assert np.max(np.abs(JJ - JJr) / scale) < 1e-13
    assert np.allclose(Jy, Jyr, rtol=1e-11, atol=1e-11 * np.abs(Jyr).max())
    y_old = y + 0.1 * rng.standard_normal(m)
    dx = 0.05 * rng.standard_normal(n)
    JJ, Jy, J_after, _ = M.jtj(J, y, y_old=y_old, dx=dx)
    d = 1.0 / (dx @ dx)
    Jn = J + np.outer(-d * (J @ dx + (y_old - y)), dx)
    assert np.max(np.abs(J_after - Jn)) < 1e-13 * max(1.0, np.abs(Jn).max())
    JJr, Jyr = ref_products(J_after, y)
    assert np.max(np.abs(JJ - JJr) / np.sqrt(np.outer(np.diag(JJr), np.diag(JJr)))) < 1e-13
    Ji = rng.integers(-4, 5, size=(2000, 144)).astype(np.float64); yi = rng.integers(-4, 5, size=2000).astype(np.float64)
    JJ, Jy, _, _ = M.jtj(Ji, yi)
    assert np.array_equal(JJ, Ji.T @ Ji) and np.array_equal(Jy, Ji.T @ yi)


@pytest.mark.parametrize("m,n", [(3001, 96), (3001, 97), (40000, 127), (777, 1), (5000, 31),
                                 # 128 < n <= 256 off the eight-wave ring's grid (n % 16 != 0 or odd m): the plain flavour of k_jtj_fdp8
                                 (3000, 200), (4001, 250), (5000, 129), (4001, 256), (2000, 161), (1, 255), (30001, 208)])
def test_jtj_exact_integers(m, n):
    """small integers: every partial sum is exact, so the result must be bit-exact (odd n: the element-wise producer of jtj_fdp.h)."""
    rng = np.random.default_rng(5)
    J = rng.integers(-8, 9, size=(m, n)).astype(np.float64)
    y = rng.integers(-8, 9, size=m).astype(np.float64)
    JJ, Jy, _, _ = M.jtj(J, y)
    assert np.array_equal(JJ, J.T @ J)
    assert np.array_equal(Jy, J.T @ y)


@pytest.mark.parametrize("m,n", [(5, 2), (1000, 16), (4097, 33), (30000, 128), (2500, 100)])
def test_broyden_fused_update(m, n):
    rng = np.random.default_rng(n)
    J = rng.standard_normal((m, n))
    y_old = rng.standard_normal(m)
    y = y_old + 0.1 * rng.standard_normal(m)
    dx = 0.05 * rng.standard_normal(n)
    JJ, Jy, J_after, _ = M.jtj(J, y, y_old=y_old, dx=dx)
    # LS:1002-1006 in the reference's operation order
    d = 1.0 / (dx @ dx)
    mB = y_old - y
    mB = J @ dx + mB
    mB = -d * mB
    Jn = J + np.outer(mB, dx)
    assert np.max(np.abs(J_after - Jn)) < 1e-13 * max(1.0, np.abs(Jn).max())
    JJr, Jyr = ref_products(J_after, y)         # products must use the UPDATED J
    scale = np.sqrt(np.outer(np.diag(JJr), np.diag(JJr)))
    assert np.max(np.abs(JJ - JJr) / scale) < 1e-13
    assert np.allclose(Jy, Jyr, rtol=1e-11, atol=1e-11 * np.abs(Jyr).max())
    # secant condition J_new dx = y - y_old
    assert np.allclose(J_after @ dx, y - y_old, rtol=1e-9, atol=1e-12)


def test_jtj_float32():
    rng = np.random.default_rng(9)
    m, n = 5000, 64
    J = rng.standard_normal((m, n)).astype(np.float32)
    y = rng.standard_normal(m).astype(np.float32)
    JJ, Jy, _, _ = M.jtj(J, y, dtype=np.float32)
    JJr = J.astype(np.float64).T @ J.astype(np.float64)
    assert np.max(np.abs(JJ - JJr)) < 2e-4 * np.abs(JJr).max()
    Ji = rng.integers(-4, 5, size=(777, 40)).astype(np.float32)
    yi = rng.integers(-4, 5, size=777).astype(np.float32)
    JJ, Jy, _, _ = M.jtj(Ji, yi, dtype=np.float32)
    assert np.array_equal(JJ, Ji.T @ Ji) and np.array_equal(Jy, Ji.T @ yi)


@pytest.mark.parametrize("m,n", [(70001, 128), (64, 4), (63, 8), (5000, 100), (20000, 36), (4097, 64), (1, 16), (300000, 128),
                                 (9000, 126), (9000, 33)])
def test_jtj_float32_producer_consumer_kernel(m, n):
    """k_jtj_pc32 (jtj_pc32.h, round 3): the f32 J^T J on v_mfma_f32_16x16x4 with producer / consumer waves, n % 4 == 0; the
    other n (126, 33) still take the register-streaming kernel. Small integers: every partial sum is exact in f32, so the result
    is bit-exact; random data: against float64 numpy at the f32 summation tolerance; and symmetric, J untouched."""
    rng = np.random.default_rng(m + n)
    Ji = rng.integers(-3, 4, size=(m, n)).astype(np.float32)
    Ji[:, 0] = (np.arange(m) % 5 - 2).astype(np.float32)        # asymmetric column: catches a transposed accumulator map
    yi = rng.integers(-3, 4, size=m).astype(np.float32)
    JJ, Jy, J_after, _ = M.jtj(Ji, yi, dtype=np.float32)
    ref = Ji.astype(np.float64).T @ Ji.astype(np.float64)
    if np.abs(ref).max() < 2 ** 24:                                 # every partial sum representable
        assert np.array_equal(JJ, ref.astype(np.float32)) and np.array_equal(Jy, (Ji.astype(np.float64).T @ yi).astype(np.float32))
    assert np.array_equal(J_after, Ji) and np.array_equal(JJ, JJ.T)
    J = rng.standard_normal((m, n)).astype(np.float32)
    y = rng.standard_normal(m).astype(np.float32)
    JJ, Jy, _, _ = M.jtj(J, y, dtype=np.float32)
    JJr = J.astype(np.float64).T @ J.astype(np.float64)
    Jyr = J.astype(np.float64).T @ y.astype(np.float64)
    scale = np.sqrt(np.outer(np.diag(JJr), np.diag(JJr)))
    assert np.max(np.abs(JJ - JJr) / scale) < 3e-6                  # ~ sqrt(m) eps_f32 relative to ||col_i|| ||col_j||
    assert np.max(np.abs(Jy - Jyr)) < 3e-6 * np.sqrt(m) * np.linalg.norm(J.astype(np.float64), axis=0).max() * np.abs(y).max() + 1e-6
    assert np.array_equal(JJ, JJ.T)


@pytest.mark.parametrize("m", [100003, 100000])     # odd m: register-streaming kernel; even m: LDS-DMA ring kernel
def test_jtj_determinism(m):
    rng = np.random.default_rng(1)
    J = rng.standard_normal((m, 128)); y = rng.standard_normal(m)
    yo = y + 0.01 * rng.standard_normal(m); dx = 1e-3 * rng.standard_normal(128)
    a = M.jtj(J, y); b = M.jtj(J, y)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    a = M.jtj(J, y, y_old=yo, dx=dx); b = M.jtj(J, y, y_old=yo, dx=dx)
    assert all(np.array_equal(u, v) for u, v in zip(a[:3], b[:3]))


# ------------------------------------------------------------------ BOXCQP / posvx on the device
def spd(n, cond, seed, scale=None):
    rng = np.random.default_rng(seed)
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    A = (Q * np.logspace(0, np.log10(cond), n)) @ Q.T
    A = (A + A.T) / 2
    if scale is not None:
        A = A * np.outer(scale, scale)
    return A


def refined_solve(A, b):
    """float64 solve polished with residuals accumulated in long double (reference solution)."""
    x = np.linalg.solve(A, b).astype(np.longdouble)
    Al, bl = A.astype(np.longdouble), b.astype(np.longdouble)
    for _ in range(4):
        r = np.asarray(bl - Al @ x, dtype=np.float64)
        x = x + np.linalg.solve(A, r)
    return np.asarray(x, dtype=np.float64)


def test_TQ_reference_unittest():
    p = P.tq()
    st, x, iters = M.solveBoxQP(p["P"], p["q"], p["l"], p["u"])
    assert st == M.BoxQPStatus.solved and iters == 1
    assert np.allclose(x, p["expect"], rtol=1e-14)             # boxcqp.d:401


@pytest.mark.parametrize("n,cond,badscale", [(1, 1, False), (2, 10, False), (3, 1e3, True), (16, 1e6, False),
                                             (33, 1e8, True), (64, 1e3, False), (128, 1e10, False), (128, 1e4, True),
                                             (129, 1e3, False), (200, 1e5, True), (256, 1e6, False), (145, 1e9, True),
                                             (250, 1e10, False), (256, 1e3, True)])
def test_unconstrained_solve_matches_oracle_posvx(oracle, n, cond, badscale):
    A = spd(n, cond, n, np.logspace(-3, 3, n) if badscale else None)
    q = np.random.default_rng(7).standard_normal(n)
    inf = np.full(n, np.inf)
    st, x, iters = M.solveBoxQP(np.tril(A), q, -inf, inf)
    so, xo, io = oracle.solve_box_qp(np.tril(A), q, -inf, inf)
    assert st == M.BoxQPStatus.solved and so == 0 and iters == io == 0
    xr = refined_solve(A, -q)
    err = np.linalg.norm(x - xr) / np.linalg.norm(xr)
    erro = np.linalg.norm(xo - xr) / np.linalg.norm(xr)
    assert err <= 10 * erro + 1e-15                               # same accuracy class (refined solutions)
    assert np.linalg.norm(x - xo) / np.linalg.norm(xo) <= 4 * max(err, erro) + 1e-15


@pytest.mark.parametrize("n,seed", [(3, 0), (8, 1), (16, 2), (40, 3), (64, 4), (128, 5), (150, 6), (208, 7), (256, 8)])
def test_boxcqp_matches_oracle(oracle, n, seed):
    rng = np.random.default_rng(seed)
    Pm = spd(n, 100.0, seed + 10)
    q = rng.standard_normal(n) * 3
    l = -np.abs(rng.standard_normal(n)) * 0.3
    u = np.abs(rng.standard_normal(n)) * 0.3
    l[::5] = -np.inf
    u[1::7] = np.inf
    Plow = np.tril(Pm) + np.triu(np.full((n, n), np.nan), 1)      # only the lower triangle may be read (QP:109)
    st, x, iters = M.solveBoxQP(Plow, q, l, u)
    so, xo, io = oracle.solve_box_qp(Plow, q, l, u)
    assert st == M.BoxQPStatus.solved and so == 0
    assert iters == io >= 1
    assert np.array_equal(x == l, xo == l) and np.array_equal(x == u, xo == u)    # same active set
    assert np.allclose(x, xo, rtol=1e-10, atol=1e-13)


def test_boxcqp_failure_codes(oracle):
    A = np.array([[1.0, 2.0], [2.0, 1.0]])                        # indefinite -> numericError (QP:212)
    st, _, _ = M.solveBoxQP(A, [1.0, 1.0], [-1.0, -1.0], [1.0, 1.0])
    assert st == M.BoxQPStatus.numericError == oracle.solve_box_qp(A, [1.0, 1.0], [-1.0, -1.0], [1.0, 1.0])[0]
    st, x, _ = M.solveBoxQP(np.zeros((0, 0)), [], [], [])
    assert st == M.BoxQPStatus.solved and x.size == 0             # QP:162-163


def test_boxcqp_float(oracle):
    p = P.tq()
    st, x, _ = M.solveBoxQP(p["P"], p["q"], p["l"], p["u"], dtype=np.float32)
    assert st == M.BoxQPStatus.solved and np.allclose(x, p["expect"], rtol=1e-5)
    for n in (96, 200):                                            # LDS fast path / panel path (factor in global memory)
        A = spd(n, 50.0, n)
        q = np.random.default_rng(n).standard_normal(n)
        l, u = np.full(n, -0.05), np.full(n, 0.05)
        st, x, it = M.solveBoxQP(np.tril(A), q, l, u, dtype=np.float32)
        so, xo, io = oracle.solve_box_qp(np.tril(A), q, l, u)       # double oracle: float result within float accuracy
        assert st == M.BoxQPStatus.solved and so == 0 and np.allclose(x, xo, atol=2e-4)
        l32, u32 = l.astype(np.float32), u.astype(np.float32)
        assert np.mean((x == l32) == (xo == l)) > 0.97 and np.mean((x == u32) == (xo == u)) > 0.97


def test_singular_matrix_reports_numeric_error_on_both_paths(oracle):
    """info > 0 from the Cholesky (QP:212): rank-deficient P, n on the LDS path and on the panel path."""
    for n in (64, 200):
        B = np.random.default_rng(n).standard_normal((n, n // 2))
        A = B @ B.T
        A[n - 1, n - 1] = -1.0                                      # a negative pivot at the very end
        q = np.ones(n)
        inf = np.full(n, np.inf)
        st, _, _ = M.solveBoxQP(np.tril(A), q, -inf, inf)
        so = oracle.solve_box_qp(np.tril(A), q, -inf, inf)[0]
        assert st == M.BoxQPStatus.numericError and so == int(M.BoxQPStatus.numericError)


def test_wave_reductions_match_the_plain_butterfly_bit_for_bit():
    """wave_sum / wave_max (DPP row operations + v_readlane; every reduction kernel and the batched kernel end with them) against
    the butterfly on __shfl_xor: same pairs of partial results, so the same bits, on 2048 x 256 lanes x 200 random inputs each
    (a fifth of them denormal-scaled). Run twice: a variant on the permlane swap instructions was right in one process and wrong
    in the next."""
    import ctypes as C
    for _ in range(2):
        bad = (C.c_int * 4)()
        assert M.lib().mir_lsq_selftest_reductions(200, C.byref(bad)) == 0
        assert list(bad) == [0, 0, 0, 0], list(bad)
