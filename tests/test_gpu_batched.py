"""BASELINE cfg 5: batched independent small fits, fp32, one wavefront per problem
(mir_optimize_least_squares_batched_s) vs the oracle's float instantiation, problem by problem."""
import numpy as np
import pytest

import mir_optim_amd as M
from mir_optim_amd import api
import problems as P

pytestmark = pytest.mark.gpu


def make_exp_decay(count, m=512):
    t = np.linspace(0.0, 4.0, m, dtype=np.float32)
    data = np.empty((count, m), dtype=np.float32)
    truth = np.empty((count, 3), dtype=np.float32)
    x0 = np.empty((count, 3), dtype=np.float32)
    for k in range(count):
        u = P.splitmix64_uniform(100 + k, m + 6)                       # per-problem seed = 100 + problem id
        truth[k] = [1.0 + u[0], 0.5 + 2.0 * u[1], 0.2 * u[2]]
        data[k] = truth[k, 0] * np.exp(-t * truth[k, 1]) + truth[k, 2] + 0.01 * (2 * u[6:] - 1)
        x0[k] = truth[k] * (1 + 0.3 * (2 * u[3:6] - 1))
    return t, data, truth, x0


def make_exp3(count, m=512):
    t = np.linspace(0.0, 4.0, m, dtype=np.float32)
    data = np.empty((count, m), dtype=np.float32)
    truth = np.empty((count, 8), dtype=np.float32)
    x0 = np.empty((count, 8), dtype=np.float32)
    for k in range(count):
        u = P.splitmix64_uniform(100 + k, m + 16)
        truth[k] = [1.0 + u[0], 0.3 + 0.2 * u[1], 0.6 + 0.5 * u[2], 1.5 + 0.5 * u[3], 0.4 + 0.3 * u[4], 5.0 + 2 * u[5], 0.1 * u[6], 0.05 * u[7]]
        p = truth[k]
        data[k] = (p[0] * np.exp(-t * p[1]) + p[2] * np.exp(-t * p[3]) + p[4] * np.exp(-t * p[5]) + p[6] + p[7] * t
                   + 0.002 * (2 * u[16:] - 1))
        x0[k] = truth[k] * (1 + 0.05 * (2 * u[8:16] - 1))
    return t, data, truth, x0


def oracle_f(model, t, d):
    t = t.astype(np.float32); d = d.astype(np.float32)
    if model == M.MODEL_EXP_DECAY:
        def f(p, y):
            y[:] = p[0] * np.exp(-t * p[1]) + p[2] - d
    else:
        def f(p, y):
            y[:] = p[0] * np.exp(-t * p[1]) + p[2] * np.exp(-t * p[3]) + p[4] * np.exp(-t * p[5]) + p[6] + p[7] * t - d
    return f


def test_cfg5_exp_decay_matches_float_oracle(oracle):
    count = 64
    t, data, truth, x0 = make_exp_decay(count)
    res, x = M.optimizeLeastSquaresBatched(M.MODEL_EXP_DECAY, x0, t, data)
    assert all(r.status >= 0 for r in res)
    assert np.allclose(x, truth, rtol=0.05, atol=0.02)                    # recovers the generating parameters
    for k in range(count):                                                # every problem against the float oracle
        ro, xo = oracle.optimize(oracle_f(M.MODEL_EXP_DECAY, t, data[k]), 512, x0[k], dtype=np.float32)
        assert ro.status >= 0
        assert np.allclose(x[k], xo, rtol=2e-3, atol=2e-4), (k, x[k], xo)
        assert np.isclose(res[k].residual, ro.residual, rtol=2e-3)


def test_cfg5_full_size_4096_problems(oracle):
    """4096 x (m = 512, n = 8), fp32. Three-exponential fits are ill-conditioned in fp32 (cond(J^T J) ~ 1/eps):
    near the noise floor the damped Cholesky can fail, which the reference algorithm reports as numericError
    (LS:1080-1085) -- the float oracle does so on MORE problems than the wave kernel (its J^T J is a plain sequential
    float sum). Parity here: every problem ends in a terminal status of the reference, reaches the noise floor, and
    on a sample does at least as well as the float oracle."""
    count = 4096
    t, data, truth, x0 = make_exp3(count)
    res, x = M.optimizeLeastSquaresBatched(M.MODEL_EXP3_AFFINE, x0, t, data)
    st = np.array([int(r.status) for r in res])
    resid = np.array([r.residual for r in res])
    S = M.LeastSquaresStatus
    assert set(np.unique(st)) <= {int(S.furtherImprovement), int(S.xConverged), int(S.gConverged), int(S.fConverged), int(S.numericError)}
    noise_floor = 512 * (0.002 ** 2) / 3                                   # E[sum (0.002 (2u-1))^2]
    assert np.all(np.isfinite(resid)) and np.mean(resid < 1.3 * noise_floor) > 0.97
    assert np.mean(st >= 0) > 0.85
    sample = list(range(0, count, 128))
    worse = ofail = gfail = 0
    for k in sample:
        ro, xo = oracle.optimize(oracle_f(M.MODEL_EXP3_AFFINE, t, data[k]), 512, x0[k], dtype=np.float32)
        ofail += ro.status < 0
        gfail += st[k] < 0
        worse += resid[k] > 1.2 * ro.residual + 1e-7
    assert gfail <= ofail and worse == 0


def test_cfg5_exp3_every_problem_both_sides_solve_is_compared(oracle):
    """EXP3_AFFINE, all 4096 problems, per problem (round-2 verdict, "what's weak" 2: the test above only looks at a 32-problem
    sample). The model is ill-conditioned in fp32 -- the float oracle ends 72 % of the problems with numericError (the damped
    Cholesky fails near the noise floor, LS:1080-1085), the wave kernel 7 % -- so the comparison runs over the problems BOTH
    sides solve (about 1100). There, per problem: the objective agrees to 2e-2 (98 % to 3e-3), and the FITTED CURVES agree to
    1e-3 = half the noise amplitude of the data (measured: <= 3.8e-4; scripts/cfg5_exp3_parity.py prints the distribution).
    The parameters themselves are not comparable: three exponentials are not identifiable at this noise level (two runs
    that reach the same curve differ by up to 0.6 relative in a rate)."""
    count = 4096
    t, data, truth, x0 = make_exp3(count)
    res, x = M.optimizeLeastSquaresBatched(M.MODEL_EXP3_AFFINE, x0, t, data)
    st = np.array([int(r.status) for r in res])
    rg = np.array([r.residual for r in res])
    so = np.empty(count, dtype=int); ro_res = np.empty(count); xo = np.empty_like(x)
    with np.errstate(over="ignore", invalid="ignore"):
        for k in range(count):
            r, xk = oracle.optimize(oracle_f(M.MODEL_EXP3_AFFINE, t, data[k]), 512, x0[k], dtype=np.float32)
            so[k], ro_res[k], xo[k] = r.status, r.residual, xk
    both = (st >= 0) & (so >= 0)
    assert both.sum() >= 800 and (st >= 0).sum() >= (so >= 0).sum()
    ratio = rg[both] / ro_res[both]
    assert np.all(np.abs(ratio - 1) <= 2e-2), (ratio.min(), ratio.max())
    assert np.mean(np.abs(ratio - 1) <= 3e-3) >= 0.98
    td = t.astype(np.float64)

    def curve(p):
        p = p.astype(np.float64)
        return (p[:, 0:1] * np.exp(-td * p[:, 1:2]) + p[:, 2:3] * np.exp(-td * p[:, 3:4]) + p[:, 4:5] * np.exp(-td * p[:, 5:6])
                + p[:, 6:7] + p[:, 7:8] * td)
    cd = np.abs(curve(x[both]) - curve(xo[both])).max(axis=1)
    assert cd.max() <= 1e-3 and np.median(cd) <= 1e-4, (cd.max(), np.median(cd))


def oracle_eval(p, t, d):
    return p[0] * np.exp(-t * p[1]) + p[2] * np.exp(-t * p[3]) + p[4] * np.exp(-t * p[5]) + p[6] + p[7] * t - d


def test_batched_bounded_problems_fall_back_to_general_solver(oracle):
    """A finite bound that the step reaches: the wave kernel hands the problem to the general solver (BOXCQP)."""
    count = 8
    t, data, truth, x0 = make_exp_decay(count)
    lo = np.array([-np.inf, 2.6, -np.inf], dtype=np.float32)             # p1 >= 2.6 > every true rate -> active bound
    x0b = x0.copy(); x0b[:, 1] = 3.0
    res, x = M.optimizeLeastSquaresBatched(M.MODEL_EXP_DECAY, x0b, t, data, l=lo)
    assert all(r.status >= 0 for r in res) and np.all(x[:, 1] >= 2.6 - 1e-6)
    for k in (0, 5):
        ro, xo = oracle.optimize(oracle_f(M.MODEL_EXP_DECAY, t, data[k]), 512, x0b[k], lower=lo, dtype=np.float32)
        assert np.allclose(x[k], xo, rtol=5e-3, atol=5e-4) and np.isclose(res[k].residual, ro.residual, rtol=5e-3)


def test_batched_validation_codes():
    t, data, truth, x0 = make_exp_decay(4)
    x0[1, 0] = np.nan
    res, x = M.optimizeLeastSquaresBatched(M.MODEL_EXP_DECAY, x0, t, data)
    assert res[1].status == M.LeastSquaresStatus.badGuess and res[0].status >= 0
    lo = np.array([5.0, -np.inf, -np.inf], dtype=np.float32)
    res, x = M.optimizeLeastSquaresBatched(M.MODEL_EXP_DECAY, np.nan_to_num(x0, nan=1.0), t, data, l=lo)
    assert all(r.status == M.LeastSquaresStatus.badBounds for r in res)
    s = M.LeastSquaresSettings(np.float32); s.minStepQuality = 2.0
    res, x = M.optimizeLeastSquaresBatched(M.MODEL_EXP_DECAY, np.nan_to_num(x0, nan=1.0), t, data, settings=s)
    assert all(r.status == M.LeastSquaresStatus.badMinStepQuality for r in res)


def test_cfg5_pad8_all_4096_problems_match_the_float_oracle(oracle):
    """BASELINE cfg 5 as specified: 4096 independent fits, m = 512, n = 8, fp32, jacobianEpsilon = 2^-11, the
    well-conditioned exponential-decay family padded to n = 8, per-problem seed 100 + id. EVERY problem is compared with
    the oracle's float instantiation (native float callback evaluating the same expression with libm).
    (The statement about the MODEL: this oracle sums sequentially without fusing and uses libm's expf; the statement about the KERNEL
    is the bit-exact test at the end of this file.)
    fp32 tolerance, stated: the two sides round differently (wave-parallel vs sequential sums, det_expf vs libm expf, device sinf)
    and stop on the flat bottom of a noisy fit, where a parameter moves by ~2e-3 between the float and the DOUBLE oracle;
    so per problem: residual rtol 1e-3, |x_gpu - x_oracle| <= 5e-2 max(1, |x|); over the set: 99 % within 5e-3 and the
    median within 1e-4."""
    import ctypes as C

    class Ctx(C.Structure):
        _fields_ = [("t", C.c_void_p), ("data", C.c_void_p)]
    count = 4096
    t, data, truth, x0 = P.cfg5_pad8(count)
    res, x = M.optimizeLeastSquaresBatched(M.MODEL_EXP_DECAY_PAD8, x0, t, data)
    st = np.array([int(r.status) for r in res])
    resid = np.array([r.residual for r in res])
    assert np.all(st >= 0), np.unique(st, return_counts=True)             # well conditioned: no numericError anywhere
    f = oracle.native_fn("wlc_exp_pad8_f_s")
    xo = np.empty_like(x)
    ro_res = np.empty(count)
    ro_st = np.empty(count, dtype=int)
    for k in range(count):
        d = np.ascontiguousarray(data[k])
        ctx = Ctx(t.ctypes.data, d.ctypes.data)
        ro, xk = oracle.optimize(f, 512, x0[k], dtype=np.float32, fctx=C.addressof(ctx))
        xo[k], ro_res[k], ro_st[k] = xk, ro.residual, ro.status
    assert np.all(ro_st >= 0)
    # the objective agrees everywhere ...
    assert np.allclose(resid, ro_res, rtol=1e-3, atol=0), np.abs(resid / ro_res - 1).max()
    # ... and so do the minimisers: to ~1e-5 for the bulk; a handful of problems with a slow decay have a flat valley
    # (p0 exp(-t p1) against the offset p2) on which two fp32 runs that stop at different passes sit up to ~2e-2 apart
    err = (np.abs(x - xo) / np.maximum(1.0, np.abs(xo))).max(axis=1)
    assert np.median(err) <= 1e-4, np.median(err)
    assert np.quantile(err, 0.99) <= 5e-3, np.quantile(err, 0.99)
    assert err.max() <= 5e-2, (err.max(), int(err.argmax()))
    noise_floor = 512 * (0.01 ** 2) / 3
    assert abs(np.mean(resid) / noise_floor - 1) < 0.05                    # and both sit on the noise floor of the data
    it = np.array([r.iterations for r in res])
    assert it.min() >= 2 and it.max() < 1000


@pytest.mark.parametrize("n", [8, 3])
def test_batched_solve_rows_against_the_float_oracle_posvx(oracle, n):
    """posvx_rows (one matrix row per lane, batched_kernel.h) on its own against the oracle's float ?posvx('E','L'), system by
    system: well-scaled and badly scaled SPD systems (the second half equilibrates: scond < 0.1), and systems whose leading
    minor of order k is not positive (info = k on both sides). The kernel fuses every multiply-add, the oracle's C does not:
    the solutions agree to rounding -- both are refined, so to a few ulp of the solution norm times a modest factor."""
    rng = np.random.default_rng(11 + n)
    count = 256
    P = np.zeros((count, n, n), dtype=np.float32)
    b = rng.standard_normal((count, n)).astype(np.float32)
    for p in range(count):
        G = rng.standard_normal((2 * n, n))
        if p >= count // 2:
            G = G * np.logspace(-2, 2, n)[None, :]                 # diagonal spread 1e8: ?laqsy scales
        A = G.T @ G + 1e-3 * np.eye(n)
        if p % 16 == 5:                                            # not positive definite from minor k on
            k = 1 + (p // 16) % n
            A[k - 1, k - 1] = -abs(A[k - 1, k - 1])
        P[p] = A
    x, info = M.batchedPosvx(P, b)
    worst = 0.0
    for p in range(count):
        o = oracle.posvx(P[p].astype(np.float64), b[p], dtype=np.float32)
        oi = 0 if o["info"] == n + 1 else o["info"]                # rcond < eps is accepted by the caller, boxcqp.d:212
        assert info[p] == oi, (p, info[p], o["info"])
        if oi != 0:
            assert not x[p].any()
            continue
        assert (o["equed"] == "Y") == (p >= count // 2) or p % 16 == 5
        err = np.linalg.norm(x[p] - o["x"]) / np.linalg.norm(o["x"])
        worst = max(worst, err)
        xr = np.linalg.solve(P[p].astype(np.float64), b[p].astype(np.float64))
        assert np.linalg.norm(x[p] - xr) <= 4 * np.linalg.norm(o["x"] - xr) + 1e-6 * np.linalg.norm(xr), p
    assert worst < 2e-4, worst


@pytest.mark.parametrize("model,maker", [(M.MODEL_EXP_DECAY_PAD8, "pad8"), (M.MODEL_EXP3_AFFINE, "exp3"), (M.MODEL_EXP_DECAY, "decay")])
def test_lambda_ladder_takes_the_steps_of_the_one_by_one_loop(model, maker):
    """By default one damped solve serves lambda and the three values the rejection rule (LS:1101-1106, 1125-1130) would give
    it next; with MIR_LSQ_BATCHED_NO_LADDER every solve is for one lambda, as the reference's loop. Same x bits, iterations,
    fCalls, status, residual and lambda on every problem."""
    count = 1024
    t, data, truth, x0 = {"pad8": P.cfg5_pad8, "exp3": make_exp3, "decay": make_exp_decay}[maker](count, 512)
    out = []
    for variant in (0, M.BATCHED_NO_LADDER):           # per call (mir_lsq_batched_options.variant): nothing is process-wide
        res, x = M.optimizeLeastSquaresBatched(model, x0, t, data, settings=M.LeastSquaresSettings(np.float32), variant=variant)
        out.append((x.view(np.uint32).copy(), [(int(r.status), r.iterations, r.fCalls, np.float32(r.residual).view(np.uint32),
                                               np.float32(r.lambda_).view(np.uint32)) for r in res]))
    assert (out[0][0] == out[1][0]).all()
    assert out[0][1] == out[1][1]
    assert sum(r[1] for r in out[0][1]) > 4 * count            # the fits did iterate


def test_repeated_launches_with_a_large_basis_table_agree():
    """Regression: with per-problem abscissae the basis table is count x m rows (2 MB here). While the launch took it from the
    stream-ordered pool (hipMallocAsync / hipFreeAsync), about one call in 150 returned wrong fits for a contiguous range of
    problems; the table now lives in ordinary memory. 120 launches, every one bit-identical with the shared-abscissae fit."""
    count = 256
    t, data, truth, x0 = P.cfg5_pad8(count, 512)
    s = M.LeastSquaresSettings(np.float32)
    res0, xa = M.optimizeLeastSquaresBatched(M.MODEL_EXP_DECAY_PAD8, x0, t, data, settings=s)
    t2 = np.tile(t, (count, 1))
    ref = xa.view(np.uint32)
    for rep in range(120):
        res1, xb = M.optimizeLeastSquaresBatched(M.MODEL_EXP_DECAY_PAD8, x0, t2, data, settings=s)
        bad = np.argwhere((xb.view(np.uint32) != ref).any(axis=1)).ravel()
        assert bad.size == 0, (rep, bad[:8])


@pytest.mark.parametrize("model,maker", [(M.MODEL_EXP_DECAY_PAD8, "pad8"), (M.MODEL_EXP_DECAY, "decay")])
def test_per_problem_abscissae_give_the_bits_of_shared_ones(model, maker):
    """t may be one vector for all problems (t_stride = 0) or count x m (t_stride = m: the basis table of a model is then
    count x m rows): the same numbers either way must give the same fits, bit for bit -- and a problem whose abscissae differ
    gets ITS basis (its fit changes, the others' do not)."""
    count = 256
    t, data, truth, x0 = {"pad8": P.cfg5_pad8, "decay": make_exp_decay}[maker](count, 512)
    s = M.LeastSquaresSettings(np.float32)
    res0, xa = M.optimizeLeastSquaresBatched(model, x0, t, data, settings=s)
    t2 = np.tile(t, (count, 1))
    res1, xb = M.optimizeLeastSquaresBatched(model, x0, t2, data, settings=s)
    assert (xa.view(np.uint32) == xb.view(np.uint32)).all()
    assert [(int(r.status), r.iterations, r.fCalls) for r in res0] == [(int(r.status), r.iterations, r.fCalls) for r in res1]
    t3 = t2.copy()
    t3[7] = t3[7] * np.float32(1.01)                                  # problem 7 sees other abscissae
    res2, xc = M.optimizeLeastSquaresBatched(model, x0, t3, data, settings=s)
    same = (xa.view(np.uint32) == xc.view(np.uint32)).all(axis=1)
    assert same[np.arange(count) != 7].all() and not same[7]


@pytest.mark.parametrize("m", [1, 37, 200, 1000])
def test_ragged_and_long_problems_match_the_float_oracle(oracle, m):
    """Rows per lane that are not a whole number of 64-row sweeps, fewer rows than lanes, more than one eight-row chunk a lane
    (m = 1000), and m = 1 < n (J^T J singular: the damping carries the solve): every problem against the float oracle."""
    count = 8
    t, data, truth, x0 = make_exp_decay(count, m)
    res, x = M.optimizeLeastSquaresBatched(M.MODEL_EXP_DECAY, x0, t, data, settings=M.LeastSquaresSettings(np.float32))
    for k in range(count):
        ro, xo = oracle.optimize(oracle_f(M.MODEL_EXP_DECAY, t, data[k]), m, x0[k], dtype=np.float32)
        assert (res[k].status >= 0) == (ro.status >= 0), (k, res[k], ro.status)
        if ro.status >= 0:
            assert np.isclose(res[k].residual, ro.residual, rtol=5e-3, atol=1e-7), (k, res[k].residual, ro.residual)
            if m >= 37:
                assert np.allclose(x[k], xo, rtol=1e-2, atol=2e-3), (k, x[k], xo)


def test_largest_row_count_of_the_n8_models(oracle):
    """A problem's J, y and trial residual live in its workgroup's LDS: (n + 2) m floats <= 160 KB - 512, m <= 4083 at n = 8
    (the four-problem workgroups of round 2 stopped at 1011). m = 4000 against the float oracle; m = 4200 is refused (-3)."""
    import ctypes as C

    class Ctx(C.Structure):
        _fields_ = [("t", C.c_void_p), ("data", C.c_void_p)]
    count, m = 6, 4000
    t, data, truth, x0 = P.cfg5_pad8(count, m)
    res, x = M.optimizeLeastSquaresBatched(M.MODEL_EXP_DECAY_PAD8, x0, t, data)
    f = oracle.native_fn("wlc_exp_pad8_f_s")
    for k in range(count):
        d = np.ascontiguousarray(data[k])
        ctx = Ctx(t.ctypes.data, d.ctypes.data)
        ro, xk = oracle.optimize(f, m, x0[k], dtype=np.float32, fctx=C.addressof(ctx))
        assert res[k].status >= 0 and ro.status >= 0
        assert np.isclose(res[k].residual, ro.residual, rtol=1e-3), (k, res[k].residual, ro.residual)
        assert (np.abs(x[k] - xk) / np.maximum(1.0, np.abs(xk))).max() <= 5e-2
    t2, data2, _, x02 = P.cfg5_pad8(2, 4200)
    with pytest.raises(RuntimeError, match="-3"):
        M.optimizeLeastSquaresBatched(M.MODEL_EXP_DECAY_PAD8, x02, t2, data2)


@pytest.mark.parametrize("n", [8, 3])
def test_batched_solve_is_bit_identical_with_the_fused_float_posvx(oracle, n):
    """posvx_rows states ?posvx('E','L') with every multiply-add fused and IEEE division / square root; the oracle's
    lmo_posvx_fused_s is the oracle's float ?posvx with fmaf in the same loops. Same operations in the same order on both sides:
    the solutions are equal BIT FOR BIT -- well-scaled systems, systems that ?laqsy equilibrates, and the info of systems whose
    leading minor of order k is not positive."""
    rng = np.random.default_rng(23 + n)
    count = 512
    Pm = np.zeros((count, n, n), dtype=np.float32)
    b = rng.standard_normal((count, n)).astype(np.float32)
    for p in range(count):
        G = rng.standard_normal((2 * n, n))
        if p % 2:
            G = G * np.logspace(-2, 2, n)[None, :]
        A = G.T @ G + 1e-3 * np.eye(n)
        if p % 32 == 9:
            k = 1 + (p // 32) % n
            A[k - 1, k - 1] = -abs(A[k - 1, k - 1])
        Pm[p] = A
    x, info = M.batchedPosvx(Pm, b)
    scaled = 0
    for p in range(count):
        oi, xo, eq = oracle.posvx_fused_s(Pm[p], b[p])
        assert info[p] == oi, (p, info[p], oi)
        scaled += int(eq)
        if oi == 0:
            assert (x[p].view(np.uint32) == xo.view(np.uint32)).all(), (p, x[p], xo)
    assert scaled > count // 3                                       # the equilibration branch was exercised


def test_cfg5_pad8_all_4096_fits_are_bit_identical_with_the_fused_float_oracle(oracle):
    """Round-3 review, "what's weak" 2: the damped solve of the wave-per-problem kernel was bit-pinned (lmo_posvx_fused_s), its sums
    and its residual were not, which left |x_gpu - x_oracle| <= 5e-2 for the worst of 4096 fits. Now the WHOLE fit is pinned:
    oracle/lm_batched_fused.c restates k_lm_batched's arithmetic operation for operation (fused where the kernel fuses, 64 per-lane
    partial sums + the wave butterfly, det_expf, the fused ?posvx) and every one of the 4096 problems of BASELINE cfg 5 must give
    the same status, iterations and fCalls and the SAME BITS of x, residual and lambda. The only input taken from the device is the
    basis table (sinf / cosf of the abscissae, tabulated once a launch: parameter-independent data). No tolerance."""
    import ctypes as C
    count, m, n = 4096, 512, 8
    t, data, truth, x0 = P.cfg5_pad8(count)
    L = api.lib()
    s = M.LeastSquaresSettings(np.float32)
    dt_, dd, dx = api.DeviceBuffer(t), api.DeviceBuffer(data), api.DeviceBuffer(x0)
    dlo = api.DeviceBuffer(np.full(n, -np.inf, dtype=np.float32)); dup = api.DeviceBuffer(np.full(n, np.inf, dtype=np.float32))
    dres = api.DeviceBuffer(nbytes=count * 24, dtype=np.uint8, shape=(count * 24,))
    dbasis = api.DeviceBuffer(nbytes=m * 4 * 4, dtype=np.float32, shape=(m, 4))      # caller-owned: filled by the launch
    st = api.Stream()
    opt = api.BatchedOptions(stream=st.handle, basis=dbasis.ptr, basis_bytes=m * 16)
    assert L.mir_lsq_batched_kernel_s(C.byref(s), count, m, M.MODEL_EXP_DECAY_PAD8, dx.ptr, dlo.ptr, dup.ptr, dt_.ptr, 0, dd.ptr,
                                      dres.ptr, C.byref(opt)) == 0
    st.synchronize()
    raw = np.frombuffer(dres.download().tobytes(), dtype=np.dtype([("status", "<i4"), ("iterations", "<u4"), ("fCalls", "<u4"),
                                                                   ("gCalls", "<u4"), ("residual", "<f4"), ("lambda", "<f4")])).copy()
    x = dx.download().reshape(count, n).copy()
    basis = dbasis.download().reshape(m, 4).copy()
    # the table is what it says (device sinf / cosf against numpy's, to float rounding)
    ref = np.stack([np.sin(2 * t.astype(np.float64)), np.cos(2 * t.astype(np.float64)), np.sin(5 * t.astype(np.float64)),
                    np.cos(5 * t.astype(np.float64))], axis=1)
    assert np.abs(basis - ref).max() < 1e-6
    for b in (dt_, dd, dx, dlo, dup, dres, dbasis):
        b.free()
    bad = []
    for k in range(count):
        (so, ito, fco, gco, ro, lo_), xo = oracle.optimize_batched_fused_pad8_s(s, t, basis, data[k], x0[k])
        same = (so == raw["status"][k] and ito == raw["iterations"][k] and fco == raw["fCalls"][k] and gco == raw["gCalls"][k]
                and np.float32(ro).tobytes() == raw["residual"][k].tobytes() and np.float32(lo_).tobytes() == raw["lambda"][k].tobytes()
                and xo.tobytes() == x[k].tobytes())
        if not same:
            bad.append((k, so, int(raw["status"][k]), ito, int(raw["iterations"][k]), fco, int(raw["fCalls"][k]),
                        float(np.abs(xo - x[k]).max())))
    assert not bad, (len(bad), bad[:8])
    assert np.all(raw["status"] >= 0) and raw["iterations"].sum() > 3 * count


def test_ragged_random_fits_are_bit_identical_with_the_fused_float_oracle(oracle):
    """The same bit-for-bit comparison off the benchmark's shape: 40 launches of scripts/fuzz_batched.py -- row counts 1 .. 1400
    (1, 2, 7 .. 9, 63 .. 65, 127 .. 129, 511 .. 513 and random: ragged against the 64 lanes and the chunks of eight loads), 1 .. 48
    problems a launch, noise 0 .. 0.1, near and far starts, maxIterations from 1, maxAge 0 / 1 / 4. (200 launches = 5022 fits: 0 differ,
    profiles/r04/fuzz_batched_200.txt.)"""
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "fuzz_batched.py")
    spec = importlib.util.spec_from_file_location("fuzz_batched", path)
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    fits, bad = mod.run(40, 700, verbose=False)
    assert fits > 500 and bad == 0, (fits, bad)
