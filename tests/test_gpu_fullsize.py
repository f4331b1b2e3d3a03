"""BASELINE cfg 3 at FULL size in the GPU test tier: tanh-linear NLS, m = 1e6 x n = 128, fp64, finite-difference
Jacobian through the batched device callbacks -- the bench.py workload -- against the oracle (the CPU restatement of
least_squares.d:877-1176 with OpenBLAS for syrk / gemv / ger / posvx) on the same inputs.

The oracle needs ~3 s per accepted iteration on a 16..64-core host (syrk dominates), so it is bounded:
  * absTolerance = 1e-5 (bench.py's setting): the whole solve is 6 accepted iterations; x, residual, status, fCalls compared;
  * absTolerance = 1e-9 (SURVEY 8d's setting): the last acceptance compares rounding noise, so the
    status is xConverged OR furtherImprovement on either side; the minimiser is the same to 1e-6 relative regardless.
Tolerances (north star): |x_gpu - x_oracle| <= 1e-6 |x|_inf, residual rtol 1e-9."""
import os
import subprocess
import sys

import numpy as np
import pytest

import mir_optim_amd as M
from mir_optim_amd import workloads as W

pytestmark = pytest.mark.gpu

M_ROWS, N = 1_000_000, 128


@pytest.fixture(scope="module")
def cfg3():
    data = W.tanh_linear_data(M_ROWS, N)
    prob = W.TanhLinear(data["A"], data["b"])
    yield data, prob
    prob.dA.free(); prob.db.free()


class OracleResult:
    pass


def oracle_run(tmp_path, abs_tolerance, max_iterations, m=M_ROWS, n=N):
    """The oracle in a process of its own (tests/oracle_fullsize_worker.py), same inputs (counter RNG)."""
    threads = min(os.cpu_count() or 1, 32)
    out = str(tmp_path / "oracle.npz")
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_fullsize_worker.py")
    p = subprocess.run([sys.executable, worker, str(m), str(n), repr(abs_tolerance), str(max_iterations), str(threads), out],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    z = np.load(out)
    r = OracleResult()
    r.status, r.iterations, r.fCalls, r.residual = int(z["status"]), int(z["iterations"]), int(z["fCalls"]), float(z["residual"])
    r.trace = [tuple(t) for t in z["trace"]]
    return r, z["x"]


def test_cfg3_full_size_bench_setting_matches_oracle(cfg3, tmp_path):
    data, prob = cfg3
    s = M.LeastSquaresSettings(); s.absTolerance = 1e-5
    st = M.Stats()
    tr = M.Trace(256)
    res, x = prob.solve(data["x0"], settings=s, batched=True, stats=st, trace=tr)
    ro, xo = oracle_run(tmp_path, 1e-5, 1000)
    # pass by pass (mir_lsq_trace vs the oracle's trace): the same events in the same order -- two full refreshes, four Broyden
    # updates, six accepted passes -- with the same damping, residuals and step lengths. Tolerances as measured at this size
    # (printed on failure): lambda is a product of exact constants until rho enters it; the sums of squares of 1e6 terms
    # agree to ~1e-13; dx.dx of the last, 1e-6-long steps carries the finite-difference noise of J (h = 2^-26).
    got = tr.records()
    assert [(int(g[0]), int(g[1])) for g in got] == [(int(e[0]), int(e[1])) for e in ro.trace], (got, ro.trace)
    for g, e in zip(got, ro.trace):
        assert np.isclose(g[2], e[2], rtol=1e-6), (g, e)
        assert np.allclose(g[3:5], e[3:5], rtol=1e-9), (g, e)
        assert np.isclose(g[5], e[5], rtol=1e-4), (g, e)
    assert int(res.status) == ro.status == M.LeastSquaresStatus.xConverged
    assert (res.iterations, res.fCalls) == (ro.iterations, ro.fCalls)
    assert np.abs(x - xo).max() <= 1e-6 * np.abs(xo).max(), np.abs(x - xo).max()
    assert np.isclose(res.residual, ro.residual, rtol=1e-9, atol=0)
    assert st.jacobian_full == 2 and st.accepted == res.iterations
    # the minimiser really is one: the gradient of the data's generating point is O(noise), x is within 1e-2 of it
    assert np.abs(x - data["xstar"]).max() < 5e-2


def test_cfg3_full_size_survey_setting_matches_oracle(cfg3, tmp_path):
    data, prob = cfg3
    s = M.LeastSquaresSettings(); s.absTolerance = 1e-9
    res, x = prob.solve(data["x0"], settings=s, batched=True)
    ro, xo = oracle_run(tmp_path, 1e-9, 14)
    ok = (M.LeastSquaresStatus.xConverged, M.LeastSquaresStatus.furtherImprovement)
    assert res.status in ok, res
    assert ro.status in (0, 1, -1)               # -1: the bounded oracle sample stopped at maxIterations = 14
    assert np.abs(x - xo).max() <= 1e-6 * np.abs(xo).max(), np.abs(x - xo).max()
    assert np.isclose(res.residual, ro.residual, rtol=1e-9, atol=0)
    assert 10 <= res.iterations <= 14


def test_cfg4_shape_quarter_million_rows_first_iterations_match_oracle(tmp_path):
    """cfg 4's per-GPU kernels (n = 256: k_jtj_fdp8, the eight-column-pair sweep, the global-memory-factor solve) against the
    ORACLE above the m = 40 004 the sharded tests reach (round-2 verdict, "what's weak" 3): m = 250 000 x n = 256, the first
    three accepted iterations (maxIterations = 3 on both sides: one FD refresh of 512 residual evaluations + Broyden passes),
    pass by pass and at the end. Tolerances as for cfg 3 at full size."""
    m, n = 250_000, 256
    data = W.tanh_linear_data(m, n)
    prob = W.TanhLinear(data["A"], data["b"])
    s = M.LeastSquaresSettings(); s.absTolerance = 1e-9; s.maxIterations = 3
    tr = M.Trace(64)
    res, x = prob.solve(data["x0"], settings=s, batched=True, trace=tr)
    prob.dA.free(); prob.db.free()
    ro, xo = oracle_run(tmp_path, 1e-9, 3, m, n)
    assert int(res.status) == ro.status == M.LeastSquaresStatus.maxIterations
    assert (res.iterations, res.fCalls) == (ro.iterations, ro.fCalls) == (3, ro.fCalls)
    got = tr.records()
    assert [(int(g[0]), int(g[1])) for g in got] == [(int(e[0]), int(e[1])) for e in ro.trace], (got, ro.trace)
    for g, e in zip(got, ro.trace):
        assert np.isclose(g[2], e[2], rtol=1e-6), (g, e)
        assert np.allclose(g[3:5], e[3:5], rtol=1e-9), (g, e)
        assert np.isclose(g[5], e[5], rtol=1e-5), (g, e)
    assert np.abs(x - xo).max() <= 1e-6 * np.abs(xo).max(), np.abs(x - xo).max()
    assert np.isclose(res.residual, ro.residual, rtol=1e-9, atol=0)


def test_cfg4_per_gpu_shape_full_size_matches_oracle(tmp_path):
    """BASELINE cfg 4's PER-GPU problem at full size -- m = 1e6 x n = 256, the shape every rank of the weak-scaled configuration
    solves -- against the oracle on the same inputs (round-3 review, "what's weak" 1: until now the oracle only saw this n at
    250 000 rows and the full size was checked GPU against GPU). bench.py's setting (absTolerance = 1e-5): the whole solve, six
    accepted iterations, two finite-difference refreshes of 512 residual evaluations; the oracle needs ~80 s on 8 cores.
    Pass by pass and at the end; tolerances as for cfg 3 at full size, dx.dx to 1e-3 (its last entries are 1e-11 .. 1e-14:
    squares of steps that are themselves the finite-difference noise of J)."""
    m, n = 1_000_000, 256
    data = W.tanh_linear_data(m, n)
    prob = W.TanhLinear(data["A"], data["b"])
    s = M.LeastSquaresSettings(); s.absTolerance = 1e-5
    st = M.Stats()
    tr = M.Trace(256)
    res, x = prob.solve(data["x0"], settings=s, batched=True, stats=st, trace=tr)
    prob.dA.free(); prob.db.free()
    ro, xo = oracle_run(tmp_path, 1e-5, 1000, m, n)
    got = tr.records()
    assert [(int(g[0]), int(g[1])) for g in got] == [(int(e[0]), int(e[1])) for e in ro.trace], (got, ro.trace)
    for g, e in zip(got, ro.trace):
        assert np.isclose(g[2], e[2], rtol=1e-6), (g, e)
        assert np.allclose(g[3:5], e[3:5], rtol=1e-9), (g, e)
        assert np.isclose(g[5], e[5], rtol=1e-3), (g, e)
    assert int(res.status) == ro.status == M.LeastSquaresStatus.xConverged
    assert (res.iterations, res.fCalls) == (ro.iterations, ro.fCalls) == (6, 519)
    assert np.abs(x - xo).max() <= 1e-6 * np.abs(xo).max(), np.abs(x - xo).max()
    assert np.isclose(res.residual, ro.residual, rtol=1e-9, atol=0)
    assert st.jacobian_full == 2 and st.accepted == res.iterations
