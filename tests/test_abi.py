"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/mir_optim_amd.h declares, pins the reference struct layouts, and the host
logic that needs no GPU (validation order LS:930-943, lengths LS:642-656, strings LS:528-557,
defaults LS:93-122) agrees with the oracle. No kernel is launched here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import mir_optim_amd as M
from mir_optim_amd import api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "mir_optim_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = set(re.findall(r"\b(mir_[a-z0-9_]+)\s*\(", src))
    typedefs = set(re.findall(r"\(\*\s*(mir_[a-z0-9_]+)\s*\)", src))
    return sorted(names - typedefs)


def test_library_exports_every_declared_symbol():
    L = api.lib()
    names = declared_functions()
    assert len(names) >= 30
    for required in ("mir_optimize_least_squares_d", "mir_optimize_least_squares_s", "mir_least_squares_work_length",
                     "mir_least_squares_iwork_length", "mir_least_squares_status_string", "mir_least_squares_init_d",
                     "mir_least_squares_init_s", "mir_least_squares_reset_d", "mir_least_squares_reset_s",
                     "mir_box_qp_work_length", "mir_box_qp_iwork_length"):
        assert required in names      # the reference's 11 extern(C) symbols
    for nm in names:
        assert hasattr(L, nm), nm
    assert api.workloads_lib() is not None


def test_struct_layouts_match_reference():
    S = api._Sd
    assert C.sizeof(S) == 128 and C.alignment(S) == 8
    offs = {n: getattr(S, n).offset for n, _ in S._fields_}
    assert offs["maxIterations"] == 0 and offs["maxAge"] == 4 and offs["jacobianEpsilon"] == 8
    assert offs["absTolerance"] == 16 and offs["relTolerance"] == 24 and offs["gradTolerance"] == 32
    assert offs["maxGoodResidual"] == 40 and offs["maxStep"] == 48 and offs["maxLambda"] == 56
    assert offs["minLambda"] == 64 and offs["minStepQuality"] == 72 and offs["goodStepQuality"] == 80
    assert offs["lambdaIncrease"] == 88 and offs["lambdaDecrease"] == 96 and offs["qpSettings"] == 104
    assert api._QPd.maxIterations.offset == 16
    assert C.sizeof(api._Ss) == 68 and api._Ss.qpSettings.offset == 56
    assert C.sizeof(api._Rd) == 32 and api._Rd.residual.offset == 16 and api._Rd.lambda_.offset == 24
    assert C.sizeof(api._Rs) == 24
    assert C.sizeof(api._SliceD) == 16 and C.sizeof(api._Task) == 16


def test_lengths_strings_defaults_agree_with_oracle(oracle):
    OL = oracle.lib()
    for m, n in [(1, 1), (2, 2), (100, 3), (5, 7), (1000000, 128), (8000000, 256)]:
        assert M.mir_least_squares_work_length(m, n) == OL.lmo_work_length(m, n)
        assert M.mir_least_squares_iwork_length(m, n) == OL.lmo_iwork_length(m, n)
        assert M.mir_box_qp_work_length(n) == OL.lmo_box_qp_work_length(n)
        assert M.mir_box_qp_iwork_length(n) == OL.lmo_box_qp_iwork_length(n)
    for st in M.LeastSquaresStatus:
        assert M.leastSquaresStatusString(st) == OL.lmo_status_string(int(st)).decode()
    for dt in (np.float64, np.float32):
        a, b = M.LeastSquaresSettings(dt), oracle.default_settings(dt)
        for name, _ in type(a)._fields_[:-1]:
            assert getattr(a, name) == getattr(b, name), name
        for name in ("relTolerance", "absTolerance", "maxIterations"):
            assert getattr(a.qpSettings, name) == getattr(b.qpSettings, name)
    s = M.LeastSquaresSettings()
    s.maxIterations = 7
    api.lib().mir_least_squares_reset_d(C.byref(s))
    assert s.maxIterations == 1000


def rosen(x, y):
    y[0] = 10 * (x[1] - x[0] ** 2)
    y[1] = 1 - x[0]


def test_validation_codes_need_no_gpu(oracle):
    """LS:930-943 run on the host before any device work (quirk Q9): same codes as the oracle."""
    def st(**kw):
        s = kw.pop("settings", None)
        m = kw.pop("m", 2)
        x0 = kw.pop("x0", [0.0, 0.0])
        res, _ = M.optimizeLeastSquares(rosen, m, x0, kw.get("l"), kw.get("u"), settings=s)
        res_o, _ = oracle.optimize(rosen, m, x0, lower=kw.get("l"), upper=kw.get("u"),
                                   settings=kw.get("osettings"))
        return res, res_o
    cases = [dict(x0=[np.nan, 0.0]), dict(x0=[np.inf, 0.0]), dict(m=0), dict(l=[1.0, -1.0], u=[2.0, 2.0])]
    for kw in cases:
        a, b = st(**kw)
        assert int(a.status) == b.status < -26
        assert a.residual == np.inf and a.lambda_ == 0 and a.iterations == 0 and a.fCalls == 0
    for field, val in [("minStepQuality", 1.0), ("minStepQuality", -0.1), ("goodStepQuality", 1.5),
                       ("goodStepQuality", 0.05), ("lambdaIncrease", 0.5), ("lambdaDecrease", 2.0)]:
        s = M.LeastSquaresSettings(); setattr(s, field, val)
        so = oracle.default_settings(); setattr(so, field, val)
        a, b = st(settings=s, osettings=so)
        assert int(a.status) == b.status, field
    with pytest.raises(M.LeastSquaresException) as ei:
        M.optimize(rosen, 2, [np.nan, 1.0])
    assert ei.value.status == M.LeastSquaresStatus.badGuess
    assert "Initial guess must be an array of finite numbers." in str(ei.value)


@pytest.mark.skipif(api.device_count() > 0, reason="GPU present: the loud-failure path is not reachable")
def test_no_gpu_means_loud_numeric_error_not_a_cpu_path(capfd):
    res, x = M.optimizeLeastSquares(rosen, 2, [-1.2, 1.0])
    assert res.status == M.LeastSquaresStatus.numericError and res.iterations == 0 and res.fCalls == 0
    assert np.array_equal(x, [-1.2, 1.0])
    assert "no usable HIP device" in capfd.readouterr().err
    st, _, _ = M.solveBoxQP(np.eye(2), [1.0, 1.0], [-1.0, -1.0], [1.0, 1.0])
    assert st == M.BoxQPStatus.numericError


def test_gpu_options_trace_field_is_appended():
    """mir_lsq_gpu_options is versioned by struct_size: `trace` was appended after `stats`, `fbRowMajor` after `trace`,
    `fbRowMajorDiff` after that, `stats_size` last; older callers (64-, 72-, 80- and 88-byte structs) stay valid. Offsets as
    declared in include/mir_optim_amd.h. mir_lsq_stats is versioned by `stats_size` (or by the era of struct_size): the sizes
    the library falls back to are the historic ones (tests/test_gpu_launch_budget.py checks the writes with a canary).
    Round 4 removed the two trailing members of a retired experiment (fd_windows became reserved0, the window callback is
    gone): the struct is 96 bytes; a caller that still passes 104 is simply read up to what this build knows."""
    import ctypes as C
    from mir_optim_amd import api
    assert C.sizeof(api.GpuOptions) == 96 and api.GpuOptions.reserved0.offset == 92 and api.GpuOptions.stats.offset == 56 and api.GpuOptions.trace.offset == 64
    assert api.GpuOptions.fbRowMajor.offset == 72 and api.GpuOptions.fbRowMajorDiff.offset == 80
    assert api.GpuOptions.stats_size.offset == 88 and api.GpuOptions().stats_size == C.sizeof(api.Stats)
    assert api.Stats.qp_active_set_passes.offset + 8 == 120 and api.Stats.jtj_fd_launches.offset + 8 == 144
    assert api.Stats.trial_callback_points.offset + 8 == 264 and C.sizeof(api.Stats) == 264 + 15 * 8 and api.Stats.fused_rounds.offset == 264 + 12 * 8
    assert C.sizeof(api.TraceRecord) == 40
    assert C.sizeof(api.BatchedOptions) == 40 and api.BatchedOptions.basis.offset == 16 and api.BatchedOptions.timing.offset == 32
    t = api.Trace(8)
    assert t.count == 0 and t.records() == []


def test_ilp64_integer_workspace_lengths():
    """64-bit lapackint builds of the reference (`*-ilp` dub configurations): QP:47-50 with lapackint.sizeof = 8."""
    L = api.lib()
    for n in (1, 7, 8, 9, 128, 1000):
        assert L.mir_box_qp_iwork_length_ilp64(n) == n + (n + 7) // 8
        assert L.mir_least_squares_iwork_length_ilp64(5, n) == max(n + (n + 7) // 8, n)
        assert L.mir_box_qp_iwork_length(n) == n + (n + 3) // 4
