"""A gcc-compiled C program calls the drop-in boundary directly (tests/c_harness/harness.c): by-value Slice structs and the
sret result of mir_optimize_least_squares_d (least_squares.d:705-724) without ctypes / libffi in the loop.
CPU tier: it compiles and links against the header and the library, struct sizes agree with the ctypes mirror and the
reference layout, the helpers answer, and without a GPU the solve reports numericError loudly (no CPU fallback).
GPU tier: the reference's unittests T2 (FD + thread manager) and T3b (bounds -> BOXCQP) through that binary."""
import ctypes as C
import os
import subprocess

import pytest

import mir_optim_amd as M
from mir_optim_amd import api, build as hipbuild

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c_harness", "harness.c")
EXE = os.path.join(ROOT, "tests", "c_harness", "harness")


@pytest.fixture(scope="module")
def harness():
    libdir = os.path.dirname(hipbuild.SOLVER_LIB)
    if (not os.path.exists(EXE) or os.path.getmtime(EXE) < max(os.path.getmtime(SRC), os.path.getmtime(hipbuild.SOLVER_LIB),
                                                                     os.path.getmtime(os.path.join(ROOT, "include", "mir_optim_amd.h")))):
        subprocess.check_call(["gcc", "-O1", "-std=c11", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), SRC,
                               "-o", EXE, "-L", libdir, "-lmir_optim_amd", "-lm", "-Wl,-rpath," + libdir,
                               "-Wl,-rpath,/opt/rocm/lib", "-Wl,--allow-shlib-undefined"])

    def run(mode):
        p = subprocess.run([EXE, mode], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        assert p.returncode == 0, p.stderr.decode()
        out = {}
        for line in p.stdout.decode().splitlines():
            k, _, v = line.partition(" ")
            out[k] = v
        return out, p.stderr.decode()
    return run


def test_c_sizes_equal_ctypes_mirror_and_reference_layout(harness):
    o, _ = harness("sizes")
    assert (int(o["settings_d"]), int(o["settings_s"]), int(o["result_d"]), int(o["result_s"])) == (128, 68, 32, 24)
    assert int(o["slice_d"]) == 16 and int(o["task"]) == 16
    assert int(o["settings_d_qp"]) == 104 and int(o["settings_s_qp"]) == 56 and int(o["result_d_residual"]) == 16
    assert int(o["gpu_options"]) == C.sizeof(api.GpuOptions) and int(o["stats"]) == C.sizeof(api.Stats)
    assert int(o["trace_record"]) == C.sizeof(api.TraceRecord)
    assert int(o["options_variant"]) == api.GpuOptions.variant.offset
    assert int(o["options_fbRowMajor"]) == api.GpuOptions.fbRowMajor.offset


def test_c_helpers(harness):
    o, _ = harness("helpers")
    assert int(o["maxIterations"]) == 1000 and float(o["jacobianEpsilon"]) == 2.0 ** -26
    assert abs(float(o["lambdaDecrease"]) - 0.30901699437494745) < 1e-17
    assert int(o["work_1e6_128"]) == M.mir_least_squares_work_length(1000000, 128) == 2 * 128 * 128 + 8 * 128 + 5 * 128 + 128 * 128 + 128 * 1000000 + 2000000
    assert int(o["qp_work_128"]) == 2 * 128 * 128 + 8 * 128 and int(o["qp_iwork_128"]) == 128 + 32
    assert o["string_numericError"] == "Numeric Error" and o["string_xConverged"] == "X converged"


def test_c_solve_without_gpu_is_a_loud_numeric_error(harness):
    if M.device_count() > 0:
        pytest.skip("a GPU is visible")
    o, err = harness("t2")
    assert int(o["status"]) == M.LeastSquaresStatus.numericError and "no usable HIP device" in err
    assert int(o["host_f_calls"]) == 0                     # nothing was computed anywhere


@pytest.mark.gpu
def test_c_caller_T2_and_T3b(harness, oracle):
    import numpy as np
    import problems as P
    o, _ = harness("t2")                                      # LS:248-273
    p = P.t2()
    ro, xo = oracle.optimize(p["f"], 2, p["x0"])
    x = np.array([float(o["x0"]), float(o["x1"])])
    assert np.linalg.norm(x - [1.0, 1.0]) < 1e-6              # LS:272
    assert (int(o["status"]), int(o["iterations"]), int(o["fCalls"])) == (ro.status, ro.iterations, ro.fCalls) == (3, 19, 38)
    assert int(o["tm_calls"]) == 4 and int(o["task_calls"]) == 8          # 4 FD refreshes x n = 2 columns through the manager
    o, _ = harness("t3b")                                     # LS:321-330
    p = P.t3b()
    ro, xo = oracle.optimize(p["f"], 2, p["x0"], lower=p["lower"], upper=p["upper"], g=p["g"])
    x = np.array([float(o["x0"]), float(o["x1"])])
    assert np.linalg.norm(x - [10.0, 100.0]) < 1e-5 and np.all(x >= 10)   # LS:329-330
    assert (int(o["status"]), int(o["iterations"]), int(o["fCalls"]), int(o["gCalls"])) == (ro.status, ro.iterations, ro.fCalls, ro.gCalls)
    assert abs(float(o["residual"]) - 81.0) < 1e-9
