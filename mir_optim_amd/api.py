"""ctypes binding of libmir_optim_amd.so, mirroring the reference's D API for the
`mir.optim.least_squares` path (names, argument meaning, error behaviour):

  reference (source/mir/optim/least_squares.d)          here
  ---------------------------------------------------   ------------------------------------
  LeastSquaresStatus                LS:20-46            LeastSquaresStatus
  LeastSquaresSettings!T            LS:85-123           LeastSquaresSettings(dtype)
  LeastSquaresResult!T              LS:128-143          LeastSquaresResult
  optimize!(f, g, tm) (throws)      LS:165-215          optimize(...)      raises LeastSquaresException
  optimizeLeastSquares!(f, g, tm)   LS:459-519          optimizeLeastSquares(...)
  leastSquaresStatusString          LS:528-557          leastSquaresStatusString
  mir_least_squares_*_length        LS:642-656          mir_least_squares_work_length / _iwork_length
  solveBoxQP (boxcqp.d:85-102)                          solveBoxQP(...)
  BoxQPSettings!T (boxcqp.d:56-71)                      BoxQPSettings(dtype)

No arithmetic happens in Python. The HIP library is mandatory: a missing library raises at
import time, and a missing GPU makes every solve return status numericError (never a CPU path).
"""
import ctypes as C
import enum
import os

import numpy as np

from . import build as _build

__all__ = [
    "LeastSquaresStatus", "BoxQPStatus", "LeastSquaresSettings", "LeastSquaresResult", "BoxQPSettings",
    "LeastSquaresException", "optimize", "optimizeLeastSquares", "solveBoxQP", "leastSquaresStatusString",
    "mir_least_squares_work_length", "mir_least_squares_iwork_length", "mir_box_qp_work_length",
    "mir_box_qp_iwork_length", "GpuOptions", "Stats", "lib", "workloads_lib", "device_count",
    "DeviceBuffer", "Stream", "jtj", "fd_jtj", "DEVICE_CALLBACKS", "TIME_KERNELS", "optimizeLeastSquaresBatched", "batchedPosvx", "BATCHED_NO_LADDER",
    "MODEL_EXP_DECAY", "MODEL_EXP3_AFFINE", "MODEL_EXP_DECAY_PAD8", "ResultS", "Trace", "TraceRecord", "Spline", "FitSplineResult", "fitSpline",
    "fit_spline_residuals", "variant_lr_cap",
    "VARIANT_BROYDEN_REWRITE", "VARIANT_FD_SEPARATE_FILL", "VARIANT_NO_SPECULATION", "VARIANT_NO_NULL_SKIP",
    "VARIANT_SOLVE_BOUNDED", "VARIANT_DEBUG_SOLVE", "VARIANT_HOST_PROFILE", "VARIANT_SOLVE_GENERIC", "VARIANT_SOLVE_ONE_WORKGROUP", "VARIANT_DEBUG_HELPERS_ABSENT", "VARIANT_FD_HOST_COLUMNS",
    "VARIANT_NO_PIPELINE", "BatchedOptions",
]

MODEL_EXP_DECAY = 0      # n = 3: p0 exp(-t p1) + p2
MODEL_EXP3_AFFINE = 1    # n = 8: p0 exp(-t p1) + p2 exp(-t p3) + p4 exp(-t p5) + p6 + p7 t
MODEL_EXP_DECAY_PAD8 = 2  # n = 8: p0 exp(-t p1) + p2 + p3 sin 2t + p4 cos 2t + p5 sin 5t + p6 cos 5t + p7 t   (cfg 5)

DEVICE_CALLBACKS = 1
TIME_KERNELS = 2

# MIR_LSQ_VARIANT_* (include/mir_optim_amd.h): literal restatements the tests compare the product path with, and
# diagnostics; 0 = the product path
VARIANT_BROYDEN_REWRITE = 1 << 0
VARIANT_FD_SEPARATE_FILL = 1 << 1
VARIANT_NO_SPECULATION = 1 << 4
VARIANT_NO_NULL_SKIP = 1 << 5
VARIANT_SOLVE_BOUNDED = 1 << 6
VARIANT_DEBUG_SOLVE = 1 << 7
VARIANT_HOST_PROFILE = 1 << 8
VARIANT_SOLVE_GENERIC = 1 << 10
VARIANT_SOLVE_ONE_WORKGROUP = 1 << 11
VARIANT_DEBUG_HELPERS_ABSENT = 1 << 12
VARIANT_FD_HOST_COLUMNS = 1 << 13
VARIANT_NO_PIPELINE = 1 << 22


def variant_lr_cap(k):
    """Fold the pending Broyden terms into J after k (1..16) updates."""
    return (int(k) & 31) << 16


class LeastSquaresStatus(enum.IntEnum):  # LS:20-46
    maxIterations = -1
    furtherImprovement = 0
    xConverged = 1
    gConverged = 2
    fConverged = 3
    badBounds = -32
    badGuess = -31
    badMinStepQuality = -30
    badGoodStepQuality = -29
    badStepQuality = -28
    badLambdaParams = -27
    numericError = -26


class BoxQPStatus(enum.IntEnum):  # boxcqp.d:18-26
    solved = 0
    numericError = 1
    maxIterations = 2


class LeastSquaresException(Exception):
    """optimize() raises this for status < 0, like LS:175-179 ("mir-optim Least Squares: " ~ status string)."""

    def __init__(self, status, result=None):
        self.status = LeastSquaresStatus(status)
        self.result = result
        super().__init__("mir-optim Least Squares: " + leastSquaresStatusString(status))


class _QPd(C.Structure):
    _fields_ = [("relTolerance", C.c_double), ("absTolerance", C.c_double), ("maxIterations", C.c_uint32)]


class _QPs(C.Structure):
    _fields_ = [("relTolerance", C.c_float), ("absTolerance", C.c_float), ("maxIterations", C.c_uint32)]


_FIELDS = ["jacobianEpsilon", "absTolerance", "relTolerance", "gradTolerance", "maxGoodResidual", "maxStep",
           "maxLambda", "minLambda", "minStepQuality", "goodStepQuality", "lambdaIncrease", "lambdaDecrease"]


class _Sd(C.Structure):
    _fields_ = ([("maxIterations", C.c_uint32), ("maxAge", C.c_uint32)] + [(k, C.c_double) for k in _FIELDS]
                + [("qpSettings", _QPd)])


class _Ss(C.Structure):
    _fields_ = ([("maxIterations", C.c_uint32), ("maxAge", C.c_uint32)] + [(k, C.c_float) for k in _FIELDS]
                + [("qpSettings", _QPs)])


class _Rd(C.Structure):
    _fields_ = [("status", C.c_int32), ("iterations", C.c_uint32), ("fCalls", C.c_uint32), ("gCalls", C.c_uint32),
                ("residual", C.c_double), ("lambda_", C.c_double)]


class _Rs(C.Structure):
    _fields_ = [("status", C.c_int32), ("iterations", C.c_uint32), ("fCalls", C.c_uint32), ("gCalls", C.c_uint32),
                ("residual", C.c_float), ("lambda_", C.c_float)]


ResultS = _Rs


class _SliceD(C.Structure):
    _fields_ = [("length", C.c_size_t), ("ptr", C.c_void_p)]


class _Task(C.Structure):
    _fields_ = [("context", C.c_void_p), ("function", C.c_void_p)]


class Stats(C.Structure):
    _fields_ = [("passes", C.c_uint64), ("accepted", C.c_uint64), ("rejected", C.c_uint64),
                ("step_guard_rejects", C.c_uint64), ("jacobian_full", C.c_uint64), ("jacobian_broyden", C.c_uint64),
                ("jtj_launches", C.c_uint64), ("jtj_broyden_launches", C.c_uint64), ("jtj_ms", C.c_double),
                ("jtj_broyden_ms", C.c_double), ("solve_ms", C.c_double), ("solve_launches", C.c_uint64),
                ("fd_ms", C.c_double), ("total_ms", C.c_double), ("qp_active_set_passes", C.c_uint64),
                ("broyden_lr_columns", C.c_uint64),
                ("jtj_fd_ms", C.c_double), ("jtj_fd_launches", C.c_uint64),
                ("elided_evaluations", C.c_uint64),
                ("allreduce_calls", C.c_uint64 * 3), ("allreduce_elems", C.c_uint64 * 3),
                ("broyden_flushes", C.c_uint64), ("jtj_resyncs", C.c_uint64),
                ("fd_callback_ms", C.c_double), ("fd_callback_calls", C.c_uint64), ("fd_callback_points", C.c_uint64),
                ("trial_callback_ms", C.c_double), ("trial_callback_calls", C.c_uint64),
                ("trial_callback_points", C.c_uint64),
                ("library_launches", C.c_uint64), ("round_launches", C.c_uint64 * 3), ("rounds", C.c_uint64 * 3),
                ("fd_host_wall_ms", C.c_double), ("fd_host_f_ms", C.c_double), ("fd_host_columns", C.c_uint64),
                ("host_f_ms", C.c_double), ("host_f_calls", C.c_uint64),
                ("fused_rounds", C.c_uint64), ("fused_passes", C.c_uint64), ("coop_timeouts", C.c_uint64)]

    def as_dict(self):
        return {k: (list(getattr(self, k)) if isinstance(getattr(self, k), C.Array) else getattr(self, k)) for k, _ in self._fields_}


class TraceRecord(C.Structure):
    _fields_ = [("event", C.c_int32), ("iterations", C.c_uint32), ("lambda_", C.c_double), ("residual", C.c_double),
                ("trial_residual", C.c_double), ("dx_dot", C.c_double)]


class _TraceHeader(C.Structure):
    _fields_ = [("records", C.POINTER(TraceRecord)), ("capacity", C.c_uint64), ("count", C.c_uint64)]


class Trace:
    """mir_lsq_trace: per-pass records (event, iterations, lambda, residual, trial_residual, dx_dot)."""

    EVENTS = {0: "jacobian_full", 1: "jacobian_broyden", 2: "rejected", 3: "accepted", 4: "step_guard"}

    def __init__(self, capacity=4096):
        self._buf = (TraceRecord * capacity)()
        self.header = _TraceHeader(C.cast(self._buf, C.POINTER(TraceRecord)), capacity, 0)

    @property
    def count(self):
        return int(self.header.count)

    def records(self):
        k = min(self.count, int(self.header.capacity))
        return [(r.event, r.iterations, r.lambda_, r.residual, r.trial_residual, r.dx_dot) for r in self._buf[:k]]


class GpuOptions(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("flags", C.c_uint32), ("stream", C.c_void_p), ("comm", C.c_void_p),
                ("workspace", C.c_void_p), ("fbContext", C.c_void_p), ("fb", C.c_void_p), ("fd_batch", C.c_uint32),
                ("variant", C.c_uint32), ("stats", C.POINTER(Stats)), ("trace", C.POINTER(_TraceHeader)),
                ("fbRowMajor", C.c_void_p), ("fbRowMajorDiff", C.c_void_p), ("stats_size", C.c_uint32),
                ("reserved0", C.c_uint32)]

    def __init__(self, **kw):
        super().__init__(**kw)
        self.struct_size = C.sizeof(GpuOptions)
        self.stats_size = C.sizeof(Stats)


class BatchedOptions(C.Structure):
    """mir_lsq_batched_options: per-call options of the batched entries (variant bits, stream, caller-owned basis table,
    profiling buffer)."""
    _fields_ = [("struct_size", C.c_uint32), ("variant", C.c_uint32), ("stream", C.c_void_p), ("basis", C.c_void_p),
                ("basis_bytes", C.c_size_t), ("timing", C.c_void_p)]

    def __init__(self, **kw):
        super().__init__(**kw)
        self.struct_size = C.sizeof(BatchedOptions)


TASK_FN = C.CFUNCTYPE(None, _Task, C.c_uint32, C.c_uint32, C.c_uint32)
TM_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_uint32, _Task, TASK_FN)
ALLREDUCE_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)


def _ftype(dtype):
    ct = C.c_double if dtype == np.float64 else C.c_float
    return C.CFUNCTYPE(None, C.c_void_p, C.c_size_t, C.c_size_t, C.POINTER(ct), C.POINTER(ct))


_lib = None
_wl = None


def _load(path):
    if not os.path.exists(path):
        raise ImportError(f"{path} is missing: build it with `python -m mir_optim_amd.build` "
                          "(hipcc --offload-arch=gfx950). mir_optim_amd has no CPU fallback.")
    return C.CDLL(path, mode=C.RTLD_GLOBAL)


def lib():
    """The solver library (include/mir_optim_amd.h). Raises ImportError when it is not built."""
    global _lib
    if _lib is None:
        L = _load(_build.SOLVER_LIB)
        sz = C.c_size_t
        for name in ("mir_least_squares_work_length", "mir_least_squares_iwork_length"):
            getattr(L, name).restype = sz
            getattr(L, name).argtypes = [sz, sz]
        for name in ("mir_box_qp_work_length", "mir_box_qp_iwork_length"):
            getattr(L, name).restype = sz
            getattr(L, name).argtypes = [sz]
        L.mir_box_qp_iwork_length_ilp64.restype = sz
        L.mir_box_qp_iwork_length_ilp64.argtypes = [sz]
        L.mir_least_squares_iwork_length_ilp64.restype = sz
        L.mir_least_squares_iwork_length_ilp64.argtypes = [sz, sz]
        L.mir_least_squares_status_string.restype = C.c_char_p
        L.mir_least_squares_status_string.argtypes = [C.c_int]
        for suf, S, R in (("d", _Sd, _Rd), ("s", _Ss, _Rs)):
            getattr(L, "mir_least_squares_init_" + suf).argtypes = [C.POINTER(S)]
            getattr(L, "mir_least_squares_reset_" + suf).argtypes = [C.POINTER(S)]
            fn = getattr(L, "mir_optimize_least_squares_" + suf)
            fn.restype = R
            fn.argtypes = [C.POINTER(S), sz, sz, C.c_void_p, C.c_void_p, C.c_void_p, _SliceD, _SliceD,
                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
            fn = getattr(L, "mir_optimize_least_squares_gpu_" + suf)
            fn.restype = R
            fn.argtypes = [C.POINTER(S), sz, sz, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(GpuOptions),
                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
            fn = getattr(L, "mir_solve_box_qp_gpu_" + suf)
            fn.restype = C.c_int
            fn.argtypes = [C.c_void_p, sz, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                           C.POINTER(C.c_int)]
            fn = getattr(L, "mir_lsq_jtj_" + suf)
            fn.restype = C.c_int
            fn.argtypes = [sz, sz, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                           C.c_void_p, C.POINTER(C.c_float)]
        L.mir_lsq_comm_create_local_group.restype = C.c_int
        L.mir_lsq_comm_create_local_group.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        L.mir_lsq_fd_diff_jtj_d.restype = C.c_int
        L.mir_lsq_fd_diff_jtj_d.argtypes = [sz, sz, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.POINTER(C.c_float)]
        L.mir_lsq_fd_jtj_d.restype = C.c_int
        L.mir_lsq_fd_jtj_d.argtypes = [sz, sz, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p]
        L.mir_optimize_least_squares_batched_s.restype = C.c_int
        L.mir_optimize_least_squares_batched_s.argtypes = [C.POINTER(_Ss), sz, sz, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                                           C.c_void_p, sz, C.c_void_p, C.c_void_p, C.POINTER(BatchedOptions)]
        L.mir_lsq_batched_kernel_s.restype = C.c_int
        L.mir_lsq_batched_kernel_s.argtypes = [C.POINTER(_Ss), sz, sz, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                               C.c_void_p, sz, C.c_void_p, C.c_void_p, C.POINTER(BatchedOptions)]
        L.mir_lsq_selftest_reductions.restype = C.c_int
        L.mir_lsq_selftest_reductions.argtypes = [C.c_int, C.POINTER(C.c_int * 4)]
        L.mir_lsq_batched_posvx_s.restype = C.c_int
        L.mir_lsq_batched_posvx_s.argtypes = [sz, sz, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.mir_lsq_comm_describe.restype = C.c_int
        L.mir_lsq_comm_describe.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
        L.mir_lsq_workspace_create.restype = C.c_void_p
        L.mir_lsq_workspace_create.argtypes = [sz, sz, sz]
        L.mir_lsq_workspace_destroy.argtypes = [C.c_void_p]
        L.mir_lsq_rccl_unique_id.restype = C.c_int
        L.mir_lsq_rccl_unique_id.argtypes = [C.c_void_p]
        L.mir_lsq_comm_create_rccl.restype = C.c_void_p
        L.mir_lsq_comm_create_rccl.argtypes = [C.c_int, C.c_int, C.c_void_p]
        L.mir_lsq_comm_create_callback.restype = C.c_void_p
        L.mir_lsq_comm_create_callback.argtypes = [C.c_int, C.c_int, ALLREDUCE_FN, C.c_void_p]
        L.mir_lsq_comm_destroy.argtypes = [C.c_void_p]
        L.mir_lsq_comm_record.restype = C.c_int
        L.mir_lsq_comm_record.argtypes = [C.c_void_p, C.c_void_p, sz]
        L.mir_lsq_comm_recorded.restype = sz
        L.mir_lsq_comm_recorded.argtypes = [C.c_void_p]
        L.mir_lsq_comm_create_replay.restype = C.c_void_p
        L.mir_lsq_comm_create_replay.argtypes = [C.c_int, C.c_int, C.c_void_p, sz, C.c_void_p]
        L.mir_lsq_comm_replay_rewind.restype = C.c_int
        L.mir_lsq_comm_replay_rewind.argtypes = [C.c_void_p]
        L.mir_lsq_comm_replay_set_delay.restype = C.c_int
        L.mir_lsq_comm_replay_set_delay.argtypes = [C.c_void_p, C.c_uint]
        L.mir_lsq_comm_ranks.restype = C.c_int
        L.mir_lsq_comm_ranks.argtypes = [C.c_void_p]
        for name in ("mir_lsq_comm_allreduce_d", "mir_lsq_comm_allreduce_s"):
            getattr(L, name).restype = C.c_int
            getattr(L, name).argtypes = [C.c_void_p, C.c_void_p, sz, C.c_void_p]
        L.mir_lsq_device_count.restype = C.c_int
        L.mir_lsq_device_malloc.restype = C.c_void_p
        L.mir_lsq_device_malloc.argtypes = [sz]
        L.mir_lsq_device_free.argtypes = [C.c_void_p]
        for name in ("mir_lsq_memcpy_h2d", "mir_lsq_memcpy_d2h", "mir_lsq_memcpy_d2d"):
            getattr(L, name).restype = C.c_int
            getattr(L, name).argtypes = [C.c_void_p, C.c_void_p, sz, C.c_void_p]
        L.mir_lsq_stream_create.restype = C.c_void_p
        L.mir_lsq_stream_destroy.argtypes = [C.c_void_p]
        L.mir_lsq_stream_synchronize.restype = C.c_int
        L.mir_lsq_stream_synchronize.argtypes = [C.c_void_p]
        L.mir_lsq_version.restype = C.c_char_p
        for suf, S, R, ct in (("d", _Sd, _Rd, C.c_double), ("s", _Ss, _Rs, C.c_float)):
            fn = getattr(L, "mir_fit_spline_" + suf)
            fn.restype = C.c_int
            fn.argtypes = [C.POINTER(S), sz, C.c_void_p, sz, C.c_void_p, C.c_void_p, C.c_void_p, ct, C.c_void_p,
                           C.c_void_p, C.c_void_p, C.POINTER(R)]
            getattr(L, "mir_spline_c2_derivatives_" + suf).argtypes = [sz, C.c_void_p, C.c_void_p, C.c_void_p]
            getattr(L, "mir_spline_eval_" + suf).argtypes = [sz, C.c_void_p, C.c_void_p, C.c_void_p, ct, C.c_void_p]
        L.mir_fit_spline_residuals_d.argtypes = [sz, C.c_void_p, sz, C.c_void_p, C.c_double, C.c_void_p, sz, C.c_void_p]
        _lib = L
    return _lib


def workloads_lib():
    """Device residual callbacks of the synthetic workloads (csrc/workloads.hip)."""
    global _wl
    if _wl is None:
        W = _load(_build.WORKLOADS_LIB)
        W.wl_uniform.argtypes = [C.c_uint64, C.c_uint64, C.c_size_t, C.c_void_p]
        W.wl_tanh_linear_generate.argtypes = [C.c_size_t, C.c_size_t, C.c_size_t, C.c_double, C.c_void_p, C.c_void_p,
                                              C.c_void_p, C.c_void_p]
        _wl = W
    return _wl


def device_count():
    return lib().mir_lsq_device_count()


def leastSquaresStatusString(st):
    return lib().mir_least_squares_status_string(int(st)).decode()


def mir_least_squares_work_length(m, n):
    return lib().mir_least_squares_work_length(m, n)


def mir_least_squares_iwork_length(m, n):
    return lib().mir_least_squares_iwork_length(m, n)


def mir_box_qp_work_length(n):
    return lib().mir_box_qp_work_length(n)


def mir_box_qp_iwork_length(n):
    return lib().mir_box_qp_iwork_length(n)


def LeastSquaresSettings(dtype=np.float64):
    """LeastSquaresSettings!T with the reference defaults (LS:93-122) written by mir_least_squares_init_*."""
    s = _Sd() if dtype == np.float64 else _Ss()
    (lib().mir_least_squares_init_d if dtype == np.float64 else lib().mir_least_squares_init_s)(C.byref(s))
    return s


def BoxQPSettings(dtype=np.float64):
    return LeastSquaresSettings(dtype).qpSettings


class LeastSquaresResult:
    """LeastSquaresResult!T (LS:128-143)."""

    def __init__(self, raw):
        self.status = LeastSquaresStatus(raw.status)
        self.iterations = raw.iterations
        self.fCalls = raw.fCalls
        self.gCalls = raw.gCalls
        self.residual = raw.residual
        self.lambda_ = raw.lambda_

    def __repr__(self):
        return (f"LeastSquaresResult(status={self.status.name}, iterations={self.iterations}, fCalls={self.fCalls}, "
                f"gCalls={self.gCalls}, residual={self.residual!r}, lambda={self.lambda_!r})")


class Stream:
    def __init__(self):
        self.handle = lib().mir_lsq_stream_create()
        if not self.handle:
            raise RuntimeError("hipStreamCreate failed (no GPU?)")

    def synchronize(self):
        if lib().mir_lsq_stream_synchronize(self.handle) != 0:
            raise RuntimeError("stream synchronize failed")

    def close(self):
        if self.handle:
            lib().mir_lsq_stream_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceBuffer:
    """A device allocation filled from / read back to numpy (plumbing for tests and the bench)."""

    def __init__(self, array=None, nbytes=None, dtype=None, shape=None):
        if array is not None:
            array = np.ascontiguousarray(array)
            nbytes, dtype, shape = array.nbytes, array.dtype, array.shape
        self.nbytes, self.dtype, self.shape = int(nbytes), np.dtype(dtype), tuple(shape)
        self.ptr = lib().mir_lsq_device_malloc(max(self.nbytes, 8))
        if not self.ptr:
            raise MemoryError(f"hipMalloc({self.nbytes}) failed (no GPU?)")
        if array is not None:
            self.upload(array)

    def upload(self, array):
        array = np.ascontiguousarray(array, dtype=self.dtype)
        assert array.nbytes == self.nbytes
        if lib().mir_lsq_memcpy_h2d(self.ptr, array.ctypes.data, self.nbytes, None) != 0:
            raise RuntimeError("H2D copy failed")

    def upload_at(self, byte_offset, array):
        """Copy `array` into the allocation at `byte_offset` (building a large device array piece by piece)."""
        array = np.ascontiguousarray(array, dtype=self.dtype)
        assert 0 <= byte_offset and byte_offset + array.nbytes <= self.nbytes
        if lib().mir_lsq_memcpy_h2d(self.ptr + byte_offset, array.ctypes.data, array.nbytes, None) != 0:
            raise RuntimeError("H2D copy failed")

    def download(self):
        out = np.empty(self.shape, dtype=self.dtype)
        if lib().mir_lsq_memcpy_d2h(out.ctypes.data, self.ptr, self.nbytes, None) != 0:
            raise RuntimeError("D2H copy failed")
        return out

    def free(self):
        if self.ptr:
            lib().mir_lsq_device_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _as_fnptr(cb, ftype, wrap, keep):
    if cb is None:
        return C.c_void_p(None)
    if callable(cb):
        c = ftype(wrap(cb))
        keep.append(c)
        return C.cast(c, C.c_void_p)
    return C.c_void_p(int(cb))


def optimizeLeastSquares(f, m, x, l=None, u=None, g=None, tm=None, settings=None, dtype=np.float64,
                         fContext=None, gContext=None, options=None, gpu_entry=None):
    """High level nothrow API (LS:459-519). `x` is updated in place (numpy array) and returned with the result.

    f, g : python callables f(x, y) / g(x, J) working on numpy views of HOST memory (y and J are
           zero-filled before the call like the template tier does, LS:469, LS:482), or integer
           addresses of native callbacks (then fContext / gContext are passed through).
    tm   : optional python thread manager tm(count, task) where task(totalThreads, threadId, i)
           must be called for every i in [0, count) (LS:575-578).
    options : GpuOptions for the additive entry point (device callbacks, comm, stats, ...)."""
    L = lib()
    dbl = dtype == np.float64
    x = np.ascontiguousarray(x, dtype=dtype)
    n = x.size
    lo = np.full(n, -np.inf, dtype=dtype) if l is None else np.ascontiguousarray(l, dtype=dtype)
    up = np.full(n, np.inf, dtype=dtype) if u is None else np.ascontiguousarray(u, dtype=dtype)
    if settings is None:
        settings = LeastSquaresSettings(dtype)
    ft = _ftype(dtype)
    keep = []
    errors = []      # exceptions raised inside Python callbacks: ctypes would print and swallow them and the solve would go
                     # on with garbage -- they are recorded, the outputs poisoned with NaN (the solver then ends with
                     # numericError at its next check, LS:990 / 1117) and the first one is re-raised after the C call returns

    def wrap_f(fn):
        def cb(_ctx, m_, n_, xp, yp):
            yv = np.ctypeslib.as_array(yp, shape=(m_,))
            try:
                xv = np.ctypeslib.as_array(xp, shape=(n_,))
                yv[:] = 0
                fn(xv, yv)
            except BaseException as e:      # noqa: BLE001 -- re-raised below
                errors.append(e)
                yv[:] = np.nan
        return cb

    def wrap_g(fn):
        def cb(_ctx, m_, n_, xp, Jp):
            Jv = np.ctypeslib.as_array(Jp, shape=(m_ * n_,)).reshape(m_, n_)
            try:
                xv = np.ctypeslib.as_array(xp, shape=(n_,))
                Jv[:] = 0
                fn(xv, Jv)
            except BaseException as e:      # noqa: BLE001
                errors.append(e)
                Jv[:] = np.nan
        return cb

    fptr = _as_fnptr(f, ft, wrap_f, keep)
    gptr = _as_fnptr(g, ft, wrap_g, keep)
    tmptr = C.c_void_p(None)
    if tm is not None:
        def tm_cb(_ctx, count, task, taskfn):
            try:
                tm(count, lambda total, tid, i: taskfn(task, total, tid, i))
            except BaseException as e:      # noqa: BLE001
                errors.append(e)
        tmc = TM_FN(tm_cb)
        keep.append(tmc)
        tmptr = C.cast(tmc, C.c_void_p)
    use_gpu_entry = gpu_entry if gpu_entry is not None else (options is not None)
    suf = "d" if dbl else "s"
    if use_gpu_entry:
        fn = getattr(L, "mir_optimize_least_squares_gpu_" + suf)
        raw = fn(C.byref(settings), m, n, x.ctypes.data, lo.ctypes.data, up.ctypes.data,
                 C.byref(options) if options is not None else None, fContext, fptr, gContext, gptr, None, tmptr)
    else:
        # the reference's own entry point (LS:705-724): work / iwork are sized by the reference formulas
        wl = L.mir_least_squares_work_length(m, n)
        iwl = L.mir_least_squares_iwork_length(m, n)
        iwork = np.zeros(iwl + 4, dtype=np.int32)
        work_slice = _SliceD(wl, None)      # never dereferenced by this implementation
        iwork_slice = _SliceD(iwl, iwork.ctypes.data)
        fn = getattr(L, "mir_optimize_least_squares_" + suf)
        raw = fn(C.byref(settings), m, n, x.ctypes.data, lo.ctypes.data, up.ctypes.data, work_slice, iwork_slice,
                 fContext, fptr, gContext, gptr, None, tmptr)
    del keep
    if errors:
        raise errors[0]
    return LeastSquaresResult(raw), x


def optimize(f, m, x, l=None, u=None, g=None, tm=None, settings=None, dtype=np.float64, taskPool=None, **kw):
    """High level throwing API (LS:165-215): raises LeastSquaresException when status < 0.

    taskPool: optional concurrent.futures-like executor with `_max_workers`; mirrors the task-pool
    overload LS:184-215 (finite-difference columns are evaluated by the pool's threads)."""
    if taskPool is not None and tm is None:
        import threading
        workers = max(1, getattr(taskPool, "_max_workers", 1))
        ids = {}
        lock = threading.Lock()

        def tm(count, task):
            def run(i):
                with lock:
                    tid = ids.setdefault(threading.get_ident(), len(ids))
                task(workers, tid % workers, i)
            list(taskPool.map(run, range(count)))
    res, xo = optimizeLeastSquares(f, m, x, l, u, g, tm, settings, dtype, **kw)
    if res.status < 0:
        raise LeastSquaresException(res.status, res)
    return res, xo


def optimizeLeastSquaresBatched(model, x, t, data, l=None, u=None, settings=None, variant=0):
    """Many independent small fits, one wavefront per problem (mir_optimize_least_squares_batched_s, fp32).
    x: count x n starts (a copy is updated and returned), t: m (shared) or count x m, data: count x m.
    variant: BATCHED_* bits (per call). Returns (list of LeastSquaresResult, x)."""
    L = lib()
    x = np.array(x, dtype=np.float32, order="C")
    count, n = x.shape
    data = np.ascontiguousarray(data, dtype=np.float32)
    m = data.shape[1]
    t = np.ascontiguousarray(t, dtype=np.float32)
    t_stride = 0 if t.ndim == 1 else m
    lo = np.full(n, -np.inf, dtype=np.float32) if l is None else np.ascontiguousarray(l, dtype=np.float32)
    up = np.full(n, np.inf, dtype=np.float32) if u is None else np.ascontiguousarray(u, dtype=np.float32)
    if settings is None:
        settings = LeastSquaresSettings(np.float32)
    raw = (_Rs * count)()
    rc = L.mir_optimize_least_squares_batched_s(C.byref(settings), count, m, int(model), x.ctypes.data, lo.ctypes.data,
                                                up.ctypes.data, t.ctypes.data, t_stride, data.ctypes.data, raw,
                                                C.byref(BatchedOptions(variant=variant)))
    if rc != 0:
        raise RuntimeError(f"mir_optimize_least_squares_batched_s failed: {rc}")
    return [LeastSquaresResult(r) for r in raw], x


BATCHED_NO_LADDER = 1


def batchedPosvx(P, rhs):
    """The damped solve of the wave-per-problem kernel on its own (mir_lsq_batched_posvx_s): P count x n x n (lower
    triangles read), rhs count x n, n in (3, 8), fp32. Returns (x count x n, info count)."""
    L = lib()
    P = np.asarray(P, dtype=np.float32)
    count, n = P.shape[0], P.shape[1]
    Pp = np.zeros((count, 8, 8), dtype=np.float32); Pp[:, :n, :n] = P
    bp = np.zeros((count, 8), dtype=np.float32); bp[:, :n] = rhs
    dP, db = DeviceBuffer(Pp), DeviceBuffer(bp)
    dx = DeviceBuffer(nbytes=count * 32, dtype=np.float32, shape=(count, 8))
    di = DeviceBuffer(nbytes=count * 4, dtype=np.int32, shape=(count,))
    st = Stream()
    rc = L.mir_lsq_batched_posvx_s(count, n, dP.ptr, db.ptr, dx.ptr, di.ptr, st.handle)
    if rc != 0:
        raise RuntimeError(f"mir_lsq_batched_posvx_s failed: {rc}")
    st.synchronize()
    x, info = dx.download()[:, :n].copy(), di.download().copy()
    for b in (dP, db, dx, di):
        b.free()
    return x, info


def solveBoxQP(P, q, l, u, x=None, settings=None, dtype=np.float64, unconstrainedSolution=False):
    """argmin_x(1/2 xPx + qx) : l <= x <= u on the device (boxcqp.d:85-102 / 122-379).
    P: only the lower triangle is read. Returns (BoxQPStatus, x, active-set iterations)."""
    L = lib()
    P = np.ascontiguousarray(P, dtype=dtype)
    n = P.shape[0]
    q = np.ascontiguousarray(q, dtype=dtype)
    l = np.ascontiguousarray(l, dtype=dtype)
    u = np.ascontiguousarray(u, dtype=dtype)
    xo = np.zeros(n, dtype=dtype) if x is None else np.ascontiguousarray(x, dtype=dtype).copy()
    if settings is None:
        settings = BoxQPSettings(dtype)
    it = C.c_int(0)
    fn = L.mir_solve_box_qp_gpu_d if dtype == np.float64 else L.mir_solve_box_qp_gpu_s
    st = fn(C.byref(settings), n, P.ctypes.data, q.ctypes.data, l.ctypes.data, u.ctypes.data, xo.ctypes.data,
            1 if unconstrainedSolution else 0, C.byref(it))
    return BoxQPStatus(st), xo, it.value


def jtj(J, y, y_old=None, dx=None, dtype=np.float64):
    """Unit-level access to the J^T J + J^T y kernels (mir_lsq_jtj_*; with dx: the Broyden REWRITE kernels, the literal
    restatement of LS:1003-1006). Returns (JJ full symmetric, Jy, J_after, kernel_ms)."""
    L = lib()
    J = np.ascontiguousarray(J, dtype=dtype)
    m, n = J.shape
    dJ = DeviceBuffer(J)
    dy = DeviceBuffer(np.ascontiguousarray(y, dtype=dtype))
    broyden = dx is not None
    dyo = DeviceBuffer(np.ascontiguousarray(y_old if broyden else y, dtype=dtype))
    ddx = DeviceBuffer(np.ascontiguousarray(dx if broyden else np.zeros(n), dtype=dtype))
    dJJ = DeviceBuffer(nbytes=n * n * J.itemsize, dtype=dtype, shape=(n, n))
    dJy = DeviceBuffer(nbytes=n * J.itemsize, dtype=dtype, shape=(n,))
    ms = C.c_float(0)
    fn = L.mir_lsq_jtj_d if dtype == np.float64 else L.mir_lsq_jtj_s
    rc = fn(m, n, dJ.ptr, dy.ptr, dyo.ptr, ddx.ptr, 1 if broyden else 0, dJJ.ptr, dJy.ptr, None, C.byref(ms))
    if rc != 0:
        raise RuntimeError(f"mir_lsq_jtj failed: {rc}")
    out = dJJ.download(), dJy.download(), dJ.download(), ms.value
    for b in (dJ, dy, dyo, ddx, dJJ, dJy):
        b.free()
    return out


def fd_jtj(Yrm, twh, y, diff=False):
    """Unit-level access to the J^T J kernel with the finite-difference fill fused in (mir_lsq_fd_jtj_d).
    Yrm: m x 2n row-major (+h / -h residual pairs) -- or, diff=True, the m x n difference panel (mir_lsq_fd_diff_jtj_d).
    Returns (J, JJ full symmetric, Jy, kernel_ms)."""
    L = lib()
    Yrm = np.ascontiguousarray(Yrm, dtype=np.float64)
    m, n2 = Yrm.shape
    n = n2 if diff else n2 // 2
    dY = DeviceBuffer(Yrm)
    dt = DeviceBuffer(np.ascontiguousarray(twh, dtype=np.float64))
    dy = DeviceBuffer(np.ascontiguousarray(y, dtype=np.float64))
    dJ = DeviceBuffer(nbytes=m * n * 8, dtype=np.float64, shape=(m, n))
    dJJ = DeviceBuffer(nbytes=n * n * 8, dtype=np.float64, shape=(n, n))
    dJy = DeviceBuffer(nbytes=n * 8, dtype=np.float64, shape=(n,))
    ms = C.c_float(0)
    rc = (L.mir_lsq_fd_diff_jtj_d if diff else L.mir_lsq_fd_jtj_d)(m, n, dY.ptr, dt.ptr, dy.ptr, dJ.ptr, dJJ.ptr, dJy.ptr, None, C.byref(ms))
    if rc != 0:
        raise RuntimeError(f"mir_lsq_fd_jtj_d failed: {rc}")
    out = dJ.download(), dJJ.download(), dJy.download(), ms.value
    for b in (dY, dt, dy, dJ, dJJ, dJy):
        b.free()
    return out


# ---- fitSpline (fit_splie.d:26-85): a caller of the LM path -----------------------------------------------

class Spline:
    """C2 cubic spline with not-a-knot ends in Hermite form (what mir.interpolate.spline's default configuration
    builds); evaluation and derivatives go through the library's host code (mir_spline_*)."""

    def __init__(self, x, values, dtype=np.float64):
        self.dtype = dtype
        self.x = np.ascontiguousarray(x, dtype=dtype)
        self.values = np.ascontiguousarray(values, dtype=dtype)
        self.derivatives = np.zeros_like(self.values)
        suf = "d" if dtype == np.float64 else "s"
        getattr(lib(), "mir_spline_c2_derivatives_" + suf)(self.x.size, self.x.ctypes.data, self.values.ctypes.data,
                                                         self.derivatives.ctypes.data)
        self._eval = getattr(lib(), "mir_spline_eval_" + suf)

    def withTwoDerivatives(self, t):
        out = np.zeros(3, dtype=self.dtype)
        self._eval(self.x.size, self.x.ctypes.data, self.values.ctypes.data, self.derivatives.ctypes.data, t, out.ctypes.data)
        return out

    def __call__(self, t):
        if np.ndim(t) == 0:
            return self.withTwoDerivatives(float(t))[0]
        return np.array([self.withTwoDerivatives(float(v))[0] for v in np.asarray(t).ravel()]).reshape(np.shape(t))


class FitSplineResult:
    """FitSplineResult!T (fit_splie.d:7-13)."""

    def __init__(self, leastSquaresResult, spline):
        self.leastSquaresResult = leastSquaresResult
        self.spline = spline


def fit_spline_residuals(points, x, lambda_, splineY):
    """The residual vector fitSpline minimises (FS:60-84), from the library's host code (double)."""
    points = np.ascontiguousarray(points, dtype=np.float64)
    x = np.ascontiguousarray(x, dtype=np.float64)
    v = np.ascontiguousarray(splineY, dtype=np.float64)
    m = points.shape[0] + (1 if lambda_ == 0 else 0)
    y = np.zeros(m)
    lib().mir_fit_spline_residuals_d(points.shape[0], points.ctypes.data, x.size, x.ctypes.data, lambda_, v.ctypes.data, m,
                                     y.ctypes.data)
    return y


def fitSpline(settings, points, x, l, u, lambda_=0.0, dtype=np.float64):
    """fitSpline (fit_splie.d:26-85): least-squares fit of the spline values at the fixed knots `x` to `points`
    (k x 2), bounds l <= spline(x) <= u, optional smoothness weight lambda_. Raises like the reference: Exception for
    too few points (FS:47-51), LeastSquaresException for a negative LM status (through `optimize`, LS:175-179)."""
    L = lib()
    points = np.ascontiguousarray(points, dtype=dtype)
    x = np.ascontiguousarray(x, dtype=dtype)
    lo = np.ascontiguousarray(l, dtype=dtype)
    up = np.ascontiguousarray(u, dtype=dtype)
    if not lambda_ >= 0:
        raise ValueError("fitSpline: lambda has to be non-negative")        # the reference's in-contract
    y = np.zeros(x.size, dtype=dtype)
    d = np.zeros(x.size, dtype=dtype)
    dbl = dtype == np.float64
    raw = (_Rd if dbl else _Rs)()
    fn = L.mir_fit_spline_d if dbl else L.mir_fit_spline_s
    rc = fn(C.byref(settings) if settings is not None else None, points.shape[0], points.ctypes.data, x.size,
            x.ctypes.data, lo.ctypes.data, up.ctypes.data, lambda_, None, y.ctypes.data, d.ctypes.data, C.byref(raw))
    if rc == 1:
        raise Exception("fitSpline: points.length has to be greater or equal x.length when lambda is 0.0")
    if rc != 0:
        raise ValueError("fitSpline: bad argument")
    res = LeastSquaresResult(raw)
    if res.status < 0:
        raise LeastSquaresException(int(res.status), res)
    return FitSplineResult(res, Spline(x, y, dtype))
