"""Synthetic workloads of SURVEY.md section 8d as DEVICE-callback problems (plumbing around
csrc/workloads.hip): build the inputs on the host with the counter RNG, upload them, and expose the
native callback addresses + context that mir_optimize_least_squares_gpu_* needs."""
import ctypes as C

import numpy as np

from . import api


class _TanhCtx(C.Structure):
    _fields_ = [("A", C.c_void_p), ("b", C.c_void_p), ("stream", C.c_void_p), ("read_a_once", C.c_int)]


class _CurveCtx(C.Structure):
    _fields_ = [("t", C.c_void_p), ("data", C.c_void_p), ("stream", C.c_void_p), ("kind", C.c_int)]


def uniform(seed, count, offset=0):
    out = np.empty(count, dtype=np.float64)
    api.workloads_lib().wl_uniform(seed, offset, count, out.ctypes.data)
    return out


def _addr(name):
    return C.cast(getattr(api.workloads_lib(), name), C.c_void_p).value


def tanh_linear_data(m, n, row_offset=0, noise=1e-3):
    """Host arrays of the cfg-3/4 family for rows [row_offset, row_offset + m) (bit-identical for
    the GPU run and the CPU baseline)."""
    A = np.empty((m, n), dtype=np.float64)
    b = np.empty(m, dtype=np.float64)
    xs = np.empty(n, dtype=np.float64)
    x0 = np.empty(n, dtype=np.float64)
    api.workloads_lib().wl_tanh_linear_generate(m, n, row_offset, noise, A.ctypes.data, b.ctypes.data,
                                                xs.ctypes.data, x0.ctypes.data)
    return dict(A=A, b=b, xstar=xs, x0=x0, m=m, n=n)


class DeviceProblem:
    """Common part: stream, context, callback addresses, GpuOptions."""

    suffix = "d"

    def options(self, flags=0, stats=None, comm=None, workspace=None, batched=False, fd_batch=0, trace=None, variant=0):
        o = api.GpuOptions()
        o.flags = api.DEVICE_CALLBACKS | flags
        o.variant = variant
        o.stream = self.stream.handle
        o.comm = getattr(comm, "handle", comm)       # a raw handle or a parallel.HostAllreduceComm
        o.workspace = workspace
        if batched and self.fb is not None:
            o.fbContext = C.addressof(self.ctx)
            o.fb = self.fb
            o.fd_batch = fd_batch
            if getattr(self, "fbr", None) is not None and batched != "pointmajor":
                o.fbRowMajor = self.fbr          # row-major FD panel: the fill is fused into the J^T J kernel
            if getattr(self, "fbd", None) is not None and batched not in ("pointmajor", "rowmajor"):
                o.fbRowMajorDiff = self.fbd      # m x n difference panel: half the bytes between the two kernels
        if stats is not None:
            o.stats = C.pointer(stats)
        if trace is not None:
            o.trace = C.pointer(trace.header)
        return o

    def solve(self, x0, l=None, u=None, settings=None, analytic=False, **opt_kw):
        opts = self.options(**opt_kw)
        out = api.optimizeLeastSquares(self.f, self.m, np.array(x0, dtype=self.dtype), l, u,
                                       g=self.g if analytic else None, settings=settings, dtype=self.dtype,
                                       fContext=C.addressof(self.ctx), gContext=C.addressof(self.ctx), options=opts)
        comm = opt_kw.get("comm")
        if hasattr(comm, "check"):
            comm.check()         # an exception inside the all-reduce callback: the solve ended with numericError, raise the cause
        return out


class TanhLinear(DeviceProblem):
    def __init__(self, A, b, dtype=np.float64, stream=None):
        self.dtype = dtype
        self.m, self.n = A.shape
        self.stream = stream or api.Stream()
        self.dA = api.DeviceBuffer(np.ascontiguousarray(A, dtype=dtype))
        self.db = api.DeviceBuffer(np.ascontiguousarray(b, dtype=dtype))
        self.ctx = _TanhCtx(self.dA.ptr, self.db.ptr, self.stream.handle)
        suf = "d" if dtype == np.float64 else "s"
        self.f = _addr("wl_tanh_linear_f_" + suf)
        self.g = _addr("wl_tanh_linear_g_" + suf)
        self.fb = _addr("wl_tanh_linear_fb_d") if dtype == np.float64 else None
        self.fbr = _addr("wl_tanh_linear_fbr_d") if dtype == np.float64 else None
        self.fbd = _addr("wl_tanh_linear_fbd_d") if dtype == np.float64 else None


class TanhLinearView(TanhLinear):
    """The tanh-linear problem over rows [row0, row0 + m) of device arrays that already hold A (row-major, n columns) and b
    (f64): several row shards -- and the unsharded problem -- as views of one resident data set."""

    def __init__(self, dA, db, row0, m, n, stream=None):
        self.dtype = np.float64
        self.m, self.n = int(m), int(n)
        self.stream = stream or api.Stream()
        self.dA, self.db = dA, db                      # kept alive; not owned
        self.ctx = _TanhCtx(dA.ptr + int(row0) * int(n) * 8, db.ptr + int(row0) * 8, self.stream.handle)
        self.f = _addr("wl_tanh_linear_f_d")
        self.g = _addr("wl_tanh_linear_g_d")
        self.fb = _addr("wl_tanh_linear_fb_d")
        self.fbr = _addr("wl_tanh_linear_fbr_d")
        self.fbd = _addr("wl_tanh_linear_fbd_d")


class Curve(DeviceProblem):
    """kind = 'gauss_sum' | 'exp_decay0' | 'exp_decay1'."""

    def __init__(self, kind, t, data, dtype=np.float64, stream=None):
        self.dtype = dtype
        self.m = len(t)
        self.stream = stream or api.Stream()
        self.dt = api.DeviceBuffer(np.ascontiguousarray(t, dtype=dtype))
        self.dd = api.DeviceBuffer(np.ascontiguousarray(data, dtype=dtype))
        k = {"gauss_sum": 0, "exp_decay0": 0, "exp_decay1": 1}[kind]
        self.ctx = _CurveCtx(self.dt.ptr, self.dd.ptr, self.stream.handle, k)
        suf = "d" if dtype == np.float64 else "s"
        self.f = _addr(("wl_gauss_sum_f_" if kind == "gauss_sum" else "wl_exp_decay_f_") + suf)
        self.g = None
        gs64 = kind == "gauss_sum" and dtype == np.float64
        self.fb = _addr("wl_gauss_sum_fb_d") if gs64 else None       # batched residuals: one launch for the 2n FD points
        self.fbr = _addr("wl_gauss_sum_fbr_d") if gs64 else None
