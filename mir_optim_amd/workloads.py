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


# ---- resident-J solver (include/mir_optim_amd_resident.hpp; models in csrc/workloads_resident.hip) --------------------

class ResidentStats(C.Structure):
    """mir_lsq_resident_stats (times in 10 ns ticks of workgroup 0)."""
    _fields_ = ([(k, C.c_uint64) for k in ("rounds", "passes", "accepted", "rejected", "step_guard_rejects", "jacobian_full",
                                            "jacobian_broyden", "qp_active_set_passes", "elided_evaluations", "t_total", "t_stage",
                                            "t_worker", "t_group", "t_total_wait", "t_solver", "t_solve_body", "t_cmd_wait", "t_w_eval", "t_w_fd",
                                            "t_w_prod", "t_w_mma", "lookahead_rejections", "t_look", "t_unpack", "t_publish")]
                + [(k, C.c_uint32) for k in ("abort_code", "grid", "rows", "groups")])

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


class ResidentOptions(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("variant", C.c_uint32), ("stream", C.c_void_p), ("workspace", C.c_void_p),
                ("workspace_bytes", C.c_size_t), ("trace_records", C.c_void_p), ("trace_capacity", C.c_uint32),
                ("max_workgroups", C.c_uint32), ("trace_count", C.c_void_p), ("stats", C.c_void_p)]

    def __init__(self, **kw):
        super().__init__(**kw)
        self.struct_size = C.sizeof(ResidentOptions)


RESIDENT_NO_NULL_SKIP = 1
RESIDENT_UNBOUNDED = 2
RESIDENT_NO_LOOKAHEAD = 4
RESIDENT_NO_STAMPS = 8
RESIDENT_ANALYTIC_JACOBIAN = 32
RESIDENT_DEBUG_DROP_WORKGROUP = 16
RESIDENT_MODELS = {"gauss5": 0, "tanh32": 1, "gauss3": 2, "exp_decay1": 3}


class ResidentDoesNotFit(Exception):
    """The problem's slices do not fit the chip's LDS (launch_resident returns -3): use the launch-chain path."""


class Resident:
    """A problem on the resident-J path: ONE cooperative launch runs the whole LM loop with J, y and the row data in LDS.
    `rowdata`: m x nd table of the model (gauss*: [t_i, data_i]; tanh32: [a_i (32), b_i]; exp_decay1: [t_i, data_i])."""

    def __init__(self, model, rowdata, stream=None, max_workgroups=0, fallback=None):
        self.model = RESIDENT_MODELS[model] if isinstance(model, str) else int(model)
        rowdata = np.ascontiguousarray(rowdata, dtype=np.float64)
        self.m = rowdata.shape[0]
        self.stream = stream or api.Stream()
        # a DeviceProblem with the same residual (the launch chain): used when the slices do not fit the LDS, and when a launch
        # gave up on a hand-off (abort_code != 0: a device shared with other work -- a scheduling fact, not a numeric one)
        self.fallback = fallback
        self.max_workgroups = int(max_workgroups)
        WL = api.workloads_lib()
        out4 = (C.c_int * 4)()
        out2 = (C.c_size_t * 2)()
        self.plan_rc = WL.wl_resident_plan(C.c_int(self.model), C.c_size_t(self.m), C.c_int(self.max_workgroups or 256), out4, out2)
        self.grid, self.rows, self.groups, self.n = (int(v) for v in out4)
        self.lds_bytes, self.workspace_bytes = (int(v) for v in out2)
        if self.plan_rc == 0:
            self.d_rows = api.DeviceBuffer(rowdata)
            self.d_ws = api.DeviceBuffer(nbytes=self.workspace_bytes, dtype=np.uint8, shape=(self.workspace_bytes,))
            self.d_x = api.DeviceBuffer(nbytes=8 * 3 * self.n, dtype=np.float64, shape=(3 * self.n,))     # x | lower | upper
            # result record | trace count | statistics in ONE allocation: one copy back per solve
            self._out_bytes = 64 + C.sizeof(ResidentStats)
            self.d_out = api.DeviceBuffer(nbytes=self._out_bytes, dtype=np.uint8, shape=(self._out_bytes,))
            self.d_trace = None

    @classmethod
    def gauss_sum(cls, t, data, K=5, **kw):
        return cls({5: "gauss5", 3: "gauss3"}[K], np.stack([t, data], axis=1), **kw)

    @classmethod
    def tanh_linear(cls, A, b, **kw):
        assert A.shape[1] == 32
        return cls("tanh32", np.concatenate([A, np.asarray(b)[:, None]], axis=1), **kw)

    def upload_point(self, x0, l=None, u=None):
        n = self.n
        xlu = np.empty(3 * n)
        xlu[:n] = x0
        xlu[n:2 * n] = -np.inf if l is None else l
        xlu[2 * n:] = np.inf if u is None else u
        self.d_x.upload(xlu)

    def launch(self, settings=None, trace_capacity=0, variant=0):
        """Enqueue one solve from the uploaded point (asynchronous). Returns launch_resident's code."""
        n = self.n
        s = settings if settings is not None else api.LeastSquaresSettings()
        o = ResidentOptions()
        o.variant = variant
        o.stream = self.stream.handle
        o.workspace = self.d_ws.ptr
        o.workspace_bytes = self.workspace_bytes
        o.max_workgroups = self.max_workgroups
        if trace_capacity:
            if self.d_trace is None or self.d_trace.nbytes < 40 * trace_capacity:
                self.d_trace = api.DeviceBuffer(nbytes=40 * trace_capacity, dtype=np.uint8, shape=(40 * trace_capacity,))
            o.trace_records = self.d_trace.ptr
            o.trace_capacity = trace_capacity
        o.trace_count = self.d_out.ptr + 32
        o.stats = self.d_out.ptr + 64
        st = C.c_int(0)
        rc = api.workloads_lib().wl_resident_launch_d(
            C.c_int(self.model), C.byref(s), C.c_size_t(self.m), C.c_void_p(self.d_x.ptr), C.c_void_p(self.d_x.ptr + 8 * n),
            C.c_void_p(self.d_x.ptr + 16 * n), C.c_void_p(self.d_rows.ptr), C.c_void_p(self.d_out.ptr), C.byref(o), C.byref(st))
        self.last_status_out = st.value
        return rc

    def solve(self, x0, l=None, u=None, settings=None, trace=None, variant=0, **fallback_kw):
        """(LeastSquaresResult, x, stats dict). trace: an api.Trace to fill (same events as the launch-chain path's)."""
        if self.plan_rc != 0:
            if self.plan_rc == -3 and self.fallback is not None:
                res, x = self.fallback.solve(x0, l, u, settings=settings, trace=trace, **fallback_kw)
                return res, x, None
            raise ResidentDoesNotFit(f"resident plan failed with {self.plan_rc}")
        n = self.n
        self.upload_point(x0, l, u)
        cap = int(trace.header.capacity) if trace is not None else 0
        rc = self.launch(settings, cap, variant)
        if rc != 0:
            if rc == -1 and self.last_status_out:
                raw = api._Rd(self.last_status_out, 0, 0, 0, np.inf, 0.0)
                return api.LeastSquaresResult(raw), np.array(x0, dtype=np.float64), None
            raise RuntimeError(f"launch_resident failed with {rc}")
        self.stream.synchronize()
        out = self.d_out.download().tobytes()
        raw = api._Rd.from_buffer_copy(out[:32])
        x = self.d_x.download()[:n].copy()
        stats = ResidentStats.from_buffer_copy(out[64:64 + C.sizeof(ResidentStats)])
        if stats.abort_code and self.fallback is not None:
            # the launch gave up on a hand-off between its workgroups: the same problem through the launch chain (ordinary
            # launches that wait for nobody), in a fresh launch of this process; the abort code stays in the statistics
            res, xf = self.fallback.solve(x0, l, u, settings=settings, trace=trace, **fallback_kw)
            d = stats.as_dict()
            d["fallback"] = "launch chain"
            return res, xf, d
        if trace is not None:
            cnt = int(np.frombuffer(out[32:36], dtype=np.uint32)[0])
            trace.header.count = cnt
            k = min(cnt, cap)
            if k:
                C.memmove(trace._buf, self.d_trace.download().tobytes()[:40 * k], 40 * k)
        return api.LeastSquaresResult(raw), x, stats.as_dict()
