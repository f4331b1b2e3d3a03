"""ctypes binding of the CPU ORACLE (oracle/liblm_oracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py. Nothing under mir_optim_amd/ imports this module.

The oracle restates /root/reference/source/mir/optim/least_squares.d:877-1176 and
boxcqp.d:122-379 (see lm_oracle.h). Parity pin: the reference's own known-answer
unittests T1-T6 / TQ (tests/test_oracle_reference_kats.py).
"""
import ctypes as C
import glob
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liblm_oracle.so")


def build(force=False):
    """Compile the oracle with gcc (seconds). `make` decides what is stale: the Makefile names every source of the library
    (an edit to lm_batched_fused.c alone rebuilds it too); `force` rebuilds regardless."""
    subprocess.check_call(["make", "-C", _HERE] + (["-B"] if force else []) + ["liblm_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


class QPSettingsD(C.Structure):
    _fields_ = [("relTolerance", C.c_double), ("absTolerance", C.c_double), ("maxIterations", C.c_uint32)]


class QPSettingsS(C.Structure):
    _fields_ = [("relTolerance", C.c_float), ("absTolerance", C.c_float), ("maxIterations", C.c_uint32)]


_FIELDS = ["jacobianEpsilon", "absTolerance", "relTolerance", "gradTolerance", "maxGoodResidual",
           "maxStep", "maxLambda", "minLambda", "minStepQuality", "goodStepQuality",
           "lambdaIncrease", "lambdaDecrease"]


class SettingsD(C.Structure):
    _fields_ = ([("maxIterations", C.c_uint32), ("maxAge", C.c_uint32)]
                + [(k, C.c_double) for k in _FIELDS] + [("qpSettings", QPSettingsD)])


class SettingsS(C.Structure):
    _fields_ = ([("maxIterations", C.c_uint32), ("maxAge", C.c_uint32)]
                + [(k, C.c_float) for k in _FIELDS] + [("qpSettings", QPSettingsS)])


class ResultD(C.Structure):
    _fields_ = [("status", C.c_int32), ("iterations", C.c_uint32), ("fCalls", C.c_uint32),
                ("gCalls", C.c_uint32), ("residual", C.c_double), ("lambda_", C.c_double)]


class ResultS(C.Structure):
    _fields_ = [("status", C.c_int32), ("iterations", C.c_uint32), ("fCalls", C.c_uint32),
                ("gCalls", C.c_uint32), ("residual", C.c_float), ("lambda_", C.c_float)]


F_D = C.CFUNCTYPE(None, C.c_void_p, C.c_size_t, C.c_size_t, C.POINTER(C.c_double), C.POINTER(C.c_double))
F_S = C.CFUNCTYPE(None, C.c_void_p, C.c_size_t, C.c_size_t, C.POINTER(C.c_float), C.POINTER(C.c_float))
TRACE = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_uint32, C.c_double, C.c_double, C.c_double, C.c_double)
ALLREDUCE = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_double), C.c_size_t)


class Options(C.Structure):
    _fields_ = [("trace", TRACE), ("trace_ctx", C.c_void_p),
                ("allreduce", ALLREDUCE), ("allreduce_ctx", C.c_void_p),
                ("use_openblas", C.c_int)]


class TanhLinearCtx(C.Structure):
    _fields_ = [("A", C.c_void_p), ("b", C.c_void_p)]


class GaussSumCtx(C.Structure):
    _fields_ = [("t", C.c_void_p), ("data", C.c_void_p)]


class ExpDecayCtx(C.Structure):
    _fields_ = [("t", C.c_void_p), ("data", C.c_void_p), ("kind", C.c_int)]


STATUS = {-1: "maxIterations", 0: "furtherImprovement", 1: "xConverged", 2: "gConverged", 3: "fConverged",
          -32: "badBounds", -31: "badGuess", -30: "badMinStepQuality", -29: "badGoodStepQuality",
          -28: "badStepQuality", -27: "badLambdaParams", -26: "numericError"}

_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.lmo_work_length.restype = C.c_size_t
        L.lmo_work_length.argtypes = [C.c_size_t, C.c_size_t]
        L.lmo_iwork_length.restype = C.c_size_t
        L.lmo_iwork_length.argtypes = [C.c_size_t, C.c_size_t]
        L.lmo_box_qp_work_length.restype = C.c_size_t
        L.lmo_box_qp_work_length.argtypes = [C.c_size_t]
        L.lmo_box_qp_iwork_length.restype = C.c_size_t
        L.lmo_box_qp_iwork_length.argtypes = [C.c_size_t]
        L.lmo_status_string.restype = C.c_char_p
        L.lmo_status_string.argtypes = [C.c_int]
        L.lmo_optimize_d.restype = ResultD
        L.lmo_optimize_d.argtypes = [C.POINTER(SettingsD), C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.POINTER(Options)]
        L.lmo_optimize_s.restype = ResultS
        L.lmo_optimize_s.argtypes = [C.POINTER(SettingsS), C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.POINTER(Options)]
        for suf, qs in (("d", QPSettingsD), ("s", QPSettingsS)):
            fn = getattr(L, "lmo_solve_box_qp_" + suf)
            fn.restype = C.c_int
            fn.argtypes = [C.POINTER(qs), C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                           C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
            fn = getattr(L, "lmo_posvx_" + suf)
            fn.restype = C.c_int
            fn.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_char_p, C.c_void_p,
                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.lmo_openblas_load.restype = C.c_int
        L.lmo_openblas_load.argtypes = [C.c_char_p]
        L.lmo_openblas_set_threads.restype = C.c_int
        L.lmo_openblas_set_threads.argtypes = [C.c_int]
        L.lmo_set_omp_threads.restype = None
        L.lmo_set_omp_threads.argtypes = [C.c_int]
        L.wlc_uniform.restype = None
        L.wlc_uniform.argtypes = [C.c_uint64, C.c_uint64, C.c_size_t, C.c_void_p]
        _lib = L
    return _lib


def default_settings(dtype=np.float64):
    s = SettingsD() if dtype == np.float64 else SettingsS()
    (lib().lmo_settings_init_d if dtype == np.float64 else lib().lmo_settings_init_s)(C.byref(s))
    return s


def uniform(seed, count, offset=0):
    out = np.empty(count, dtype=np.float64)
    lib().wlc_uniform(seed, offset, count, out.ctypes.data)
    return out


def _wrap_callback(fn, m, n, out_len, ftype, np_dtype):
    """Wrap a python callable fn(x: ndarray[n], out: ndarray) as a C callback."""
    def cb(_ctx, m_, n_, xp, yp):
        x = np.ctypeslib.as_array(xp, shape=(n_,))
        y = np.ctypeslib.as_array(yp, shape=(out_len(m_, n_),))
        fn(x, y)
    return ftype(cb)


def native_fn(name):
    """Address of a native callback exported by the oracle library (e.g. 'wlc_tanh_linear_f')."""
    return C.cast(getattr(lib(), name), C.c_void_p).value


def optimize(f, m, x0, lower=None, upper=None, g=None, settings=None, dtype=np.float64,
             trace=None, allreduce=None, use_openblas=False, fctx=None, gctx=None):
    """Run the oracle LM solver.

    f, g: python callables f(x, y) / g(x, J) filling numpy views in place, OR integer
          addresses of native callbacks (with fctx/gctx = address of their context).
    Returns (ResultD|ResultS, x)."""
    L = lib()
    dbl = dtype == np.float64
    x = np.array(x0, dtype=dtype).copy()
    n = x.size
    lo = np.full(n, -np.inf, dtype=dtype) if lower is None else np.array(lower, dtype=dtype)
    up = np.full(n, np.inf, dtype=dtype) if upper is None else np.array(upper, dtype=dtype)
    if settings is None:
        settings = default_settings(dtype)
    work = np.empty(L.lmo_work_length(m, n), dtype=dtype)
    iwork = np.zeros(L.lmo_iwork_length(m, n) + 4, dtype=np.int32)
    ft = F_D if dbl else F_S
    keep = []
    if callable(f):
        fcb = _wrap_callback(f, m, n, lambda m_, n_: m_, ft, dtype)
        keep.append(fcb)
        fptr = C.cast(fcb, C.c_void_p)
    else:
        fptr = C.c_void_p(f)
    if g is None:
        gptr = C.c_void_p(None)
    elif callable(g):
        def g2(xv, Jflat, g=g, n=n):
            g(xv, Jflat.reshape(-1, n))
        gcb = _wrap_callback(g2, m, n, lambda m_, n_: m_ * n_, ft, dtype)
        keep.append(gcb)
        gptr = C.cast(gcb, C.c_void_p)
    else:
        gptr = C.c_void_p(g)
    opt = Options()
    if trace is not None:
        tcb = TRACE(lambda _c, ev, it, lam, res, tres, dxd: trace(ev, it, lam, res, tres, dxd))
        keep.append(tcb)
        opt.trace = tcb
    if allreduce is not None:
        def ar(_c, p, cnt):
            buf = np.ctypeslib.as_array(p, shape=(cnt,))
            allreduce(buf)
        acb = ALLREDUCE(ar)
        keep.append(acb)
        opt.allreduce = acb
    opt.use_openblas = 1 if use_openblas else 0
    fn = L.lmo_optimize_d if dbl else L.lmo_optimize_s
    res = fn(C.byref(settings), m, n, x.ctypes.data, lo.ctypes.data, up.ctypes.data,
             work.ctypes.data, iwork.ctypes.data, fctx, fptr, gctx, gptr, C.byref(opt))
    del keep
    return res, x


def solve_box_qp(P, q, l, u, settings=None, dtype=np.float64, x0=None, unconstrained_solution=False):
    """solveBoxQP (boxcqp.d:122-379). P row-major, lower triangle meaningful. Returns (status, x, iters)."""
    L = lib()
    dbl = dtype == np.float64
    P = np.array(P, dtype=dtype, order="C").copy()
    n = P.shape[0]
    q = np.array(q, dtype=dtype); l = np.array(l, dtype=dtype); u = np.array(u, dtype=dtype)
    x = np.zeros(n, dtype=dtype) if x0 is None else np.array(x0, dtype=dtype).copy()
    if settings is None:
        settings = default_settings(dtype).qpSettings
    work = np.empty(L.lmo_box_qp_work_length(n), dtype=dtype)
    iwork = np.zeros(L.lmo_box_qp_iwork_length(n) + 4, dtype=np.int32)
    it = C.c_int(0)
    fn = L.lmo_solve_box_qp_d if dbl else L.lmo_solve_box_qp_s
    st = fn(C.byref(settings), n, P.ctypes.data, q.ctypes.data, l.ctypes.data, u.ctypes.data,
            x.ctypes.data, 1 if unconstrained_solution else 0, work.ctypes.data, iwork.ctypes.data, 1, C.byref(it))
    return st, x, it.value


def posvx(A, b, dtype=np.float64):
    """Restated ?posvx('E','L'). A symmetric (full); returns dict like scipy's dposvx outputs."""
    L = lib()
    dbl = dtype == np.float64
    a = np.array(A, dtype=dtype, order="F").copy()
    n = a.shape[0]
    af = np.zeros((n, n), dtype=dtype, order="F")
    s = np.zeros(n, dtype=dtype)
    bb = np.array(b, dtype=dtype).copy()
    x = np.zeros(n, dtype=dtype)
    sc = (C.c_double if dbl else C.c_float)
    rcond, ferr, berr = sc(0), sc(0), sc(0)
    work = np.zeros(3 * n + 1, dtype=dtype)
    iwork = np.zeros(n + 1, dtype=np.int32)
    equed = C.create_string_buffer(2)
    fn = L.lmo_posvx_d if dbl else L.lmo_posvx_s
    info = fn(n, a.ctypes.data, n, af.ctypes.data, n, equed, s.ctypes.data, bb.ctypes.data, x.ctypes.data,
              C.addressof(rcond), C.addressof(ferr), C.addressof(berr), work.ctypes.data, iwork.ctypes.data)
    return dict(info=info, x=x, equed=equed.value[:1].decode(), s=s, rcond=rcond.value, berr=berr.value,
                a_s=a, af=af)


def posvx_fused_s(A, b):
    """lmo_posvx_fused_s: float ?posvx('E','L') with fused multiply-adds (the device's posvx_rows arithmetic).
    A symmetric (full or lower); returns (info, x, equilibrated)."""
    L = lib()
    a = np.array(A, dtype=np.float32, order="F").copy()
    n = a.shape[0]
    bb = np.array(b, dtype=np.float32).copy()
    x = np.zeros(n, dtype=np.float32)
    eq = C.c_int(0)
    L.lmo_posvx_fused_s.restype = C.c_int
    L.lmo_posvx_fused_s.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
    info = L.lmo_posvx_fused_s(n, a.ctypes.data, n, bb.ctypes.data, x.ctypes.data, C.byref(eq))
    return info, x, bool(eq.value)


def optimize_batched_fused_pad8_s(settings, t, basis, data, x0, lower=None, upper=None):
    """lmo_optimize_batched_fused_pad8_s (lm_batched_fused.c): ONE float fit of cfg 5's padded exponential-decay model with the
    arithmetic of the device's wave-per-problem kernel. settings: an lmo_settings_s-layout ctypes structure (LeastSquaresSettings
    of the product has the same layout: LS:85-123); basis: m x 4 float32 as the device tabulated it.
    Returns ((status, iterations, fCalls, gCalls, residual, lambda), x)."""
    L = lib()

    class Res(C.Structure):
        _fields_ = [("status", C.c_int32), ("iterations", C.c_uint32), ("fCalls", C.c_uint32), ("gCalls", C.c_uint32),
                    ("residual", C.c_float), ("lam", C.c_float)]
    t = np.ascontiguousarray(t, dtype=np.float32)
    basis = np.ascontiguousarray(basis, dtype=np.float32)
    data = np.ascontiguousarray(data, dtype=np.float32)
    m = t.size
    assert basis.size == 4 * m and data.size == m
    x = np.array(x0, dtype=np.float32).copy()
    assert x.size == 8
    lo = np.full(8, -np.inf, dtype=np.float32) if lower is None else np.ascontiguousarray(lower, dtype=np.float32)
    up = np.full(8, np.inf, dtype=np.float32) if upper is None else np.ascontiguousarray(upper, dtype=np.float32)
    r = Res()
    L.lmo_optimize_batched_fused_pad8_s.restype = C.c_int
    L.lmo_optimize_batched_fused_pad8_s.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 7
    rc = L.lmo_optimize_batched_fused_pad8_s(C.addressof(settings), m, t.ctypes.data, basis.ctypes.data, data.ctypes.data,
                                             x.ctypes.data, lo.ctypes.data, up.ctypes.data, C.addressof(r))
    if rc != 0:
        raise MemoryError("lmo_optimize_batched_fused_pad8_s")
    return (r.status, r.iterations, r.fCalls, r.gCalls, r.residual, r.lam), x


def openblas_path():
    import scipy
    cands = glob.glob(os.path.join(os.path.dirname(os.path.dirname(scipy.__file__)), "scipy.libs", "libscipy_openblas*.so"))
    return cands[0] if cands else None


def load_openblas(threads=None):
    p = openblas_path()
    if p is None or lib().lmo_openblas_load(p.encode()) != 0:
        return False
    if threads:
        lib().lmo_openblas_set_threads(int(threads))
    return True
