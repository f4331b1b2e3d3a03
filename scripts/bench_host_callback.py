"""Reference-ABI mode at cfg 3: `mir_optimize_least_squares_d` with a HOST residual callback (x, y host pointers), the
PCIe-inclusive path a caller of the unmodified reference API gets (run on the GPU box).
usage: python scripts/bench_host_callback.py [m] [n]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mir_optim_amd as M
from mir_optim_amd import api, workloads as W

m = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1000000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 128
w = W.tanh_linear_data(m, n)


class HostCtx(C.Structure):
    _fields_ = [("A", C.c_void_p), ("b", C.c_void_p)]


ctx = HostCtx(w["A"].ctypes.data, w["b"].ctypes.data)
f = C.cast(api.workloads_lib().wl_tanh_linear_f_host_d, C.c_void_p).value
s = M.LeastSquaresSettings(); s.absTolerance = 1e-9
# time of one host residual call (the user's own cost) for reference
y = np.zeros(m); x = w["x0"].copy()
fn = C.CFUNCTYPE(None, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p)(f)
fn(C.addressof(ctx), m, n, x.ctypes.data, y.ctypes.data)
t0 = time.perf_counter()
for _ in range(5):
    fn(C.addressof(ctx), m, n, x.ctypes.data, y.ctypes.data)
tf = (time.perf_counter() - t0) / 5
t0 = time.perf_counter()
res, xs = M.optimizeLeastSquares(f, m, w["x0"].copy(), settings=s, fContext=C.addressof(ctx), gpu_entry=False)
dt = time.perf_counter() - t0
print("host residual call: %.2f ms" % (tf * 1e3))
print("reference-ABI solve: %.2f s  %s" % (dt, res))
print("-> %.2f LM iterations/s; %d residual calls = %.2f s of host residual work (%.0f %% of the solve)"
      % (res.iterations / dt, res.fCalls, res.fCalls * tf, 100 * res.fCalls * tf / dt))
