"""Time the single-point and multi-point residual callbacks of the tanh-linear workload (run on the GPU box)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mir_optim_amd import api, workloads as W

m, n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1000000, int(sys.argv[2]) if len(sys.argv) > 2 else 128
w = W.tanh_linear_data(m, n)
prob = W.TanhLinear(w["A"], w["b"])
WL = api.workloads_lib()
for p in (1, 8):
    X = w["xstar"][None, :] + 0.1 * np.random.default_rng(1).standard_normal((p, n))
    dX = api.DeviceBuffer(X); dY = api.DeviceBuffer(np.zeros((p, m)))
    if p == 1:
        call = lambda: WL.wl_tanh_linear_f_d(C.c_void_p(C.addressof(prob.ctx)), C.c_size_t(m), C.c_size_t(n), C.c_void_p(dX.ptr), C.c_void_p(dY.ptr))
    else:
        call = lambda: WL.wl_tanh_linear_fb_d(C.c_void_p(C.addressof(prob.ctx)), C.c_size_t(m), C.c_size_t(n), C.c_size_t(p), C.c_void_p(dX.ptr), C.c_void_p(dY.ptr))
    for _ in range(3):
        call()
    prob.stream.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        call()
    prob.stream.synchronize()
    dt = (time.perf_counter() - t0) / 50
    err = np.abs(dY.download()[:, :100000] - (np.tanh(X @ w["A"][:100000].T) - w["b"][None, :100000])).max()
    print("m=%d n=%d p=%d: %.3f ms (%.2f TB/s)  maxerr %.1e" % (m, n, p, dt * 1e3, (m * n * 8.0 + (p + 1) * m * 8.0) / dt / 1e12, err))
