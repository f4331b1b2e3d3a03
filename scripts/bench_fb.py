"""Time the batched residual callback of the tanh-linear workload (run on the GPU box).
WL_BATCHED_V1=1 selects the first (VGPR-staged) kernel, default is the LDS-DMA kernel."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mir_optim_amd import api, workloads as W

m, n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1000000, int(sys.argv[2]) if len(sys.argv) > 2 else 128
w = W.tanh_linear_data(m, n)
A, b, xstar, x0 = w["A"], w["b"], w["xstar"], w["x0"]
prob = W.TanhLinear(A, b)
WL = api.workloads_lib()
rng = np.random.default_rng(1)
for p in (n, 2 * n):
    X = xstar[None, :] + 0.1 * rng.standard_normal((p, n))
    dX = api.DeviceBuffer(X)
    dY = api.DeviceBuffer(np.zeros((p, m)))
    call = lambda: WL.wl_tanh_linear_fb_d(C.c_void_p(C.addressof(prob.ctx)), C.c_size_t(m), C.c_size_t(n), C.c_size_t(p),
                                          C.c_void_p(dX.ptr), C.c_void_p(dY.ptr))
    for _ in range(3):
        call()
    prob.stream.synchronize()
    t0 = time.perf_counter()
    reps = 20
    for _ in range(reps):
        call()
    prob.stream.synchronize()
    dt = (time.perf_counter() - t0) / reps
    Y = dY.download()
    ref = np.tanh(X[:4] @ A[:200000].T) - b[None, :200000]
    err = np.abs(Y[:4, :200000] - ref).max()
    print("m=%d n=%d p=%d: %.3f ms  (%.1f TFLOP/s, %.2f TB/s A+Y)  maxerr %.2e" % (
        m, n, p, dt * 1e3, 2.0 * m * n * p / dt / 1e12, (m * n * 8.0 * ((p + 127) // 128) + p * m * 8.0) / dt / 1e12, err))
