import ctypes as C, os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
from mir_optim_amd import api, workloads as W
m, n = 1000000, 128
w = W.tanh_linear_data(m, n)
prob = W.TanhLinear(w["A"], w["b"])
WL = api.workloads_lib()
for p in (2, 4, 8):
    X = w["xstar"][None, :] + 0.1 * np.random.default_rng(1).standard_normal((p, n))
    dX = api.DeviceBuffer(X)
    for trial in range(4):
        dY = api.DeviceBuffer(np.zeros((p, m + 64 * trial)))
        call = lambda: WL.wl_tanh_linear_fb_d(C.c_void_p(C.addressof(prob.ctx)), C.c_size_t(m), C.c_size_t(n), C.c_size_t(p), C.c_void_p(dX.ptr), C.c_void_p(dY.ptr))
        ts = []
        for rep in range(5):
            for _ in range(3): call()
            prob.stream.synchronize()
            t0 = time.perf_counter()
            for _ in range(20): call()
            prob.stream.synchronize()
            ts.append((time.perf_counter() - t0) / 20 * 1e3)
        print("p=%d buffer %d (ptr %% 2MB = %d KB): %s ms" % (p, trial, (dY.ptr % (2 << 20)) >> 10, " ".join("%.3f" % t for t in ts)), flush=True)
