"""Time of one launch of 4096 fits (m = 512) for each compiled-in model of the wave-per-problem kernel (run on the GPU box)."""
import sys, time, ctypes as C
import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import mir_optim_amd as M
from mir_optim_amd import api
import problems as P
import test_gpu_batched as T
L = api.lib()
s = M.LeastSquaresSettings(np.float32)
for name, model, maker, n in (("EXP_DECAY n=3", M.MODEL_EXP_DECAY, T.make_exp_decay, 3), ("EXP3_AFFINE n=8", M.MODEL_EXP3_AFFINE, T.make_exp3, 8), ("EXP_DECAY_PAD8 n=8", M.MODEL_EXP_DECAY_PAD8, P.cfg5_pad8, 8)):
    count, m = 4096, 512
    t, data, truth, x0 = maker(count, m)
    dt_, dd, dx0, dx = api.DeviceBuffer(t), api.DeviceBuffer(data), api.DeviceBuffer(x0), api.DeviceBuffer(x0)
    dlo = api.DeviceBuffer(np.full(n, -np.inf, dtype=np.float32)); dup = api.DeviceBuffer(np.full(n, np.inf, dtype=np.float32))
    dres = api.DeviceBuffer(nbytes=count * 24, dtype=np.uint8, shape=(count * 24,))
    st = api.Stream()
    opt = api.BatchedOptions(stream=st.handle)
    def step():
        L.mir_lsq_memcpy_d2d(dx.ptr, dx0.ptr, count * n * 4, st.handle)
        assert L.mir_lsq_batched_kernel_s(C.byref(s), count, m, model, dx.ptr, dlo.ptr, dup.ptr, dt_.ptr, 0, dd.ptr, dres.ptr, C.byref(opt)) == 0
    for _ in range(3): step()
    st.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): step()
    st.synchronize()
    print(name, "%.3f ms per 4096 fits" % ((time.perf_counter() - t0) / 20 * 1e3))
